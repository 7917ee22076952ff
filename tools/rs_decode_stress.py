"""Stress run of the RS(255,223) wave decoder (rs_decode_bin_kernel, hand-written Berlekamp-Massey / Horner / claiming code):
2^20 codewords per case at exactly 0, t and t + 1 errors, with and without erasures, at the GFA_RS_WPS setting of the
environment (the library reads it once per process: tests/test_gpu_rs.py starts one process per setting).
Checks on EVERY word: decodable cases return the transmitted codeword and the exact count; words beyond the radius are all
reported as failures (-1) -- a miscorrection has probability ~3e-14 per word -- and a 512-word sample of every case goes
through the oracle (decoded word, count) as well.  Prints `rs decode stress ok`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import galois_amd as ga
from oracle import gf_oracle as O

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
rs = ga.ReedSolomon(255, 223)
F = O.OracleField(2, 8, 285, 2, lookup=True)
OR = O.OracleRS(F, 255, 223)
rng = np.random.default_rng(int(os.environ.get("GFA_RS_WPS", "8")))
M = rng.integers(0, 256, (B, 223), dtype=np.uint8)
C = rs.encode(M).numpy()
t = 16


def corrupt(ne, nu):
    """ne errors and nu erasures per word at distinct random positions (erased symbols get random values)"""
    keys = rng.random((B, 255), dtype=np.float32)
    order = np.argsort(keys, axis=1)[:, :ne + nu]
    R = C.copy()
    rows = np.arange(B)[:, None]
    if ne:
        R[rows, order[:, :ne]] ^= rng.integers(1, 256, (B, ne), dtype=np.uint8)
    E = None
    if nu:
        E = np.zeros((B, 255), dtype=bool)
        E[rows, order[:, ne:]] = True
        R[rows, order[:, ne:]] = rng.integers(0, 256, (B, nu), dtype=np.uint8)
    return R, E


cases = [(0, 0), (t, 0), (t + 1, 0), (0, 6), (13, 6), (14, 6), (0, 32), (1, 32), (1, 30)]
for ne, nu in cases:
    R, E = corrupt(ne, nu)
    dec, nerr = rs.decode(R, erasures=E, output="codeword", errors=True)
    dec = dec.numpy()
    ok = 2 * ne + nu <= 32
    blind = nu == 32 and ne > 0  # every parity symbol spent on erasures: ANY word decodes to some codeword, nothing can be detected
    sel = rng.choice(B, 512, replace=False)
    od, on = OR.decode_u8(R[sel], None if E is None else E[sel])
    assert np.array_equal(dec[sel], od) and np.array_equal(nerr[sel], on), f"oracle sample differs at ({ne} errors, {nu} erasures)"
    if ok:
        assert np.array_equal(dec, C), f"({ne}, {nu}): {np.count_nonzero((dec != C).any(axis=1))} words not restored"
        if nu == 0:
            assert (nerr == ne).all(), f"({ne}, 0): counts {np.unique(nerr)[:8]}"
        else:  # an erased symbol that kept its value by chance is not counted: the oracle sample above pins the rule
            assert nerr.min() >= ne and nerr.max() <= ne + nu, f"({ne}, {nu}): counts {np.unique(nerr)[:8]}"
    elif not blind:
        assert (nerr == -1).all(), f"({ne}, {nu}): {np.count_nonzero(nerr != -1)} words beyond the radius not reported as failures"
    print(f"  {ne:2d} errors {nu:2d} erasures: {'restored' if ok else ('oracle sample only' if blind else 'all reported as failures')}; counts {sorted(set(int(v) for v in np.unique(nerr)))[:6]}", flush=True)
print("rs decode stress ok", B, "words per case, GFA_RS_WPS =", os.environ.get("GFA_RS_WPS", "default"))
