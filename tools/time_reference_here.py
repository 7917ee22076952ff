"""Sanity anchor (CPU baseline item 3): the ACTUAL reference code, imported in place from
/root/reference/src through oracle/ref_shim (pure-Python "python-calculate" mode -- Numba is not installable here), timed in
the build container on small samples.  Not runnable on the GPU box (no /root/reference there); the numbers are committed
under profiles/.  They bound the reference from below: its Numba-compiled mode is 100-1000x faster."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_shim"))
import numpy as np
import load_reference

galois = load_reference.load()
print(f"host: {os.cpu_count()} vCPUs; reference imported from /root/reference/src in python-calculate mode")
rng = np.random.default_rng(1)
GF = load_reference.ref_field(2**8, irreducible_poly=galois.matlab_primitive_poly(2, 8))
n = 1_000_000
x, y = GF(rng.integers(0, 256, n, dtype=np.uint8)), GF(rng.integers(0, 256, n, dtype=np.uint8))
t = time.perf_counter(); z = x * y; dt = time.perf_counter() - t; dt_mul = dt
print(f"GF(2^8) multiply, {n} elements: {dt:.2f} s = {n / dt / 1e6:.2f} Mop/s")
P = load_reference.ref_field(65537)
v = P(rng.integers(0, 65537, 4096))
t = time.perf_counter(); X = np.fft.fft(v); dt = time.perf_counter() - t; dt_ntt = dt
print(f"4096-point NTT over GF(65537): {dt:.3f} s = {1 / dt:.2f} transforms/s")
rs = galois.ReedSolomon(255, 223, field=GF)
M = GF(rng.integers(0, 256, (16, 223), dtype=np.uint8))
t = time.perf_counter(); C = rs.encode(M); te = time.perf_counter() - t
R = np.array(C)
for i in range(16):
    pos = rng.choice(255, i + 1, replace=False)
    R[i, pos] ^= rng.integers(1, 256, i + 1, dtype=np.uint8)
t = time.perf_counter(); D, ne = rs.decode(GF(R), errors=True); td = time.perf_counter() - t
assert np.array_equal(np.array(D), np.array(M)) and list(ne) == list(range(1, 17))
import json
rec = {"where": f"build container, {os.cpu_count()} vCPUs, one thread", "mode": "python-calculate (Numba is not installable here)",
       "gf256_mul_Mop/s": round(n / dt_mul / 1e6, 3), "gf256_mul_sample": f"{n} elements",
       "ntt_4096_gf65537_transforms/s": round(1 / dt_ntt, 2), "rs_255_223_encode_kB/s": round(16 * 255 / te / 1e3, 2),
       "rs_255_223_decode_kB/s": round(16 * 255 / td / 1e3, 2), "rs_sample": "16 codewords with 1..16 errors"}
if len(sys.argv) > 1:
    json.dump(rec, open(sys.argv[1], "w"), indent=1)
print(f"RS(255,223) 16 codewords: encode {te:.2f} s = {16 * 255 / te / 1e3:.2f} kB/s, decode (1..16 errors) {td:.2f} s = {16 * 255 / td / 1e3:.2f} kB/s")
