"""GF(65537) transforms of 2^13 .. 2^16 points on the grouped one-workgroup kernel (r06): every output of a few rows against the oracle
for batches that are and are not multiples of the group size, the scaled inverse as a round trip, random roots; then timings."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
from oracle import gf_oracle as O
lib = L.lib(); st = torch.cuda.current_stream().cuda_stream
p = 65537
P = ga.GF(p); F = O.OracleField(p, 1, None, int(P.primitive_element))
rng = np.random.default_rng(7)
for logn in (10, 11, 12, 13, 14, 15, 16):
    n = 1 << logn; G = 65536 // n
    for batch in (64 * G, 64 * G + 1, 70 * G + G - 1, 300 * G + 3):
        w = pow(P._root_of_unity_int(n), int(rng.integers(0, n // 2)) * 2 + 1, p)
        x = rng.integers(0, p, (batch, n), dtype=np.uint32)
        x[0] = p - 1; x[-1, ::2] = 0; x[-1, 1::2] = p - 1
        xt = torch.from_numpy(x.view(np.int32)).cuda(); out = torch.full_like(xt, -1)
        L.check(lib.gfa_ntt(P._handle, xt.data_ptr(), out.data_ptr(), n, batch, w, 0, L.U32, st))
        got = out.cpu().numpy().view(np.uint32)
        for i in {0, 1, batch // 2, batch - 2, batch - 1}:
            assert np.array_equal(got[i], F.ntt_u32_pow2(x[i], w)), (logn, batch, i)
        back = torch.empty_like(xt)
        L.check(lib.gfa_ntt(P._handle, out.data_ptr(), back.data_ptr(), n, batch, pow(w, p - 2, p), 1, L.U32, st))
        assert torch.equal(back, xt), ("inverse", logn, batch)
    print(f"2^{logn}: ok", flush=True)
ms = ctypes.c_float()
for logn, batch in ((10, 65536), (10, 262144), (11, 32768), (11, 131072), (12, 16384), (12, 65536), (13, 8192), (14, 4096), (15, 2048), (16, 1024), (13, 32768), (14, 16384), (15, 8192), (16, 4096)):
    n = 1 << logn
    x = torch.from_numpy(rng.integers(0, p, (batch, n), dtype=np.uint32).view(np.int32)).cuda(); o = torch.empty_like(x)
    L.check(lib.gfa_time_ntt(P._handle, x.data_ptr(), o.data_ptr(), n, batch, P._root_of_unity_int(n), L.U32, st, 20, ctypes.byref(ms)))
    print(f"p=65537 2^{logn} x {batch}: {ms.value:.4f} ms  {8.0 * n * batch / (ms.value * 1e-3) / 8e12:.3f}", flush=True)
