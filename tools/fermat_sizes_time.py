"""GF(65537) transforms of 2^10 .. 2^16 points, 2^26 points per launch: timings only (tuning aid; GALOIS_AMD_LIB selects a variant library)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
lib = L.lib(); st = torch.cuda.current_stream().cuda_stream
p = 65537; P = ga.GF(p); ms = ctypes.c_float(); rng = np.random.default_rng(1)
out = []
for logn in (10, 12, 14, 15, 16):
    n = 1 << logn; batch = (1 << 26) >> logn
    x = torch.from_numpy(rng.integers(0, p, (batch, n), dtype=np.uint32).view(np.int32)).cuda(); o = torch.empty_like(x)
    best = 1e9
    for _ in range(2):
        L.check(lib.gfa_time_ntt(P._handle, x.data_ptr(), o.data_ptr(), n, batch, P._root_of_unity_int(n), L.U32, st, 20, ctypes.byref(ms)))
        best = min(best, ms.value)
    out.append(f"2^{logn} {8.0 * n * batch / (best * 1e-3) / 8e12:.3f}")
print("  ".join(out), flush=True)
