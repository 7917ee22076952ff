"""Time of the 2^20-point x 64 transform over GF(7340033) and of 16 x 2^16 over GF(65537) (gfa_time_ntt, HIP events)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
for p, logn, batch in ((7340033, 20, 64), (65537, 16, 1024), (469762049, 20, 64)):
    P = ga.GF(p); N = 1 << logn
    x = torch.from_numpy(np.random.default_rng(3).integers(0, p, (batch, N), dtype=np.uint32).view(np.int32)).cuda()
    o = torch.empty_like(x)
    ms = ctypes.c_float()
    L.check(lib.gfa_time_ntt(P._handle, x.data_ptr(), o.data_ptr(), N, batch, P._root_of_unity_int(N), L.U32, st, 20, ctypes.byref(ms)))
    print(f"p={p} 2^{logn} x {batch}: {ms.value:.4f} ms")
