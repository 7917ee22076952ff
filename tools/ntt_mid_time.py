"""2^11 .. 2^16-point transforms over GF(p), p < 2^26, 2^26 points per launch: the one-workgroup-per-transform kernels of gfa_ntt_m32.hip (one pass over
HBM: three register networks, two LDS exchanges).  `ntt_mid_time.py 16`: the 2^16-point kernel only, at 1024 and 4096 transforms.  HIP events via
gfa_time_ntt."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
lib = L.lib(); st = torch.cuda.current_stream().cuda_stream
cases = [(7340033, 11, 0), (7340033, 12, 0), (65537, 12, 0), (7340033, 13, 0), (7340033, 14, 0), (7340033, 15, 0), (7340033, 16, 0), (7340033, 16, 4096)]
if len(sys.argv) > 1 and sys.argv[1] == "16":
    cases = [(7340033, 16, 1024), (7340033, 16, 4096), (33292289, 16, 1024)]
for p, logn, nb in cases:
    batch = nb or (1 << 26) >> logn
    P = ga.GF(p); N = 1 << logn
    x = torch.from_numpy(np.random.default_rng(3).integers(0, p, (batch, N), dtype=np.uint32).view(np.int32)).cuda()
    o = torch.empty_like(x); ms = ctypes.c_float()
    L.check(lib.gfa_time_ntt(P._handle, x.data_ptr(), o.data_ptr(), N, batch, P._root_of_unity_int(N), L.U32, st, 20, ctypes.byref(ms)))
    print(f"p={p} 2^{logn} x {batch}: {ms.value:.4f} ms  {8.0*N*batch/(ms.value*1e-3)/8e12:.3f}", flush=True)
