"""Runs the 2^20-point x 64 NTT a few times (target for rocprofv3 kernel-trace / PMC passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
lib = L.lib()
p, logn, batch = 7340033, 20, 64
P = ga.GF(p); N = 1 << logn
omega = P._root_of_unity_int(N)
x = torch.from_numpy(np.random.default_rng(3).integers(0, p, (batch, N), dtype=np.uint32).view(np.int32)).cuda()
o = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    L.check(lib.gfa_ntt(P._handle, x.data_ptr(), o.data_ptr(), N, batch, omega, 0, L.U32, st))
torch.cuda.synchronize()
