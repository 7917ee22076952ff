"""Runs only reciprocals (and quotients) of GF(7^7) / GF(7^6) uint32 arrays a few times: target for rocprofv3 --pmc passes over packed_divt_kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
lib = L.lib()
n = 1 << 26
st = torch.cuda.current_stream().cuda_stream
for q in ((int(sys.argv[2]),) if len(sys.argv) > 2 else (7**7, 7**6)):
    GF = ga.GF(q)
    x = torch.from_numpy(np.random.default_rng(1).integers(1, q, n, dtype=np.uint32)).cuda()
    y = torch.from_numpy(np.random.default_rng(2).integers(1, q, n, dtype=np.uint32)).cuda()
    o = torch.empty_like(x)
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
        L.check(lib.gfa_unary(GF._handle, L.OP_RECIP, y.data_ptr(), o.data_ptr(), n, L.U32, st, None))
        L.check(lib.gfa_binary(GF._handle, L.OP_DIV, x.data_ptr(), 1, y.data_ptr(), 1, o.data_ptr(), n, L.U32, st, None))
    torch.cuda.synchronize()
