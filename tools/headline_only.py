"""Runs only the headline kernel (GF(2^8) multiply, 1e8 uint8 elements) a few times: target for rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
lib = L.lib()
GF = ga.GF(2**8)
n = 100_000_000
x = torch.from_numpy(np.random.default_rng(1).integers(0, 256, n, dtype=np.uint8)).cuda()
y = torch.from_numpy(np.random.default_rng(2).integers(0, 256, n, dtype=np.uint8)).cuda()
o = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    L.check(lib.gfa_binary(GF._handle, L.OP_MUL, x.data_ptr(), 1, y.data_ptr(), 1, o.data_ptr(), n, L.U8, st, None))
torch.cuda.synchronize()
