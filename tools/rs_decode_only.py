"""Runs only the RS(255,223) decoder (2^17 codewords, e ~ U{0..16}) a few times: target for rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
rs = ga.ReedSolomon(255, 223)
B = 1 << 17
rng = np.random.default_rng(4)
C = rs.encode(rng.integers(0, 256, (B, 223), dtype=np.uint8)).numpy()
ne = rng.integers(0, 17, B)
order = np.argsort(rng.random((B, 255), dtype=np.float32), axis=1)[:, :16]
mask = np.arange(16)[None, :] < ne[:, None]
rows = np.repeat(np.arange(B), 16).reshape(B, 16)
R = C.copy()
R[rows[mask], order[mask]] ^= rng.integers(1, 256, (B, 16), dtype=np.uint8)[mask]
Rd = torch.from_numpy(R).cuda(); Dd = torch.empty_like(Rd); Ed = torch.empty(B, dtype=torch.int64, device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    L.check(lib.gfa_rs_decode(rs._handle, Rd.data_ptr(), None, 255, Dd.data_ptr(), Ed.data_ptr(), B, L.U8, st))
torch.cuda.synchronize()
assert np.array_equal(Dd.cpu().numpy(), C)
