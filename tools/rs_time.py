"""RS(255,223), 2^17 codewords: encode (full codewords), decode at e ~ U{0..16}, decode of clean words -- kernel time from
gfa_time_rs_* (HIP events on the launch stream), with a parity check of every output.  Knobs are read from the environment by
the library (GFA_RS_LFSR_REP4, GFA_RS_WPS)."""
import ctypes, os, sys
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
M = rng.integers(0, 256, (B, 223), dtype=np.uint8)
Md = torch.from_numpy(M).cuda()
Cd = torch.empty((B, 255), dtype=torch.uint8, device="cuda")
ms = ctypes.c_float()
L.check(lib.gfa_time_rs_encode(rs._handle, Md.data_ptr(), 223, Cd.data_ptr(), B, L.U8, st, 20, ctypes.byref(ms)))
C = Cd.cpu().numpy()
assert np.array_equal(C[:, :223], M)
te = ms.value
ne = rng.integers(0, 17, B)
order = np.argsort(rng.random((B, 255), dtype=np.float32), axis=1)[:, :16]
mask = np.arange(16)[None, :] < ne[:, None]
rows = np.repeat(np.arange(B), 16).reshape(B, 16)
R = C.copy()
R[rows[mask], order[mask]] ^= rng.integers(1, 256, (B, 16), dtype=np.uint8)[mask]
Rd = torch.from_numpy(R).cuda(); Dd = torch.empty_like(Rd); Ed = torch.empty(B, dtype=torch.int64, device="cuda")
L.check(lib.gfa_time_rs_decode(rs._handle, Rd.data_ptr(), 255, Dd.data_ptr(), Ed.data_ptr(), B, L.U8, st, 10, ctypes.byref(ms)))
assert np.array_equal(Dd.cpu().numpy(), C) and np.array_equal(Ed.cpu().numpy(), ne)
td = ms.value
L.check(lib.gfa_time_rs_decode(rs._handle, Cd.data_ptr(), 255, Dd.data_ptr(), Ed.data_ptr(), B, L.U8, st, 10, ctypes.byref(ms)))
assert np.array_equal(Dd.cpu().numpy(), C)
tc = ms.value
gb = B * 255 / 1e9
print(f"rep4={os.environ.get('GFA_RS_LFSR_REP4', 'auto')} wps={os.environ.get('GFA_RS_WPS', 'auto')}: "
      f"encode {te:.4f} ms = {gb / te * 1e3:.0f} GB/s   decode e~U{{0..16}} {td:.4f} ms = {gb / td * 1e3:.0f} GB/s   clean {tc:.4f} ms = {gb / tc * 1e3:.0f} GB/s")
