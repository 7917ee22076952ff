"""RS(255,223), 2^17 codewords: decode time against the number of errors per word (every word the same count) -- where the wave decoder's time
goes: the fixed part (syndromes from the remainder, 32 Berlekamp-Massey steps) against the part that grows with the error count (Chien, Forney,
corrections).  Kernel time from gfa_time_rs_decode (HIP events on the launch stream); every output checked."""
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
C = rs.encode(ga.GF(2**8)(M)).numpy()
order = np.argsort(rng.random((B, 255), dtype=np.float32), axis=1)[:, :16]
vals = rng.integers(1, 256, (B, 16), dtype=np.uint8)
rows = np.repeat(np.arange(B), 16).reshape(B, 16)
ms = ctypes.c_float()
gb = B * 255 / 1e9
for e in [int(v) for v in sys.argv[1:]] or [0, 1, 2, 4, 8, 12, 16]:
    mask = np.broadcast_to(np.arange(16)[None, :] < e, (B, 16))
    R = C.copy()
    R[rows[mask], order[mask]] ^= vals[mask]
    Rd = torch.from_numpy(R).cuda(); Dd = torch.empty_like(Rd); Ed = torch.empty(B, dtype=torch.int64, device="cuda")
    L.check(lib.gfa_time_rs_decode(rs._handle, Rd.data_ptr(), 255, Dd.data_ptr(), Ed.data_ptr(), B, L.U8, st, 10, ctypes.byref(ms)))
    assert np.array_equal(Dd.cpu().numpy(), C) and (Ed.cpu().numpy() == e).all()
    print(f"errors per word {e:2d}: {ms.value * 1e3:7.1f} us = {gb / ms.value * 1e3:5.0f} GB/s", flush=True)
