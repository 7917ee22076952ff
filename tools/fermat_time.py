"""Timing of the one-pass GF(65537) 2^16-point kernel over a range of batch sizes (gfa_time_ntt, HIP events)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
P = ga.GF(65537); N = 1 << 16
w = P._root_of_unity_int(N)
batches = [int(b) for b in sys.argv[1:]] or [64, 128, 256, 512, 1024, 2048, 4096]
for batch in batches:
    x = torch.from_numpy(np.random.default_rng(3).integers(0, 65537, (batch, N), dtype=np.uint32).view(np.int32)).cuda()
    o = torch.empty_like(x)
    ms = ctypes.c_float()
    L.check(lib.gfa_time_ntt(P._handle, x.data_ptr(), o.data_ptr(), N, batch, w, L.U32, st, 20, ctypes.byref(ms)))
    pts = batch * N
    print(f"batch {batch:5d}: {ms.value:.4f} ms  {8 * pts / ms.value / 1e6:.0f} GB/s algorithmic  frac {8 * pts / ms.value / 1e6 / 8000:.3f}  "
          f"{ms.value * 1e3 / ((batch + 255) // 256):.1f} us per round of 256")
