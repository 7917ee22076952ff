"""Three-pass Goldilocks transforms (2^22 .. 2^28 points, one transform): HIP-event time per size."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import galois_amd as ga
from galois_amd import _lib as L
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
P = ga.GF(2**64 - 2**32 + 1)
ms = ctypes.c_float()
for logn in [int(a) for a in sys.argv[1:]] or [22, 24, 26, 28]:
    N = 1 << logn
    x = torch.empty((1, N), dtype=torch.int64, device="cuda").random_(0, 2**62)
    o = torch.empty_like(x)
    L.check(lib.gfa_time_ntt(P._handle, x.data_ptr(), o.data_ptr(), N, 1, P._root_of_unity_int(N), L.U64, st, 10, ctypes.byref(ms)))
    print(f"2^{logn}: {ms.value:.4f} ms  frac {16 * N / ms.value / 1e6 / 8000:.3f}")
