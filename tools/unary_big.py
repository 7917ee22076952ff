"""GF(2^8) reciprocal on 1e9 elements (2 GB of traffic per launch, no Infinity Cache reuse)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
lib = L.lib(); GF = ga.GF(2**8)
n = 1_000_000_000
a = torch.empty(n, dtype=torch.uint8, device="cuda").random_(1, 256)
o = torch.empty_like(a)
ms = ctypes.c_float()
L.check(lib.gfa_time_unary(GF._handle, L.OP_RECIP, a.data_ptr(), o.data_ptr(), n, L.U8, torch.cuda.current_stream().cuda_stream, 10, ctypes.byref(ms)))
chk = (GF._wrap(o[-100000:], np.uint8) * GF._wrap(a[-100000:], np.uint8)).numpy()
print(f"reciprocal 1e9: {ms.value:.4f} ms  {2.0 * n / ms.value / 1e6:.0f} GB/s  ok={bool((chk == 1).all())}")
