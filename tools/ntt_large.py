"""Single large transforms (2^20 .. 2^28 points): time per launch and algorithmic GB/s (8 or 16 bytes per point)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L

lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(7)
logs = [int(a) for a in sys.argv[1:]] or [20, 21, 22, 24, 26]
only = int(os.environ.get('NTT_P', '0'))
for p, dt, tdt, width in ((469762049, L.U32, torch.int32, 4), (2013265921, L.U32, torch.int32, 4), (3221225473, L.U32, torch.int32, 4), (2**64 - 2**32 + 1, L.U64, torch.int64, 8)):
    if only and p != only:
        continue
    GF = ga.GF(p)
    for lg in logs:
        n = 1 << lg
        if (p - 1) % n:
            continue
        if width == 4:
            x = torch.from_numpy(rng.integers(0, p, n, dtype=np.uint32).view(np.int32)).cuda()
        else:
            x = torch.from_numpy((rng.integers(0, 2**63, n, dtype=np.uint64) % np.uint64(p)).view(np.int64)).cuda()
        o = torch.empty_like(x)
        ms = ctypes.c_float()
        L.check(lib.gfa_time_ntt(GF._handle, x.data_ptr(), o.data_ptr(), n, 1, GF._root_of_unity_int(n), dt, st, 3, ctypes.byref(ms)))
        print(f"p={p} n=2^{lg}: {ms.value:.3f} ms  {2.0 * width * n / ms.value / 1e6:.0f} GB/s algorithmic", flush=True)
        del x, o
