"""GF(2^8) / GF(2^16) matmul: Karatsuba bit planes on the matrix cores against the other kernels (LDS product table / shift-and-xor), cube sizes
256 .. 1024: where the threshold GFA_MFMA_BITS_MIN_LOG belongs.  Run twice: GFA_MFMA_BITS_MIN_LOG=16 (planes always) and =62 (never)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
ms = ctypes.c_float()
rng = np.random.default_rng(0)
for m, npdt, gdt, tdt in ((8, np.uint8, L.U8, np.uint8), (16, np.uint16, L.U16, np.int16)):
    GF = ga.GF(2**m)
    for n in (256, 384, 512, 640, 768, 1024):
        a = torch.from_numpy(rng.integers(0, 2**m, (n, n), dtype=np.uint64).astype(npdt).view(tdt)).cuda()
        b = torch.from_numpy(rng.integers(0, 2**m, (n, n), dtype=np.uint64).astype(npdt).view(tdt)).cuda()
        o = torch.empty_like(a)
        L.check(lib.gfa_time_matmul(GF._handle, a.data_ptr(), b.data_ptr(), o.data_ptr(), 1, n, n, n, gdt, st, 5, ctypes.byref(ms)))
        print(f"GFA_MFMA_BITS_MIN_LOG={os.environ.get('GFA_MFMA_BITS_MIN_LOG', 'default')}  GF(2^{m}) {n}^3: {ms.value * 1e3:8.1f} us  {n**3 / ms.value / 1e9:7.3f} TMAC/s", flush=True)
