"""GF(2^8) multiply (the headline kernel) and a plain device copy at several array sizes: the 1e8-element BASELINE config
has a 300 MB working set, comparable with the 256 MiB Infinity Cache, so larger sizes show the HBM-only rate."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
lib = L.lib()
GF = ga.GF(2**8)
st = torch.cuda.current_stream().cuda_stream
ms = ctypes.c_float()
for n in (10**7, 10**8, 4 * 10**8, 10**9, 2 * 10**9):
    x = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 256)
    y = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 256)
    o = torch.empty_like(x)
    L.check(lib.gfa_time_binary(GF._handle, L.OP_MUL, x.data_ptr(), y.data_ptr(), o.data_ptr(), n, L.U8, st, 20, ctypes.byref(ms)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    o.copy_(x); e0.record()
    for _ in range(20):
        o.copy_(x)
    e1.record(); e1.synchronize()
    cp = e0.elapsed_time(e1) / 20
    print(f"n = {n:>11d}: multiply {ms.value * 1e3:9.1f} us = {3.0 * n / ms.value / 1e9:6.2f} TB/s ({3.0 * n / ms.value / 1e9 / 8 * 100:4.1f} % of 8 TB/s);  copy {cp * 1e3:9.1f} us = {2.0 * n / cp / 1e9:6.2f} TB/s")
    del x, y, o
