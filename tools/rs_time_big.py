"""RS(255,223) encode / decode kernel time at several batch sizes (the LFSR kernels' persistent regime needs many words per CU)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
lib = L.lib(); st = torch.cuda.current_stream().cuda_stream
rs = ga.ReedSolomon(255, 223); ms = ctypes.c_float()
for logb in (17, 18, 20, 22):
    B = 1 << logb
    Md = torch.empty((B, 223), dtype=torch.uint8, device="cuda").random_(0, 256)
    Cd = torch.empty((B, 255), dtype=torch.uint8, device="cuda")
    L.check(lib.gfa_time_rs_encode(rs._handle, Md.data_ptr(), 223, Cd.data_ptr(), B, L.U8, st, 10, ctypes.byref(ms))); te = ms.value
    Dd = torch.empty_like(Cd); Ed = torch.empty(B, dtype=torch.int64, device="cuda")
    L.check(lib.gfa_time_rs_decode(rs._handle, Cd.data_ptr(), 255, Dd.data_ptr(), Ed.data_ptr(), B, L.U8, st, 10, ctypes.byref(ms))); tc = ms.value
    assert bool(torch.equal(Dd, Cd)) and int(Ed.abs().sum()) == 0
    print(f"2^{logb} words: encode {te:.4f} ms = {B * 255 / te / 1e6:.0f} GB/s   clean decode (pre-pass only) {tc:.4f} ms = {B * 255 / tc / 1e6:.0f} GB/s", flush=True)
