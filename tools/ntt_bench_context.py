"""Why does the 2^20 x 64 transform run 15 % slower inside bench.py than in tools/m32_time.py?  The same gfa_time_ntt call in a fresh
process, after a large element-wise workload, after freeing torch's cache, after a pause, and with buffers allocated early / late."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
p, logn, batch = 7340033, 20, 64
P = ga.GF(p); N = 1 << logn
ms = ctypes.c_float()
def make():
    x = torch.from_numpy(np.random.default_rng(3).integers(0, p, (batch, N), dtype=np.uint32).view(np.int32)).cuda()
    return x, torch.empty_like(x)
def t(tag, x, o, iters=10):
    L.check(lib.gfa_time_ntt(P._handle, x.data_ptr(), o.data_ptr(), N, batch, P._root_of_unity_int(N), L.U32, st, iters, ctypes.byref(ms)))
    print(f"{tag:58s} {ms.value:.4f} ms  frac {8.0 * N * batch / (ms.value * 1e-3) / 8e12:.3f}   in@{x.data_ptr():#x} out@{o.data_ptr():#x}", flush=True)
x, o = make()
t("fresh process", x, o); t("again", x, o); t("again, 20 iterations", x, o, 20)
# a bench-like load: GF(2^8) products over 1e8 elements, 300 launches, plus a few GiB of other tensors
G = ga.GF(2**8)
a = torch.randint(0, 256, (100_000_000,), dtype=torch.uint8, device="cuda"); b = torch.randint(1, 256, (100_000_000,), dtype=torch.uint8, device="cuda")
c = torch.empty_like(a)
for _ in range(300):
    L.check(lib.gfa_binary(G._handle, L.OP_MUL, a.data_ptr(), 1, b.data_ptr(), 1, c.data_ptr(), a.numel(), L.U8, st, None))
torch.cuda.synchronize()
t("after 300 element-wise launches (same buffers)", x, o)
junk = [torch.empty(256 << 20, dtype=torch.uint8, device="cuda") for _ in range(12)]
x2, o2 = make()
t("new buffers allocated after 3 GiB of other tensors", x2, o2)
t("old buffers again", x, o)
del junk, a, b, c
torch.cuda.empty_cache()
t("after empty_cache (new buffers)", x2, o2)
time.sleep(2.0)
t("after a 2 s pause (new buffers)", x2, o2)
x3, o3 = make()
t("third pair of buffers", x3, o3)
