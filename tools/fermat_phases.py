"""Phase timeline of the one-pass GF(65537) kernel: 8 timestamps (100 MHz) per (round, workgroup) from wave 0."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
lib = ctypes.CDLL(L.LIB_PATH)
st = torch.cuda.current_stream().cuda_stream
P = ga.GF(65537); N = 1 << 16
w = P._root_of_unity_int(N)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = torch.from_numpy(np.random.default_rng(3).integers(0, 65537, (batch, N), dtype=np.uint32).view(np.int32)).cuda()
o = torch.empty_like(x)
for _ in range(3):
    L.check(L.lib().gfa_ntt(P._handle, x.data_ptr(), o.data_ptr(), N, batch, w, 0, L.U32, st))
dbg = torch.zeros(batch * 8, dtype=torch.int64, device="cuda")
lib.gfa_debug_fermat_stamps.argtypes = [ctypes.c_void_p]
lib.gfa_debug_fermat_stamps(dbg.data_ptr())
L.check(L.lib().gfa_ntt(P._handle, x.data_ptr(), o.data_ptr(), N, batch, w, 0, L.U32, st))
torch.cuda.synchronize()
lib.gfa_debug_fermat_stamps(None)
t = dbg.cpu().numpy().reshape(-1, 8).astype(np.float64)
t0 = t[:, 0].min()
t = (t - t0) / 100.0  # us
names = ["top", "net0", "tw1a", "W0 tw1b B R0 B W1", "net1a B R1 net1b", "B W0 B R0 B W1", "net2a+st", "ld B R1 net2b+st ld"]
grid = min(batch, 256)
for rnd in range((batch + grid - 1) // grid):
    blk = t[rnd * grid:(rnd + 1) * grid]
    print(f"round {rnd}: " + "  ".join(f"{names[i]} {np.median(blk[:, i]):.1f}" for i in range(8)))
    d = np.diff(blk, axis=1)
    print("   phase durations (median us): " + "  ".join(f"{names[i + 1]} {np.median(d[:, i]):.2f}" for i in range(7)))
    print(f"   start spread: min {blk[:, 0].min():.1f} max {blk[:, 0].max():.1f}; end spread: min {blk[:, 7].min():.1f} max {blk[:, 7].max():.1f}")
