"""Phase timeline of the paired-workgroup GF(65537) kernel (two 512-thread workgroups per CU): 8 timestamps (100 MHz) per
(round, workgroup) from wave 0, plus each workgroup's placement (HW_ID, LDS_ALLOC), so that the two workgroups sharing a CU
can be laid side by side.  usage: fermat_pair_phases.py [batch]"""
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
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
x = torch.from_numpy(np.random.default_rng(3).integers(0, 65537, (batch, N), dtype=np.uint32).view(np.int32)).cuda()
o = torch.empty_like(x)
for _ in range(3):
    L.check(L.lib().gfa_ntt(P._handle, x.data_ptr(), o.data_ptr(), N, batch, w, 0, L.U32, st))
grid = min(batch, int(os.environ.get("GFA_NTT_FERMAT_GRID", "512")))
rounds = (batch + grid - 1) // grid
dbg = torch.zeros(rounds * grid * 8 + grid, dtype=torch.int64, device="cuda")
lib.gfa_debug_fermat_stamps.argtypes = [ctypes.c_void_p]
lib.gfa_debug_fermat_stamps(dbg.data_ptr())
L.check(L.lib().gfa_ntt(P._handle, x.data_ptr(), o.data_ptr(), N, batch, w, 0, L.U32, st))
torch.cuda.synchronize()
lib.gfa_debug_fermat_stamps(None)
raw = dbg.cpu().numpy()
t = raw[:rounds * grid * 8].reshape(rounds, grid, 8).astype(np.float64)
place = raw[rounds * grid * 8:].astype(np.uint64)
hwid = (place & np.uint64(0xffffffff)).astype(np.int64); ldsa = (place >> np.uint64(32)).astype(np.int64)
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
t0 = t[t > 0].min()
t = (t - t0) / 100.0
names = ["top", "net0a+tw", "net0b", "X1(+net1 0..2)", "net1(3)", "X2(+net2 0..2,ld)", "net2(3)", "tw+ld"]
for r in range(rounds):
    blk = t[r]
    d = np.diff(blk, axis=1)
    print(f"round {r}: start med {np.median(blk[:, 0]):.1f} (min {blk[:, 0].min():.1f} max {blk[:, 0].max():.1f})  end med {np.median(blk[:, 7]):.1f}  "
          + "  ".join(f"{names[i + 1]} {np.median(d[:, i]):.2f}" for i in range(7)))
print("LDS_ALLOC values seen (hex):", sorted({hex(int(v)) for v in ldsa})[:8])
# workgroups that share (blockIdx % 8 = XCD, se, sh, cu)
key = {}
for b in range(grid):
    key.setdefault((b % 8, int(se[b]), int(sh[b]), int(cu[b])), []).append(b)
sizes = np.bincount([len(v) for v in key.values()])
print("workgroups per (xcd, se, sh, cu):", {i: int(c) for i, c in enumerate(sizes) if c})
shown = 0
for k, bs in sorted(key.items()):
    if len(bs) == 2 and shown < 4:
        shown += 1
        for b in bs:
            print(f"  cu {k} block {b:4d} lds {hex(int(ldsa[b]))}: " + " | ".join(" ".join(f"{t[r, b, i]:6.1f}" for i in range(8)) for r in range(rounds)))
# overlap measure: fraction of a workgroup's compute span [0..5] during which its CU mate is also inside a compute span
ov = []
for k, bs in key.items():
    if len(bs) != 2:
        continue
    a, b = bs
    for r in range(rounds):
        a0, a1 = t[r, a, 0], t[r, a, 5]
        tot = 0.0
        for r2 in range(rounds):
            b0, b1 = t[r2, b, 0], t[r2, b, 5]
            tot += max(0.0, min(a1, b1) - max(a0, b0))
        ov.append(tot / max(a1 - a0, 1e-9))
if ov:
    print(f"fraction of a compute span shared with the CU mate's compute span: median {np.median(ov):.2f}")
