"""Per-GPU compute part of BASELINE config 5 (Goldilocks, one 2^26-point NTT over 8 GPUs): the two kernels each rank
runs around the single all-to-all -- gfa_ntt_columns on its (8192 x 1024) column block and the batched row transform
(1024 rows of 8192) -- plus, for reference, batched 2^20-point Goldilocks transforms.  The exchange itself (8 MiB per peer
pair over xGMI) cannot be measured on a 1-GPU box."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
from galois_amd import _dist as D

P = 2**64 - 2**32 + 1
GF = ga.GF(P)
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(6)
G = 8
omega = GF._root_of_unity_int(1 << 26)


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


for n1, n2 in ((1 << 13, 1 << 13), (1 << 10, 1 << 16), (1 << 16, 1 << 10)):
    cols, rows = n2 // G, n1 // G
    x = torch.from_numpy((rng.integers(0, 2**63, (n1, cols), dtype=np.uint64) % np.uint64(P)).view(np.int64)).cuda()
    try:
        a = D._device_column_pass(GF, x, n1, cols, 3 * cols, n1 * n2, omega)
    except Exception as e:
        print(f"split {n1} x {n2}: column pass unsupported ({e})")
        continue
    t_cols = timed(lambda: D._device_column_pass(GF, x, n1, cols, 3 * cols, n1 * n2, omega))
    mine = a.reshape(rows, n2).contiguous()
    om2 = GF._scalar(L.OP_POW, omega, n1)
    t_rows = timed(lambda: D._device_row_pass(GF, mine, n2, om2))
    pts = n1 * cols
    print(f"C5 per-rank compute, split n1 x n2 = {n1} x {n2} ({pts} points per rank):")
    print(f"  column pass ({n1}-point columns x {cols}, + twiddle): {t_cols:.3f} ms  {16 * pts / t_cols / 1e6:.0f} GB/s algorithmic")
    print(f"  row pass    ({rows} rows of {n2})                   : {t_rows:.3f} ms  {16 * pts / t_rows / 1e6:.0f} GB/s algorithmic")
    print(f"  => per-rank compute {t_cols + t_rows:.3f} ms (+ one all-to-all: 7 x 8 MiB per rank, ~0.11 ms at 77 GB/s per xGMI link and direction)")
    del x, a, mine
ms = ctypes.c_float()
N, B = 1 << 20, 16
xb = torch.from_numpy((rng.integers(0, 2**63, (B, N), dtype=np.uint64) % np.uint64(P)).view(np.int64)).cuda()
ob = torch.empty_like(xb)
L.check(lib.gfa_time_ntt(GF._handle, xb.data_ptr(), ob.data_ptr(), N, B, GF._root_of_unity_int(N), L.U64, st, 5, ctypes.byref(ms)))
print(f"Goldilocks 2^20-point NTT x {B}: {ms.value:.3f} ms per launch = {B / ms.value * 1e3:.0f} transforms/s, {16.0 * N * B / ms.value / 1e6:.0f} GB/s algorithmic ({16.0 * N * B / ms.value / 1e6 / 80:.1f} % of 8 TB/s)")
