"""Throughput survey of the field linear-algebra kernels (gfa_matmul / gfa_row_reduce / gfa_plu_decompose) with the
oracle's C matmul timed beside it.  Usage: python tools/linalg_bench.py"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
from oracle import gf_oracle as O

lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
ms = ctypes.c_float()
rng = np.random.default_rng(0)
NP2T = {np.uint8: torch.uint8, np.uint16: torch.int16, np.uint32: torch.int32, np.uint64: torch.int64}


def dev(a):
    return torch.from_numpy(a.view({1: np.uint8, 2: np.int16, 4: np.int32, 8: np.int64}[a.itemsize])).cuda()


def mm(tag, GF, npdt, gdt, batch, M, K, N, iters=10):
    q = GF.order
    hi = min(q, 2**63)
    a = rng.integers(0, hi, (batch, M, K), dtype=np.uint64).astype(npdt)
    b = rng.integers(0, hi, (batch, K, N), dtype=np.uint64).astype(npdt)
    da, db = dev(a), dev(b)
    do = torch.empty((batch, M, N), dtype=da.dtype, device="cuda")
    L.check(lib.gfa_time_matmul(GF._handle, da.data_ptr(), db.data_ptr(), do.data_ptr(), batch, M, K, N, gdt, st, iters, ctypes.byref(ms)))
    macs = batch * M * K * N
    isz = a.itemsize
    gbs = (a.nbytes + b.nbytes + batch * M * N * isz) / (ms.value * 1e-3) / 1e9
    print(f"{tag:34s} batch {batch:6d} {M}x{K}x{N}: {ms.value:9.4f} ms  {macs / (ms.value * 1e-3) / 1e12:8.3f} TMAC/s  {gbs:8.1f} GB/s algorithmic")
    return a, b, do


G8 = ga.GF(2**8)
a, b, do = mm("GF(2^8) u8 (27 planes, MFMA)", G8, np.uint8, L.U8, 1, 4096, 4096, 4096, 3)
F8 = O.OracleField(2, 8, 285, 2, lookup=True)
t = time.perf_counter(); ref = F8.matmul(a[0, :256, :256], b[0, :256, :256]); dt = time.perf_counter() - t
print(f"  oracle C port, 1 thread: 256^3 in {dt * 1e3:.1f} ms = {256**3 / dt / 1e9:.3f} GMAC/s")
for sz in (1024, 2048, 8192):  # r06: bit planes on the matrix cores from 2^30 multiply-adds
    mm("GF(2^8) u8 (27 planes, MFMA)", G8, np.uint8, L.U8, 1, sz, sz, sz, 3)
mm("GF(2^4) u8 (9 planes, MFMA)", ga.GF(2**4), np.uint8, L.U8, 1, 4096, 4096, 4096, 3)
mm("GF(2^8) u8 (27 planes, ragged)", G8, np.uint8, L.U8, 1, 4000, 4100, 3900, 3)
mm("GF(2^16) u16 (81 planes, MFMA)", ga.GF(2**16), np.uint16, L.U16, 1, 4096, 4096, 4096, 3)
mm("GF(2^16) u16 (81 planes, MFMA)", ga.GF(2**16), np.uint16, L.U16, 1, 1024, 1024, 1024, 3)
mm("GF(2^16) u16 (81 planes, MFMA)", ga.GF(2**16), np.uint16, L.U16, 1, 512, 512, 512, 3)
mm("GF(2^8) u8 (27 planes, MFMA)", G8, np.uint8, L.U8, 1, 512, 512, 512, 3)
mm("GF(2^8) u8 (LDS table: N < 256)", G8, np.uint8, L.U8, 1, 4096, 4096, 128, 3)
mm("GF(3^5) u8 (22 digit planes, MFMA)", ga.GF(3**5), np.uint8, L.U8, 1, 1024, 1024, 1024, 3)
mm("GF(3^5) u8 (22 digit planes, MFMA)", ga.GF(3**5), np.uint8, L.U8, 1, 4096, 4096, 4096, 3)
mm("GF(7^3) u16 (8 digit planes, MFMA)", ga.GF(7**3), np.uint16, L.U16, 1, 4096, 4096, 4096, 3)
mm("GF(251^2) u16 (3 planes, MFMA)", ga.GF(251**2), np.uint16, L.U16, 1, 4096, 4096, 4096, 3)
mm("GF(3^10) u16 (digit planes, MFMA)", ga.GF(3**10), np.uint16, L.U16, 1, 2048, 2048, 2048, 3)
mm("Goldilocks u64 (10 limbs, MFMA)", ga.GF(2**64 - 2**32 + 1), np.uint64, L.U64, 1, 2048, 2048, 2048, 3)
mm("GF(2^8) u8 small stack", G8, np.uint8, L.U8, 16384, 16, 16, 16)
mm("GF(2^8) u8 RS-encode shape", G8, np.uint8, L.U8, 1, 131072, 223, 32)
mm("GF(65537) u32 (3 limbs, MFMA)", ga.GF(65537), np.uint32, L.U32, 1, 4096, 4096, 4096, 3)
mm("GF(2^31-1) u32 (5 limbs, MFMA)", ga.GF(2147483647), np.uint32, L.U32, 1, 4096, 4096, 4096, 3)
mm("GF(65537) u32 (VALU, below thresh)", ga.GF(65537), np.uint32, L.U32, 1, 400, 400, 400, 3)
mm("GF(31) u8", ga.GF(31), np.uint8, L.U8, 1, 4096, 4096, 4096, 3)
mm("GF(251) u8", ga.GF(251), np.uint8, L.U8, 1, 8192, 8192, 8192, 3)
mm("GF(2) u8", ga.GF(2), np.uint8, L.U8, 1, 8192, 8192, 8192, 3)
mm("GF(31) u8 stack", ga.GF(31), np.uint8, L.U8, 64, 512, 512, 512, 3)
mm("Goldilocks u64", ga.GF(2**64 - 2**32 + 1), np.uint64, L.U64, 1, 1024, 1024, 1024, 3)
mm("GF(3^5) u8 (Zech tables)", ga.GF(3**5), np.uint8, L.U8, 1, 1024, 1024, 1024, 3)
mm("GF(2^32) u32 (243 planes, MFMA)", ga.GF(2**32), np.uint32, L.U32, 1, 2048, 2048, 2048, 3)
mm("GF(2^20) u32 (bit planes, MFMA)", ga.GF(2**20), np.uint32, L.U32, 1, 2048, 2048, 2048, 3)
mm("GF(2^32) u32 (243 planes, MFMA)", ga.GF(2**32), np.uint32, L.U32, 1, 1024, 1024, 1024, 3)

# elimination: stacks of small systems and one large matrix
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps

S = G8(rng.integers(0, 256, (65536, 16, 16), dtype=np.uint8))
print(f"inv_batched GF(2^8) 65536 x 16x16: {timed(lambda: ga.linalg.inv_batched(S[:1]) if False else ga.linalg.row_reduce_batched(S)) * 1e3:.3f} ms (row_reduce_batched)")
print(f"det_batched GF(2^8) 65536 x 16x16: {timed(lambda: ga.linalg.det_batched(S)) * 1e3:.3f} ms")
P = ga.GF(65537)
A = P(rng.integers(0, 65537, (1024, 1024), dtype=np.uint32))
print(f"inv GF(65537) 1024x1024 (two kernels per column): {timed(lambda: np.linalg.inv(A), 1) * 1e3:.1f} ms")
S2 = P(rng.integers(0, 65537, (1024, 64, 64), dtype=np.uint32))
print(f"inv_batched GF(65537) 1024 x 64x64: {timed(lambda: ga.linalg.inv_batched(S2)) * 1e3:.2f} ms")
