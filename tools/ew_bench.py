"""Element-wise throughput across field kinds / dtypes / modes (Gop/s and fraction of the 8 TB/s HBM peak)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
ms = ctypes.c_float()
rows = []
cases = [(31, np.uint8, "auto"), (31, np.uint8, "jit-lookup"), (2**8, np.uint8, "jit-calculate"), (2**8, np.int64, "jit-lookup"),
         (65537, np.uint32, "auto"), (7340033, np.uint32, "auto"), (2147483647, np.uint32, "auto"),
         (2**64 - 2**32 + 1, None, "auto"), (2**16, np.uint16, "auto"), (2**16, np.uint16, "jit-calculate"), (2**32, np.uint32, "auto"),
         (3**5, np.uint8, "auto"), (3**5, np.uint8, "jit-calculate"), (251**3, np.uint32, "auto"),
         (2**10, np.uint16, "auto"), (2**10, np.uint16, "jit-calculate"), (2**12, np.uint16, "jit-calculate"), (3**7, np.uint16, "auto"), (3**7, np.uint16, "jit-calculate"),
         (2**10, np.uint16, "jit-lookup"), (2**13, np.uint16, "auto"), (2**13, np.uint16, "jit-lookup"), (8191, np.uint16, "auto"),
         (8191, np.uint16, "jit-lookup"), (5**5, np.uint16, "auto"), (2**16, np.uint16, "jit-lookup"),
         (2**20, np.uint32, "jit-calculate"), (2**24, np.uint32, "auto"),
         (3**7, np.int64, "auto"), (2**12, np.uint32, "auto"),
         (2**14, np.uint16, "auto"), (2**15, np.uint16, "auto"), (3**9, np.uint16, "auto"), (3**10, np.uint16, "auto"), (65521, np.uint16, "auto"), (65521, np.uint16, "jit-lookup")]
if len(sys.argv) > 1 and sys.argv[1] == "--widestore":  # table fields in uint32 / int64 storage (dtype=int of the reference docs)
    cases = [(3**7, np.int64, "auto"), (3**7, np.uint32, "auto"), (2**12, np.int64, "auto"), (7919, np.uint32, "auto"), (2**15, np.int64, "auto")]
if len(sys.argv) > 1 and sys.argv[1] == "--widestore16":  # r05: 32768 < q <= 65536 in uint32 / int64 storage (narrow -> staged LDS tables -> widen)
    cases = [(65521, np.uint32, "jit-lookup"), (65521, np.int64, "jit-lookup"), (3**10, np.uint32, "auto"), (3**10, np.int64, "auto"), (3**10, np.uint16, "auto")]
if len(sys.argv) > 1 and sys.argv[1] == "--packed":  # r05: sums of odd-characteristic extension fields, 8192 < q <= 2^20, as packed digits
    cases = [(3**10, np.uint16, "auto"), (3**9, np.uint16, "auto"), (7**7, np.uint32, "auto"), (5**8, np.uint32, "auto"), (13**5, np.uint32, "auto"),
             (997**2, np.uint32, "auto"), (7**7, np.uint32, "jit-calculate")]
if len(sys.argv) > 1 and sys.argv[1] == "--band16":  # r06: GF(p^m), 32768 < q <= 65536, on the digit tables
    cases = [(251**2, np.uint16, "auto"), (251**2, np.uint32, "auto"), (37**3, np.uint16, "auto"), (251**2, np.uint16, "jit-lookup"), (3**10, np.uint16, "auto")]
if len(sys.argv) > 1 and sys.argv[1] == "--div3":  # r06: degree-3 quotients by Cramer's rule
    cases = [(97**3, np.uint32, "auto"), (41**3, np.uint32, "auto"), (97**3, np.uint32, "jit-lookup")]
if len(sys.argv) > 1 and sys.argv[1] == "--pow24":  # r06: scalar powers of table fields above 65536 elements: per-call table + one gather
    cases = [(7**7, np.uint32, "auto"), (5**8, np.uint32, "auto"), (3**11, np.uint32, "auto"), (2**20, np.uint32, "auto"), (2**17, np.uint32, "auto"), (97**3, np.uint32, "auto")]
if len(sys.argv) > 1 and sys.argv[1] == "--inv16":  # r06: 32768 < q <= 65536 on uint16: reciprocals / quotients through one inverse table in LDS
    cases = [(2**16, np.uint16, "auto"), (65521, np.uint16, "auto"), (65521, np.uint16, "jit-lookup"), (3**10, np.uint16, "auto"), (251**2, np.uint16, "auto"),
             (2**16, np.uint32, "jit-lookup"), (65521, np.uint32, "jit-lookup")]
if len(sys.argv) > 1 and sys.argv[1] == "--divwide":  # r06: GF(p^2) / GF(p^3) above 2^20 elements (no tables): quotients by the norm / Cramer's rule, exact digit split
    def irr2(p):  # x^2 + x + c, irreducible iff 1 - 4 c is a non-residue
        c = next(c for c in range(1, p) if pow((1 - 4 * c) % p, (p - 1) // 2, p) == p - 1)
        return [1, 1, c]
    def irr3(p):  # x^3 + x + c without a root
        c = next(c for c in range(1, p) if all((x * x * x + x + c) % p for x in range(p)))
        return [1, 0, 1, c]
    cases = [(251**3, np.uint32, "auto"), ((1021**3, irr3(1021)), np.uint32, "auto"), ((1621**3, irr3(1621)), np.uint32, "auto"),
             ((1031**2, irr2(1031)), np.uint32, "auto"), ((8191**2, irr2(8191)), np.uint32, "auto"), ((37813**2, irr2(37813)), np.uint32, "auto"),
             (251**3, np.uint32, "jit-calculate"), ((8191**2, irr2(8191)), np.uint32, "jit-calculate")]  # pinned: the digit-vector kernels these fields ran on before
if len(sys.argv) > 1 and sys.argv[1] == "--bininv":  # r06: GF(2^17) .. GF(2^20): quotients / reciprocals through the 3-byte inverse table, powers through LOG / EXP
    cases = [(2**20, np.uint32, "auto"), (2**17, np.uint32, "auto"), (2**20, np.uint32, "jit-calculate"), (2**20, np.uint32, "jit-lookup")]
if len(sys.argv) > 1 and sys.argv[1] == "--divt":  # r06: quotients of degrees 4 .. 8: one gather from the 3-byte inverse table + the digit-table product
    cases = [(7**7, np.uint32, "auto"), (5**8, np.uint32, "auto"), (13**5, np.uint32, "auto"), (31**4, np.uint32, "auto"), (7**6, np.uint32, "auto"), (3**11, np.uint32, "auto"), (3**12, np.uint32, "auto"), (7**7, np.uint32, "jit-lookup")]
if len(sys.argv) > 1 and sys.argv[1] == "--div2":  # r06: degree-2 quotients by the norm
    cases = [(997**2, np.uint32, "auto"), (257**2, np.uint32, "auto"), (509**2, np.uint32, "auto")]
if len(sys.argv) > 1 and sys.argv[1] == "--packed2":  # r06: GF(3^11), GF(3^12): two packed words
    cases = [(3**11, np.uint32, "auto"), (3**12, np.uint32, "auto"), (3**11, np.uint32, "jit-calculate"), (7**7, np.uint32, "auto")]
if len(sys.argv) > 1 and sys.argv[1] == "--extcalc":  # r05: extension fields above 65536 elements pinned to explicit calculation (routing decision)
    cases = [(997**2, np.uint32, "jit-calculate"), (97**3, np.uint32, "jit-calculate"), (31**4, np.uint32, "jit-calculate"), (13**5, np.uint32, "jit-calculate"),
             (5**8, np.uint32, "jit-calculate"), (7**7, np.uint32, "jit-calculate"), (3**11, np.uint32, "jit-calculate"), (3**11, np.uint32, "auto")]
if len(sys.argv) > 1 and sys.argv[1] == "--ext":  # GF(p^m) pinned to explicit calculation (per-degree kernels up to m = 8)
    cases = [(3**5, np.uint8, "jit-calculate"), (3**7, np.uint16, "jit-calculate"), (3**8, np.uint16, "jit-calculate"), (251**3, np.uint32, "auto"), (7**7, np.uint32, "auto")]
if len(sys.argv) > 1 and sys.argv[1] == "--mid":
    cases = [c for c in cases if 256 < c[0] <= 2**16]
if len(sys.argv) > 1 and sys.argv[1] == "--big16":  # 32768 < q <= 65536: one table in LDS at a time
    cases = [(2**16, np.uint16, "auto"), (3**10, np.uint16, "auto"), (65521, np.uint16, "jit-lookup")]
if len(sys.argv) > 1 and sys.argv[1] == "--calc":  # fields whose default route is explicit calculation
    cases = [c for c in cases if c[2] == "jit-calculate" and c[0] > 2**16 or c[0] in (65537, 2**64 - 2**32 + 1)] + [(251**3, np.uint32, "auto")]
if len(sys.argv) > 1 and sys.argv[1] == "--lutpow":  # fields whose np.power with an exponent array runs on the generic table kernel
    cases = [(31, np.uint8, "auto"), (3**5, np.uint8, "auto"), (2**8, np.int64, "jit-lookup"), (3**10, np.uint16, "auto"), (65521, np.uint16, "jit-lookup")]
if len(sys.argv) > 1 and sys.argv[1] == "--bin":
    cases = [c for c in cases if c[0] in (2**20, 2**24, 2**32)]
n = 50_000_000
for order, dt, mode in cases:
    if isinstance(order, tuple):  # (order, irreducible polynomial): fields outside the shipped Conway table
        order, irr = order
        GF = ga.GF(order, irreducible_poly=irr)
    else:
        GF = ga.GF(order)
    GF.compile(mode)
    esize = 8 if dt is None else np.dtype(dt).itemsize
    rng = np.random.default_rng(1)
    if order > 2**63:
        a = torch.from_numpy((rng.integers(0, 2**63, n, dtype=np.uint64) % np.uint64(order)).view(np.int64)).cuda()
        b = torch.from_numpy(((rng.integers(0, 2**63, n, dtype=np.uint64) % np.uint64(order - 1)) + np.uint64(1)).view(np.int64)).cuda()
    else:
        sd = {1: np.uint8, 2: np.int16, 4: np.int32, 8: np.int64}[esize]
        a = torch.from_numpy(rng.integers(0, order, n, dtype=np.uint64).astype(dt).view(sd)).cuda()
        b = torch.from_numpy(rng.integers(1, order, n, dtype=np.uint64).astype(dt).view(sd)).cuda()
    o = torch.empty_like(a)
    code = {1: L.U8, 2: L.U16, 4: L.U32, 8: L.U64}[esize]
    r = {"field": GF.name, "dtype": "u%d" % (8 * esize), "mode": GF.ufunc_mode}
    for name, op in (("add", L.OP_ADD), ("mul", L.OP_MUL), ("div", L.OP_DIV)):
        L.check(lib.gfa_time_binary(GF._handle, op, a.data_ptr(), b.data_ptr(), o.data_ptr(), n, code, st, 5, ctypes.byref(ms)))
        r[name] = f"{n / ms.value / 1e6:.0f} Gop/s ({3 * esize * n / ms.value / 1e6 / 8000:.2f})"
    L.check(lib.gfa_time_unary(GF._handle, L.OP_RECIP, b.data_ptr(), o.data_ptr(), n, code, st, 5, ctypes.byref(ms)))
    r["recip"] = f"{n / ms.value / 1e6:.0f} Gop/s ({2 * esize * n / ms.value / 1e6 / 8000:.2f})"
    # np.power (north_star names it): a scalar exponent and one exponent per element (int64 array, 8 more bytes per element)
    ek = torch.tensor([12345], dtype=torch.int64, device="cuda")
    ev = torch.from_numpy(rng.integers(-50, 1000, n, dtype=np.int64)).cuda()
    for name, e, se, extra in (("pow_scalar", ek, 0, 0), ("pow_array", ev, 1, 8)):
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
        err = torch.zeros(1, dtype=torch.int32, device="cuda")
        L.check(lib.gfa_power(GF._handle, b.data_ptr(), 1, e.data_ptr(), se, o.data_ptr(), n, code, st, err.data_ptr()))
        ev0.record()
        for _ in range(3):
            L.check(lib.gfa_power(GF._handle, b.data_ptr(), 1, e.data_ptr(), se, o.data_ptr(), n, code, st, err.data_ptr()))
        ev1.record(); ev1.synchronize()
        t = ev0.elapsed_time(ev1) / 3
        r[name] = f"{n / t / 1e6:.0f} Gop/s ({(2 * esize + extra) * n / t / 1e6 / 8000:.2f})"
    GF.compile("auto")
    rows.append(r)
    print(json.dumps(r), flush=True)
    del a, b, o
