"""Pass-shape sweep of the three-pass transform: n = 2^log0 * 2^log1 * 2^log2 (GFA_NTT3_LOG0 / GFA_NTT3_LOG1)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L

lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(7)
for p, dt, width in ((2013265921, L.U32, 4), (2**64 - 2**32 + 1, L.U64, 8)):
    GF = ga.GF(p)
    for lg in (22, 24, 26):
        n = 1 << lg
        if width == 4:
            x = torch.from_numpy(rng.integers(0, p, n, dtype=np.uint32).view(np.int32)).cuda()
        else:
            x = torch.from_numpy((rng.integers(0, 2**63, n, dtype=np.uint64) % np.uint64(p)).view(np.int64)).cuda()
        o = torch.empty_like(x)
        om = GF._root_of_unity_int(n)
        res = []
        i = 0
        for l0 in range(4, 11):
            for l1 in range(4, 11):
                l2 = lg - l0 - l1
                if l2 < 4 or l2 > 10:
                    continue
                os.environ["GFA_NTT3_LOG0"] = str(l0); os.environ["GFA_NTT3_LOG1"] = str(l1)
                i += 1
                w = GF._scalar(L.OP_POW, om, 2 * i + 1)
                ms = ctypes.c_float()
                rc = lib.gfa_time_ntt(GF._handle, x.data_ptr(), o.data_ptr(), n, 1, w, dt, st, 3, ctypes.byref(ms))
                if rc:
                    continue
                res.append((ms.value, l0, l1, l2))
        res.sort()
        print(f"p={p} n=2^{lg}: best " + "  ".join(f"({a},{b},{c}) {t:.3f}" for t, a, b, c in res[:6]) + f"  | worst {res[-1]}", flush=True)
        del x, o
