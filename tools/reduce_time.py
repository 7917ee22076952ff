"""gfa_reduce / gfa_accumulate over ONE long row, device time (HIP events around back-to-back C-ABI calls, no host synchronisation between
them): element rate and fraction of 8 TB/s at the bytes the call has to move (reduce: the input; accumulate: input + output)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
for q, n in ((2**8, 10**8), (3**5, 10**8), (31, 10**8), (7340033, 5 * 10**7), (2**16, 5 * 10**7), (2**20, 5 * 10**7), (2**64 - 2**32 + 1, 2 * 10**7)):
    GF = ga.GF(q)
    x = GF.Random(n, low=1, seed=2)
    t = x._t
    es = t.element_size()
    code = {1: L.U8, 2: L.U16, 4: L.U32, 8: L.U64}[es]
    out1 = torch.empty(1, dtype=t.dtype, device="cuda")
    outn = torch.empty_like(t)
    row = {"field": GF.name, "n": n}
    for name, op in (("add", L.OP_ADD), ("multiply", L.OP_MUL)):
        for kind in ("reduce", "accumulate"):
            def call():
                if kind == "reduce":
                    L.check(lib.gfa_reduce(GF._handle, op, t.data_ptr(), out1.data_ptr(), 1, n, code, st, None))
                else:
                    L.check(lib.gfa_accumulate(GF._handle, op, t.data_ptr(), outn.data_ptr(), 1, n, code, st, None))
            call(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                call()
            e1.record(); e1.synchronize()
            ms = e0.elapsed_time(e1) / 5
            byt = es * n * (1 if kind == "reduce" else 2)
            row[f"{name}.{kind}"] = f"{n / ms / 1e6:.0f} Gel/s ({byt / ms / 1e6 / 8000:.2f})"
    print(row, flush=True)
