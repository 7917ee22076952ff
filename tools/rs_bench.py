"""RS(255,223) encode/decode timing for three error distributions (benchmarks/test_fec.py style extremes + uniform)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import galois_amd as ga
from galois_amd import _lib as L

lib = L.lib()
stream = torch.cuda.current_stream().cuda_stream
rs = ga.ReedSolomon(255, 223)
B = 1 << 17
rng = np.random.default_rng(4)
M = rng.integers(0, 256, (B, 223), dtype=np.uint8)
Md = torch.from_numpy(M).cuda()
Cd = torch.empty((B, 255), dtype=torch.uint8, device="cuda")
ms = ctypes.c_float()
L.check(lib.gfa_time_rs_encode(rs._handle, Md.data_ptr(), 223, Cd.data_ptr(), B, L.U8, stream, 10, ctypes.byref(ms)))
out = {"encode_ms": round(ms.value, 4), "encode_GB/s": round(255.0 * B / ms.value / 1e6, 1)}
C = Cd.cpu().numpy()
for tag, ne in (("e=0", np.zeros(B, dtype=int)), ("e=16", np.full(B, 16)), ("e~U{0..16}", rng.integers(0, 17, B))):
    keys = rng.random((B, 255), dtype=np.float32)
    order = np.argsort(keys, axis=1)[:, :16]
    R = C.copy()
    mask = np.arange(16)[None, :] < ne[:, None]
    rows = np.repeat(np.arange(B), 16).reshape(B, 16)
    vals = rng.integers(1, 256, (B, 16), dtype=np.uint8)
    R[rows[mask], order[mask]] ^= vals[mask]
    Rd = torch.from_numpy(R).cuda()
    Dd = torch.empty_like(Rd)
    Ed = torch.empty(B, dtype=torch.int64, device="cuda")
    L.check(lib.gfa_time_rs_decode(rs._handle, Rd.data_ptr(), 255, Dd.data_ptr(), Ed.data_ptr(), B, L.U8, stream, 10, ctypes.byref(ms)))
    ok = bool(np.array_equal(Dd.cpu().numpy(), C) and np.array_equal(Ed.cpu().numpy(), ne))
    out[tag] = {"decode_ms": round(ms.value, 4), "decode_GB/s": round(255.0 * B / ms.value / 1e6, 1), "ok": ok}
print(json.dumps(out))
