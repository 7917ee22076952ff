"""A/B timing of the NTT tuning knobs (GFA_NTT_WIDE, GFA_NTT_SUBBATCH_MB) on the GPU box: one subprocess per setting."""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch

    import galois_amd as ga
    from galois_amd import _lib as L
    from oracle import gf_oracle as O

    lib = L.lib()
    stream = torch.cuda.current_stream().cuda_stream
    out = {}
    for tag, p, logn, batch in (("2^20", 7340033, 20, 64), ("2^16", 65537, 16, 1024), ("2^10", 7340033, 10, 65536),
                                ("2^18", 7340033, 18, 256)):
        P = ga.GF(p)
        N = 1 << logn
        omega = P._root_of_unity_int(N)
        xh = np.random.default_rng(3).integers(0, p, (batch, N), dtype=np.uint32)
        xd = torch.from_numpy(xh.view(np.int32)).cuda()
        od = torch.empty_like(xd)
        ms = ctypes.c_float()
        best = 1e9
        for _ in range(3):
            L.check(lib.gfa_time_ntt(P._handle, xd.data_ptr(), od.data_ptr(), N, batch, omega, L.U32, stream, 10, ctypes.byref(ms)))
            best = min(best, ms.value)
        FP = O.OracleField(p, 1, None, P._primitive_element_int)
        ok = bool(np.array_equal(od[batch - 1].cpu().numpy().view(np.uint32), FP.ntt_u32_pow2(xh[batch - 1], omega)))
        out[tag] = {"ms": round(best, 4), "Gpt/s": round(batch * N / best / 1e6, 1), "frac_8TBs": round(8.0 * batch * N / (best * 1e-3) / 8e12, 4), "ok": ok}
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    for wide, sub, xcd in ((0, 0, 0), (0, 0, 1), (1, 0, 1)):
        if True:
            env = dict(os.environ, GFA_NTT_WIDE=str(wide), GFA_NTT_SUBBATCH_MB=str(sub), GFA_NTT_XCD=str(xcd))
            r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]
            print(f"wide={wide} sub_mb={sub} xcd={xcd}: {line}", flush=True)
