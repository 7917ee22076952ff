"""A/B timing of the GF(2^8) table-kernel variants and of the plain XOR (GF add) stream as the practical ceiling."""
import ctypes, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def child():
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    import galois_amd as ga
    from galois_amd import _lib as L
    lib = L.lib(); GF = ga.GF(2**8)
    n = 100_000_000
    x = torch.from_numpy(np.random.default_rng(1).integers(0, 256, n, dtype=np.uint8)).cuda()
    y = torch.from_numpy(np.random.default_rng(2).integers(1, 256, n, dtype=np.uint8)).cuda()
    o = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    ms = ctypes.c_float(); out = {}
    for name, op in (("mul", L.OP_MUL), ("div", L.OP_DIV), ("add_xor", L.OP_ADD)):
        best = 1e9
        for _ in range(3):
            L.check(lib.gfa_time_binary(GF._handle, op, x.data_ptr(), y.data_ptr(), o.data_ptr(), n, L.U8, st, 30, ctypes.byref(ms)))
            best = min(best, ms.value)
        out[name] = {"us": round(best * 1e3, 2), "TB/s": round(3e8 / best / 1e9, 3)}
    best = 1e9
    for _ in range(3):
        L.check(lib.gfa_time_unary(GF._handle, L.OP_RECIP, y.data_ptr(), o.data_ptr(), n, L.U8, st, 30, ctypes.byref(ms)))
        best = min(best, ms.value)
    out["recip"] = {"us": round(best * 1e3, 2), "TB/s": round(2e8 / best / 1e9, 3)}
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    o.copy_(x); torch.cuda.synchronize(); t0.record()
    for _ in range(30): o.copy_(x)
    t1.record(); torch.cuda.synchronize()
    out["torch_copy"] = {"us": round(t0.elapsed_time(t1) / 30 * 1e3, 2), "TB/s": round(2e8 / (t0.elapsed_time(t1) / 30) / 1e9, 3)}
    print(json.dumps(out))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(); sys.exit(0)
    for var in ("1n", "1t", "2n", "2t", "4n", "4t"):
        env = dict(os.environ, GFA_TAB8_VARIANT=var)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print(var, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
