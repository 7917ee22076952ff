#!/usr/bin/env python
"""
bench.py -- headline benchmark of the finite-field hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU; RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the env)

Workload (BASELINE.json configs[1]): GF(2^8) multiply, EXP/LOG-class lookup mode (full LDS product table), 1e8 uint8
elements PER GPU, inputs resident in HBM (x: seed 1, y: seed 2, as SURVEY.md section 8(d)).  A "step" is one pass of
np.multiply over the 1e8-element batch through the C-ABI (gfa_binary).  Weak scaling: every rank owns an independent
1e8-element shard; there is no data-path collective, only the timing barrier and a MAX all-reduce of the elapsed time.

One JSON line is printed by rank 0.  Besides the contract keys it carries
  "roofline":     dominant kernel (tab8_binary_kernel) -- algorithmic 3 B/element x 1e8 per launch / HIP-event time
  "cpu_baseline": the C port of the reference's lookup ufunc (oracle/gf_oracle.c, 1 thread) on the same data
  "extra":        the other two parts of BASELINE.json's composite metric (2^20-point NTT/s, RS(255,223) GB/s),
                  measured after the timed region on this rank's GPU, each with its own roofline fraction.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
N_ELEMENTS = 100_000_000


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: ~0.1 s of warm-up steps and ~0.1 s timed (wall clock around 20 steps = 1 ms carries the launch / sync latency)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--no-extras", action="store_true", help="skip the NTT / Reed-Solomon side measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the two rocprofv3 --pmc passes that measure the HBM traffic")
    ap.add_argument("--dist-extras", action="store_true",
                    help="run the multi-GPU side measurements (sharded RS / NTT, distributed C5 transform) even at world size 1")
    ap.add_argument("--print-launch", action="store_true",
                    help="print the torch.distributed.run command `--gpus N` re-executes itself under, and exit")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher around it: start the N ranks ourselves (one process per GPU over RCCL)
        # and pass rank 0's JSON line through.  Refuses to pretend: fewer visible GPUs than ranks is an error unless the
        # single-device plumbing knob is set.
        sys.exit(self_launch(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("GFA_BENCH_SINGLE_DEVICE"):  # plumbing runs only: every rank on the one visible GPU
        local_rank = int(os.environ["GFA_BENCH_SINGLE_DEVICE"])
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.dist_extras:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        backend = os.environ.get("GFA_BENCH_BACKEND", "nccl")  # "gloo": plumbing runs with several ranks on one GPU
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        assert dist.get_world_size() == max(args.gpus, 1) or args.dist_extras, (dist.get_world_size(), args.gpus)

    if not os.path.exists(os.path.join(ROOT, "galois_amd", "libgalois_amd.so")):
        # a checkout without build artefacts: build the product (rank 0 builds, the others wait at the barrier)
        import subprocess

        if rank == 0:
            subprocess.run([sys.executable, os.path.join(ROOT, "galois_amd", "build.py")], check=True, stdout=sys.stderr)
        if dist is not None:
            dist.barrier()
    import galois_amd as ga
    from galois_amd import _lib as L

    lib = L.lib()
    GF = ga.GF(2**8)  # irreducible x^8+x^4+x^3+x^2+1, alpha = 2; default mode = lookup
    assert GF.ufunc_mode == "jit-lookup"
    n = N_ELEMENTS
    # identical bits to the reference's GF.Random(seed=...) = default_rng(seed).integers(0, 256, n, uint8)
    x_h = np.random.default_rng(1 + 1000 * rank).integers(0, 256, n, dtype=np.uint8)
    y_h = np.random.default_rng(2 + 1000 * rank).integers(0, 256, n, dtype=np.uint8)
    x = torch.from_numpy(x_h).cuda()
    y = torch.from_numpy(y_h).cuda()
    out = torch.empty_like(x)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        rc = lib.gfa_binary(GF._handle, L.OP_MUL, x.data_ptr(), 1, y.data_ptr(), 1, out.data_ptr(), n, L.U8, stream, None)
        if rc:
            L.check(rc, "gfa_binary")

    # As-measured figure with the protocol of rounds 1-3 (5 warm-up + 20 timed steps straight after the idle set-up phase, no clock
    # pre-warm): kept NEXT to the warm figure so that rounds stay comparable
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t_cold = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    cold_elapsed = time.perf_counter() - t_cold

    # Clock pre-warm, untimed and outside the W + K steps: the set-up above leaves the GPU idle for seconds, and it then needs tens
    # of milliseconds of load before its clocks are back up -- a kernel timed straight away reads 3-15 % slow
    # (profiles/r04_bench_clock_ramp.txt).  ~0.1 s of the same launch; reported in the JSON line as "clock_prewarm".
    prewarm_steps, t_pw = 0, time.perf_counter()
    while time.perf_counter() - t_pw < 0.1:
        for _ in range(64):
            step()
        torch.cuda.synchronize()
        prewarm_steps += 64
    prewarm_ms = (time.perf_counter() - t_pw) * 1e3
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # parity gate on this rank's shard (sampled every 997th element + both ends) against the oracle port
    from oracle import gf_oracle as O

    F = O.OracleField(2, 8, 285, 2, lookup=True)
    idx = np.unique(np.concatenate([np.arange(0, n, 997), np.arange(0, 4096), np.arange(n - 4096, n)]))
    got = out.cpu().numpy()[idx]
    assert np.array_equal(got, F.ufunc_u8(O.MUL, x_h[idx], y_h[idx])), "GPU result differs from the oracle"

    # roofline of the dominant kernel: HIP events on the launch stream, inside the library (gfa_time_binary)
    ms = ctypes.c_float()
    L.check(lib.gfa_time_binary(GF._handle, L.OP_MUL, x.data_ptr(), y.data_ptr(), out.data_ptr(), n, L.U8, stream, 50,
                                ctypes.byref(ms)), "gfa_time_binary")
    alg_bytes = 3.0 * n
    achieved = alg_bytes / (ms.value * 1e-3) / 1e9
    # HBM traffic per launch from the PMC counters: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
    # tools/headline_only.py (same kernel, same 1e8-element inputs), corrected as MI355X_MICROARCH.md prescribes;
    # the raw counter files and the derivation are committed under profiles/ (PMC passes cannot run inside this process)
    traffic, traffic_source = None, None
    if rank == 0 and world == 1 and not args.no_pmc:
        traffic, traffic_source = measure_traffic()  # live: two rocprofv3 --pmc passes over tools/headline_only.py
    if traffic is None:
        for name in ("r02_pmc_headline.json", "r01_pmc_headline.json"):
            pmc_path = os.path.join(ROOT, "profiles", name)
            if os.path.exists(pmc_path):
                traffic = json.load(open(pmc_path))["traffic_bytes_per_launch"]
                traffic_source = f"profiles/{name}"
                break
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_live": bool(traffic_source and traffic_source.startswith("live")),
                "kernel": "tab8_binary_kernel", "kernel_ms": round(ms.value, 5), "algorithmic_bytes_per_launch": alg_bytes}

    result = None
    if rank == 0:
        value = world * n * args.steps / elapsed / 1e9
        result = {
            "metric": "GF(2^8) mul Gop/s (lookup mode, 1e8 uint8 per GPU)",
            "value": round(value, 2),
            "unit": "Gop/s",
            "n_gpus": world,
            "rccl_ranks": (dist.get_world_size() if dist is not None else 1),
            "steps": args.steps,
            "warmup": args.warmup,
            "clock_prewarm": {"steps": prewarm_steps, "ms": round(prewarm_ms, 1), "timed": False},
            "as_measured_without_prewarm": {"value": round(n * 20 / cold_elapsed / 1e9, 2), "unit": "Gop/s", "steps": 20, "warmup": 5},
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "GF(2^8) mul, lookup mode, 1e8 uint8 per GPU, poly 0x11D (BASELINE.json configs[1])",
                       "elements_per_gpu": n, "parallelism": f"batch-shard x{world}, no collectives"},
            "roofline": roofline,
        }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # bounded CPU sample: the same 1e8-element batch, repeated until >= 10 s of single-thread work
        t_cpu, reps = 0.0, 0
        F.ufunc_u8(O.MUL, x_h[:1_000_000], y_h[:1_000_000])
        while t_cpu < 10.0 and reps < 400:
            t1 = time.perf_counter()
            F.ufunc_u8(O.MUL, x_h, y_h)
            t_cpu += time.perf_counter() - t1
            reps += 1
        result["cpu_baseline"] = {"value": round(n * reps / t_cpu / 1e9, 4), "unit": "Gop/s", "cores": 1, "kind": "port",
                                  "sample": f"{reps} x the 1e8-element batch, oracle/gf_oracle.c, 1 of {os.cpu_count()} host threads"}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline_reference"] = reference_timing()

    if rank == 0 and world == 1 and not args.no_extras:
        result["extra"] = extras(ga, L, lib, stream, world == 1 and not args.no_cpu_baseline)
        if not args.no_cpu_baseline:
            # the same port over every host core (ctypes releases the GIL): what an OpenMP'd reference loop could reach
            from concurrent.futures import ThreadPoolExecutor

            cores = os.cpu_count() or 1
            cuts = np.linspace(0, n, cores + 1).astype(np.int64)
            outs = np.empty(n, dtype=np.uint8)

            def work(i):
                outs[cuts[i]:cuts[i + 1]] = F.ufunc_u8(O.MUL, x_h[cuts[i]:cuts[i + 1]], y_h[cuts[i]:cuts[i + 1]])

            with ThreadPoolExecutor(cores) as pool:
                list(pool.map(work, range(cores)))
                t1, reps = time.perf_counter(), 0
                while time.perf_counter() - t1 < 5.0:
                    list(pool.map(work, range(cores)))
                    reps += 1
                dt = time.perf_counter() - t1
            result["extra"]["cpu_baseline_all_cores"] = {"value": round(n * reps / dt / 1e9, 3), "unit": "Gop/s", "cores": cores,
                                                         "kind": "port", "sample": f"{reps} x 1e8 elements"}

    if dist is not None and not args.no_extras and (world > 1 or args.dist_extras):
        # the other parts of the composite metric at N GPUs: every rank takes part, rank 0 reports
        ex = extras_distributed(ga, L, lib, stream, dist, rank, world)
        if rank == 0:
            result.setdefault("extra", {}).update(ex)
    if rank == 0:
        # the figures BASELINE.json's composite metric names, as the LAST key of the line (whatever keeps only a tail keeps this)
        ex = result.pop("extra", None)
        if ex is not None:
            result["extra"] = ex
            ns = {"gf256_mul_Gop/s": result["value"], "gf256_mul_frac": result["roofline"]["frac"]}
            for tag in ("ntt_2^20_gf7340033", "ntt_2^20_gf469762049", "ntt_2^20_gf2013265921"):
                if tag in ex:
                    ns[tag] = {"tps": ex[tag]["transforms_per_s"], "frac": ex[tag]["roofline_frac"], "physical_frac": ex[tag].get("physical_frac")}
            for tag in ("ntt_16x2^16_gf65537", "ntt_2^14_gf65537", "ntt_2^16_gf7340033", "ntt_2^14_gf7340033"):
                if tag in ex:
                    ns[tag + "_frac"] = ex[tag]["roofline_frac"]
            if "ntt_16x2^16_gf65537" in ex:
                ns["ntt_2^16_gf65537_hbm_only_4096_frac"] = ex["ntt_16x2^16_gf65537"]["hbm_only_batch_4096"]["roofline_frac"]
            for tag in ("ntt_single_2^26_gf469762049", "ntt_single_2^26_goldilocks", "ntt_single_2^27_gf2013265921"):
                if tag in ex:
                    ns[tag + "_frac"] = ex[tag]["roofline_frac"]
            if "rs_255_223" in ex:
                ns["rs_enc_GB/s"] = ex["rs_255_223"]["encode_GB/s"]
                ns["rs_dec_GB/s"] = ex["rs_255_223"]["decode_GB/s"]
            # N > 1: the sharded legs (whole-job figures over all ranks) and the distributed 2^26-point Goldilocks transform
            if "rs_255_223_sharded" in ex:
                ns["rs_sharded_enc_GB/s"] = ex["rs_255_223_sharded"]["encode_GB/s"]
                ns["rs_sharded_dec_GB/s"] = ex["rs_255_223_sharded"]["decode_GB/s"]
            for tag in ("ntt_2^20_gf7340033_sharded", "ntt_16x2^16_gf65537_sharded"):
                if tag in ex:
                    ns[tag] = {"tps": ex[tag]["transforms_per_s"], "frac_per_gpu": ex[tag]["roofline_frac_per_gpu"]}
            c5 = ex.get("c5_goldilocks_2^26_distributed")
            if c5 and "forward" in c5:
                ns["c5_goldilocks_2^26_forward_ms"] = c5["forward"].get("wall_ms")
            result["north_star"] = ns
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def self_launch(args):
    """`python bench.py --gpus N` without WORLD_SIZE in the environment: re-execute under torch.distributed.run with N ranks
    on this node (127.0.0.1 rendezvous, a free port) and return its exit code.  The children print the JSON line."""
    import socket
    import subprocess

    single = os.environ.get("GFA_BENCH_SINGLE_DEVICE")
    if not args.print_launch and single is None:
        ndev = torch.cuda.device_count()
        if ndev < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) visible (set GFA_BENCH_SINGLE_DEVICE=<ordinal> and "
                  f"GFA_BENCH_BACKEND=gloo for a plumbing run with every rank on one GPU)", file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != "--print-launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    if args.print_launch:
        print(" ".join(cmd))
        return 0
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def reference_timing():
    """the ACTUAL reference on a stated subsample.  The reference is pure Python, is not vendored
    (and must not be) and /root/reference does not exist on a GPU box, so it cannot be timed in this process: this is the committed
    record of tools/time_reference_here.py -- the reference imported in place in the build container (python-calculate mode; its
    Numba mode is not installable there), labelled with where and on what it ran."""
    rec_path = os.path.join(ROOT, "profiles", "r03_reference_timings.json")
    if not os.path.exists(rec_path):
        return None
    rec = json.load(open(rec_path))
    keep = {k: v for k, v in rec.items() if not isinstance(v, str)}
    return {"kind": "reference", "timed_in_this_run": False, "source": "profiles/r03_reference_timings.json", "mode": "python-calculate", **keep}


def measure_traffic():
    """HBM bytes per launch of the headline kernel from the PMC counters, measured NOW: separate rocprofv3 passes for
    FETCH_SIZE and WRITE_SIZE over tools/headline_only.py (same kernel, same 1e8-element inputs; --kernel-trace only, as the
    guide prescribes), corrected as MI355X_MICROARCH.md's HBM section says for gfx950: FETCH_SIZE counts a wide coalesced
    streaming read at half its bytes.  Returns (bytes, description) or (None, None) when rocprofv3 is not usable here."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, None
    vals = {}
    tmp = tempfile.mkdtemp(prefix="gfa_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run([exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                                sys.executable, os.path.join(ROOT, "tools", "headline_only.py"), "6"],
                               cwd="/tmp", env=env, capture_output=True, timeout=300)
            got = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "tab8_binary" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                        got.append(float(row["Counter_Value"]))
            if r.returncode != 0 or not got:
                return None, None
            vals[counter] = sum(got) / len(got)  # KiB per dispatch
        read_b = 2.0 * vals["FETCH_SIZE"] * 1024.0
        write_b = vals["WRITE_SIZE"] * 1024.0
        return read_b + write_b, f"live: FETCH_SIZE {vals['FETCH_SIZE']:.1f} x 2 + WRITE_SIZE {vals['WRITE_SIZE']:.1f} KiB"
    except Exception:
        return None, None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def extras_distributed(ga, L, lib, stream, dist, rank, world):
    """BASELINE.json configs[3] and [4] and the batched NTTs at N GPUs: codewords and whole
    transforms shard across ranks with no data-path collective; the single 2^26-point Goldilocks transform uses the
    four-step split with ONE all-to-all.  Every figure is whole-job: units of all ranks / MAX over ranks of the time."""
    from galois_amd import dist as gdist

    ms = ctypes.c_float()

    def job_ms(local_ms):
        t = torch.tensor([local_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    ex = {}
    # ---- RS(255,223): 2^20 codewords batch-sharded over the ranks, e ~ U{0..16} errors per codeword ----
    rs = ga.ReedSolomon(255, 223)
    total = 1 << 20
    lo, hi = gdist.shard_range(total, rank, world)
    B = hi - lo
    rng = np.random.default_rng(4 + 1000 * rank)
    M = rng.integers(0, 256, (B, 223), dtype=np.uint8)
    Md = torch.from_numpy(M).cuda()
    Cd = torch.empty((B, 255), dtype=torch.uint8, device="cuda")
    dist.barrier()
    L.check(lib.gfa_time_rs_encode(rs._handle, Md.data_ptr(), 223, Cd.data_ptr(), B, L.U8, stream, 20, ctypes.byref(ms)))
    enc_ms = job_ms(ms.value)
    # errors are planted on the device: ne[i] positions per codeword, distinct, non-zero values
    g = torch.Generator(device="cuda").manual_seed(5 + rank)
    ne = torch.randint(0, 17, (B,), device="cuda", generator=g)
    order = torch.rand((B, 255), device="cuda", generator=g).argsort(dim=1)[:, :16]
    vals = torch.randint(1, 256, (B, 16), device="cuda", generator=g, dtype=torch.int64).to(torch.uint8)
    mask = torch.arange(16, device="cuda")[None, :] < ne[:, None]
    Rd = Cd.clone()
    rowsel = torch.arange(B, device="cuda")[:, None].expand(B, 16)
    Rd[rowsel[mask], order[mask]] ^= vals[mask]
    Dd = torch.empty_like(Rd)
    Ed = torch.empty(B, dtype=torch.int64, device="cuda")
    dist.barrier()
    L.check(lib.gfa_time_rs_decode(rs._handle, Rd.data_ptr(), 255, Dd.data_ptr(), Ed.data_ptr(), B, L.U8, stream, 20, ctypes.byref(ms)))
    dec_ms = job_ms(ms.value)
    assert torch.equal(Dd, Cd) and torch.equal(Ed, ne), "RS round trip failed"
    ex["rs_255_223_sharded"] = {
        "codewords_total": total, "codewords_per_rank": B, "errors_per_codeword": "uniform 0..16",
        "encode_GB/s": round(255.0 * total / (enc_ms * 1e-3) / 1e9, 2), "decode_GB/s": round(255.0 * total / (dec_ms * 1e-3) / 1e9, 2),
        "encode+decode_GB/s": round(255.0 * total / ((enc_ms + dec_ms) * 1e-3) / 1e9, 2),
        "encode_ms": round(enc_ms, 4), "decode_ms": round(dec_ms, 4), "collectives": "none (timing only)"}
    del Md, Cd, Rd, Dd, Ed, order, vals, mask, rowsel
    # ---- batched NTTs: whole transforms per rank ----
    for tag, p, logn, batch in (("ntt_2^20_gf7340033_sharded", 7340033, 20, 64), ("ntt_16x2^16_gf65537_sharded", 65537, 16, 1024)):
        P = ga.GF(p)
        N = 1 << logn
        xd = torch.from_numpy(np.random.default_rng(3 + rank).integers(0, p, (batch, N), dtype=np.uint32).view(np.int32)).cuda()
        od = torch.empty_like(xd)
        dist.barrier()
        L.check(lib.gfa_time_ntt(P._handle, xd.data_ptr(), od.data_ptr(), N, batch, P._root_of_unity_int(N), L.U32, stream, 10, ctypes.byref(ms)))
        t = job_ms(ms.value)
        ex[tag] = {"transforms_per_s": round(world * batch / (t * 1e-3), 1), "batch_per_rank": batch, "ms_per_launch": round(t, 4),
                   "2^20_points_per_s_equiv": round(world * batch * N / (1 << 20) / (t * 1e-3), 1),
                   "roofline_frac_per_gpu": round(8.0 * batch * N / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "collectives": "none (timing only)"}
        del xd, od
    # ---- C5: ONE 2^26-point Goldilocks transform over all ranks: column pass, one all-to-all, row pass ----
    p = 2**64 - 2**32 + 1
    P = ga.GF(p)
    N = 1 << 26
    if world >= 2 and (world & (world - 1)) == 0 or world == 1:
        n1, n2 = gdist.choose_split(N, world) if world > 1 else (1 << 10, 1 << 16)
        cols = n2 // world
        xl = torch.empty((n1, cols), dtype=torch.int64, device="cuda").random_(0, 2**62)  # < p: valid field elements
        omega = P._root_of_unity_int(N)
        fw, bw = {}, {}
        X = gdist.ntt_four_step_distributed(P, xl, n1, n2, omega=omega)  # warm-up: plans, RCCL channels
        back = gdist.intt_four_step_distributed(P, X, n1, n2, omega=omega)
        assert torch.equal(back, xl), "distributed inverse(forward(x)) != x"
        # X[0] = sum of all inputs: the one output that is cheap to recompute independently
        part = int(np.add.reduce(P._wrap(xl.reshape(-1), np.object_)))
        parts = [None] * world
        dist.all_gather_object(parts, part)
        if rank == 0:
            assert int(X[0, 0].item()) % 2**64 == sum(parts) % p, "X[0] differs from the sum of the inputs"
        for _ in range(5):
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            X = gdist.ntt_four_step_distributed(P, xl, n1, n2, omega=omega, timings=fw)
            torch.cuda.synchronize()
            fw.setdefault("wall_ms", []).append((time.perf_counter() - t0) * 1e3)
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gdist.intt_four_step_distributed(P, X, n1, n2, omega=omega, timings=bw)
            torch.cuda.synchronize()
            bw.setdefault("wall_ms", []).append((time.perf_counter() - t0) * 1e3)
        entry = {"points": N, "split": f"{n1} x {n2}", "ranks": world, "bytes_per_peer_pair": 8 * (n1 // world) * cols}
        for name, tm in (("forward", fw), ("inverse", bw)):
            stages = {k: round(job_ms(float(np.median(v))), 4) for k, v in sorted(tm.items())}
            if name == "inverse":
                kern = stages.get("column_pass_ms", 0) + stages.get("row_pass_ms", 0) + stages.get("relayout_ms", 0)
                stages["kernels_ms"] = round(kern, 4)
            stages["points_per_s"] = round(N / (stages["wall_ms"] * 1e-3), 0)
            entry[name] = stages
        ex["c5_goldilocks_2^26_distributed"] = entry
    return ex


def _all_cores(fn_of_index, seconds=3.0):
    """Calls fn_of_index(i) from one thread per host core for about `seconds` (the oracle's C entry points release the GIL):
    (calls completed, elapsed seconds, threads) -- the all-cores leg of a cpu_baseline."""
    import concurrent.futures as cf
    import itertools
    import threading

    cores = max(1, min(os.cpu_count() or 1, 128))
    counter = itertools.count()
    done = [0] * cores
    stop = threading.Event()

    def worker(w):
        while not stop.is_set():
            fn_of_index(next(counter))
            done[w] += 1

    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(max_workers=cores) as ex_:
        futs = [ex_.submit(worker, w) for w in range(cores)]
        time.sleep(seconds)
        stop.set()
        for f in futs:
            f.result()
    return sum(done), time.perf_counter() - t0, cores


def extras(ga, L, lib, stream, with_cpu):
    """The NTT and Reed-Solomon parts of the composite metric, on this rank's GPU (not part of the timed region)."""
    from oracle import gf_oracle as O

    ex = {"cpu_baselines": "oracle/gf_oracle.c (kind: port), ~3 s samples of the same inputs"}
    ms = ctypes.c_float()
    GF = ga.GF(2**8)
    n = N_ELEMENTS
    # ---- measured device-to-device copy (1 read + 1 write of 1e8 bytes): the practical HBM ceiling on this box ----
    src = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 256)
    dst = torch.empty_like(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        dst.copy_(src)
    e0.record()
    for _ in range(50):
        dst.copy_(src)
    e1.record()
    e1.synchronize()
    copy_gbs = 2.0 * n / (e0.elapsed_time(e1) / 50 * 1e-3) / 1e9
    ex["device_copy"] = {"GB/s": round(copy_gbs, 1), "frac_of_peak": round(copy_gbs / HBM_PEAK_GBS, 4)}
    del src, dst
    # ---- the headline kernel on arrays far larger than the 256 MiB Infinity Cache (the 1e8-element BASELINE config has a
    # 300 MB working set and is partly served by it): the HBM-only rate ----
    nbig = 1_000_000_000
    xb = torch.empty(nbig, dtype=torch.uint8, device="cuda").random_(0, 256)
    yb = torch.empty(nbig, dtype=torch.uint8, device="cuda").random_(0, 256)
    ob = torch.empty_like(xb)
    L.check(lib.gfa_time_binary(GF._handle, L.OP_MUL, xb.data_ptr(), yb.data_ptr(), ob.data_ptr(), nbig, L.U8, stream, 10, ctypes.byref(ms)))
    F8b = O.OracleField(2, 8, 285, 2, lookup=True)
    sl = slice(nbig - 100_000, nbig)
    assert np.array_equal(ob[sl].cpu().numpy(), F8b.ufunc_u8(O.MUL, xb[sl].cpu().numpy(), yb[sl].cpu().numpy()))
    e0.record()
    for _ in range(10):
        ob.copy_(xb)
    e1.record()
    e1.synchronize()
    ex["gf256_mul_1e9_elements"] = {"Gop/s": round(nbig / (ms.value * 1e-3) / 1e9, 1), "kernel_ms": round(ms.value, 4),
                                    "algorithmic_GB/s": round(3.0 * nbig / (ms.value * 1e-3) / 1e9, 1),
                                    "roofline_frac": round(3.0 * nbig / (ms.value * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "device_copy_GB/s_same_size": round(2.0 * nbig / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e9, 1)}
    del xb, yb, ob
    # ---- the headline op at the reference docs' dtype=int (int64 storage, 24 B/element), same field and kernel family ----
    n64 = 25_000_000
    a64 = torch.from_numpy(np.random.default_rng(1).integers(0, 256, n64, dtype=np.int64)).cuda()
    b64 = torch.from_numpy(np.random.default_rng(2).integers(0, 256, n64, dtype=np.int64)).cuda()
    o64 = torch.empty_like(a64)
    L.check(lib.gfa_time_binary(GF._handle, L.OP_MUL, a64.data_ptr(), b64.data_ptr(), o64.data_ptr(), n64, L.U64, stream, 20,
                                ctypes.byref(ms)))
    F8 = O.OracleField(2, 8, 285, 2, lookup=True)
    chk = slice(0, 100_000)
    assert np.array_equal(o64[chk].cpu().numpy().astype(np.uint8),
                          F8.ufunc_u8(O.MUL, a64[chk].cpu().numpy().astype(np.uint8), b64[chk].cpu().numpy().astype(np.uint8)))
    gbs = 24.0 * n64 / (ms.value * 1e-3) / 1e9
    ex["gf256_mul_int64"] = {"Gop/s": round(n64 / (ms.value * 1e-3) / 1e9, 2), "elements": n64, "kernel_ms": round(ms.value, 5),
                             "algorithmic_GB/s": round(gbs, 1), "roofline_frac": round(gbs / HBM_PEAK_GBS, 4)}
    del a64, b64, o64
    # ---- reciprocal, 2 B/element ----
    a = torch.from_numpy(np.random.default_rng(2).integers(1, 256, n, dtype=np.uint8)).cuda()
    o = torch.empty_like(a)
    L.check(lib.gfa_time_unary(GF._handle, L.OP_RECIP, a.data_ptr(), o.data_ptr(), n, L.U8, stream, 50, ctypes.byref(ms)))
    gbs = 2.0 * n / (ms.value * 1e-3) / 1e9
    ex["gf256_reciprocal"] = {"Gop/s": round(n / (ms.value * 1e-3) / 1e9, 2), "kernel_ms": round(ms.value, 5),
                              "roofline_frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_GB/s": round(gbs, 1)}
    del a, o
    # ---- fields above 256 elements on uint16 storage (EXP / LOG / Zech as 16-bit tables in LDS): 6 B/element ----
    for tag, order, ops in (("gf3^7_uint16", 3**7, (("add", L.OP_ADD), ("mul", L.OP_MUL), ("div", L.OP_DIV))),
                            ("gf2^16_uint16", 2**16, (("mul", L.OP_MUL), ("div", L.OP_DIV)))):
        T = ga.GF(order)
        nt_ = 50_000_000
        rng = np.random.default_rng(4)
        ah, bh = rng.integers(0, order, nt_, dtype=np.uint16), rng.integers(1, order, nt_, dtype=np.uint16)
        at, bt = torch.from_numpy(ah.view(np.int16)).cuda(), torch.from_numpy(bh.view(np.int16)).cuda()
        ot = torch.empty_like(at)
        FT = O.OracleField(T.characteristic, T.degree, int(T.irreducible_poly), int(T.primitive_element), lookup=True)
        entry = {"elements": nt_, "mode": T.ufunc_mode}
        for name, op in ops:
            L.check(lib.gfa_time_binary(T._handle, op, at.data_ptr(), bt.data_ptr(), ot.data_ptr(), nt_, L.U16, stream, 10, ctypes.byref(ms)))
            want = getattr(FT, name)(ah[:200_000].astype(np.uint64), bh[:200_000].astype(np.uint64))
            assert np.array_equal(ot[:200_000].cpu().numpy().view(np.uint16).astype(np.uint64), want), f"{tag} {name} differs from the oracle"
            entry[name] = {"Gop/s": round(nt_ / (ms.value * 1e-3) / 1e9, 1), "roofline_frac": round(6.0 * nt_ / (ms.value * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        ex[tag] = entry
        del at, bt, ot
    # ---- NTT: 2^20 points over GF(7340033) (the modulus galois.ntt picks for that size), batch of 64 ----
    # kernel names / launch shapes behind every entry, ceilings and their sources: DESIGN.md section 5 (the JSON line carries numbers only)
    for tag, p, logn, batch in (("ntt_2^20_gf7340033", 7340033, 20, 64), ("ntt_2^20_gf469762049", 469762049, 20, 64),
                                ("ntt_2^20_gf2013265921", 2013265921, 20, 64),  # galois.ntt's default modulus from 2^27 points (lazy-Shoup kernels)
                                ("ntt_16x2^16_gf65537", 65537, 16, 16 * 64),
                                ("ntt_2^14_gf65537", 65537, 14, 4096),  # r06: four transforms per workgroup of the one-pass kernel
                                ("ntt_2^16_gf7340033", 7340033, 16, 1024), ("ntt_2^14_gf7340033", 7340033, 14, 4096)):
        P = ga.GF(p)
        N = 1 << logn
        omega = P._root_of_unity_int(N)
        xh = np.random.default_rng(3).integers(0, p, (batch, N), dtype=np.uint32)
        xd = torch.from_numpy(xh.view(np.int32)).cuda()
        od = torch.empty_like(xd)
        L.check(lib.gfa_time_ntt(P._handle, xd.data_ptr(), od.data_ptr(), N, batch, omega, L.U32, stream, 10, ctypes.byref(ms)))
        points = batch * N
        gbs = 8.0 * points / (ms.value * 1e-3) / 1e9
        entry = {"transforms_per_s": round(batch / (ms.value * 1e-3), 1), "ms_per_launch": round(ms.value, 4), "batch": batch,
                 "roofline_frac": round(gbs / HBM_PEAK_GBS, 4)}  # of 8 TB/s at the algorithmic 8 B/point
        if logn == 20:
            # what the memory system moves over both passes (PMC FETCH_SIZE x 2 + WRITE_SIZE, profiles/r04_pmc_ntt_m32_two_pass.txt; static)
            if p < (1 << 29):
                entry["physical_bytes_per_point"] = 17.0
                entry["physical_frac"] = round(17.0 * points / (ms.value * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            entry["ceiling_frac_static"] = 0.333
        elif p == 65537 and logn == 16:
            entry["ceiling_frac_static"] = 0.73
            # the same kernel on a batch far larger than the 256 MiB Infinity Cache (4096 transforms: 1 GiB in, 1 GiB out)
            big = 4096
            xb = torch.empty((big, N), dtype=torch.int32, device="cuda").random_(0, p)
            ob = torch.empty_like(xb)
            L.check(lib.gfa_time_ntt(P._handle, xb.data_ptr(), ob.data_ptr(), N, big, omega, L.U32, stream, 5, ctypes.byref(ms)))
            entry["hbm_only_batch_4096"] = {"ms_per_launch": round(ms.value, 4), "transforms_per_s": round(big / (ms.value * 1e-3), 1),
                                            "roofline_frac": round(8.0 * big * N / (ms.value * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            del xb, ob
        # parity of one transform against the oracle port
        FP = O.OracleField(p, 1, None, P._primitive_element_int)
        ref = FP.ntt_u32_pow2(xh[0], omega)
        assert np.array_equal(od[0].cpu().numpy().view(np.uint32), ref), "NTT differs from the oracle"
        if with_cpu:
            t1 = time.perf_counter()
            reps = 0
            while time.perf_counter() - t1 < 3.0:
                FP.ntt_u32_pow2(xh[reps % batch], omega)
                reps += 1
            dt = time.perf_counter() - t1
            entry["cpu_baseline"] = {"transforms_per_s": round(reps / dt, 2), "cores": 1}
            if tag in ("ntt_2^20_gf7340033", "ntt_16x2^16_gf65537"):  # (the all-cores leg on the two transforms BASELINE.json names)
                calls, dt_all, cores = _all_cores(lambda i: FP.ntt_u32_pow2(xh[i % batch], omega))
                entry["cpu_baseline_all_cores"] = {"transforms_per_s": round(calls / dt_all, 2), "cores": cores}
        ex[tag] = entry
        del xd, od
    # ---- ONE long transform (three passes of the register kernel) and a long polynomial product that needs the CRT route ----
    for tag, p, logn, dt, tdt in (("ntt_single_2^26_gf469762049", 469762049, 26, L.U32, torch.int32),
                                  ("ntt_single_2^27_gf2013265921", 2013265921, 27, L.U32, torch.int32),  # = galois.ntt(x) of 2^27 points
                                  ("ntt_single_2^26_goldilocks", 2**64 - 2**32 + 1, 26, L.U64, torch.int64)):
        P = ga.GF(p)
        N = 1 << logn
        xd = torch.empty(N, dtype=tdt, device="cuda").random_(0, min(p, 2**62))
        od = torch.empty_like(xd)
        L.check(lib.gfa_time_ntt(P._handle, xd.data_ptr(), od.data_ptr(), N, 1, P._root_of_unity_int(N), dt, stream, 5, ctypes.byref(ms)))
        width = 4 if dt == L.U32 else 8
        gbs = 2.0 * width * N / (ms.value * 1e-3) / 1e9
        ex[tag] = {"ms": round(ms.value, 4), "roofline_frac": round(gbs / HBM_PEAK_GBS, 4), "passes": 3}
        if dt == L.U32:
            if p == 469762049:
                ex[tag]["physical_bytes_per_point"] = 25.2  # profiles/r05_pmc_ntt_m32_three_pass.txt (static)
            # parity of the three-pass form in this very run: the inverse transform restores the input
            bk = torch.empty_like(xd)
            w = P._root_of_unity_int(N)
            L.check(lib.gfa_ntt(P._handle, xd.data_ptr(), od.data_ptr(), N, 1, w, 0, dt, stream))
            L.check(lib.gfa_ntt(P._handle, od.data_ptr(), bk.data_ptr(), N, 1, pow(w, p - 2, p), 1, dt, stream))
            assert torch.equal(bk, xd), "2^26-point round trip"
            del bk
        del xd, od
    M31 = ga.GF(2**31 - 1)
    ca = M31(np.random.default_rng(8).integers(0, 2**31 - 1, 1 << 20, dtype=np.int64))
    cb = M31(np.random.default_rng(9).integers(0, 2**31 - 1, 1 << 20, dtype=np.int64))
    cc = np.convolve(ca, cb)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        cc = np.convolve(ca, cb)
    torch.cuda.synchronize()
    conv_ms = (time.perf_counter() - t1) / 5 * 1e3
    assert int(cc[0]) == int(ca[0] * cb[0]) and int(cc[-1]) == int(ca[-1] * cb[-1])
    ex["convolve_2^20x2^20_gf2147483647"] = {"ms": round(conv_ms, 3), "multiply_adds_equivalent_per_s": round(2.0**40 / (conv_ms * 1e-3), 0),
                                            "route": "3-prime CRT"}
    del ca, cb, cc
    # ---- RS(255,223): 2^17 codewords per GPU (= 2^20 over 8 GPUs), e ~ U{0..16} errors per codeword ----
    rs = ga.ReedSolomon(255, 223)
    B = 1 << 17
    rng = np.random.default_rng(4)
    M = rng.integers(0, 256, (B, 223), dtype=np.uint8)
    Md = torch.from_numpy(M).cuda()
    Cd = torch.empty((B, 255), dtype=torch.uint8, device="cuda")
    L.check(lib.gfa_time_rs_encode(rs._handle, Md.data_ptr(), 223, Cd.data_ptr(), B, L.U8, stream, 20, ctypes.byref(ms)))
    enc_ms = ms.value
    C = Cd.cpu().numpy()
    rng5 = np.random.default_rng(5)
    R = C.copy()
    ne = rng5.integers(0, 17, B)
    for i in range(B):
        if ne[i]:
            pos = rng5.choice(255, ne[i], replace=False)
            R[i, pos] ^= rng5.integers(1, 256, ne[i], dtype=np.uint8)
    Rd = torch.from_numpy(R).cuda()
    Dd = torch.empty_like(Rd)
    Ed = torch.empty(B, dtype=torch.int64, device="cuda")
    L.check(lib.gfa_time_rs_decode(rs._handle, Rd.data_ptr(), 255, Dd.data_ptr(), Ed.data_ptr(), B, L.U8, stream, 20,
                                   ctypes.byref(ms)))
    dec_ms = ms.value
    assert np.array_equal(Dd.cpu().numpy(), C) and np.array_equal(Ed.cpu().numpy(), ne), "RS round trip failed"
    F = O.OracleField(2, 8, 285, 2, lookup=True)
    OR = O.OracleRS(F, 255, 223)
    assert np.array_equal(OR.encode_u8(M[:256]), C[:256]), "RS encode differs from the oracle"
    ex["rs_255_223"] = {
        "codewords": B, "errors_per_codeword": "uniform 0..16",
        "encode_GB/s": round(255.0 * B / (enc_ms * 1e-3) / 1e9, 2), "encode_ms": round(enc_ms, 4),
        "decode_GB/s": round(255.0 * B / (dec_ms * 1e-3) / 1e9, 2), "decode_ms": round(dec_ms, 4),
        "encode_roofline_frac": round(478.0 * B / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
        "decode_roofline_frac": round(486.0 * B / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
    }
    # the two extremes benchmarks/test_fec.py uses: no errors, and t = 16 errors in every codeword
    Rd.copy_(Cd)
    L.check(lib.gfa_time_rs_decode(rs._handle, Rd.data_ptr(), 255, Dd.data_ptr(), Ed.data_ptr(), B, L.U8, stream, 20,
                                   ctypes.byref(ms)))
    ex["rs_255_223"]["decode_GB/s_no_errors"] = round(255.0 * B / (ms.value * 1e-3) / 1e9, 2)
    R16 = C.copy()
    cols = np.argsort(rng5.random((B, 255)), axis=1)[:, :16]
    rows = np.arange(B)[:, None]
    R16[rows, cols] ^= rng5.integers(1, 256, (B, 16), dtype=np.uint8)
    Rd.copy_(torch.from_numpy(R16))
    L.check(lib.gfa_time_rs_decode(rs._handle, Rd.data_ptr(), 255, Dd.data_ptr(), Ed.data_ptr(), B, L.U8, stream, 20,
                                   ctypes.byref(ms)))
    assert np.array_equal(Dd.cpu().numpy(), C) and bool((Ed == 16).all()), "RS t=16 round trip failed"
    ex["rs_255_223"]["decode_GB/s_16_errors"] = round(255.0 * B / (ms.value * 1e-3) / 1e9, 2)
    # the whole C4 batch on ONE GPU (2^20 words: 267 MB per array, beyond the 256 MiB Infinity Cache): the HBM-only rates
    Bf = 1 << 20
    Mf = torch.empty((Bf, 223), dtype=torch.uint8, device="cuda").random_(0, 256)
    Cf = torch.empty((Bf, 255), dtype=torch.uint8, device="cuda")
    L.check(lib.gfa_time_rs_encode(rs._handle, Mf.data_ptr(), 223, Cf.data_ptr(), Bf, L.U8, stream, 5, ctypes.byref(ms)))
    encf_ms = ms.value
    Rf = Cf.clone()
    nef = torch.randint(0, 17, (Bf,), device="cuda")
    posf = torch.argsort(torch.rand((Bf, 255), device="cuda"), dim=1)[:, :16]
    valf = torch.randint(1, 256, (Bf, 16), device="cuda", dtype=torch.uint8)
    valf = torch.where(torch.arange(16, device="cuda")[None, :] < nef[:, None], valf, torch.zeros_like(valf))
    Rf.scatter_(1, posf, torch.gather(Rf, 1, posf) ^ valf)
    Df = torch.empty_like(Rf)
    Ef = torch.empty(Bf, dtype=torch.int64, device="cuda")
    L.check(lib.gfa_time_rs_decode(rs._handle, Rf.data_ptr(), 255, Df.data_ptr(), Ef.data_ptr(), Bf, L.U8, stream, 5, ctypes.byref(ms)))
    assert bool(torch.equal(Df, Cf)) and bool(torch.equal(Ef, nef)), "RS round trip failed at 2^20 codewords"
    ex["rs_255_223"]["hbm_only_2^20_codewords"] = {"encode_GB/s": round(255.0 * Bf / (encf_ms * 1e-3) / 1e9, 2),
                                                   "decode_GB/s": round(255.0 * Bf / (ms.value * 1e-3) / 1e9, 2),
                                                   "encode_ms": round(encf_ms, 4), "decode_ms": round(ms.value, 4)}
    del Mf, Cf, Rf, Df, Ef, posf, valf
    if with_cpu:
        t1 = time.perf_counter()
        OR.encode_u8(M[:2048])
        te = time.perf_counter() - t1
        t1 = time.perf_counter()
        OR.decode_u8(R[:2048])
        td = time.perf_counter() - t1
        ex["rs_255_223"]["cpu_baseline"] = {"encode_GB/s": round(255.0 * 2048 / te / 1e9, 5), "decode_GB/s": round(255.0 * 2048 / td / 1e9, 5),
                                            "cores": 1}
        ce, dte, cores = _all_cores(lambda i: OR.encode_u8(M[(i % 256) * 256:(i % 256) * 256 + 256]))
        cd_, dtd, _ = _all_cores(lambda i: OR.decode_u8(R[(i % 256) * 256:(i % 256) * 256 + 256]))
        ex["rs_255_223"]["cpu_baseline_all_cores"] = {"encode_GB/s": round(255.0 * 256 * ce / dte / 1e9, 4), "decode_GB/s": round(255.0 * 256 * cd_ / dtd / 1e9, 4),
                                                      "cores": cores}
    # ---- binary BCH(255, 223), t = 4: same entry points, symbols in GF(2), syndromes in GF(2^8) ----
    bch = ga.BCH(255, 223)
    Mb = rng.integers(0, 2, (B, 223), dtype=np.uint8)
    Mbd = torch.from_numpy(Mb).cuda()
    L.check(lib.gfa_time_rs_encode(bch._handle, Mbd.data_ptr(), 223, Cd.data_ptr(), B, L.U8, stream, 20, ctypes.byref(ms)))
    benc_ms = ms.value
    Cb = Cd.cpu().numpy()
    Rb = Cb.copy()
    neb = rng5.integers(0, 5, B)
    colsb = np.argsort(rng5.random((B, 255)), axis=1)[:, :4]
    flip = np.arange(4)[None, :] < neb[:, None]
    Rb[np.repeat(np.arange(B)[:, None], 4, axis=1)[flip], colsb[flip]] ^= 1
    Rd.copy_(torch.from_numpy(Rb))
    L.check(lib.gfa_time_rs_decode(bch._handle, Rd.data_ptr(), 255, Dd.data_ptr(), Ed.data_ptr(), B, L.U8, stream, 20,
                                   ctypes.byref(ms)))
    assert np.array_equal(Dd.cpu().numpy(), Cb) and np.array_equal(Ed.cpu().numpy(), neb), "BCH round trip failed"
    ex["bch_255_223"] = {"codewords": B, "errors_per_codeword": "uniform 0..4",
                         "encode_GB/s": round(255.0 * B / (benc_ms * 1e-3) / 1e9, 2),
                         "decode_GB/s": round(255.0 * B / (ms.value * 1e-3) / 1e9, 2),
                         "encode_ms": round(benc_ms, 4), "decode_ms": round(ms.value, 4)}
    if with_cpu:
        FB = O.OracleField(2, 8, 285, 2, lookup=True)
        OB = O.OracleBCH(FB, 255, 223)
        assert np.array_equal(OB.encode(Mb[:64]), Cb[:64]), "BCH encode differs from the oracle"
        t1 = time.perf_counter()
        od, on = OB.decode(Rb[:2048])
        td = time.perf_counter() - t1
        assert np.array_equal(od, Cb[:2048]) and np.array_equal(on, neb[:2048])
        ex["bch_255_223"]["cpu_baseline"] = {"decode_GB/s": round(255.0 * 2048 / td / 1e9, 5), "cores": 1}
    return ex


if __name__ == "__main__":
    main()
