"""`python bench.py --gpus N` must start N ranks by itself (the driver calls it without a launcher around it): the command it
re-executes under, the refusal when fewer GPUs than ranks are visible, and -- on the one-GPU box -- a real two-rank plumbing run
(both ranks on the one device, gloo for the timing collectives) whose JSON line says n_gpus = 2."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def test_gpus_n_without_a_launcher_re_executes_under_torch_distributed_run():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "7", "--warmup", "3", "--print-launch"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    cmd = r.stdout.strip().splitlines()[-1].split()
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "3"]


def test_more_ranks_than_visible_gpus_is_refused_not_faked():
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GFA_BENCH_SINGLE_DEVICE")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "only" in r.stderr and not r.stdout.strip()


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_report_n_gpus_2():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(GFA_BENCH_SINGLE_DEVICE="0", GFA_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-extras", "--no-pmc",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["steps"] == 5 and d["value"] > 0
