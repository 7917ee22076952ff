"""Tests that need MORE THAN ONE GPU: they skip on a one-GPU box (every gpurun box so far) and run the day two devices are
visible -- the RCCL paths proper (torch.distributed over RCCL, the grouped ncclSend / ncclRecv halves inside gfa_ntt_dist, both
GFA_DIST_NSUB settings, bench.py --gpus 2).  The same code runs at world size 1 in tests/test_gpu_ntt.py and over gloo with
oracle stand-ins in tests/test_dist_gloo.py; what only these tests see is two ranks exchanging data over xGMI."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpus() -> int:
    import torch

    return torch.cuda.device_count()


def _launch(script_args, world, port, extra_env=None, timeout=900):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_rccl_selfcheck_on_one_gpu():
    """tools/rccl_selfcheck.py: the collective library loads and runs the calls bench.py and the distributed transform make
    (world size 1; needs no second GPU, so this one always runs)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_selfcheck.py")], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl self-check ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_world_check_script_at_world_size_one():
    """The worker of the two-GPU tests below, run with ONE rank (always possible): keeps the script itself -- the unique-id
    hand-off, the raw communicator, the layout bookkeeping of its comparisons -- from rotting between multi-GPU runs."""
    r = _launch([os.path.join(ROOT, "tools", "dist_world_check.py")], 1, 29570)
    assert r.returncode == 0 and "dist world check ok 1" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_world_check_script_with_eight_ranks_on_one_gpu_over_gloo():
    """r05: EIGHT processes, all on cuda:0, exchange over gloo through the host -- the 8-rank layout
    bookkeeping of galois_amd.dist (column slices, the all-to-all's [peer][row][column] chunks, the row pass reading them in
    place, the inverse consuming that layout) driven by real processes and the REAL kernels (gfa_ntt_columns_pitched,
    gfa_ntt_chunked, ...), every rank's block against the whole transform; C5's own shape (2^10 x 2^16 over GF(469762049)) and
    Goldilocks included.  What stays for real devices: the grouped ncclSend / ncclRecv halves inside gfa_ntt_dist."""
    r = _launch([os.path.join(ROOT, "tools", "dist_world_check.py")], 8, 29590, {"GFA_DIST_CHECK_ONE_GPU": "1"}, timeout=1200)
    assert r.returncode == 0 and "dist world check ok 8" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("nsub", ["2", "1"])
def test_distributed_transform_over_two_gpus(nsub):
    """tools/dist_world_check.py at world size 2: torch.distributed collectives over RCCL, the four-step transform and its
    inverse through galois_amd.dist, and gfa_ntt_dist / gfa_intt_dist with a raw RCCL communicator -- the overlapped
    two-half exchange (GFA_DIST_NSUB=2) and the single ncclAllToAll (1), every rank's block against the whole transform."""
    if _gpus() < 2:
        pytest.skip("needs two GPUs")
    r = _launch([os.path.join(ROOT, "tools", "dist_world_check.py")], 2, 29571 + int(nsub), {"GFA_DIST_NSUB": nsub})
    assert r.returncode == 0 and "dist world check ok 2" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_bench_over_two_gpus():
    """python bench.py --gpus 2 over RCCL: one JSON line, n_gpus = 2, both ranks' work in the aggregate, the C4 / C5 extras
    present with their per-stage times."""
    if _gpus() < 2:
        pytest.skip("needs two GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2"], cwd=ROOT,
                       capture_output=True, text=True, timeout=1500, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["value"] > 0 and rec["scaling"] == "weak"
