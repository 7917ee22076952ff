"""Builds galois_amd/libgalois_amd.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

    python galois_amd/build.py [--force]

hipcc cross-compiles without a GPU, so this also runs on the CPU-only build container.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libgalois_amd.so")
SOURCES = ["gfa_field.hip", "gfa_elementwise.hip", "gfa_elementwise_mid.hip", "gfa_elementwise_packed.hip", "gfa_ntt.hip", "gfa_ntt_m32.hip", "gfa_ntt_fermat.hip", "gfa_dist.hip", "gfa_wide.hip", "gfa_big.hip", "gfa_rs.hip", "gfa_linalg.hip", "gfa_matmul_mfma.hip", "gfa_dlog.hip", "gfa_conv_crt.hip", "gfa_rs_wide.hip"]
import glob
HEADERS = sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc"))) + [
    os.path.join(HERE, "..", "include", "galois_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-result"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _deps(o: str, s: str) -> list[str]:
    """Headers the object was built from (the compiler's own -MMD record); every header when there is no record yet."""
    d = o[:-2] + ".d"
    if not os.path.exists(d):
        return [s] + HEADERS
    txt = open(d).read().replace("\\\n", " ")
    return [s] + [t for t in txt.split(":", 1)[-1].split() if os.path.exists(t)]


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, _deps(o, s)):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + ["-MMD", "-MF", o[:-2] + ".d", "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs + [os.path.join(CSRC, "exports.map")]):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
