#!/bin/bash
# r05 GPU run 6: fuzz campaign of the round-5 paths (three seeds), short fuzz test
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
for s in 5 17 23; do ( timeout 200 python tools/fuzz_r05.py 90 $s 2>&1 | tail -4 ) ; done > gpurun_out/r05_fuzz.txt 2>&1
( timeout 300 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k r05 2>&1 | tail -5 ) >> gpurun_out/r05_fuzz.txt 2>&1
cat gpurun_out/r05_fuzz.txt
