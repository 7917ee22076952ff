#!/bin/bash
# r05 GPU run 7: packed-digit sums -- parity tests and throughput
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_elementwise.py -q -m gpu -k "packed_digit or table or sage" 2>&1 | tail -40 ) > gpurun_out/r05/run7_pytest.txt 2>&1
( timeout 400 python tools/ew_bench.py --packed 2>/dev/null | grep field ) > gpurun_out/r05_ew_packed.txt
tail -6 gpurun_out/r05/run7_pytest.txt; cat gpurun_out/r05_ew_packed.txt
