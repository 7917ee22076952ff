#!/bin/bash
# r05 GPU run 9: AUTO routing of extension-field products (digit tables, lazy product) -- parity and throughput
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_elementwise.py -q -m gpu -k "packed_digit or auto_mode_products" 2>&1 | tail -30 ) > gpurun_out/r05/run9_pytest.txt 2>&1
( timeout 400 python tools/ew_bench.py --packed 2>/dev/null | grep field ) > gpurun_out/r05_ew_packed_after.txt
tail -5 gpurun_out/r05/run9_pytest.txt; cat gpurun_out/r05_ew_packed_after.txt
