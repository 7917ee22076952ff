#!/bin/bash
# r05 GPU run 15: from_packed written out (no run-time loop) -- parity of the packed kernels and throughput
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_elementwise.py -q -m gpu -k "packed_digit or auto_mode_products or sage or table" 2>&1 | tail -5 ) > gpurun_out/r05/run15_pytest.txt 2>&1
( timeout 200 python tools/fuzz_fields.py 45 1111 2>&1 | grep -v amdgpu | tail -1 ) >> gpurun_out/r05/run15_pytest.txt 2>&1
( timeout 400 python tools/ew_bench.py --packed 2>/dev/null | grep field ) > gpurun_out/r05_ew_packed_v2.txt
( timeout 400 python tools/ew_bench.py --ext 2>/dev/null | grep field ) >> gpurun_out/r05_ew_packed_v2.txt
tail -4 gpurun_out/r05/run15_pytest.txt; cut -c1-210 gpurun_out/r05_ew_packed_v2.txt
