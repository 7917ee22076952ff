#!/bin/bash
# r05 GPU run 13: packed kernels for pinned-calculate fields of any order (uint8 / uint16 / uint32) -- parity and throughput
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_elementwise.py tests/test_gpu_linalg.py tests/test_gpu_poly.py tests/test_gpu_rs.py tests/test_gpu_bch.py -q -m gpu 2>&1 | tail -12 ) > gpurun_out/r05/run13_pytest.txt 2>&1
( timeout 200 python tools/fuzz_fields.py 60 1010 2>&1 | grep -v amdgpu | tail -2 ) >> gpurun_out/r05/run13_pytest.txt 2>&1
( timeout 400 python tools/ew_bench.py --ext 2>/dev/null | grep field ) > gpurun_out/r05_ew_ext.txt
tail -6 gpurun_out/r05/run13_pytest.txt; cut -c1-220 gpurun_out/r05_ew_ext.txt
