#!/bin/bash
# r05 GPU run 8: packed sums over the remaining fields; calculate-mode figures of the fields above 65536 elements (routing decision)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_elementwise.py -q -m gpu -k "packed_digit" 2>&1 | tail -8 ) > gpurun_out/r05/run8_pytest.txt 2>&1
( timeout 400 python tools/ew_bench.py --packed 2>/dev/null | grep field ) > gpurun_out/r05_ew_packed.txt
( timeout 400 python tools/ew_bench.py --extcalc 2>/dev/null | grep field ) > gpurun_out/r05_ew_extcalc.txt
tail -4 gpurun_out/r05/run8_pytest.txt; cat gpurun_out/r05_ew_packed.txt gpurun_out/r05_ew_extcalc.txt
