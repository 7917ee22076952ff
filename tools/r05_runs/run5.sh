#!/bin/bash
# r05 GPU run 5: tests fixed / added after run 4, wide-storage figures, the bench line with the driver's arguments and with the defaults
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_big.py tests/test_gpu_wide.py tests/test_gpu_elementwise.py -q -m gpu \
    -k "big or wide_storage or array_surface or k_limb or factory" 2>&1 | tail -80 ) > gpurun_out/r05/run5_pytest.txt 2>&1
( timeout 300 python tools/ew_bench.py --widestore16 2>/dev/null | grep field ) > gpurun_out/r05_ew_widestore16.txt
( timeout 400 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r05_bench_driver_args.json 2> gpurun_out/r05/run5_bench1.err
( timeout 400 python bench.py ) > gpurun_out/r05_bench_final.json 2> gpurun_out/r05/run5_bench2.err
tail -8 gpurun_out/r05/run5_pytest.txt; cat gpurun_out/r05_ew_widestore16.txt; head -c 300 gpurun_out/r05_bench_driver_args.json; echo; head -c 300 gpurun_out/r05_bench_final.json
