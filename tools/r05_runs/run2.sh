#!/bin/bash
# r05 GPU run 2: parity tests of the round (no -x), m32 tuning sweep, fresh PMC passes (RS decoder, three-pass NTT)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_ufunc_kwargs.py tests/test_gpu_wide.py tests/test_gpu_ntt.py tests/test_gpu_elementwise.py tests/test_gpu_poly.py tests/test_gpu_linalg.py -q -m gpu \
    -k "ufunc_kwargs or wide or signed_montgomery or 2e16_points or three_pass or 2e26_points_over_a_32 or ordering_and_editing or out_keyword or convolve or sage" 2>&1 | tail -150 ) > gpurun_out/r05/run2_pytest.txt 2>&1
( timeout 600 python tools/m32_tune3.py ) > gpurun_out/r05/run2_m32_tune3.txt 2>&1
( timeout 300 bash tools/pmc_run.sh r05_pmc_rs_decode rs_ -- python tools/rs_decode_only.py 5 ) > /dev/null 2>&1
( NTT_P=469762049 timeout 300 bash tools/pmc_run.sh r05_pmc_ntt_m32_three_pass ntt_m32 -- python tools/ntt_large.py 26 ) > /dev/null 2>&1
tail -12 gpurun_out/r05/run2_pytest.txt; grep -c . gpurun_out/r05/run2_m32_tune3.txt; tail -5 gpurun_out/r05_pmc_ntt_m32_three_pass.txt
