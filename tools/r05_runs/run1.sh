#!/bin/bash
# r05 GPU run 1: new parity tests (where= / initial=, primes up to 2^29 on the m32 kernels, three-pass), packed-intermediate ubench, bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_ufunc_kwargs.py tests/test_gpu_ntt.py tests/test_gpu_elementwise.py tests/test_gpu_poly.py -x -q -m gpu \
    -k "ufunc_kwargs or signed_montgomery or 2e16_points or three_pass or 2e26_points_over_a_32 or ordering_and_editing or out_keyword or convolve or power" 2>&1 | tail -25 ) > gpurun_out/r05/run1_pytest.txt 2>&1
( timeout 120 tools/ubench/ntt_packed 64 ) > gpurun_out/r05/run1_ntt_packed.txt 2>&1
( NTT_P=469762049 timeout 200 python tools/ntt_large.py 20 21 22 24 26 ) > gpurun_out/r05/run1_ntt_large.txt 2>&1
( timeout 300 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r05/run1_bench.json 2> gpurun_out/r05/run1_bench.err
tail -5 gpurun_out/r05/run1_pytest.txt; cat gpurun_out/r05/run1_ntt_packed.txt gpurun_out/r05/run1_ntt_large.txt; tail -c 600 gpurun_out/r05/run1_bench.err
