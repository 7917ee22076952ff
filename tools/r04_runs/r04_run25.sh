set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_dist.py -x -q -m gpu --durations=5 > $O/ntt_tests.txt 2>&1
timeout 200 python tools/goldi_time.py > $O/goldi_time.txt 2>&1
timeout 600 bash tools/pmc_run.sh r04_pmc_ntt_goldilocks ntt_reg_kernel_gl -- python tools/goldi_time.py > $O/pmc.log 2>&1
tail -3 $O/ntt_tests.txt; cat $O/goldi_time.txt
