set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ntt.py -x -q -m gpu > $O/ntt_tests3.txt 2>&1
for r in 1 2; do timeout 200 python tools/goldi_time.py 2>&1 | grep -v amdgpu >> $O/goldi_time4.txt; done
timeout 600 bash tools/pmc_run.sh r04_pmc_ntt_goldilocks ntt_reg_kernel_gl -- python tools/goldi_time.py > $O/pmc.log 2>&1
grep -n "passed\|failed" $O/ntt_tests3.txt; cat $O/goldi_time4.txt
