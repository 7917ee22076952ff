cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ntt.py -x -q 2>&1 | tail -3
timeout 300 python tools/m32_time.py 2>&1 | grep "p=" > gpurun_out/r03_m32_time.txt
cat gpurun_out/r03_m32_time.txt
bash tools/pmc_run.sh r03_pmc_ntt_2e20x64 ntt_m32_kernel -- python tools/m32_time.py 1 > /dev/null 2>&1
cat gpurun_out/r03_pmc_ntt_2e20x64.txt
