cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.txt 2>&1
tail -5 gpurun_out/gpu_tests.txt
timeout 600 python tools/rs_time.py > gpurun_out/rs_time.txt 2>&1; tail -12 gpurun_out/rs_time.txt
