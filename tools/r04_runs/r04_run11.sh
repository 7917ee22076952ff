set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_elementwise.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/test_ew.txt 2>&1
timeout 600 python tools/ew_bench.py > $O/ew_bench.txt 2>&1
