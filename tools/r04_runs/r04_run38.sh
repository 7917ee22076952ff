set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_elementwise.py tests/test_gpu_fuzz.py tests/test_gpu_poly.py tests/test_gpu_linalg.py -x -q -m gpu > $O/ew_tests5.txt 2>&1
timeout 300 python tools/ew_bench.py --ext 2>/dev/null | grep field > $O/ew_ext.txt
grep -n "passed\|failed" $O/ew_tests5.txt | tail -2; cat $O/ew_ext.txt
