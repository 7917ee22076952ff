set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ntt.py -x -q -m gpu --durations=5 > $O/ntt_tests2.txt 2>&1
timeout 200 python tools/goldi_time.py 2>&1 | grep -v amdgpu > $O/goldi_time3.txt
timeout 600 python -m pytest tests/test_gpu_elementwise.py tests/test_gpu_fuzz.py tests/test_gpu_linalg.py tests/test_gpu_poly.py -x -q -m gpu > $O/ew_tests2.txt 2>&1
timeout 300 python tools/ew_bench.py 2>/dev/null | grep 18446744069414584321 > $O/ew_goldi.txt
tail -3 $O/ntt_tests2.txt; cat $O/goldi_time3.txt; tail -3 $O/ew_tests2.txt; cat $O/ew_goldi.txt
