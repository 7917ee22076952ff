set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_elementwise.py tests/test_gpu_fuzz.py -x -q -m gpu --durations=5 > $O/ew_tests3.txt 2>&1
timeout 400 python tools/ew_bench.py --calc 2>/dev/null | grep field > $O/ew_calc.txt
grep -n "passed\|failed\|Error" $O/ew_tests3.txt | head; cat $O/ew_calc.txt
