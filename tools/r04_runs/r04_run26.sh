set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ntt.py -x -q -m gpu --durations=5 > $O/ntt_tests.txt 2>&1
GALOIS_AMD_LIB=$PWD/_variants/lib_nonop.so timeout 900 python -m pytest tests/test_gpu_ntt.py -x -q -m gpu -k "goldilocks or Goldilocks or 2e26 or c5 or C5" > $O/ntt_tests_nonop.txt 2>&1
for r in 1 2; do
echo "== main" >> $O/goldi_time2.txt; timeout 200 python tools/goldi_time.py 2>&1 | grep -v amdgpu >> $O/goldi_time2.txt
echo "== nonop" >> $O/goldi_time2.txt; GALOIS_AMD_LIB=$PWD/_variants/lib_nonop.so timeout 200 python tools/goldi_time.py 2>&1 | grep -v amdgpu >> $O/goldi_time2.txt
done
tail -3 $O/ntt_tests.txt; tail -3 $O/ntt_tests_nonop.txt; cat $O/goldi_time2.txt
