set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04w; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rs.py tests/test_gpu_bch.py -x -q -m gpu -k "not stress" > $O/test_rs.txt 2>&1
timeout 300 python tools/fuzz_r04.py 20 8 > $O/fuzz.txt 2>&1
timeout 200 python tools/rs_time.py 2>&1 | grep -v amdgpu >> $O/rs_time.txt; timeout 300 python tools/rs_time_big.py 2>&1 | grep -v amdgpu >> $O/rs_time.txt
