set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rs.py tests/test_gpu_bch.py -x -q -m gpu -k "not stress" > $O/test_rs.txt 2>&1
timeout 300 python tools/fuzz_r04.py 30 7 > $O/fuzz.txt 2>&1
for v in 0 1; do echo "== GFA_RS_LFSR_REG=$v" >> $O/rs_time.txt; GFA_RS_LFSR_REG=$v timeout 200 python tools/rs_time.py 2>&1 | grep -v amdgpu >> $O/rs_time.txt; GFA_RS_LFSR_REG=$v timeout 300 python tools/rs_time_big.py 2>&1 | grep -v amdgpu >> $O/rs_time.txt; done
timeout 600 python -m pytest tests/test_gpu_rs.py -x -q -m gpu -k "stress and 8" > $O/test_stress.txt 2>&1
