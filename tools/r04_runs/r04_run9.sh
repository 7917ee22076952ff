set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
for rep in 1 2; do
for nib in 0 1; do echo "== NIB=$nib" >> $O/rs_time.txt; GFA_RS_LFSR_NIB=$nib timeout 200 python tools/rs_time.py 2>&1 | grep -v amdgpu >> $O/rs_time.txt; done
done
timeout 900 python -m pytest tests/test_gpu_rs.py tests/test_gpu_bch.py -x -q -m gpu -k "not stress" > $O/test_rs.txt 2>&1
