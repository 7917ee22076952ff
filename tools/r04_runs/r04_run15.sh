set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04o; mkdir -p $O
for v in 0 1; do echo "== GFA_RS_LFSR_REG=$v" >> $O/rs_time.txt; GFA_RS_LFSR_REG=$v timeout 200 python tools/rs_time.py 2>&1 | grep -v amdgpu >> $O/rs_time.txt; GFA_RS_LFSR_REG=$v timeout 300 python tools/rs_time_big.py 2>&1 | grep -v amdgpu >> $O/rs_time.txt; done
