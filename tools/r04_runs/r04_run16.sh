set -u
cd $GRAFT_REPO_ROOT
timeout 600 bash tools/pmc_run.sh r04_pmc_rs_lfsr_reg rs_lfsr_reg -- python tools/rs_time_big.py > gpurun_out/r04p_pmc.log 2>&1
