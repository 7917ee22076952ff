set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ntt.py -x -q -m gpu -k fermat > $O/test_pair.txt 2>&1; echo "test rc=$?" >> $O/test_pair.txt
for cfg in "PAIR=0" "PAIR=1 PSTAGGER=0" "PAIR=1 PSTAGGER=3" "PAIR=1 PSTAGGER=6" "PAIR=1 PSTAGGER=10"; do
  envs=""; for kv in $cfg; do envs="$envs GFA_NTT_FERMAT_$kv"; done
  for rep in 1 2; do
    echo "== $cfg rep $rep" >> $O/fermat_time.txt
    env $envs timeout 120 python tools/fermat_time.py 256 512 1024 2048 4096 >> $O/fermat_time.txt 2>&1
  done
done
timeout 600 bash tools/pmc_run.sh r04a/pmc_m32_one ntt_m32_one -- python tools/ntt_mid_time.py > $O/pmc_m32_one.log 2>&1
