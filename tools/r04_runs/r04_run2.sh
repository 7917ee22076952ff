set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
for cfg in "PSTAGGER=0" "PSTAGGER=6"; do
  echo "== $cfg" >> $O/pair_phases.txt
  env GFA_NTT_FERMAT_$cfg timeout 120 python tools/fermat_pair_phases.py 2048 >> $O/pair_phases.txt 2>&1
done
for cfg in "SKEL=0" "SKEL=1" "SKEL=2" "SKEL=0 GRID=256" "SKEL=1 GRID=256" "SKEL=2 GRID=256"; do
  envs=""; for kv in $cfg; do envs="$envs GFA_NTT_FERMAT_$kv"; done
  echo "== $cfg" >> $O/pair_skel.txt
  env $envs timeout 120 python tools/fermat_time.py 1024 4096 >> $O/pair_skel.txt 2>&1
done
