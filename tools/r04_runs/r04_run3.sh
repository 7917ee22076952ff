set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
OBJ=galois_amd/_obj
for i in 1 2 3 4 5 6; do
  objs=$(ls $OBJ/*.o | grep -v gfa_ntt_fermat.o | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=galois_amd/csrc/exports.map -o _variants/lib_v$i.so $objs _variants/fermat_v$i.o
done
run() { echo "== $1" >> $O/fermat_sched.txt; shift; env "$@" timeout 120 python tools/fermat_time.py 1024 4096 2>&1 | grep batch >> $O/fermat_sched.txt; }
for rep in 1 2; do
run "base SCHED=0" GFA_NTT_FERMAT_SCHED=0
for i in 1 2 3 4 5 6; do run "v$i" GALOIS_AMD_LIB=$PWD/_variants/lib_v$i.so; done
done
timeout 120 python -m pytest tests/test_gpu_ntt.py -x -q -m gpu -k fermat > $O/test_sched1.txt 2>&1
GALOIS_AMD_LIB=$PWD/_variants/lib_v3.so timeout 120 python -m pytest tests/test_gpu_ntt.py -x -q -m gpu -k fermat >> $O/test_sched1.txt 2>&1
timeout 120 python tools/fermat_phases.py 1024 > $O/phases_v1.txt 2>&1
GFA_NTT_FERMAT_SCHED=0 timeout 120 python tools/fermat_phases.py 1024 > $O/phases_base.txt 2>&1
timeout 600 bash tools/pmc_run.sh r04c_pmc_m32_one ntt_m32_one -- python tools/ntt_mid_time.py > $O/pmc_m32_one.log 2>&1
