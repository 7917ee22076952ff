set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
OBJ=galois_amd/_obj
for i in 0 1 2 3 4; do
  objs=$(ls $OBJ/*.o | grep -v gfa_ntt_fermat.o | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=galois_amd/csrc/exports.map -o _variants/lib_v$i.so $objs _variants/fermat_v$i.o
done
run() { echo "== $1" >> $O/fermat_s2.txt; shift; env "$@" timeout 120 python tools/fermat_time.py 1024 4096 2>&1 | grep batch >> $O/fermat_s2.txt; }
for rep in 1 2 3; do
for i in 0 1 2 3 4; do run "v$i" GALOIS_AMD_LIB=$PWD/_variants/lib_v$i.so; done
done
for i in 1 3; do GALOIS_AMD_LIB=$PWD/_variants/lib_v$i.so timeout 120 python -m pytest tests/test_gpu_ntt.py -x -q -m gpu -k fermat >> $O/test_s2.txt 2>&1; done
GALOIS_AMD_LIB=$PWD/_variants/lib_v1.so timeout 120 python tools/fermat_phases.py 1024 > $O/phases_s2.txt 2>&1
