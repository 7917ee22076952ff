set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04r; mkdir -p $O
objs=$(ls galois_amd/_obj/*.o | grep -v gfa_ntt_fermat.o | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=galois_amd/csrc/exports.map -o _variants/lib_head.so $objs _variants/fermat_head.o
timeout 200 python -m pytest tests/test_gpu_ntt.py -x -q -m gpu -k fermat > $O/test.txt 2>&1
for rep in 1 2 3; do
echo "== r03 kernel" >> $O/fermat_tw.txt; GALOIS_AMD_LIB=$PWD/_variants/lib_head.so timeout 100 python tools/fermat_time.py 1024 4096 2>&1 | grep batch >> $O/fermat_tw.txt
echo "== twiddle window requested ahead of the last loads" >> $O/fermat_tw.txt; timeout 100 python tools/fermat_time.py 1024 4096 2>&1 | grep batch >> $O/fermat_tw.txt
done
