set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m; mkdir -p $O
for i in 1 2 3 4 5; do
  objs=$(ls galois_amd/_obj/*.o | grep -v gfa_ntt_m32.o | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=galois_amd/csrc/exports.map -o _variants/lib_aux$i.so $objs _variants/m32_aux$i.o
done
for rep in 1 2; do
echo "== default (no hints)" >> $O/m32_aux.txt; timeout 100 python tools/m32_time.py 3 2>&1 | grep 'p=' >> $O/m32_aux.txt
for i in 1 2 3 4 5; do echo "== aux$i" >> $O/m32_aux.txt; GALOIS_AMD_LIB=$PWD/_variants/lib_aux$i.so timeout 100 python tools/m32_time.py 3 2>&1 | grep 'p=' >> $O/m32_aux.txt; done
done
timeout 400 bash tools/pmc_run.sh r04_pmc_ntt_m32_2e16 ntt_m32_2e16 -- python tools/ntt_mid_time.py 16 > $O/pmc1.log 2>&1
timeout 400 bash tools/pmc_run.sh r04_pmc_ntt_fermat ntt_fermat16 -- python tools/fermat_time.py 1024 > $O/pmc2.log 2>&1
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_r04 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_r04 -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-pmc > /dev/null 2>&1
  DB=$(find /tmp/prof_r04 -name "*.db" | head -1); [ -n "$DB" ] && python "$GRAFT_REPO_ROOT/tools/export_rocprof_stats.py" "$DB" "$GRAFT_REPO_ROOT/gpurun_out/r04_bench_kernel_stats.csv" )
