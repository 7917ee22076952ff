set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O/r04w
timeout 1800 python -m pytest tests -x -q -m gpu --durations=6 > $O/r04w/gpu_tests.txt 2>&1
timeout 600 python bench.py > $O/r04_bench_final.json 2> $O/r04w/bench.err
timeout 400 python tools/ew_bench.py 2>/dev/null | grep field > $O/r04_ew_bench.txt
timeout 200 python tools/rs_time.py 2>/dev/null | tail -1 > $O/r04_rs_time.txt; timeout 300 python tools/rs_time_big.py 2>/dev/null | grep words >> $O/r04_rs_time.txt
timeout 200 python tools/m32_time.py 2>/dev/null | grep "p=" > $O/r04_m32_time.txt
timeout 200 python tools/ntt_mid_time.py 2>/dev/null | grep "p=" > $O/r04_ntt_mid_sizes.txt
timeout 200 python tools/fermat_time.py 64 256 1024 4096 2>/dev/null | grep batch > $O/r04_ntt_fermat_batches.txt
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_r04 && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_r04 -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-pmc > /dev/null 2>&1
  DB=$(find /tmp/prof_r04 -name "*.db" | head -1); [ -n "$DB" ] && python "$GRAFT_REPO_ROOT/tools/export_rocprof_stats.py" "$DB" "$GRAFT_REPO_ROOT/gpurun_out/r04_bench_kernel_stats.csv" )
timeout 200 python tools/goldi_time.py 2>&1 | grep "2^" > $O/r04_goldi_time.txt; tail -3 $O/r04w/gpu_tests.txt; cat $O/r04_bench_final.json
