#!/bin/bash
# r05 GPU run 10: the whole -m gpu suite + smoke as the driver runs them, then the bench lines and the rocprof statistics of the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
( timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -60 ) > gpurun_out/r05/run10_pytest.txt 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/r05/run10_smoke.txt 2>&1
( timeout 400 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r05_bench_driver_args.json 2> gpurun_out/r05/run10_bench1.err
( timeout 400 python bench.py ) > gpurun_out/r05_bench_final.json 2> gpurun_out/r05/run10_bench2.err
( cd /tmp && rm -rf /tmp/prof_r05 && timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_r05 -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-pmc > /dev/null 2>&1
  DB=$(find /tmp/prof_r05 -name "*.db" | head -1); [ -n "$DB" ] && python "$GRAFT_REPO_ROOT/tools/export_rocprof_stats.py" "$DB" "$GRAFT_REPO_ROOT/gpurun_out/r05_bench_kernel_stats.csv" )
tail -6 gpurun_out/r05/run10_pytest.txt; cat gpurun_out/r05/run10_smoke.txt; head -c 250 gpurun_out/r05_bench_driver_args.json; echo
