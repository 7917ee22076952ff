#!/bin/bash
# r05 GPU run 3: the tests fixed after run 2, the 8-rank one-GPU check, a bench line, rocprof kernel statistics keyed by (kernel, grid)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_wide.py tests/test_gpu_elementwise.py tests/test_gpu_multi.py tests/test_gpu_ntt.py -q -m gpu \
    -k "random_and_edge or ordering_and_editing or multi or three_pass or world or rccl" 2>&1 | tail -60 ) > gpurun_out/r05/run3_pytest.txt 2>&1
( timeout 400 python bench.py ) > gpurun_out/r05/run3_bench.json 2> gpurun_out/r05/run3_bench.err
( cd /tmp && rm -rf /tmp/prof_r05 && timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_r05 -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-pmc > /dev/null 2>&1
  DB=$(find /tmp/prof_r05 -name "*.db" | head -1); [ -n "$DB" ] && python "$GRAFT_REPO_ROOT/tools/export_rocprof_stats.py" "$DB" "$GRAFT_REPO_ROOT/gpurun_out/r05_bench_kernel_stats.csv" )
tail -8 gpurun_out/r05/run3_pytest.txt; tail -c 400 gpurun_out/r05/run3_bench.err; head -c 600 gpurun_out/r05/run3_bench.json; echo; head -8 gpurun_out/r05_bench_kernel_stats.csv
