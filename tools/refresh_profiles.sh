#!/bin/bash
# Regenerates the text profiles under profiles/ in ONE call on the GPU box (about 2 GPU-minutes):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r02'
# then copy gpurun_out/<round>_*.txt / .json / .csv into profiles/.  PMC passes are separate.
set -u
R=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT"
python bench.py > "$OUT/${R}_bench_final.json" 2> "$OUT/${R}_bench_final.err"
python tools/c5_local_bench.py > "$OUT/${R}_c5_goldilocks_local.txt" 2>/dev/null
python tools/ntt_large.py 20 21 22 24 26 28 > "$OUT/${R}_ntt_large.txt" 2>/dev/null
python tools/ntt3_tune.py > "$OUT/${R}_ntt3_tune.txt" 2>/dev/null
python tools/wide_codes_bench.py > "$OUT/${R}_wide_codes_bench.txt" 2>/dev/null
python tools/linalg_bench.py > "$OUT/${R}_linalg_bench.txt" 2>/dev/null
python tools/headline_sizes.py > "$OUT/${R}_headline_sizes.txt" 2>/dev/null
{ echo "# GFA_CONVOLVE_CRT=0 (direct kernel)"; GFA_CONVOLVE_CRT=0 python tools/convolve_bench.py 2>/dev/null
  echo "# GFA_CONVOLVE_CRT_MIN=0 (CRT route wherever it applies)"; GFA_CONVOLVE_CRT_MIN=0 python tools/convolve_bench.py 256 1024 4096 16384 65536 1048576 2>/dev/null; } > "$OUT/${R}_convolve_bench.txt"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$R && rocprofv3 --kernel-trace --stats -d /tmp/prof_$R -o bench -- python "$ROOT/bench.py" --no-cpu-baseline > /dev/null 2>&1
  DB=$(find /tmp/prof_$R -name "*.db" | head -1); [ -n "$DB" ] && python "$ROOT/tools/export_rocprof_stats.py" "$DB" "$OUT/${R}_bench_kernel_stats.csv" )
ls -la "$OUT" | tail -12
