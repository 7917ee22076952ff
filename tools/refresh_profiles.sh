#!/bin/bash
# Regenerates the text profiles under profiles/ in ONE call on the GPU box (about 5 GPU-minutes):
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/refresh_profiles.sh r04'
# then copy gpurun_out/<round>_*.txt / .json / .csv into profiles/.
set -u
R=${1:-r06}
ONLY=${2:-all} # "codes": only the files the Reed-Solomon / BCH and Goldilocks kernels feed (about 2 GPU-minutes)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT"
if [ "$ONLY" = codes ]; then
    python bench.py > "$OUT/${R}_bench_final.json" 2> "$OUT/${R}_bench_final.err"
    python tools/goldi_time.py 2>/dev/null | grep "2^" > "$OUT/${R}_ntt_goldilocks.txt"
    python tools/rs_time.py 2>/dev/null | tail -1 > "$OUT/${R}_rs_time.txt"
    python tools/ntt_large.py 20 21 22 24 26 28 > "$OUT/${R}_ntt_large.txt" 2>/dev/null
    ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$R && rocprofv3 --kernel-trace --stats -d /tmp/prof_$R -o bench -- python "$ROOT/bench.py" --no-cpu-baseline --no-pmc > /dev/null 2>&1
      DB=$(find /tmp/prof_$R -name "*.db" | head -1); [ -n "$DB" ] && python "$ROOT/tools/export_rocprof_stats.py" "$DB" "$OUT/${R}_bench_kernel_stats.csv" )
    bash tools/pmc_run.sh ${R}_pmc_ntt_goldilocks ntt_reg_kernel_gl -- python tools/goldi_time.py > /dev/null 2>&1
    bash tools/pmc_run.sh ${R}_pmc_rs_decode rs_ -- python tools/rs_decode_only.py 5 > /dev/null 2>&1
    ls -la "$OUT" | tail -12
    exit 0
fi
python bench.py > "$OUT/${R}_bench_final.json" 2> "$OUT/${R}_bench_final.err"
python bench.py --steps 50 --warmup 5 --dist-extras --no-cpu-baseline --no-pmc > "$OUT/${R}_bench_dist_world1.json" 2>/dev/null
python tools/ew_bench.py 2>/dev/null | grep field > "$OUT/${R}_ew_bench.txt"
python tools/fermat_time.py 64 256 1024 4096 2>/dev/null | grep batch > "$OUT/${R}_ntt_fermat_batches.txt"
python tools/fermat_phases.py 1024 2>/dev/null | grep -E "round|phase|spread" > "$OUT/${R}_ntt_fermat_phases.txt"
python tools/goldi_time.py 2>/dev/null | grep "2^" > "$OUT/${R}_ntt_goldilocks.txt"
python tools/ntt_time.py 2>/dev/null | grep "p=" > "$OUT/${R}_ntt_time.txt"
python tools/m32_time.py 2>/dev/null | grep "p=" > "$OUT/${R}_m32_time.txt"
python tools/ew_bench.py --widestore 2>/dev/null | grep field > "$OUT/${R}_ew_widestore.txt"
python tools/ew_bench.py --ext 2>/dev/null | grep field > "$OUT/${R}_ew_ext_calculate.txt"
python tools/ntt_mid_time.py 2>/dev/null | grep "p=" > "$OUT/${R}_ntt_mid_sizes.txt"
./tools/ubench/ntt_access 64 > "$OUT/${R}_ntt_access_skeleton.txt" 2>/dev/null
./tools/ubench/ntt_fused_skel 64 > "$OUT/${R}_ntt_fused_skeleton.txt" 2>/dev/null
./tools/ubench/mfma_dft16 > "$OUT/${R}_mfma_dft16.txt" 2>/dev/null
python tools/rs_time.py 2>/dev/null | tail -1 > "$OUT/${R}_rs_time.txt"
python tools/c5_local_bench.py > "$OUT/${R}_c5_goldilocks_local.txt" 2>/dev/null
python tools/ntt_large.py 20 21 22 24 26 28 > "$OUT/${R}_ntt_large.txt" 2>/dev/null
./tools/ubench/valu_rates2 > "$OUT/${R}_valu_issue_rates.txt" 2>/dev/null
python tools/fft_small_bench.py 2>/dev/null | grep size > "$OUT/${R}_fft_reference_benchmark_sizes.txt"
python tools/wide_codes_bench.py > "$OUT/${R}_wide_codes_bench.txt" 2>/dev/null
python tools/linalg_bench.py > "$OUT/${R}_linalg_bench.txt" 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$R && rocprofv3 --kernel-trace --stats -d /tmp/prof_$R -o bench -- python "$ROOT/bench.py" --no-cpu-baseline --no-pmc > /dev/null 2>&1
  DB=$(find /tmp/prof_$R -name "*.db" | head -1); [ -n "$DB" ] && python "$ROOT/tools/export_rocprof_stats.py" "$DB" "$OUT/${R}_bench_kernel_stats.csv" )
bash tools/pmc_run.sh ${R}_pmc_headline tab8_binary -- python tools/headline_only.py 6 > /dev/null 2>&1
bash tools/pmc_run.sh ${R}_pmc_ntt_fermat ntt_fermat16 -- python tools/fermat_time.py 1024 > /dev/null 2>&1
bash tools/pmc_run.sh ${R}_pmc_ntt_2e20x64 ntt_m32_kernel -- python tools/m32_time.py 1 > /dev/null 2>&1
bash tools/pmc_run.sh ${R}_pmc_ntt_m32_one ntt_m32_one -- python tools/ntt_mid_time.py > /dev/null 2>&1
bash tools/pmc_run.sh ${R}_pmc_ntt_m32_2e16 ntt_m32_2e16 -- python tools/ntt_mid_time.py 16 > /dev/null 2>&1
bash tools/pmc_run.sh ${R}_pmc_ntt_goldilocks ntt_reg_kernel_gl -- python tools/goldi_time.py > /dev/null 2>&1
bash tools/pmc_run.sh ${R}_pmc_rs_decode rs_ -- python tools/rs_decode_only.py 5 > /dev/null 2>&1
ls -la "$OUT" | tail -30
# r05 additions: packed-digit sums / digit-table products of the extension fields above the LDS table sizes, wide storage of the 2^16 band,
# the three-pass / last-pass tuning sweep, the packed-intermediate skeletons
python tools/ew_bench.py --packed 2>/dev/null | grep field > "$OUT/${R}_ew_packed.txt"
python tools/ew_bench.py --widestore16 2>/dev/null | grep field > "$OUT/${R}_ew_widestore16.txt"
python tools/m32_tune3.py > "$OUT/${R}_m32_tune3.txt" 2>/dev/null
./tools/ubench/ntt_packed 64 > "$OUT/${R}_ntt_packed_intermediate.txt" 2>/dev/null
# r06 additions: GF(65537) transforms of 2^10 .. 2^16 points (grouped one-pass kernel), two-word packed sums, GF(p^2) quotients by the norm,
# the 2^15 .. 2^16-element band, strided-piece skeleton, the differential fuzzer
python tools/fermat_sizes_time.py 2>/dev/null | tail -1 > "$OUT/${R}_fermat_sizes.txt"
python tools/fermat_grouped_check.py 2>/dev/null | grep -E "ok|p=" > "$OUT/${R}_fermat_grouped_sizes.txt"
python tools/ew_bench.py --packed2 2>/dev/null | grep field > "$OUT/${R}_ew_packed_two_words.txt"
python tools/ew_bench.py --div2 2>/dev/null | grep field > "$OUT/${R}_ew_div2.txt"
python tools/ew_bench.py --div3 2>/dev/null | grep field > "$OUT/${R}_ew_div3.txt"
python tools/ew_bench.py --divt 2>/dev/null | grep field > "$OUT/${R}_ew_divt.txt"
python tools/ew_bench.py --divwide 2>/dev/null | grep field > "$OUT/${R}_ew_div_wide.txt"
python tools/ew_bench.py --inv16 2>/dev/null | grep field > "$OUT/${R}_ew_inv16.txt"
python tools/ew_bench.py --pow24 2>/dev/null | grep field > "$OUT/${R}_ew_pow24.txt"
python tools/ew_bench.py --bininv 2>/dev/null | grep field > "$OUT/${R}_ew_bin_inverse_table.txt"
python tools/ew_bench.py --band16 2>/dev/null | grep field > "$OUT/${R}_ew_band16.txt"
./_variants/strided_pieces > "$OUT/${R}_strided_pieces.txt" 2>/dev/null
for sd in 61 62 63; do python tools/fuzz_r06.py 60 $sd 2>/dev/null | tail -1; done > "$OUT/${R}_fuzz.txt"
# r06, second half: the array functions beside the element-wise ufuncs (folds / scans of one long row, plane convolutions, polynomial evaluation),
# GF(2^m) / GF(p^m) matmul on the matrix cores
python tools/reduce_time.py 2>/dev/null | grep field > "$OUT/${R}_reduce_time.txt"
python tools/misc_bench.py 2>/dev/null | grep field > "$OUT/${R}_misc_bench.txt"
{ GFA_MFMA_BITS_MIN_LOG=16 python tools/matmul_bits_crossover.py; GFA_MFMA_BITS_MIN_LOG=62 python tools/matmul_bits_crossover.py; } 2>/dev/null | grep GFA_ > "$OUT/${R}_matmul_bits_crossover.txt"
python tools/rs_decode_by_errors.py 2>/dev/null | grep errors > "$OUT/${R}_rs_decode_by_errors.txt"
