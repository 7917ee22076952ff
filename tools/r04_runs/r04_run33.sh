set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04y; mkdir -p $O
timeout 300 python tools/ntt_bench_context.py 2>&1 | grep -v amdgpu > $O/clock_ramp2.txt
for w in "20 200" "2000 200" "2000 2000" "20 200" "4000 200"; do set -- $w; echo "== warmup $1 steps $2" >> $O/headline_warm.txt; timeout 300 python bench.py --no-extras --no-cpu-baseline --no-pmc --warmup $1 --steps $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'])" >> $O/headline_warm.txt; done
cat $O/clock_ramp2.txt $O/headline_warm.txt
