set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04v; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench.err
grep -n "passed\|failed" $O/gpu_tests.txt | tail -1; tail -1 $O/smoke.txt
python - <<P
import json
d=json.loads(open("$O/bench_driver_args.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["steps"], d["warmup"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["roofline"]["traffic"])
ex=d["extra"]
print({k:(v.get("roofline_frac") or v.get("encode_GB/s")) for k,v in ex.items() if isinstance(v,dict) and ("roofline_frac" in v or "encode_GB/s" in v)})
P
