cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 ./tools/ubench/ntt_access 64 > gpurun_out/ntt_access.txt 2>&1
cat gpurun_out/ntt_access.txt
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/tools/m32_time.py 1 > /tmp/kt.log 2>&1)
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - $f <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'ntt_m32' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
print("dispatch durations us (first 16):", [round(x,1) for x in d[:16]])
ev=d[0::2]; od=d[1::2]
print("even avg %.1f  odd avg %.1f"%(sum(ev)/len(ev), sum(od)/len(od)))
PY
