cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ntt.py -x -q -k "signed_montgomery or power_of_two or unreduced or batched" 2>&1 | tail -3
{
echo "== default (split pass1, nosplit pass2)"; timeout 300 python tools/m32_time.py 4
echo "== split both"; GFA_M32_SPLIT=3 timeout 300 python tools/m32_time.py 3
echo "== nosplit both"; GFA_M32_SPLIT=0 timeout 300 python tools/m32_time.py 3
echo "== order 2"; GFA_M32_ORDER=2 timeout 300 python tools/m32_time.py 3
echo "== order1 only for >=64 tiles"; GFA_M32_ORDER_MIN=64 timeout 300 python tools/m32_time.py 3
echo "== 1024 threads, split pass 1"; GFA_M32_THREADS=1024 timeout 300 python tools/m32_time.py 3
} 2>&1 | grep -v "amdgpu.ids" > gpurun_out/m32_time2.txt
cat gpurun_out/m32_time2.txt
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/tools/m32_time.py 1 > /tmp/kt.log 2>&1)
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - $f <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'ntt_m32' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
ev=d[0::2]; od=d[1::2]
print("pass 1 avg %.1f us, pass 2 avg %.1f us"%(sum(ev)/len(ev), sum(od)/len(od)))
PY
