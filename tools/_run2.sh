cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/tools/m32_time.py 1 > /tmp/kt.log 2>&1)
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/m32_kernel_stats.csv
head -8 $f
bash tools/pmc_run.sh m32_pmc ntt_m32_kernel -- python tools/m32_time.py 1
