#!/bin/bash
# PMC passes over one command (run on the GPU box):  bash tools/pmc_run.sh <tag> <kernel-name-substring> -- <command...>
# Every pass collects a few SQ / TCC counters with --kernel-trace only (never combined with the API / memory trace domains);
# the per-kernel averages of all passes are written to gpurun_out/<tag>.txt.
set -u
TAG=$1; KERN=$2; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    ( cd "$ROOT" && rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$TAG/p$i -o p -- "$@" > /tmp/pmc_$TAG.log$i 2>&1 )
done
python - "$TAG" "$KERN" > "$OUT/$TAG.txt" <<'PY'
import csv, glob, sys, collections
tag, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"/tmp/pmc_{tag}/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if kern in name:
            acc[name[:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, cs in acc.items():
    n = max(len(v) for v in cs.values())
    print(f"{name}   ({n} dispatches; per-dispatch averages, whole chip)")
    for c, v in sorted(cs.items()):
        print(f"    {c:24s} {sum(v) / len(v):16.1f}")
PY
cat "$OUT/$TAG.txt"
