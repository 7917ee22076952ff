#!/bin/bash
# r05 GPU run 11: longer campaigns of every differential fuzzer on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
{
for job in "fuzz_fields.py 120 101" "fuzz_fields.py 120 202" "fuzz_table_fields.py 60 303" "fuzz_ntt_linalg.py 60 404" "fuzz_conv_ntt3.py 60 505" "fuzz_r04.py 60 606" "fuzz_r05.py 120 707" "fuzz_codes.py 60 808"; do
    set -- $job
    echo "== tools/$1 $2 s, seed $3"
    timeout $(( $2 + 240 )) python tools/$1 $2 $3 2>&1 | grep -v amdgpu | tail -3
done
} > gpurun_out/r05_fuzz_long.txt 2>&1
cat gpurun_out/r05_fuzz_long.txt
