set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04s; mkdir -p $O
timeout 400 python tools/fuzz_r04.py 90 101 > $O/fuzz.txt 2>&1
timeout 400 python tools/fuzz_r04.py 60 102 >> $O/fuzz.txt 2>&1
timeout 300 python tools/fuzz_fields.py 40 1101 >> $O/fuzz.txt 2>&1
timeout 300 python tools/fuzz_codes.py 40 177 >> $O/fuzz.txt 2>&1
timeout 300 python tools/fuzz_ntt_linalg.py 40 131 >> $O/fuzz.txt 2>&1
timeout 300 python tools/fuzz_table_fields.py 30 15 >> $O/fuzz.txt 2>&1
