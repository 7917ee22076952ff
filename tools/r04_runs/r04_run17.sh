set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_elementwise.py -x -q -m gpu -k "staged_in_turn or reference_outputs_for_table" > $O/test_big16.txt 2>&1
timeout 300 python tools/ew_bench.py --big16 > $O/ew_big16.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_rs.py -x -q -m gpu -k "full_size" > $O/test_rs_full.txt 2>&1
