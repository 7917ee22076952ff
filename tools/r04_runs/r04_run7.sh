set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_rs.py tests/test_gpu_multi.py -x -q -m gpu -k "stress or assembled or multi or world or selfcheck or two_gpus" > $O/test_new_rs_multi.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ntt.py -x -q -m gpu -k "c5_full or 2e26" --durations=5 > $O/test_c5_oracle.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_elementwise.py tests/test_gpu_wide.py -x -q -m gpu -k "ordering_and_editing or reductions_of_the_big" > $O/test_misc.txt 2>&1
