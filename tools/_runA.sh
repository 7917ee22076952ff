cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_elementwise.py -x -q -k "uint32_and_int64 or small_fields_in_wide or mid_size or fields_up_to_2e16" 2>&1 | tail -2
echo "== new"; timeout 300 python tools/ew_bench.py --widestore 2>/dev/null | grep field
echo "== old (generic kernels, tables in L2)"; GFA_MID_LDS=0 timeout 300 python tools/ew_bench.py --widestore 2>/dev/null | grep field
