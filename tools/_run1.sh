set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ntt.py -x -q -k "signed_montgomery or power_of_two or unreduced or reference_generated or known_answers or batched" 2>&1 | tail -15 > gpurun_out/m32_tests.txt
cat gpurun_out/m32_tests.txt
{
echo "== old"; GFA_NTT_M32=0 timeout 300 python tools/m32_time.py
echo "== m32 512 split"; timeout 300 python tools/m32_time.py
echo "== m32 512 nosplit"; GFA_M32_SPLIT=0 timeout 300 python tools/m32_time.py 3
echo "== m32 1024 split"; GFA_M32_THREADS=1024 timeout 300 python tools/m32_time.py 3
echo "== m32 1024 nosplit"; GFA_M32_THREADS=1024 GFA_M32_SPLIT=0 timeout 300 python tools/m32_time.py 3
echo "== m32 256 split"; GFA_M32_THREADS=256 timeout 300 python tools/m32_time.py 3
echo "== m32 256 nosplit"; GFA_M32_THREADS=256 GFA_M32_SPLIT=0 timeout 300 python tools/m32_time.py 3
echo "== m32 512 split noxcd"; GFA_NTT_XCD=0 timeout 300 python tools/m32_time.py 3
} 2>&1 | grep -v "^+" > gpurun_out/m32_time.txt
cat gpurun_out/m32_time.txt
