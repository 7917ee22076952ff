cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for tw in 0 1 2; do
GFA_M32_TW=$tw timeout 600 python -m pytest tests/test_gpu_ntt.py -x -q -k "signed_montgomery or power_of_two" 2>&1 | tail -2
done
{
for tw in 0 1 2; do
echo "== TW=$tw"; GFA_M32_TW=$tw timeout 300 python tools/m32_time.py 3
done
echo "== TW=2 nosplit"; GFA_M32_TW=2 GFA_M32_SPLIT=0 timeout 300 python tools/m32_time.py 3
echo "== TW=0 nosplit"; GFA_M32_TW=0 GFA_M32_SPLIT=0 timeout 300 python tools/m32_time.py 3
} 2>&1 | grep -v "amdgpu.ids" > gpurun_out/m32_time3.txt
cat gpurun_out/m32_time3.txt
