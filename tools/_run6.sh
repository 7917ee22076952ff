cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for v in default w4 w5 w4nodef w6nodef w8nodef; do
  if [ $v = default ]; then unset GALOIS_AMD_LIB; else export GALOIS_AMD_LIB=$PWD/_variants/lib_m32_$v.so; fi
  for tw in 0 1 2; do for sp in 0 1 3; do
    echo "$v tw=$tw split=$sp: $(GFA_M32_TW=$tw GFA_M32_SPLIT=$sp timeout 120 python tools/m32_time.py 1 2>/dev/null | grep 'p=' | awk '{print $5, $6}')"
  done; done
done
} > gpurun_out/m32_sweep.txt 2>&1
cat gpurun_out/m32_sweep.txt
