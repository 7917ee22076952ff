set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04y; mkdir -p $O; rm -f $O/goldi_split.txt
for cfg in "X=0" "GFA_NTT3_LOG0=8 GFA_NTT3_LOG1=9" "GFA_NTT3_LOG0=9 GFA_NTT3_LOG1=8" "GFA_NTT3_LOG0=10 GFA_NTT3_LOG1=8" "GFA_NTT3_LOG0=8 GFA_NTT3_LOG1=10" "GFA_NTT3_LOG0=8 GFA_NTT3_LOG1=8" "GFA_NTT3_LOG0=9 GFA_NTT3_LOG1=9" "GFA_NTT3_LOG0=10 GFA_NTT3_LOG1=9" "GFA_NTT3_LOG0=9 GFA_NTT3_LOG1=10"; do
echo "== $cfg" >> $O/goldi_split.txt
env $cfg timeout 200 python tools/goldi_time3.py 26 2>&1 | grep "2^" >> $O/goldi_split.txt
env $cfg timeout 200 python tools/goldi_time3.py 26 2>&1 | grep "2^" >> $O/goldi_split.txt
done
for cfg in "X=0" "GFA_NTT3_LOG0=8 GFA_NTT3_LOG1=8" "GFA_NTT3_LOG0=7 GFA_NTT3_LOG1=8" "GFA_NTT3_LOG0=8 GFA_NTT3_LOG1=7" "GFA_NTT3_LOG0=6 GFA_NTT3_LOG1=9" "GFA_NTT3_LOG0=9 GFA_NTT3_LOG1=9"; do
echo "== $cfg" >> $O/goldi_split.txt
env $cfg timeout 200 python tools/goldi_time3.py 24 2>&1 | grep "2^" >> $O/goldi_split.txt
done
for cfg in "X=0" "GFA_NTT3_LOG0=9 GFA_NTT3_LOG1=10" "GFA_NTT3_LOG0=10 GFA_NTT3_LOG1=9" "GFA_NTT3_LOG0=8 GFA_NTT3_LOG1=10"; do
echo "== $cfg" >> $O/goldi_split.txt
env $cfg timeout 200 python tools/goldi_time3.py 28 2>&1 | grep "2^" >> $O/goldi_split.txt
done
cat $O/goldi_split.txt
