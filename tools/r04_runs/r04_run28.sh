set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04y; mkdir -p $O
rm -f $O/goldi_knobs.txt
for rep in 1 2; do
for cfg in "GFA_GL_T512=0 GFA_GL_TABLE=1" "GFA_GL_T512=0 GFA_GL_TABLE=0" "GFA_GL_T512=1 GFA_GL_TABLE=1" "GFA_GL_T512=1 GFA_GL_TABLE=0" "GFA_GL_T512=2 GFA_GL_TABLE=1" "GFA_GL_T512=0 GFA_GL_TABLE=1 GFA_GL_W4=0" "GFA_GL_T512=0 GFA_GL_TABLE=1 GFA_GL_W4=2" "GFA_GL_T512=1 GFA_GL_TABLE=1 GFA_GL_W4=0"; do
echo "== $cfg" >> $O/goldi_knobs.txt
env $cfg timeout 200 python tools/goldi_time.py 2>&1 | grep -v amdgpu >> $O/goldi_knobs.txt
done; done
cat $O/goldi_knobs.txt
