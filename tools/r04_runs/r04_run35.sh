set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04y; mkdir -p $O; rm -f $O/goldi_t512b.txt
for rep in 1 2; do for k in 0 1 2 3; do echo "== GFA_GL_T512B=$k" >> $O/goldi_t512b.txt; GFA_GL_T512B=$k timeout 200 python tools/goldi_time.py 2>&1 | grep "2^" >> $O/goldi_t512b.txt; GFA_GL_T512B=$k timeout 200 python tools/goldi_time3.py 22 2>&1 | grep "2^" >> $O/goldi_t512b.txt; done; done
cat $O/goldi_t512b.txt
