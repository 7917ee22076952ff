set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
OBJ=galois_amd/_obj
for i in 1 2 3 4 5; do
  objs=$(ls $OBJ/*.o | grep -v gfa_ntt_m32.o | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=galois_amd/csrc/exports.map -o _variants/lib_v$i.so $objs _variants/m32_v$i.o
done
cat > /tmp/t16.py <<'P'
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
lib = L.lib(); st = torch.cuda.current_stream().cuda_stream
for p, logn, batch in [(7340033, 16, 1024), (7340033, 16, 4096)]:
    P = ga.GF(p); N = 1 << logn
    x = torch.from_numpy(np.random.default_rng(3).integers(0, p, (batch, N), dtype=np.uint32).view(np.int32)).cuda()
    o = torch.empty_like(x); ms = ctypes.c_float(); best = 1e9
    for _ in range(3):
        L.check(lib.gfa_time_ntt(P._handle, x.data_ptr(), o.data_ptr(), N, batch, P._root_of_unity_int(N), L.U32, st, 20, ctypes.byref(ms)))
        best = min(best, ms.value)
    print(f"p={p} 2^{logn} x {batch}: {best:.4f} ms   {8.0 * N * batch / (best * 1e-3) / 8e12:.3f}", flush=True)
P
for rep in 1 2; do for i in 1 2 3 4 5; do echo "== v$i" >> $O/time_2e16.txt; GALOIS_AMD_LIB=$PWD/_variants/lib_v$i.so timeout 120 python /tmp/t16.py 2>&1 | grep -v amdgpu >> $O/time_2e16.txt; done; done
timeout 600 python -m pytest tests/test_gpu_ntt.py -x -q -m gpu -k "2e16" > $O/test_2e16.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_elementwise.py tests/test_gpu_wide.py -x -q -m gpu -k "ordering_and_editing or reductions_of_the_big" > $O/test_misc.txt 2>&1
