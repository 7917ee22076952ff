set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l; mkdir -p $O
objs=$(ls galois_amd/_obj/*.o | grep -v gfa_elementwise.o | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=galois_amd/csrc/exports.map -o _variants/lib_b16.so $objs _variants/ew_b16.o
cat > /tmp/ewp.py <<'P'
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
lib = L.lib(); st = torch.cuda.current_stream().cuda_stream; ms = ctypes.c_float()
n = 50_000_000
for p in (65537, 7340033, 2147483647, 4294967291):
    GF = ga.GF(p); rng = np.random.default_rng(1)
    a = torch.from_numpy(rng.integers(0, p, n, dtype=np.uint32).view(np.int32)).cuda()
    b = torch.from_numpy(rng.integers(1, p, n, dtype=np.uint32).view(np.int32)).cuda()
    o = torch.empty_like(a)
    L.check(lib.gfa_time_unary(GF._handle, L.OP_RECIP, b.data_ptr(), o.data_ptr(), n, L.U32, st, 10, ctypes.byref(ms))); tr = ms.value
    L.check(lib.gfa_time_binary(GF._handle, L.OP_DIV, a.data_ptr(), b.data_ptr(), o.data_ptr(), n, L.U32, st, 10, ctypes.byref(ms))); td = ms.value
    print(f"GF({p}): recip {n / tr / 1e6:.0f} Gop/s ({8 * n / tr / 1e6 / 8000:.2f})  div {n / td / 1e6:.0f} Gop/s ({12 * n / td / 1e6 / 8000:.2f})", flush=True)
P
for rep in 1 2; do
echo "== batch 32" >> $O/ew_batch.txt; timeout 200 python /tmp/ewp.py 2>&1 | grep -v amdgpu >> $O/ew_batch.txt
echo "== batch 16" >> $O/ew_batch.txt; GALOIS_AMD_LIB=$PWD/_variants/lib_b16.so timeout 200 python /tmp/ewp.py 2>&1 | grep -v amdgpu >> $O/ew_batch.txt
done
timeout 1500 python -m pytest tests -x -q -m gpu --durations=10 > $O/gpu_tests.txt 2>&1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
