set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04u; mkdir -p $O
cat > /tmp/enc.py <<'P'
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
lib = L.lib(); st = torch.cuda.current_stream().cuda_stream
rs = ga.ReedSolomon(255, 223); ms = ctypes.c_float()
for logb in (20, 22):
    B = 1 << logb
    Md = torch.empty((B, 223), dtype=torch.uint8, device="cuda").random_(0, 256)
    Cd = torch.empty((B, 255), dtype=torch.uint8, device="cuda")
    L.check(lib.gfa_time_rs_encode(rs._handle, Md.data_ptr(), 223, Cd.data_ptr(), B, L.U8, st, 10, ctypes.byref(ms)))
    print(f"2^{logb} words: encode {ms.value:.4f} ms = {B * 255 / ms.value / 1e6:.0f} GB/s", flush=True)
P
for v in p1 p2 t512 t256; do
  objs=$(ls galois_amd/_obj/*.o | grep -v gfa_rs.o | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=galois_amd/csrc/exports.map -o _variants/lib_$v.so $objs _variants/rs_$v.o
done
echo "== full kernel" >> $O/probe.txt; timeout 100 python /tmp/enc.py 2>&1 | grep words >> $O/probe.txt
for v in p1 p2 t512 t256; do echo "== $v" >> $O/probe.txt; GALOIS_AMD_LIB=$PWD/_variants/lib_$v.so timeout 100 python /tmp/enc.py 2>&1 | grep words >> $O/probe.txt; done
