set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04v; mkdir -p $O
cat > /tmp/enc2.py <<'P'
import ctypes, os, sys, time
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
    Pd = torch.empty((B, 32), dtype=torch.uint8, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): L.check(lib.gfa_rs_encode(rs._handle, Md.data_ptr(), 223, Pd.data_ptr(), B, 1, L.U8, st))
    e0.record()
    for _ in range(10): L.check(lib.gfa_rs_encode(rs._handle, Md.data_ptr(), 223, Pd.data_ptr(), B, 1, L.U8, st))
    e1.record(); e1.synchronize()
    tp = e0.elapsed_time(e1) / 10
    print(f"2^{logb} words: full encode {ms.value:.4f} ms = {B * 255 / ms.value / 1e6:.0f} GB/s   parity only {tp:.4f} ms = {B * 255 / tp / 1e6:.0f} GB/s-equivalent", flush=True)
P
objs=$(ls galois_amd/_obj/*.o | grep -v gfa_rs.o | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=galois_amd/csrc/exports.map -o _variants/lib_p3.so $objs _variants/rs_p3.o
echo "== full kernel" >> $O/probe2.txt; timeout 100 python /tmp/enc2.py 2>&1 | grep words >> $O/probe2.txt
echo "== p3: MODE 0 without the early stores of message blocks 0..2 (timing probe, wrong output)" >> $O/probe2.txt; GALOIS_AMD_LIB=$PWD/_variants/lib_p3.so timeout 100 python /tmp/enc2.py 2>&1 | grep words >> $O/probe2.txt
