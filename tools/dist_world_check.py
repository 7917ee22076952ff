"""One rank of the multi-GPU checks (tests/test_gpu_multi.py launches WORLD of these under torch.distributed.run; needs
WORLD visible GPUs).  Every rank:
  1. the bench's collective calls over RCCL (barrier, MAX all-reduce) and an all_to_all_single across the ranks;
  2. galois_amd.dist.ntt_four_step_distributed / intt over torch.distributed (RCCL all-to-all): forward against the whole
     transform computed on this rank's own GPU, inverse as a round trip;
  3. the C-ABI form: gfa_ntt_dist / gfa_intt_dist with a communicator created directly through RCCL (ncclGetUniqueId on
     rank 0, the id handed to the others through the process group), for every GFA_DIST_NSUB setting the environment names.
Prints `dist world check ok <world>` on rank 0."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.distributed as tdist

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
# GFA_DIST_CHECK_ONE_GPU=1 (r05): every rank on cuda:0, exchange over gloo through the host -- WORLD processes drive the REAL kernels
# through the WORLD-rank layout bookkeeping of the Python path on a one-GPU box (RCCL refuses several ranks per device, so
# the raw-communicator part 3, whose per-rank kernels are the same C-ABI calls, needs real devices and is skipped)
ONE_GPU = os.environ.get("GFA_DIST_CHECK_ONE_GPU", "0") == "1"
if ONE_GPU:
    local = 0
else:
    assert torch.cuda.device_count() >= world, "needs one GPU per rank"
torch.cuda.set_device(local)
if ONE_GPU:
    tdist.init_process_group(backend="gloo", rank=rank, world_size=world)
else:
    tdist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
import galois_amd as ga
from galois_amd import _lib as L, dist as gdist

# ---- 1. plain collectives ----
tdist.barrier()
cdev = "cpu" if ONE_GPU else "cuda"
t = torch.tensor([float(rank)], dtype=torch.float64, device=cdev)
tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
assert float(t.item()) == world - 1
x = torch.arange(world * 1024, dtype=torch.int64, device=cdev) + rank * 1_000_000
y = torch.empty_like(x)
tdist.all_to_all_single(y, x)
for s in range(world):
    assert torch.equal(y[s * 1024:(s + 1) * 1024], torch.arange(rank * 1024, (rank + 1) * 1024, dtype=torch.int64, device=cdev) + s * 1_000_000)

# ---- a raw RCCL communicator next to torch's ----
comm = ctypes.c_void_p()
rccl = None
if not ONE_GPU:
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    rccl = ctypes.CDLL(path if os.path.exists(path) else "librccl.so", mode=ctypes.RTLD_GLOBAL)

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    uid = UniqueId()
    if rank == 0:
        assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    box = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).cuda()
    tdist.broadcast(box, src=0)
    ctypes.memmove(ctypes.byref(uid), bytes(box.cpu().numpy().tobytes()), 128)
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), world, uid, rank) == 0

lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
try:
    for order, n1, n2, dt, tdt, native in [(2**64 - 2**32 + 1, 1 << 10, 1 << 12, L.U64, torch.int64, np.uint64),
                                           (7340033, 1 << 8, 1 << 10, L.U32, torch.int32, np.uint32),
                                           (469762049, 1 << 10, 1 << 16, L.U32, torch.int32, np.uint32)]:
        GF = ga.GF(order)
        n = n1 * n2
        gen = torch.Generator(device="cpu").manual_seed(1234 + n)  # the same vector on every rank
        xs = torch.randint(0, min(order, 2**62), (n,), generator=gen, dtype=torch.int64)
        xfull = xs.to(tdt).cuda()
        omega = GF._root_of_unity_int(n)
        want = torch.empty_like(xfull)
        L.check(lib.gfa_ntt(GF._handle, xfull.data_ptr(), want.data_ptr(), n, 1, omega, 0, dt, st))
        cols, rows = n2 // world, n1 // world
        mine = xfull.view(n1, n2)[:, rank * cols:(rank + 1) * cols].contiguous()
        want_rows = want.view(n2, n1)[:, rank * rows:(rank + 1) * rows].t().contiguous()  # X[k1 + n1 k2], this rank's k1
        # ---- 2. the Python path over torch.distributed ----
        out = gdist.ntt_four_step_distributed(GF, mine, n1, n2)
        assert torch.equal(out.view(rows, n2), want_rows), f"four-step forward over RCCL, order {order}, rank {rank}"
        back = gdist.intt_four_step_distributed(GF, out, n1, n2)
        assert torch.equal(back.view(n1, cols), mine), f"four-step inverse over RCCL, order {order}, rank {rank}"
        if ONE_GPU:
            continue
        # ---- 3. the C-ABI path with its own exchange ----
        out2 = torch.empty(rows * n2, dtype=tdt, device="cuda")
        L.check(lib.gfa_ntt_dist(GF._handle, comm, rank, world, mine.data_ptr(), out2.data_ptr(), n1, n2, omega, dt, st), "gfa_ntt_dist")
        assert torch.equal(out2.view(rows, n2), want_rows), f"gfa_ntt_dist, order {order}, rank {rank}"
        back2 = torch.empty(n1 * cols, dtype=tdt, device="cuda")
        L.check(lib.gfa_intt_dist(GF._handle, comm, rank, world, out2.data_ptr(), back2.data_ptr(), n1, n2, omega, 1, dt, st), "gfa_intt_dist")
        assert torch.equal(back2.view(n1, cols), mine), f"gfa_intt_dist, order {order}, rank {rank}"
    torch.cuda.synchronize()
finally:
    if rccl is not None:
        rccl.ncclCommDestroy(comm)
tdist.barrier()
tdist.destroy_process_group()
if rank == 0:
    print("dist world check ok", world)
