"""World-size-1 RCCL round trip with exactly the calls bench.py makes for N > 1 (init with device_id, barrier, MAX all-reduce of a
float64 scalar, destroy) plus the all_to_all_single of the distributed transform: checks that the collective library loads and
runs in this image.  The N > 1 paths themselves are covered by the gloo tests."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dist.barrier()
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
x = torch.arange(1 << 20, dtype=torch.int64, device="cuda")
y = torch.empty_like(x)
dist.all_to_all_single(y, x)
torch.cuda.synchronize()
assert float(t.item()) == 1.25 and bool(torch.equal(x, y))
dist.barrier()
dist.destroy_process_group()
print("rccl self-check ok", torch.cuda.nccl.version())
