"""1-rank RCCL sanity of the exact collective pattern the data-parallel step uses (async all-reduce of slices of the
flat gradient buffer, issued between backward parts, waited before the SGD step)."""
import os
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
flat = torch.randn(109499337, device="cuda")
ref = flat.clone()
hs = []
for a, b in ((70524096, 109499337), (8543296, 23508032), (0, 225344)):
    x = torch.randn(4096, 4096, device="cuda") @ torch.randn(4096, 4096, device="cuda")   # compute to overlap with
    hs.append(dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, async_op=True))
for h in hs:
    h.wait()
torch.cuda.synchronize()
assert torch.equal(flat, ref)
dist.barrier()
dist.destroy_process_group()
print("rccl sanity ok")
