#!/usr/bin/env python3
"""Single-rank exercise of the RCCL calls the N>1 bench path makes (init, all_gather_into_tensor on a side stream,
barrier, MAX all-reduce) -- run under torchrun with --nproc-per-node 1 on the 1-GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=dev)
world, rank = dist.get_world_size(), dist.get_rank()
shard = torch.arange(6400 * 256, dtype=torch.float32, device=dev).view(6400, 256) + rank
bank = torch.empty((world * 6400, 256), device=dev)
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    work = dist.all_gather_into_tensor(bank, shard, async_op=True)
x = torch.randn(4096, 4096, device=dev) @ torch.randn(4096, 4096, device=dev)      # overlapped compute
work.wait()
torch.cuda.current_stream().wait_stream(side)
assert torch.equal(bank[:6400], shard)
t = torch.tensor([1.25 + rank], dtype=torch.float64, device=dev)
dist.barrier()
dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
print(f"rank {rank}/{world}: RCCL all_gather_into_tensor + barrier + MAX all-reduce OK ({float(t)})")
dist.destroy_process_group()
