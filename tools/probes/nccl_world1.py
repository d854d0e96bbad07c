"""Probe: RCCL through torch.distributed on one GPU (world size 1) -- init, a device-named barrier and an async
all-reduce, i.e. the calls bench.py / cfun_amd.dist make at N > 1."""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dist.barrier(device_ids=[0])
t = torch.ones(4, device="cuda")
w = dist.all_reduce(t, async_op=True)
w.wait()
torch.cuda.synchronize()
print("nccl world-1 ok", t.tolist())
dist.destroy_process_group()
