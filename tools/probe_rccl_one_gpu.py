import os, sys, torch, torch.distributed as dist
r = int(os.environ["RANK"]); w = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=r, world_size=w)
    t = torch.ones(4, device="cuda:0") * (r + 1)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("rank", r, "all_reduce ->", t.tolist(), flush=True)
except Exception as e:
    print("rank", r, "FAILED:", str(e)[:300].replace("\n", " | "), flush=True)
