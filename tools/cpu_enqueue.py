import sys, time, torch
sys.path.insert(0, '/root/repo')
from cfun_amd import step as S, config as C
cfg = C.heart_config('finetune', 256, 256, 128)
dev = torch.device('cuda')
net = S.CFUNHotPath(cfg).to(dev)
s = S.synthetic_inputs(cfg, dev)
for _ in range(3):
    net.zero_grad(set_to_none=True); S.training_step(net, s)
torch.cuda.synchronize()
enq = []; tot = []
for _ in range(8):
    t0 = time.perf_counter()
    net.zero_grad(set_to_none=True); S.training_step(net, s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print("enqueue ms", [round(v, 1) for v in enq]); print("total ms", [round(v, 1) for v in tot])
