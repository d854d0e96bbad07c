#!/usr/bin/env python3
"""Which torch-side ops (copies, adds, fills) surround the HIP kernels in one cfg2 step?  torch.profiler table by
input shape -- the glue that the C-ABI kernels do not cover (tools only; not part of the product)."""
import os
import sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfun_amd import step as S, config as C
from torch.profiler import profile, ProfilerActivity
cfg = C.heart_config('finetune', 256, 256, 128)
dev = torch.device('cuda')
net = S.CFUNHotPath(cfg).to(dev)
s = S.synthetic_inputs(cfg, dev)
for _ in range(2):
    net.zero_grad(set_to_none=True); S.training_step(net, s)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    net.zero_grad(set_to_none=True); S.training_step(net, s); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.self_device_time_total)
print("%-28s %6s %10s  %s" % ("op", "calls", "self us", "input shapes"))
for e in rows[:70]:
    if e.self_device_time_total > 5:
        print("%-28s %6d %10.1f  %s" % (e.key, e.count, e.self_device_time_total, str(e.input_shapes)[:150]))
print("total aten self us: %.1f" % sum(e.self_device_time_total for e in rows))
