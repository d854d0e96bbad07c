#!/usr/bin/env python3
"""Launch-to-launch time of one conv through the C ABI: 64 back-to-back launches between one pair of HIP events (what bench.py's
roofline_hbm reports), next to the median of per-launch event pairs (what tools/bench_layers.py reports).
   python tools/bench_b2b.py [stem|pointwise]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cfun_amd import _lib, ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "stem"
dev = torch.device("cuda:0")
n, side = 4, (96, 96, 96)
if which == "stem":
    ci, co, k, pad = 1, 20, (3, 3, 3), (1, 1, 1)
else:
    ci, co, k, pad = 40, 8, (1, 1, 1), (0, 0, 0)
x = torch.randn((n,) + side + (ci,), device=dev)
w = torch.randn((co, ci) + k, device=dev)
spec = ops.ConvSpec(k=k, co=co, pad=pad)
p = ops._params(spec, x.shape, False, False, False)
lib = _lib.load()
wp = ops.pack_weight(w)
y = torch.empty((n,) + side + (co,), device=dev)
ws = _lib.workspace(lib.cfun_conv3d_fwd_workspace_bytes(C.byref(p)), x)
args = (_lib.ptr(x), _lib.ptr(wp), None, None, None, _lib.ptr(y), C.byref(p), _lib.ptr(ws), ws.numel(), _lib.stream(x))
for _ in range(4):
    _lib.check(lib.cfun_conv3d_fwd(*args), "conv3d_fwd")
res = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(64):
        lib.cfun_conv3d_fwd(*args)
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 64 * 1e3)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for a, b in ev:
    a.record()
    lib.cfun_conv3d_fwd(*args)
    b.record()
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) for a, b in ev)
print("%s: back-to-back %s us per launch; per-launch event pairs median %.1f us" % (which, " / ".join("%.1f" % r for r in res), t[10] * 1e3))
