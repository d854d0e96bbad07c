#!/usr/bin/env python3
"""Why do the l4.0 data / weight gradient take 10 - 14 % longer inside the step than in tools/bench_layers.py (VERDICT
round 3, item 2b)?  The same C-ABI calls (3x3x3 40 -> 40 @ 4 x 96^3) timed with HIP events (a) back to back, first vs
last iterations of a long run (clock / power state), (b) with a 1.1 GB device copy between the launches (cold L2 and
Infinity Cache), (c) with the InstanceNorm backward that precedes the data gradient in the step between the launches.
tools only; python tools/probe_instep.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfun_amd import _lib, ops  # noqa: E402
from cfun_amd._lib import check, ptr  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
n, side, c = 4, 96, 40
x = torch.randn(n, side, side, side, c, device=dev)
w = torch.randn(c, c, 3, 3, 3, device=dev) / (27 * c) ** 0.5
spec = ops.ConvSpec(k=(3, 3, 3), co=c, pad=(1, 1, 1))
p = ops._params(spec, x.shape, False, False, False)
wp = ops.pack_weight(w)
wpT = ops._transpose_pack(wp, c)
g = torch.randn_like(x)
dx, dwp = torch.empty_like(x), torch.empty_like(wp)
ws_d = _lib.workspace(lib.cfun_conv3d_bwd_data_workspace_bytes(C.byref(p)), x)
ws_w = _lib.workspace(lib.cfun_conv3d_bwd_weight_workspace_bytes(C.byref(p)), x)
st = _lib.stream(x)
big_a = torch.empty(280 * 1024 * 1024, device=dev)
big_b = torch.empty_like(big_a)
stats = torch.stack([torch.zeros(n, c, device=dev), torch.ones(n, c, device=dev)], dim=-1).contiguous()
nd = torch.empty_like(x)
ws_n = _lib.workspace(lib.cfun_instnorm_workspace_bytes(n, side ** 3, c), x)


def dgrad():
    check(lib.cfun_conv3d_bwd_data(ptr(g), ptr(wpT), ptr(dx), C.byref(p), ptr(ws_d), ws_d.numel(), st), "d")


def wgrad():
    check(lib.cfun_conv3d_bwd_weight(ptr(x), ptr(g), ptr(dwp), C.byref(p), ptr(ws_w), ws_w.numel(), st), "w")


def evict():
    big_b.copy_(big_a)


def norm_bwd():
    check(lib.cfun_instnorm_lrelu_bwd(ptr(x), ptr(stats), ptr(g), ptr(nd), n, side ** 3, c, 0.01, ptr(ws_n), ws_n.numel(), st), "n")


def timed(fn, iters, between=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(iters):
        if between is not None:
            between()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in ev]


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


for name, fn in (("dgrad", dgrad), ("wgrad", wgrad)):
    t = timed(fn, 300)
    print("%s back to back, 300 launches: first 20 median %.3f ms, last 20 median %.3f ms" % (name, med(t[:20]), med(t[-20:])))
    print("%s after a 1.1 GB device copy (cold caches): median %.3f ms" % (name, med(timed(fn, 30, evict))))
    print("%s after the InstanceNorm backward of a 4 x 96^3 x 40 tensor: median %.3f ms" % (name, med(timed(fn, 30, norm_bwd))))
