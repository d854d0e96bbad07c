#!/usr/bin/env python3
"""Does a captured hipGraph recover anything?  The mask head (RoIAlign of the image -> U-Net -> softmax -> CE + edge loss ->
backward: ~85 % of the cfg2 step and ~900 of its launches) eagerly vs as ONE replayed graph, same process, same kernels.
The capture needs FIXED Dropout3d masks: the kept-channel sets decide the shapes of the per-sample convs, so in training
(a new draw every step) a graph of this part cannot be replayed -- the probe only measures what launch overhead there is
to recover (tools only; not part of the product).   python tools/graph_probe.py [--iters 10]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfun_amd import config as C, ops, step as S  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
cfg = C.heart_config("finetune", 256, 256, 128)
dev = torch.device("cuda")
torch.manual_seed(0)
net = S.CFUNHotPath(cfg).to(dev)
s = S.synthetic_inputs(cfg, dev)
unet = net.mask.modified_u_net
b = cfg.UNET_MASK_BRANCH_CHANNEL
gen = torch.Generator().manual_seed(1)
unet.dropout_masks = [torch.empty(4, ch).bernoulli_(0.4, generator=gen) / 0.4 for ch in (b, 2 * b, 4 * b, 8 * b, 16 * b)]
_up = unet._upload_dropout
_cache = {}


def cached_upload(drops, device):      # fixed masks: uploaded once, outside the capture
    if "v" not in _cache:
        _cache["v"] = _up(drops, device)
    return _cache["v"]


unet._upload_dropout = cached_upload
img = ops.to_ndhwc(s["image"])[0]
params = [p for p in net.mask.parameters() if p.requires_grad]


def head_step():
    logits, probs = net.mask.forward_ndhwc(img, s["p_rois"])
    ce, edge = ops.mask_losses(logits, probs, s["mask_labels"])
    (ce + edge).backward()
    return ce, edge


def timed(fn, iters):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / iters


def eager():
    for p in params:
        p.grad = None
    return head_step()


for _ in range(3):
    eager()
t_eager = timed(eager, args.iters)
ce0, edge0 = [float(v) for v in eager()]
g0 = [p.grad.clone() for p in params]

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        eager()
torch.cuda.current_stream().wait_stream(side)
for p in params:
    p.grad = None
graph = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(graph):
        ce_g, edge_g = head_step()
except Exception as e:      # what stopped the capture is the finding
    print("capture failed: %s: %s" % (type(e).__name__, str(e)[:400]))
    sys.exit(0)
graph.replay()
torch.cuda.synchronize()
same = all(torch.equal(a, p.grad) for a, p in zip(g0, params))
t_graph = timed(graph.replay, args.iters)
print("mask head fwd + losses + bwd: eager %.2f ms, one replayed hipGraph %.2f ms (%+.2f ms); losses eager (%.6f, %.6f) graph "
      "(%.6f, %.6f); gradients bit-identical: %s" % (t_eager, t_graph, t_graph - t_eager, ce0, edge0, float(ce_g), float(edge_g), same))
