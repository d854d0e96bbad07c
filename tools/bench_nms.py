#!/usr/bin/env python3
"""cfun_nms3d call time at the proposal layer's size (1000 candidates, threshold 0.7, 500 kept at most) and at refine_detections'
(64 boxes, 0.3, 32): 50 calls between one pair of HIP events.   CFUN_NMS_SCAN_LDS=0|1 python tools/bench_nms.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfun_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
# (spread of the centres / jitter of the sizes: scattered boxes are all kept and the scan stops at max_num after ~max_num rows;
# clustered boxes -- what the RPN of the bench step produces -- suppress each other and the scan has to visit all n candidates)
for n, thr, mx, spread, jit in ((1000, 0.7, 500, 96.0, 48.0), (1000, 0.7, 500, 6.0, 4.0), (1000, 0.7, 500, 2.0, 1.0), (64, 0.3, 32, 48.0, 48.0)):
    c = torch.rand(n, 3, generator=g) * spread
    sz = 32.0 + torch.rand(n, 3, generator=g) * jit
    boxes = torch.cat([c, c + sz], dim=1).to(dev)
    scores = torch.rand(n, generator=g).to(dev)
    for _ in range(3):
        keep, count = ops.nms3d(boxes, scores, thr, mx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        keep, count = ops.nms3d(boxes, scores, thr, mx)
    e1.record()
    torch.cuda.synchronize()
    print("n = %4d  thr %.1f  max %3d  spread %5.1f / jitter %4.1f: kept %3d, %.1f us per call" % (n, thr, mx, spread, jit, int(count.item()), e0.elapsed_time(e1) * 1e3 / 50))
