#!/usr/bin/env python3
"""EXPERIMENTAL 3xBF16 conv (conv3d_b3.hip) against the exact-fp32 MFMA kernel on the cfg2 U-Net shapes: time per launch
(HIP events) and max error of both against an fp64 reference on a sub-volume (tools only)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfun_amd import ops  # noqa: E402

SHAPES = [("l4.0 40->40 @4x96^3", 4, 96, 40, 40), ("l3.0 80->80 @4x48^3", 4, 48, 80, 80),
          ("l2.0 160->160 @4x24^3", 4, 24, 160, 160), ("l1.0 320->320 @4x12^3", 4, 12, 320, 320),
          ("nlc_c2 40->40 @4x48^3", 4, 48, 40, 40), ("nlc_c5 320->320 @4x6^3", 4, 6, 320, 320),
          ("160->160 @4x12^3", 4, 12, 160, 160), ("80->80 @4x24^3", 4, 24, 80, 80), ("c1 20->20 @4x96^3", 4, 96, 20, 20),
          ("l3.1 80->40 @4x48^3", 4, 48, 80, 40), ("sparse 40->16 @1x96^3", 1, 96, 40, 16),
          ("sparse 16->40 @1x96^3", 1, 96, 16, 40), ("sparse 8->20 @1x96^3", 1, 96, 8, 20),
          ("FPN P2 128->128 @1x32^3", 1, 32, 128, 128), ("RPN 128->256 @1x32^3", 1, 32, 128, 256),
          ("RPN 128->256 @1x16^3", 1, 16, 128, 256)]
dev = torch.device("cuda")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


import ctypes as C  # noqa: E402
from cfun_amd import _lib  # noqa: E402
from cfun_amd._lib import check, ptr, stream, workspace  # noqa: E402


def wgrad_fp32(x, g, co):
    lib = _lib.load()
    p = ops._params(ops.ConvSpec(k=(3, 3, 3), co=co, pad=(1, 1, 1)), x.shape, False, False, False)
    dw = torch.empty((co, x.shape[-1], 3, 3, 3), dtype=torch.float32, device=x.device)
    ws = workspace(lib.cfun_conv3d_bwd_weight_workspace_bytes(C.byref(p)), x)
    check(lib.cfun_conv3d_bwd_weight_oidhw(ptr(x), ptr(g), ptr(dw), C.byref(p), ptr(ws), ws.numel(), stream(x)), "wgrad")
    return dw


print("%-26s %10s %10s %8s %12s %12s" % ("layer", "fp32 ms", "3xbf16 ms", "speedup", "err fp32", "err 3xbf16"))
for name, n, s, ci, co in SHAPES:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, s, s, s, ci, generator=g).to(dev)
    w = (torch.randn(co, ci, 3, 3, 3, generator=g) / (ci * 27) ** 0.5).to(dev)
    spec = ops.ConvSpec(k=(3, 3, 3), co=co, pad=(1, 1, 1))
    wp, wb3 = ops.pack_weight(w).detach(), ops.pack_weight_b3(w)
    t32 = timed(lambda: ops.conv3d(x, wp, spec))
    tb3 = timed(lambda: ops.conv3d_b3(x, wb3, co))
    y32, yb3 = ops.conv3d(x, wp, spec), ops.conv3d_b3(x, wb3, co)
    # error against fp64 on the corner block [0:k)^3 of sample 0: those outputs only read inputs [0:k+1)^3
    k = min(s, 13) - 1
    y64 = F.conv3d(x[:1, :k + 1, :k + 1, :k + 1].double().permute(0, 4, 1, 2, 3), w.double(), padding=1)
    r = y64.permute(0, 2, 3, 4, 1)[:, :k, :k, :k]
    sc = float(r.abs().max())
    e32 = float((y32[:1, :k, :k, :k].double() - r).abs().max()) / sc
    eb3 = float((yb3[:1, :k, :k, :k].double() - r).abs().max()) / sc
    fl = 2.0 * ci * co * 27 * n * s ** 3
    print("%-26s %10.3f %10.3f %8.2f %12.2e %12.2e   (%.0f -> %.0f TFLOP/s)" % (name, t32, tb3, t32 / tb3, e32, eb3,
                                                                               fl / t32 / 1e9, fl / tb3 / 1e9))
    gg = torch.randn(n, s, s, s, co, generator=g).to(dev)
    w32 = timed(lambda: wgrad_fp32(x, gg, co))
    wb = timed(lambda: ops.conv3d_b3_wgrad(x, gg, co))
    d32, db = wgrad_fp32(x, gg, co), ops.conv3d_b3_wgrad(x, gg, co)
    ed = float((d32 - db).abs().max()) / float(d32.abs().max())
    print("%-26s %10.3f %10.3f %8.2f %12s %12.2e   (%.0f -> %.0f TFLOP/s)" % ("   wgrad", w32, wb, w32 / wb, "(vs fp32:)", ed,
                                                                             fl / w32 / 1e9, fl / wb / 1e9))
