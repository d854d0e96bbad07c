#!/usr/bin/env python3
"""Per-layer timing of the conv kernels at the cfg2 shapes (U-Net b=20, 4 RoIs at 96^3; FPN/RPN at 256x256x128).

    python tools/bench_layers.py [--filter l4] [--iters 5]

Times forward, data-gradient and weight-gradient C-ABI calls separately with HIP events on the launch stream and
prints ms + useful TFLOP/s (2*Ci*Co*taps*voxels) for each; used to A/B kernel variants inside one process."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfun_amd import _lib, ops  # noqa: E402
from cfun_amd._lib import ACT_NONE, check, ptr  # noqa: E402

# name, N, (D,H,W) of the stored input, Ci, Co, k, stride, mode   (mode: "", "up2", "fold3", "fold5")
B = 20
LAYERS = [
    ("c1_1 stem 1->20 @96", 4, (96, 96, 96), 1, B, 3, 1, ""),
    ("c1_2 / lrelu_conv_c1 20->20 @96", 4, (96, 96, 96), B, B, 3, 1, ""),
    ("l4.0 40->40 @96", 4, (96, 96, 96), 2 * B, 2 * B, 3, 1, ""),
    ("l3.3 up2 40->20 @48->96 (unfolded)", 4, (48, 48, 48), 2 * B, B, 3, 1, "up2"),
    ("l3.3 up2 40->20 @48->96 (folded)", 4, (48, 48, 48), 2 * B, B, 3, 1, "fold3"),
    ("c2 s2 20->40 @96->48", 4, (96, 96, 96), B, 2 * B, 3, 2, ""),
    ("nlc_c2 40->40 @48", 4, (48, 48, 48), 2 * B, 2 * B, 3, 1, ""),
    ("l3.0 80->80 @48", 4, (48, 48, 48), 4 * B, 4 * B, 3, 1, ""),
    ("l2.3 up2 80->40 @24->48 (unfolded)", 4, (24, 24, 24), 4 * B, 2 * B, 3, 1, "up2"),
    ("l2.3 up2 80->40 @24->48 (folded)", 4, (24, 24, 24), 4 * B, 2 * B, 3, 1, "fold3"),
    ("nlc_c3 80->80 @24", 4, (24, 24, 24), 4 * B, 4 * B, 3, 1, ""),
    ("l2.0 160->160 @24", 4, (24, 24, 24), 8 * B, 8 * B, 3, 1, ""),
    ("l1.3 up2 160->80 @12->24 (folded)", 4, (12, 12, 12), 8 * B, 4 * B, 3, 1, "fold3"),
    ("nlc_c4 160->160 @12", 4, (12, 12, 12), 8 * B, 8 * B, 3, 1, ""),
    ("l1.0 320->320 @12", 4, (12, 12, 12), 16 * B, 16 * B, 3, 1, ""),
    ("l0.3 up2 320->160 @6->12 (folded)", 4, (6, 6, 6), 16 * B, 8 * B, 3, 1, "fold3"),
    ("nlc_c5 320->320 @6", 4, (6, 6, 6), 16 * B, 16 * B, 3, 1, ""),
    ("out_upscale 5^3 8->8 @96->192 (folded)", 4, (96, 96, 96), 8, 8, 5, 1, "fold5"),
    ("conv3d_l4 1x1 40->8 @96", 4, (96, 96, 96), 2 * B, 8, 1, 1, ""),
    ("sparse-dropout 20->8 @96 (1 RoI)", 1, (96, 96, 96), B, 8, 3, 1, ""),
    ("sparse-dropout 8->20 @96 (1 RoI)", 1, (96, 96, 96), 8, B, 3, 1, ""),
    ("sparse-dropout 20->12 @96 (1 RoI)", 1, (96, 96, 96), B, 12, 3, 1, ""),
    ("sparse-dropout 12->20 @96 (1 RoI)", 1, (96, 96, 96), 12, B, 3, 1, ""),
    ("sparse-dropout 40->16 @48 (1 RoI)", 1, (48, 48, 48), 2 * B, 16, 3, 1, ""),
    ("sparse-dropout 16->40 @48 (1 RoI)", 1, (48, 48, 48), 16, 2 * B, 3, 1, ""),
    ("sparse-dropout 80->32 @24 (1 RoI)", 1, (24, 24, 24), 4 * B, 32, 3, 1, ""),
    ("sparse-dropout 32->80 @24 (1 RoI)", 1, (24, 24, 24), 32, 4 * B, 3, 1, ""),
    ("P2_conv2 128->128 @16x32x32", 1, (16, 32, 32), 128, 128, 3, 1, ""),
    ("rpn.conv_shared 128->256 @16x32x32", 1, (16, 32, 32), 128, 256, 3, 1, ""),
]


def build(name, n, dhw, ci, co, k, stride, mode, dev):
    w = torch.randn(co, ci, k, k, k, device=dev) / (ci * k ** 3) ** 0.5
    x = torch.randn(n, *dhw, ci, device=dev)
    pad = (k // 2,) * 3
    taps = k ** 3
    if mode == "fold3":
        cqp = (co + 15) // 16 * 16
        spec = ops.ConvSpec(k=(3, 3, 3), co=8 * cqp, pad=(1, 1, 1), d2s=True, d2s_cq=co, tap_skip=True)
        wp = ops.pack_weight(ops.fold_up2_weight(w, cqp))
        out_vox = n * 8 * dhw[0] * dhw[1] * dhw[2]
    elif mode == "fold5":
        spec = ops.ConvSpec(k=(3, 3, 3), co=8 * co, pad=(1, 1, 1), d2s=True)
        wp = ops.pack_weight(ops.fold_up2_weight(w))
        out_vox = n * 8 * dhw[0] * dhw[1] * dhw[2]
    else:
        spec = ops.ConvSpec(k=(k, k, k), co=co, stride=stride, pad=pad, up2=(mode == "up2"))
        wp = ops.pack_weight(w)
        sh = 2 if mode == "up2" else 1
        out_vox = n * (dhw[0] * sh // stride) * (dhw[1] * sh // stride) * (dhw[2] * sh // stride)
    flops = 2.0 * ci * co * taps * out_vox      # useful FLOPs of the ORIGINAL (unfolded) convolution
    return x, wp, spec, flops


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--filter", default="")
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    print("%-44s %9s %9s %9s   %s" % ("layer", "fwd ms", "dgrad ms", "wgrad ms", "useful TFLOP/s (fwd/dgrad/wgrad)"))
    tot = [0.0, 0.0, 0.0]
    for L in LAYERS:
        if args.filter and args.filter not in L[0]:
            continue
        x, wp, spec, flops = build(*L, dev)
        p = ops._params(spec, x.shape, False, False, False)
        if spec.d2s:
            y = torch.empty((p.N, 2 * p.Do, 2 * p.Ho, 2 * p.Wo, spec.d2s_cq or p.Co // 8), device=dev)
        else:
            y = torch.empty((p.N, p.Do, p.Ho, p.Wo, p.Co), device=dev)
        g = torch.randn_like(y)
        wpT = ops._transpose_pack(wp, p.Co)
        dx = torch.empty_like(x)
        dwp = torch.empty_like(wp)
        ws_d = _lib.workspace(lib.cfun_conv3d_bwd_data_workspace_bytes(C.byref(p)), x)
        ws_w = _lib.workspace(lib.cfun_conv3d_bwd_weight_workspace_bytes(C.byref(p)), x)
        st = _lib.stream(x)
        ws_f = _lib.workspace(lib.cfun_conv3d_fwd_workspace_bytes(C.byref(p)), x)
        t_f = timeit(lambda: check(lib.cfun_conv3d_fwd(ptr(x), ptr(wp), None, None, None, ptr(y), C.byref(p), ptr(ws_f), ws_f.numel(), st), "f"), args.iters)
        t_d = timeit(lambda: check(lib.cfun_conv3d_bwd_data(ptr(g), ptr(wpT), ptr(dx), C.byref(p), ptr(ws_d), ws_d.numel(), st), "d"), args.iters)
        t_w = timeit(lambda: check(lib.cfun_conv3d_bwd_weight(ptr(x), ptr(g), ptr(dwp), C.byref(p), ptr(ws_w), ws_w.numel(), st), "w"), args.iters)
        for i, t in enumerate((t_f, t_d, t_w)):
            tot[i] += t
        extra = ""
        if L[3] == 1:      # C_in = 1 stem: HBM-bound, algorithmic bytes = input + output + weights
            nbytes = 4.0 * (x.numel() + y.numel() + wp.numel())
            extra = "   fwd %.0f GB/s" % (nbytes / t_f / 1e6)
        print("%-44s %9.3f %9.3f %9.3f   %6.1f %6.1f %6.1f%s" % (L[0], t_f, t_d, t_w, flops / t_f / 1e9, flops / t_d / 1e9, flops / t_w / 1e9, extra))
    print("%-44s %9.3f %9.3f %9.3f" % ("sum", *tot))


if __name__ == "__main__":
    main()
