#!/usr/bin/env python3
"""Mask-head loss kernels at the cfg2 shapes (4 x 192^3 x 8 logits): softmax, CE + Sobel edge forward (with the saved
coefficient field) and the fused backward, timed with HIP events; GB/s = algorithmic bytes / time against 8 TB/s.
    python tools/bench_losses.py [n] [side]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfun_amd import ops  # noqa: E402


def timeit(fn, iters=7):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 192
    dev = torch.device("cuda:0")
    c = 8
    logits = torch.randn(n, s, s, s, c, device=dev).requires_grad_(True)
    labels = torch.randint(0, c, (n, s // 8, s // 8, s // 8), device=dev, dtype=torch.uint8)
    labels = labels.repeat_interleave(8, 1).repeat_interleave(8, 2).repeat_interleave(8, 3).contiguous()
    vox = n * s ** 3
    vo = n * (s - 2) ** 3
    probs = ops.softmax_channels(logits.detach())
    t_sm = timeit(lambda: ops.softmax_channels(logits.detach()))
    state = {}

    def fwd():
        state["l"] = ops.mask_losses(logits, probs, labels)
    t_f = timeit(fwd)

    def fb():
        fwd()
        logits.grad = None
        (state["l"][0] + state["l"][1]).backward()
    t_fb = timeit(fb)
    t_b = t_fb - t_f
    b_sm = 8.0 * vox * c
    b_f = 4.0 * vox * c * 2 + 2.0 * vox + 8.0 * vo * (c - 1)         # CE reads logits, edge reads probs + labels, writes dc
    b_b = 8.0 * vo * (c - 1) + 4.0 * vox * c * 2 + vox               # reads dc, probs, labels; writes dlogits
    print("softmax            %.3f ms  %5.0f GB/s (%.2f of 8 TB/s)" % (t_sm, b_sm / t_sm / 1e6, b_sm / t_sm / 8e9))
    print("CE + edge forward  %.3f ms  %5.0f GB/s (%.2f)" % (t_f, b_f / t_f / 1e6, b_f / t_f / 8e9))
    print("fused backward     %.3f ms  %5.0f GB/s (%.2f)   (forward+backward %.3f ms)" % (t_b, b_b / t_b / 1e6, b_b / t_b / 8e9, t_fb))
    # round 6: ONE pass each way (softmax + CE + edge -> probs + 2 losses; backward recomputes the coefficients)
    st2 = {}

    def fwd1():
        st2["l"] = ops.mask_losses_fused(logits, labels)
    t_f1 = timeit(fwd1)

    def fb1():
        fwd1()
        logits.grad = None
        (st2["l"][0] + st2["l"][1]).backward()
    t_fb1 = timeit(fb1)
    t_b1 = t_fb1 - t_f1
    ub = 8.0 * n * s * (s - 2) * (s - 2) * (c - 1)
    b_f1 = 4.0 * vox * c * 2 + vox + ub       # reads logits + labels, writes probs + the backward's operand field
    b_b1 = 4.0 * vox * c * 2 + vox + ub       # reads the field, probs + labels, writes dlogits
    print("one-pass forward   %.3f ms  %5.0f GB/s (%.2f)   (was softmax + CE + edge: %.3f ms)" % (t_f1, b_f1 / t_f1 / 1e6, b_f1 / t_f1 / 8e9, t_sm + t_f))
    print("one-pass backward  %.3f ms  %5.0f GB/s (%.2f)   (was %.3f ms; forward+backward %.3f ms, was %.3f)"
          % (t_b1, b_b1 / t_b1 / 1e6, b_b1 / t_b1 / 8e9, t_b, t_fb1, t_sm + t_fb))


if __name__ == "__main__":
    main()
