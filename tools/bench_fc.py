#!/usr/bin/env python3
"""Classifier-head GEMM (csrc/fc.hip) at the heart shapes: conv1 = [R x 221 184] . [221 184 x 128], 113 MB weight streamed
once per pass.  Prints ms and GB/s (weight bytes / time, the algorithmic traffic) for forward, dW and dx, next to the library
GEMM through torch (F.linear) for the same shapes.    python tools/bench_fc.py [R ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfun_amd import _lib  # noqa: E402
from cfun_amd._lib import check, ptr  # noqa: E402


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    K, O = 128 * 12 ** 3, 128
    w = torch.randn(O, K, device=dev) * 0.01
    wbytes = 4.0 * O * K
    for R in [int(v) for v in sys.argv[1:]] or [12, 64]:
        x = torch.randn(R, K, device=dev)
        g = torch.randn(R, O, device=dev)
        y = torch.empty(R, O, device=dev)
        dw = torch.empty_like(w)
        dx = torch.empty_like(x)
        ws = _lib.workspace(lib.cfun_fc_workspace_bytes(R, K, O), x)
        st = _lib.stream(x)
        tf = timeit(lambda: check(lib.cfun_fc_fwd(ptr(x), ptr(w), None, None, ptr(y), R, K, O, 0, ptr(ws), ws.numel(), st), "f"))
        tw = timeit(lambda: check(lib.cfun_fc_bwd_weight(ptr(x), ptr(g), ptr(dw), R, K, O, st), "w"))
        td = timeit(lambda: check(lib.cfun_fc_bwd_data(ptr(g), ptr(w), ptr(dx), R, K, O, st), "d"))
        ref = torch.nn.functional.linear(x, w)
        err = float((y - ref).abs().max() / ref.abs().max())
        tl = timeit(lambda: torch.nn.functional.linear(x, w))
        tlw = timeit(lambda: g.t() @ x)
        tld = timeit(lambda: g @ w)
        print("R=%-3d fc_fwd %.3f ms (%.0f GB/s)  fc_dW %.3f ms (%.0f GB/s)  fc_dx %.3f ms (%.0f GB/s) | torch fwd %.3f dW %.3f dx "
              "%.3f ms | fwd err %.1e" % (R, tf, wbytes / tf / 1e6, tw, wbytes / tw / 1e6, td, wbytes / td / 1e6, tl, tlw, tld, err))


if __name__ == "__main__":
    main()
