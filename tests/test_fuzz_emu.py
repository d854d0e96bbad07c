"""CPU tier: a small randomised sweep of the Winograd convolution kernels on the HIP emulator -- both forms (x axis only /
x and y), every loop variant the channel count selects (16- and 32-wide tiles on the two-waves-per-SIMD loop, 48-wide on
the double-buffered one), odd spatial sizes, odd channel-chunk counts, split-K or not, random epilogues, the statistics
epilogue and the input prologue where the shape has them -- forward, data gradient and weight gradient against torch
(kernel_cases.check_conv).  Seeds are fixed; the GPU tier runs the larger sweep (test_fuzz_gpu.py)."""
import numpy as np
import pytest

import kernel_cases as kc


@pytest.mark.parametrize("seed", range(10))
def test_wino_fuzz(emu, seed):
    rng = np.random.default_rng(900 + seed)
    algo = kc.ALGO_WINO2 if seed % 2 == 0 else kc.ALGO_WINO
    ci = int(rng.choice([8, 12, 16, 20, 36, 40]))
    co = int(rng.choice([16, 24, 32, 40, 48, 64]))
    dhw = tuple(int(v) for v in rng.integers(3, 10, 3))
    n = int(rng.choice([1, 2]))
    kw = dict(algo=algo, seed=seed)
    if rng.random() < 0.5:
        kw.update(scale=True, per_n=bool(rng.random() < 0.5))
    if rng.random() < 0.5:
        kw["shift"] = True
    if rng.random() < 0.4:
        kw["res"] = True
    if rng.random() < 0.3:
        kw["pad"] = (0, 1, 1)          # a depth slab that arrives with its halo planes
        dhw = (dhw[0] + 2,) + dhw[1:]
    kw["act"] = int(rng.choice([kc.ACT_NONE, kc.ACT_RELU, kc.ACT_LRELU]))
    kc.check_conv(emu, n, dhw, ci, co, (3, 3, 3), **kw)
