"""CPU tier: small randomised sweeps on the HIP emulator -- NMS keep lists and RoIAlign (bit-exact contracts vs the oracle), and
the Winograd convolution kernels: both forms (x axis only /
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


def _boxes(rng, n, dhw, smin, smax):
    c = rng.uniform(0, 1, (n, 3)) * np.array(dhw)
    s = rng.uniform(smin, smax, (n, 3))
    return np.concatenate([c - s / 2, c + s / 2], axis=1).astype(np.float32)


@pytest.mark.parametrize("seed", range(6))
def test_nms_fuzz(emu, seed):
    """Keep lists bit-exact vs the oracle's restatement of utils.non_max_suppression: clustered near-duplicates, thresholds
    0.1 - 0.9, max_num below / above the survivor count, n around the 64-lane boundaries."""
    import torch
    from cfun_amd import ops
    from oracle import cfun_oracle as orc
    rng = np.random.default_rng(700 + seed)
    n = int([1, 2, 63, 65, 130, 300][seed])
    boxes = _boxes(rng, n, (64, 128, 128), 4, 60)
    if n > 10:
        k = n // 3
        boxes[:k] = boxes[k:2 * k] + rng.normal(0, 1.5, (k, 6)).astype(np.float32)
    scores = rng.permutation(n).astype(np.float32) / n + rng.uniform(0, 1e-4)      # tie-free
    thr = float(rng.choice([0.1, 0.3, 0.5, 0.7, 0.9]))
    max_num = int(rng.choice([1, 5, 32, 500]))
    want = orc.nms(boxes, scores, thr, max_num)
    keep, count = ops.nms3d(torch.from_numpy(boxes).to(emu), torch.from_numpy(scores).to(emu), thr, max_num)
    np.testing.assert_array_equal(keep[:int(count.item())].cpu().numpy(), want)


@pytest.mark.parametrize("seed", range(3))
def test_roi_align_fuzz(emu, seed):
    """Integer crop bounds bit-exact and values to 2e-6 for boxes that cross the borders, are inverted / empty or one voxel
    thin, on non-cubic maps and pools; gradients against the oracle."""
    rng = np.random.default_rng(800 + seed)
    dhw = [(5, 9, 13), (1, 7, 7), (6, 4, 10)][seed]
    pool = [(3, 5, 2), (2, 2, 2), (1, 4, 3)][seed]
    c = int(rng.choice([1, 4, 8]))
    fm = rng.normal(size=(c,) + dhw).astype(np.float32)
    lo = rng.uniform(-0.2, 0.9, (12, 3))
    hi = lo + rng.uniform(-0.05, 0.7, (12, 3))
    boxes = np.concatenate([lo, hi], axis=1).astype(np.float32)
    boxes[0] = [0, 0, 0, 1, 1, 1]
    boxes[1] = [0.5, 0.5, 0.5, 0.5, 0.5, 0.5]
    gy = rng.normal(size=(12, c) + pool).astype(np.float32)
    kc.check_roi_align(emu, fm, boxes, list(pool), gy=gy)
