"""GPU tier: randomised parity sweeps of the bit-exact contracts and of the conv dispatcher against the oracle /
torch-CPU references -- ragged sizes, degenerate boxes, every kernel shape, channel counts that exercise the
remainder / packed / fused code paths.  Seeds are fixed: the sweep is deterministic."""
import numpy as np
import pytest
import torch

import kernel_cases as kc
from oracle import cfun_oracle as orc

pytestmark = pytest.mark.gpu


def _boxes(rng, n, dhw, smin, smax):
    c = rng.uniform(0, 1, (n, 3)) * np.array(dhw)
    s = rng.uniform(smin, smax, (n, 3))
    return np.concatenate([c - s / 2, c + s / 2], axis=1).astype(np.float32)


@pytest.mark.parametrize("seed", range(12))
def test_nms_fuzz(gpu, seed):
    """Keep lists bit-exact vs the numpy restatement: n from 1 to 1000, clustered boxes (many overlaps), thresholds
    0.1-0.9, max_num below / above the survivor count."""
    from cfun_amd import ops
    rng = np.random.default_rng(100 + seed)
    n = int(rng.choice([1, 2, 7, 63, 64, 65, 300, 999, 1000]))
    dhw = (128, 256, 256)
    boxes = _boxes(rng, n, dhw, 4, 120)
    if n > 10:                                   # clusters of near-duplicates
        k = n // 3
        boxes[:k] = boxes[k:2 * k] + rng.normal(0, 1.5, (k, 6)).astype(np.float32)
    scores = rng.permutation(n).astype(np.float32) / n + rng.uniform(0, 1e-4)      # tie-free
    thr = float(rng.choice([0.1, 0.3, 0.5, 0.7, 0.9]))
    max_num = int(rng.choice([1, 5, 32, 500, 2000]))
    want = orc.nms(boxes, scores, thr, max_num)
    keep, count = ops.nms3d(torch.from_numpy(boxes).to(gpu), torch.from_numpy(scores).to(gpu), thr, max_num)
    np.testing.assert_array_equal(keep[:int(count.item())].cpu().numpy(), want)


@pytest.mark.parametrize("seed", range(6))
def test_roi_align_fuzz(gpu, seed):
    """Integer crop bounds bit-exact and values to 2e-6 for boxes that touch / cross the borders, are degenerate
    (lo >= hi: zero rows) or one voxel thin, on non-cubic maps and pools."""
    rng = np.random.default_rng(200 + seed)
    dhw = [(8, 16, 16), (5, 9, 13), (16, 32, 32), (1, 7, 7), (12, 12, 12), (3, 20, 6)][seed]
    pool = [(4, 4, 4), (3, 5, 2), (12, 12, 12), (2, 2, 2), (7, 7, 7), (1, 4, 3)][seed]
    c = int(rng.choice([1, 3, 4, 8]))
    fm = rng.normal(size=(c,) + dhw).astype(np.float32)
    lo = rng.uniform(-0.2, 0.9, (40, 3))
    hi = lo + rng.uniform(-0.05, 0.7, (40, 3))          # some inverted / empty
    boxes = np.concatenate([lo, hi], axis=1).astype(np.float32)
    boxes[0] = [0, 0, 0, 1, 1, 1]
    boxes[1] = [0.5, 0.5, 0.5, 0.5, 0.5, 0.5]
    gy = rng.normal(size=(40, c) + pool).astype(np.float32)
    kc.check_roi_align(gpu, fm, boxes, list(pool), gy=gy)


@pytest.mark.parametrize("seed", range(24))
def test_conv_dispatch_fuzz(gpu, seed):
    """Forward, data-gradient and weight-gradient of randomly drawn convolutions (every kernel shape of the path, odd
    spatial sizes, channel counts in {4..84} incl. 20 / 40 / 8, random epilogues) against torch on the CPU."""
    rng = np.random.default_rng(300 + seed)
    k, stride = [((3, 3, 3), 1), ((3, 3, 3), 2), ((1, 1, 1), 1), ((1, 1, 1), 2), ((1, 3, 3), 1), ((3, 1, 1), 1),
                 ((3, 3, 3), 1), ((3, 3, 3), 1)][seed % 8]
    ci = int(rng.choice([4, 8, 12, 16, 20, 36, 40, 64, 84]))
    co = int(rng.choice([4, 8, 16, 20, 24, 40, 48, 80]))
    dhw = tuple(int(v) for v in rng.integers(3, 19, 3))
    if stride == 2:
        dhw = tuple(2 * (v // 2 + 1) for v in dhw)
    n = int(rng.choice([1, 2, 3]))
    kw = dict(stride=stride, algo=kc.ALGO_AUTO, seed=seed)
    if stride == 2 and k == (1, 1, 1):
        kw["pad"] = (0, 0, 0)
    if rng.random() < 0.5:
        kw.update(scale=True, per_n=bool(rng.random() < 0.5))
    if rng.random() < 0.5:
        kw["shift"] = True
    if rng.random() < 0.4:
        kw["res"] = True
    kw["act"] = int(rng.choice([kc.ACT_NONE, kc.ACT_RELU, kc.ACT_LRELU]))
    kc.check_conv(gpu, n, dhw, ci, co, k, tol=1e-4, **kw)


@pytest.mark.parametrize("seed", range(4))
def test_mask_losses_fuzz(gpu, seed):
    """CE + edge losses and their fused backward on random logits / labels, non-cubic volumes, 8 and 3 classes."""
    rng = np.random.default_rng(400 + seed)
    c = [8, 3, 8, 3][seed]
    shape = [(2, 9, 12, 17), (1, 16, 16, 16), (3, 5, 20, 7), (2, 12, 6, 10)][seed]
    logits = (rng.normal(size=(shape[0], c) + shape[1:]) * 2).astype(np.float32)
    labels = rng.integers(0, c, shape).astype(np.uint8)
    kc.check_mask_losses(gpu, logits, labels)
