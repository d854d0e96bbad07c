"""GPU tier: the kernel parity cases of kernel_cases.py on the real libcfun_hip.so (cuda:0), plus larger
shapes (many workgroups, real channel counts) and the MFMA-vs-direct cross-check."""
import numpy as np
import pytest

import kernel_cases as kc
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(kc.CONV_CASES))
def test_conv(gpu, name):
    n, dhw, ci, co, k, kw = kc.CONV_CASES[name]
    kc.check_conv(gpu, n, dhw, ci, co, k, **kw)


@pytest.mark.parametrize("name", sorted(kc.CONV_CASES_LARGE))
def test_conv_large(gpu, name):
    n, dhw, ci, co, k, kw = kc.CONV_CASES_LARGE[name]
    kc.check_conv(gpu, n, dhw, ci, co, k, tol=5e-5, **kw)


@pytest.mark.parametrize("name", sorted(kc.NORM_CASES))
def test_instnorm_lrelu(gpu, name):
    kc.check_instnorm_lrelu(gpu, *kc.NORM_CASES[name])


def test_instnorm_lrelu_large(gpu):
    kc.check_instnorm_lrelu(gpu, 4, (48, 48, 48), 40)


def test_fold_up2_kernels(gpu):
    """cfun_fold_up2_fwd / _bwd against the fold written out as a tensor contraction (k = 3 and 5, padded parity groups)."""
    kc.check_fold_up2_kernels(gpu)


def test_fold_up2_conv5(gpu):
    kc.check_fold_up2(gpu)


@pytest.mark.parametrize("algo", ["mfma", "direct"])
def test_fold_up2_conv3(gpu, algo):
    a = kc.ALGO_MFMA if algo == "mfma" else kc.ALGO_DIRECT
    kc.check_fold_up2_conv3(gpu, 8, 20, (3, 4, 5), a)
    kc.check_fold_up2_conv3(gpu, 12, 40, (2, 3, 9), a)
    kc.check_fold_up2_conv3(gpu, 80, 40, (24, 24, 24))
    kc.check_fold_up2_conv3(gpu, 320, 160, (6, 6, 6), n=4)


def test_elementwise(gpu):
    kc.check_elementwise(gpu)


def test_norm_passthrough(gpu):
    kc.check_norm_passthrough(gpu)


def test_roi_align_slabs(gpu):
    kc.check_roi_align_slabs(gpu)


def test_fc(gpu):
    kc.check_fc(gpu)


def test_maxpool(gpu):
    kc.check_maxpool(gpu)


def test_halo(gpu):
    kc.check_halo(gpu)


def test_roi_align_golden(gpu):
    g = load_golden("roi_align")
    kc.check_roi_align(gpu, g["fm"], g["boxes"], [int(v) for v in g["pool"]], g["gy"], g["out"], g["fm_grad"])


def test_roi_align_mask_head_size(gpu):
    """96^3 crops of a raw image (C = 1), the mask-head shape (model.py:797)."""
    rng = np.random.default_rng(3)
    fm = rng.normal(size=(1, 64, 128, 128)).astype(np.float32)
    boxes = np.array([[0, .25, .25, 1, .75, .75], [0, .2, .3, 1, .7, .8], [.1, 0, 0, .6, .5, .5], [.5, .5, .5, .5, .6, .6]],
                     np.float32)
    kc.check_roi_align(gpu, fm, boxes, [96, 96, 96])


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e", "f", "tie"])
def test_nms_golden(gpu, tag):
    g = load_golden("nms")
    thr, mx = g[tag + "_cfg"]
    kc.check_nms(gpu, g[tag + "_boxes"], g[tag + "_scores"], float(thr), int(mx), g[tag + "_keep"])


def test_nms_large_vs_oracle(gpu):
    """4096 boxes (the kernel's maximum), heavy overlap; keep list bit-exact against the oracle."""
    from oracle import cfun_oracle as orc
    rng = np.random.default_rng(11)
    c = rng.uniform(0, 1, (4096, 3)) * np.array([256, 512, 512])
    s = rng.uniform(30, 160, (4096, 3))
    boxes = np.concatenate([np.clip(c - s / 2, 0, None), c + s / 2], 1).astype(np.float32)
    scores = (rng.permutation(4096).astype(np.float32) + 1) / 4097
    for thr, mx in ((0.7, 500), (0.3, 64), (0.5, 4096)):
        kc.check_nms(gpu, boxes, scores, thr, mx, orc.nms(boxes, scores, thr, mx))


def test_mask_losses_golden(gpu):
    g = load_golden("losses")
    kc.check_mask_losses(gpu, g["logits"], g["labels"], g)


def test_mask_losses_multi_segment(gpu):
    """D > 18: the z-marching edge kernels cross a segment boundary (and end on a ragged one)."""
    rng = np.random.default_rng(4)
    logits = rng.normal(size=(1, 8, 37, 5, 6)).astype(np.float32)
    labels = rng.integers(0, 8, size=(1, 37, 5, 6)).astype(np.uint8)
    kc.check_mask_losses(gpu, logits, labels)


def test_mask_losses_multi_tile(gpu):
    """Two samples, three z-segments of the marching edge kernels (one ragged), non-cubic (y, x) extent, labels in blocks
    of three along x (flat target regions: the sqrt at 0 of App. A-13 is exercised on the target side)."""
    rng = np.random.default_rng(6)
    logits = rng.normal(size=(2, 8, 37, 13, 39)).astype(np.float32)
    labels = np.repeat(rng.integers(0, 8, size=(2, 37, 13, 13)), 3, axis=3).astype(np.uint8)
    kc.check_mask_losses(gpu, logits, labels)


def test_mask_losses_3class(gpu):
    rng = np.random.default_rng(0)
    logits = rng.normal(size=(1, 3, 6, 7, 8)).astype(np.float32)
    labels = rng.integers(0, 3, size=(1, 6, 7, 8)).astype(np.uint8)
    kc.check_mask_losses(gpu, logits, labels)


def test_mask_losses_large(gpu):
    rng = np.random.default_rng(1)
    logits = (rng.normal(size=(2, 8, 40, 40, 40)) * 2).astype(np.float32)
    labels = np.repeat(np.repeat(np.repeat(rng.integers(0, 8, size=(2, 10, 10, 10)), 4, 1), 4, 2), 4, 3).astype(np.uint8)
    kc.check_mask_losses(gpu, logits, labels)


def test_mfma_equals_direct_on_device(gpu):
    """The two HIP conv families agree on the device itself (independent of any host reference)."""
    import torch
    from cfun_amd import ops
    from cfun_amd._lib import ALGO_DIRECT, ALGO_MFMA
    torch.manual_seed(0)
    x = torch.randn(2, 12, 20, 28, 40, device=gpu)
    w = torch.randn(80, 40, 3, 3, 3, device=gpu) / 33.0
    ys = []
    for algo in (ALGO_DIRECT, ALGO_MFMA):
        spec = ops.ConvSpec(k=(3, 3, 3), co=80, pad=(1, 1, 1), algo=algo)
        ys.append(ops.conv3d(x, ops.pack_weight(w), spec))
    kc.assert_close(ys[1], ys[0], "mfma vs direct", 1e-5)


def test_mask_target_labels(gpu):
    kc.check_mask_target_labels(gpu)


def test_weight_layouts(gpu):
    kc.check_weight_layouts(gpu)


def test_weight_scope(gpu):
    kc.check_weight_scope(gpu)


def test_mask_losses_lits_golden(gpu):
    kc.check_mask_losses_lits(gpu, load_golden("losses_lits"))
