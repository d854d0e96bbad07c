"""CPU tier: every HIP kernel source, compiled for the host against tests/emu (fiber-based functional HIP
emulator, incl. the v_mfma_f32_16x16x4_f32 lane layout), checked through the C ABI against plain torch /
the oracle / the golden vectors at small sizes.  The GPU tier (test_kernels_gpu.py) runs the same cases on
the real library."""
import numpy as np
import pytest

import kernel_cases as kc
from conftest import load_golden


@pytest.mark.parametrize("name", sorted(kc.CONV_CASES))
def test_conv(emu, name):
    n, dhw, ci, co, k, kw = kc.CONV_CASES[name]
    kc.check_conv(emu, n, dhw, ci, co, k, **kw)


@pytest.mark.parametrize("name", sorted(kc.NORM_CASES))
def test_instnorm_lrelu(emu, name):
    kc.check_instnorm_lrelu(emu, *kc.NORM_CASES[name])


def test_fold_up2_kernels(emu):
    """cfun_fold_up2_fwd / _bwd against the fold written out as a tensor contraction (k = 3 and 5, padded parity groups)."""
    kc.check_fold_up2_kernels(emu)


def test_fold_up2_conv5(emu):
    kc.check_fold_up2(emu)


@pytest.mark.parametrize("algo", ["mfma", "direct"])
def test_fold_up2_conv3(emu, algo):
    a = kc.ALGO_MFMA if algo == "mfma" else kc.ALGO_DIRECT
    kc.check_fold_up2_conv3(emu, 8, 20, (3, 4, 5), a)
    kc.check_fold_up2_conv3(emu, 12, 40, (2, 3, 9), a)
    kc.check_fold_up2_conv3(emu, 20, 24, (2, 2, 5), a)   # packed remainder subtile (4 taps x 4 channels)


def test_elementwise(emu):
    kc.check_elementwise(emu)


def test_norm_passthrough(emu):
    kc.check_norm_passthrough(emu)


def test_roi_align_slabs(emu):
    kc.check_roi_align_slabs(emu)


def test_fc(emu):
    kc.check_fc(emu)


def test_maxpool(emu):
    kc.check_maxpool(emu)


def test_halo(emu):
    kc.check_halo(emu)


def test_roi_align_golden(emu):
    g = load_golden("roi_align")
    kc.check_roi_align(emu, g["fm"], g["boxes"], [int(v) for v in g["pool"]], g["gy"], g["out"], g["fm_grad"])


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e", "f", "tie"])
def test_nms_golden(emu, tag):
    g = load_golden("nms")
    thr, mx = g[tag + "_cfg"]
    kc.check_nms(emu, g[tag + "_boxes"], g[tag + "_scores"], float(thr), int(mx), g[tag + "_keep"])


def test_mask_losses_golden(emu):
    g = load_golden("losses")
    kc.check_mask_losses(emu, g["logits"], g["labels"], g)


def test_mask_losses_multi_segment(emu):
    """D > 18: the z-marching edge kernels cross a segment boundary (and end on a ragged one)."""
    rng = np.random.default_rng(4)
    logits = rng.normal(size=(1, 8, 37, 5, 6)).astype(np.float32)
    labels = rng.integers(0, 8, size=(1, 37, 5, 6)).astype(np.uint8)
    kc.check_mask_losses(emu, logits, labels)


def test_mask_losses_multi_tile(emu):
    """Two samples, three z-segments of the marching edge kernels (one ragged), non-cubic (y, x) extent, labels in blocks
    of three along x (flat target regions: the sqrt at 0 of App. A-13 is exercised on the target side)."""
    rng = np.random.default_rng(6)
    logits = rng.normal(size=(2, 8, 37, 13, 39)).astype(np.float32)
    labels = np.repeat(rng.integers(0, 8, size=(2, 37, 13, 13)), 3, axis=3).astype(np.uint8)
    kc.check_mask_losses(emu, logits, labels)


def test_mask_losses_3class(emu):
    rng = np.random.default_rng(0)
    logits = rng.normal(size=(1, 3, 6, 7, 8)).astype(np.float32)
    labels = rng.integers(0, 3, size=(1, 6, 7, 8)).astype(np.uint8)
    kc.check_mask_losses(emu, logits, labels)


def test_mask_target_labels(emu):
    kc.check_mask_target_labels(emu)


def test_weight_layouts(emu):
    kc.check_weight_layouts(emu)


def test_weight_scope(emu):
    """cfun_weight_prepare (one launch for all of a pass's weight operands) against the per-conv pack / transform path."""
    kc.check_weight_scope(emu)


def test_mask_losses_lits_golden(emu):
    kc.check_mask_losses_lits(emu, load_golden("losses_lits"))
