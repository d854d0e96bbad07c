"""CPU tier, only where the reference checkout exists (this container; never on the GPU box): the drop-in recipe of
INTEGRATION.md executed for real.  The reference's OWN ``model.py`` classes (FPN, Mask, proposal_layer ...) are built
twice -- once with its ``backbone`` / ``mask_branch`` / ``utils.non_max_suppression`` / ``RoI_Align`` and once with
cfun_amd's drop-ins swapped in at the module level -- loaded with the same state dict, and must agree.  The kernels
run through the HIP emulator build (CPU tensors)."""
import os
import sys

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (GPU box)")


@pytest.fixture()
def emu_direct(emu, monkeypatch):
    monkeypatch.setenv("CFUN_CONV_ALGO", "direct")     # module-sized graphs: direct kernels on the fiber emulator
    return emu


@pytest.fixture()
def ref(emu_direct):
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    if gdir not in sys.path:
        sys.path.insert(0, gdir)
    import gen_golden as gg                      # installs the nibabel / skimage / .cuda() shims and imports the reference
    return gg


def _tiny_cfg(gg, stage="beginning"):
    return gg.make_cfg(stage, 32, 16, MASK_POOL_SIZE=[32, 32, 32], POOL_SIZE=[4, 4, 4], UNET_MASK_BRANCH_CHANNEL=4,
                       TOP_DOWN_PYRAMID_SIZE=16, RPN_CONV_CHANNELS=16, FPN_CLASSIFY_FC_LAYERS_SIZE=16,
                       RPN_ANCHOR_SCALES=(16, 32))


def test_fpn_with_dropin_backbone(ref, emu_direct):
    """model.FPN (reference class, reference P-convs) on cfun_amd.backbone.P3D19 stages == on the reference's."""
    from cfun_amd import backbone as my_backbone
    cfg = _tiny_cfg(ref)
    torch.manual_seed(0)
    r_stages = ref.ref_backbone.P3D19(config=cfg)
    m_stages = my_backbone.P3D19(config=cfg)
    m_stages.load_state_dict(r_stages.state_dict(), strict=True)          # identical keys and shapes
    r_fpn = ref.ref_model.FPN(*r_stages.stages(), out_channels=16, config=cfg).eval()
    m_fpn = ref.ref_model.FPN(*m_stages.stages(), out_channels=16, config=cfg).eval()
    m_fpn.load_state_dict(r_fpn.state_dict(), strict=True)
    x = torch.randn(1, 1, 16, 32, 32)
    with torch.no_grad():
        rp2, rp3 = r_fpn(x)
        mp2, mp3 = m_fpn(x)
    np.testing.assert_allclose(mp2.numpy(), rp2.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(mp3.numpy(), rp3.numpy(), rtol=1e-4, atol=2e-5)


def test_mask_head_with_dropin_unet_and_roi_align(ref, emu_direct, monkeypatch):
    """model.Mask built by the reference's constructor with ``mask_branch`` and ``RoI_Align`` swapped for cfun_amd's
     == the untouched reference, same weights, eval mode."""
    from cfun_amd import mask_branch as my_mask_branch
    from cfun_amd import model as my_model
    rm = ref.ref_model
    torch.manual_seed(1)
    r_mask = rm.Mask(1, [32, 32, 32], 8, 4, "beginning").eval()
    monkeypatch.setattr(rm, "mask_branch", my_mask_branch)
    monkeypatch.setattr(rm, "RoI_Align", my_model.RoI_Align)
    m_mask = rm.Mask(1, [32, 32, 32], 8, 4, "beginning").eval()
    m_mask.load_state_dict(r_mask.state_dict(), strict=True)
    img = torch.randn(1, 1, 16, 32, 32)
    rois = torch.tensor([[0.0, 0.1, 0.1, 1.0, 0.9, 0.8], [0.2, 0.0, 0.3, 0.9, 0.6, 1.0]])   # [n,6] as in training
    with torch.no_grad():
        m_logits, m_probs = m_mask([img, img], rois)
        monkeypatch.undo()
        r_logits, r_probs = r_mask([img, img], rois)
    assert tuple(m_logits.shape) == tuple(r_logits.shape) == (2, 8, 32, 32, 32)
    assert float((m_logits - r_logits).abs().max()) < 1e-3
    assert float((m_probs - r_probs).abs().max()) < 5e-4


def test_proposal_layer_with_dropin_nms(ref, emu_direct, monkeypatch):
    """model.proposal_layer (reference function) with ``utils.non_max_suppression`` swapped for the HIP NMS."""
    from cfun_amd import utils as my_utils
    from oracle import formula
    cfg = _tiny_cfg(ref)
    shapes = ref.ref_model.compute_backbone_shapes(cfg, cfg.IMAGE_SHAPE)
    anchors = ref.ref_utils.generate_pyramid_anchors(cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS, shapes,
                                                     cfg.BACKBONE_STRIDES, cfg.RPN_ANCHOR_STRIDE).astype(np.float32)
    a = anchors.shape[0]
    probs = torch.softmax(torch.from_numpy(formula.uniform("dropin.logits", (1, a, 2), -3, 3)), dim=2)
    bbox = torch.from_numpy(formula.uniform("dropin.bbox", (1, a, 6), -1, 1))
    want = ref.ref_model.proposal_layer([probs.clone(), bbox.clone()], proposal_count=12, nms_threshold=0.7,
                                        anchors=torch.from_numpy(anchors), config=cfg)
    monkeypatch.setattr(ref.ref_utils, "non_max_suppression", my_utils.non_max_suppression)
    got = ref.ref_model.proposal_layer([probs.clone(), bbox.clone()], proposal_count=12, nms_threshold=0.7,
                                       anchors=torch.from_numpy(anchors), config=cfg)
    np.testing.assert_array_equal(got.numpy(), want.numpy())
