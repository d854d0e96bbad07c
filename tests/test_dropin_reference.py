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


def test_whole_maskrcnn_training_step_with_all_dropins(ref, emu_direct, monkeypatch):
    """The reference's OWN ``MaskRCNN`` -- its build(), predict('training'), detection_target_layer, compute_losses and
    backward (model.py:1259-1304, 1391-1514, 984-1000) -- run twice on the same weights, inputs, randperm draws and
    Dropout3d masks: untouched, and with ALL FOUR swaps of INTEGRATION.md section 1 applied at once (``backbone``,
    ``mask_branch``, ``RoI_Align``, ``utils.non_max_suppression``).  The nine outputs, the six losses and the parameter
    gradients must agree."""
    import torch.nn as nn
    from cfun_amd import backbone as my_backbone
    from cfun_amd import mask_branch as my_mask_branch
    from cfun_amd import model as my_model
    from cfun_amd import utils as my_utils
    from oracle import formula
    rm = ref.ref_model
    cfg = _tiny_cfg(ref)
    cfg.MASK_SHAPE = cfg.MINI_MASK_SHAPE = (32, 32, 32)
    cfg.POST_NMS_ROIS_TRAINING = 64
    H, W, D = [int(v) for v in cfg.IMAGE_SHAPE[:3]]
    lab = np.zeros((D, H, W), np.int64)
    for k in range(1, 8):
        lab[:, :, (k - 1) * W // 7:k * W // 7] = k
    hu = np.where(lab == 0, -1000.0, (lab - 1) * 50.0) + formula.uniform("dropin.noise", (D, H, W), -50, 50)
    image = torch.from_numpy(((hu - hu.mean()) / hu.std()).astype(np.float32))[None, None]
    gt_masks = torch.from_numpy(np.stack([(lab == k) for k in range(8)], axis=0).astype(np.float32))[None]
    gt_boxes = torch.from_numpy(np.tile(np.array([0, 0, 0, D, H, W], np.float32), (7, 1)))[None]
    gt_ids = torch.from_numpy(np.arange(1, 8, dtype=np.int32))[None]

    def build():
        torch.manual_seed(5)
        net = rm.MaskRCNN(cfg, "/tmp/cfun_logs", test_flag=False)
        with torch.no_grad():                       # keep the proposals near their anchors so that positives exist
            net.rpn.conv_bbox.weight.mul_(0.05)
        return net

    def run(net, drop_masks):
        it = iter(drop_masks) if drop_masks is not None else None
        rec = []
        orig_fwd = nn.Dropout3d.forward

        def fwd(mod, inp):                          # the reference's Dropout3d: record, or replay the recorded masks
            if not mod.training:
                return inp
            if it is None:
                out = orig_fwd(mod, inp)
                n, c = inp.shape[:2]
                rec.append(torch.where(out.reshape(n, c, -1).abs().amax(-1) > 0, torch.full((n, c), 2.5), torch.zeros(n, c)))
                return out
            m = next(it)
            return inp * m[:, :, None, None, None]
        monkeypatch.setattr(nn.Dropout3d, "forward", fwd)
        with ref.PermRecorder():
            outs = net.predict([image, None, gt_ids, gt_boxes, gt_masks], "training")
        monkeypatch.setattr(nn.Dropout3d, "forward", orig_fwd)
        a = net.anchors.shape[0]
        rpn_match = torch.zeros((1, a, 1), dtype=torch.int32)
        rpn_match[0, ::5, 0] = 1
        rpn_match[0, 1::5, 0] = -1
        rpn_bbox = torch.from_numpy(formula.uniform("dropin.rpnb", (1, cfg.RPN_TRAIN_ANCHORS_PER_IMAGE, 6), -1, 1))
        losses = rm.compute_losses(rpn_match, rpn_bbox, outs[0], outs[1], outs[2], outs[3], outs[4], outs[5], outs[6],
                                   outs[7], outs[8], "beginning")
        w = cfg.LOSS_WEIGHTS
        keys = ("rpn_class_loss", "rpn_bbox_loss", "mrcnn_class_loss", "mrcnn_bbox_loss", "mrcnn_mask_loss",
                "mrcnn_mask_edge_loss")
        sum(w[k] * l for k, l in zip(keys, losses)).backward()
        return outs, losses, rec

    r_net = build()
    r_outs, r_losses, masks = run(r_net, None)
    assert r_outs[8].numel() > 0, "test setup: the reference's heads were skipped (no positive RoI)"
    monkeypatch.setattr(rm, "backbone", my_backbone)
    monkeypatch.setattr(rm, "mask_branch", my_mask_branch)
    monkeypatch.setattr(rm, "RoI_Align", my_model.RoI_Align)
    monkeypatch.setattr(ref.ref_utils, "non_max_suppression", my_utils.non_max_suppression)
    m_net = build()
    m_net.load_state_dict(r_net.state_dict(), strict=True)
    unet = m_net.mask.modified_u_net
    assert type(unet).__module__.startswith("cfun_amd") and type(m_net.fpn.C1).__module__.startswith("cfun_amd")
    unet.dropout_masks = masks                       # the drop-in U-Net takes the five masks instead of torch's RNG
    m_outs, m_losses, _ = run(m_net, masks)
    names = ("rpn_class_logits", "rpn_bbox", "target_class_ids", "mrcnn_class_logits", "target_deltas", "mrcnn_bbox",
             "target_mask", "mrcnn_mask", "mrcnn_mask_logits")
    for nm, a, b in zip(names, m_outs, r_outs):
        assert tuple(a.shape) == tuple(b.shape), nm
        tol = 1e-3 if "mask" in nm else 2e-5
        assert float((a.double() - b.double()).abs().max()) <= tol * max(1.0, float(b.double().abs().max())), nm
    for a, b in zip(m_losses, r_losses):
        assert abs(float(a) - float(b)) <= 1e-4 * max(abs(float(b)), 1e-3)
    rp, mp = dict(r_net.named_parameters()), dict(m_net.named_parameters())
    checked = 0
    for k, p in rp.items():
        if p.grad is None:
            assert mp[k].grad is None or float(mp[k].grad.abs().max()) == 0.0, k
            continue
        e = float((mp[k].grad - p.grad).norm() / (p.grad.norm() + 1e-30))
        assert e < (6e-2 if k.startswith("mask.") else 1e-3), (k, e)     # (U-Net: the 32^3 toy configuration's fp32 noise floor)
        checked += 1
    assert checked >= 90
