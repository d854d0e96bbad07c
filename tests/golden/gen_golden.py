#!/usr/bin/env python3
"""Golden-vector generator.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

Imports the reference's own Python (read-only, PYTHONDONTWRITEBYTECODE=1) with the three
shims of SURVEY.md section 8(c) -- stub nibabel/skimage, identity .cuda(), GPU_COUNT=0 -- runs
its functions/modules on closed-form inputs (oracle/formula.py) and stores inputs + outputs as
small .npz fixtures next to this script.  Only DATA is stored; no reference source travels.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py [case ...]
"""
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import formula  # noqa: E402
from oracle import cfun_oracle as orc  # noqa: E402  (only nearest_resize, for the skimage stub)

REF = "/root/reference"


def install_shims():
    for name in ("nibabel", "skimage", "skimage.transform"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["skimage"].__version__ = "0.19.0"
    sys.modules["skimage"].transform = sys.modules["skimage.transform"]

    def resize(image, output_shape, order=1, mode="constant", preserve_range=True, **kw):
        assert order == 0, "only the order=0 path is on the hot path (model.py:490)"
        t = torch.from_numpy(np.asarray(image))
        lead = t.shape[:-3]
        assert tuple(output_shape[:len(lead)]) == tuple(lead)
        return orc.nearest_resize(t, tuple(output_shape[-3:])).numpy()

    sys.modules["skimage.transform"].resize = resize
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)


install_shims()
import backbone as ref_backbone  # noqa: E402
import mask_branch as ref_mask_branch  # noqa: E402
import model as ref_model  # noqa: E402
import utils as ref_utils  # noqa: E402
import heart_main as ref_heart  # noqa: E402


def make_cfg(stage, max_dim, min_dim, **over):
    ns = dict(GPU_COUNT=0, IMAGE_MAX_DIM=max_dim, IMAGE_MIN_DIM=min_dim)
    ns.update(over)
    return type("Cfg", (ref_heart.HeartConfig,), ns)(stage)


def load_formula(module, gain=1.0):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    arrs = formula.fill_state_dict(shapes, gain)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in arrs.items()}, strict=True)
    return shapes


def shapes_to_npz(shapes):
    keys = sorted(shapes)
    return dict(sd_keys=np.array(keys), sd_shapes=np.array([",".join(map(str, shapes[k])) for k in keys]))


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024.0))


# ------------------------------------------------------------------------------------------


def case_nms():
    out = {}
    rng = np.random.default_rng(1)

    def boxes(n, dhw=(128, 256, 256), smin=8, smax=96):
        c = rng.uniform(0, 1, (n, 3)) * np.array(dhw)
        s = rng.uniform(smin, smax, (n, 3))
        lo = np.clip(c - s / 2, 0, dhw)
        hi = np.clip(c + s / 2, 0, dhw)
        return np.concatenate([lo, hi], 1).astype(np.float32)

    def scores(n):
        s = rng.permutation(n).astype(np.float32) / n  # tie-free
        return (s * 0.98 + 0.01).astype(np.float32)

    cases = [("a", boxes(1000), scores(1000), 0.7, 500),
             ("b", boxes(64, smin=40, smax=120), scores(64), 0.3, 32),
             ("c", boxes(300, smin=60, smax=160), scores(300), 0.7, 5),       # max_num early break
             ("d", boxes(1), scores(1), 0.7, 500),
             ("e", np.zeros((0, 6), np.float32), np.zeros((0,), np.float32), 0.7, 500),
             ("f", boxes(1000, smin=90, smax=200), scores(1000), 0.5, 500)]    # heavy suppression
    tie = np.array([[0, 0, 0, 10, 10, 10], [0, 0, 0, 10, 10, 10], [20, 20, 20, 30, 30, 30]], np.float32)
    cases.append(("tie", tie, np.array([0.5, 0.5, 0.4], np.float32), 0.7, 500))
    for tag, b, s, thr, mx in cases:
        keep = ref_utils.non_max_suppression(b.copy(), s.copy(), thr, mx)
        out["%s_boxes" % tag] = b
        out["%s_scores" % tag] = s
        out["%s_cfg" % tag] = np.array([thr, mx], np.float64)
        out["%s_keep" % tag] = keep
        print("nms", tag, b.shape[0], "->", keep.shape[0])
    save("nms", **out)


def case_anchors():
    out = {}
    for tag, (h, w, d) in (("cfg0", (64, 64, 32)), ("odd", (96, 64, 48))):
        cfg = make_cfg("beginning", 64, 32)
        shapes = ref_model.compute_backbone_shapes(cfg, np.array([h, w, d, 1]))
        a = ref_utils.generate_pyramid_anchors(cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS, shapes,
                                               cfg.BACKBONE_STRIDES, cfg.RPN_ANCHOR_STRIDE)
        out[tag + "_hwd"] = np.array([h, w, d])
        out[tag + "_shapes"] = shapes
        out[tag + "_anchors"] = a
    save("anchors", **out)


def case_roi_align():
    fm = torch.from_numpy(formula.uniform("roi.fm", (3, 6, 10, 12), -1, 1))
    boxes = torch.tensor([
        [0.10, 0.10, 0.10, 0.70, 0.80, 0.90],
        [0.00, 0.00, 0.00, 1.00, 1.00, 1.00],     # whole map
        [0.50, 0.50, 0.50, 0.50, 0.50, 0.50],     # z: floor(3.0)=3, ceil(3.0)=3 -> empty -> zeros
        [0.34, 0.21, 0.26, 0.49, 0.29, 0.33],     # single-element crops on some axes (in==1)
        [0.90, 0.90, 0.90, 1.20, 1.30, 1.10],     # upper bounds beyond the map (clamped by slicing)
        [0.60, 0.20, 0.10, 0.40, 0.90, 0.80],     # inverted in z -> empty -> zeros
        [0.17, 0.33, 0.41, 0.83, 0.67, 0.59],
        [0.00, 0.95, 0.00, 0.20, 1.00, 0.10],
    ], dtype=torch.float32)
    pool = [4, 5, 3]
    fm_g = fm.clone().requires_grad_(True)
    out = ref_model.RoI_Align(fm_g, pool, boxes.clone())
    gy = torch.from_numpy(formula.uniform("roi.gy", tuple(out.shape), -1, 1))
    (out * gy).sum().backward()
    # pyramid: two maps, boxes straddling the level threshold
    p2 = torch.from_numpy(formula.uniform("roi.p2", (2, 8, 8, 8), -1, 1))
    p3 = torch.from_numpy(formula.uniform("roi.p3", (2, 4, 4, 4), -1, 1))
    pb = torch.tensor([
        [0.10, 0.10, 0.10, 0.40, 0.40, 0.40],     # vol .027 -> level 2
        [0.00, 0.00, 0.00, 0.90, 0.90, 0.90],     # level 3
        [0.20, 0.20, 0.20, 0.55, 0.56, 0.57],     # ~0.045 near threshold
        [0.20, 0.20, 0.20, 0.55, 0.55, 0.55],     # 0.0429
        [0.50, 0.10, 0.30, 0.95, 0.60, 0.70],
        [0.05, 0.05, 0.05, 0.30, 0.30, 0.30],
    ], dtype=torch.float32)
    lv = (4 + (1. / 3.) * ref_model.log2((pb[:, 4] - pb[:, 1]) * (pb[:, 5] - pb[:, 2]) * (pb[:, 3] - pb[:, 0])))
    lv = lv.round().int().clamp(2, 3)
    pooled = ref_model.pyramid_roi_align([pb.clone().unsqueeze(0).squeeze(0), p2.unsqueeze(0), p3.unsqueeze(0)],
                                         [3, 3, 3])
    save("roi_align", fm=fm.numpy(), boxes=boxes.numpy(), pool=np.array(pool), out=out.detach().numpy(),
         gy=gy.numpy(), fm_grad=fm_g.grad.numpy(), p2=p2.numpy(), p3=p3.numpy(), pboxes=pb.numpy(),
         plevels=lv.numpy(), ppool=np.array([3, 3, 3]), pooled=pooled.detach().numpy())


def case_fpn_rpn():
    cfg = make_cfg("beginning", 32, 16)
    p3d = ref_backbone.P3D19(config=cfg)
    c1, c2, c3 = p3d.stages()
    fpn = ref_model.FPN(c1, c2, c3, out_channels=cfg.TOP_DOWN_PYRAMID_SIZE, config=cfg)
    rpn = ref_model.RPN(len(cfg.RPN_ANCHOR_RATIOS), cfg.RPN_ANCHOR_STRIDE, cfg.TOP_DOWN_PYRAMID_SIZE,
                        cfg.RPN_CONV_CHANNELS)
    holder = nn.Module()
    holder.fpn = fpn
    holder.rpn = rpn
    shapes = load_formula(holder)
    holder.eval()
    x = torch.from_numpy(formula.uniform("fpn.x", (1, 1, 16, 32, 32), -2, 2)).requires_grad_(True)
    feats = {}
    h = fpn.C1(x); feats["c1"] = h
    h = fpn.C2(h); feats["c2"] = h
    h = fpn.C3(h); feats["c3"] = h
    p2, p3 = fpn(x)
    outs = dict(p2=p2, p3=p3)
    for tag, p in (("l2", p2), ("l3", p3)):
        lg, pr, bb = rpn(p)
        outs["rpn_logits_" + tag] = lg
        outs["rpn_probs_" + tag] = pr
        outs["rpn_bbox_" + tag] = bb
    loss = 0
    for k in ("p2", "p3", "rpn_logits_l2", "rpn_bbox_l2", "rpn_logits_l3", "rpn_bbox_l3"):
        g = torch.from_numpy(formula.uniform("fpn.g." + k, tuple(outs[k].shape), -1, 1))
        loss = loss + (outs[k] * g).sum()
    loss.backward()
    grads = {}
    for k in ("fpn.C1.0.weight", "fpn.C1.0.bias", "fpn.C2.0.conv1.weight", "fpn.C2.0.conv2.weight",
              "fpn.C2.1.conv3.weight", "fpn.C2.0.downsample.0.weight", "fpn.C3.2.conv3.weight",
              "fpn.C3.1.conv4.bias", "fpn.P2_conv2.weight", "fpn.P3_conv1.weight", "fpn.P2_conv1.bias",
              "rpn.conv_shared.weight", "rpn.conv_class.weight", "rpn.conv_bbox.bias"):
        g = dict(holder.named_parameters())[k].grad.numpy()
        if g.size > 100000:  # big tensors: keep every 4th output and input channel
            grads["grad4:" + k] = g[::4, ::4].copy()
        else:
            grads["grad:" + k] = g
    arrs = dict(x=x.detach().numpy(), x_grad=x.grad.numpy(), **shapes_to_npz(shapes))
    arrs.update({k: v.detach().numpy() for k, v in feats.items()})
    arrs.update({k: v.detach().numpy() for k, v in outs.items()})
    arrs.update(grads)
    save("fpn_rpn", **arrs)


class DropRecorder:
    """Records Dropout3d channel multipliers (out/in per (n,c)) -- SURVEY.md App. A-4."""

    def __init__(self):
        self.masks = []
        self._orig = nn.Dropout3d.forward

    def __enter__(self):
        rec = self

        def fwd(mod, inp):
            out = rec._orig(mod, inp)
            if mod.training:
                n, c = inp.shape[:2]
                m = torch.where(out.reshape(n, c, -1).abs().amax(-1) > 0,
                                torch.full((n, c), 1.0 / (1.0 - mod.p)), torch.zeros(n, c))
                rec.masks.append(m)
            return out

        nn.Dropout3d.forward = fwd
        return self

    def __exit__(self, *a):
        nn.Dropout3d.forward = self._orig


def unet_grads_fp64(net, x, masks, gy, keys):
    """Gradients of sum(net(x) * gy) with the reference module converted to fp64; Dropout3d replays ``masks``."""
    import copy
    net64 = copy.deepcopy(net).double()
    for p in net64.parameters():
        p.grad = None
    it = iter(masks)
    orig = nn.Dropout3d.forward
    nn.Dropout3d.forward = lambda mod, inp: inp * next(it).to(inp.dtype)[:, :, None, None, None] if mod.training else inp
    try:
        x64 = x.double().requires_grad_(True)
        (net64(x64) * gy.double()).sum().backward()
    finally:
        nn.Dropout3d.forward = orig
    params = dict(net64.named_parameters())
    out = {k: params[k].grad.numpy() for k in keys if params[k].grad is not None}
    out["x"] = x64.grad.numpy()
    return out


def case_unet():
    for tag, stage, b, n, size, ncls, train in (("unet_beginning_eval", "beginning", 4, 2, 32, 8, False),
                                                ("unet_beginning_train", "beginning", 4, 2, 32, 8, True),
                                                ("unet_finetune_train", "finetune", 2, 1, 32, 8, True),
                                                ("unet_lits_eval", "beginning", 4, 1, 32, 3, False)):
        net = ref_mask_branch.Modified3DUNet(1, ncls, stage, b)
        shapes = load_formula(net, gain=1.0)
        net.train(train)
        torch.manual_seed(7)
        x = torch.from_numpy(formula.uniform(tag + ".x", (n, 1, size, size, size), -2, 2)).requires_grad_(True)
        with DropRecorder() as rec:
            y = net(x)
        arrs = dict(x=x.detach().numpy(), stage=np.array(stage), b=np.array(b), ncls=np.array(ncls),
                    **shapes_to_npz(shapes))
        if train:
            assert len(rec.masks) == 5
            for i, m in enumerate(rec.masks):
                arrs["drop%d" % i] = m.numpy()
            gy = torch.from_numpy(formula.uniform(tag + ".gy", tuple(y.shape), -1, 1))
            (y * gy).sum().backward()
            arrs["x_grad"] = x.grad.numpy()
            params = dict(net.named_parameters())
            gkeys = ("conv3d_c1_1.weight", "conv3d_c1_2.weight", "lrelu_conv_c1.1.weight", "conv3d_c2.weight",
                     "norm_lrelu_conv_c3.2.weight", "norm_lrelu_conv_c5.2.weight",
                     "norm_lrelu_upscale_conv_norm_lrelu_l0.3.weight", "conv3d_l0.weight",
                     "conv_norm_lrelu_l2.0.weight", "conv3d_l3.weight",
                     "norm_lrelu_upscale_conv_norm_lrelu_l3.3.weight", "conv_norm_lrelu_l4.0.weight",
                     "conv3d_l4.weight", "ds2_1x1_conv3d.weight", "ds3_1x1_conv3d.weight",
                     "out_upscale_conv.1.weight")
            for k in gkeys:
                if params[k].grad is not None:
                    arrs["grad:" + k] = params[k].grad.numpy()
            # the SAME reference module in fp64 (same weights, input, recorded dropout masks): the yardstick that
            # says how far the reference's own fp32 gradients are from exact -- tests hold the HIP path to that bound
            g64 = unet_grads_fp64(net, x.detach(), rec.masks, gy, gkeys)
            arrs["x_grad64"] = g64.pop("x").astype(np.float32)
            for k, v in g64.items():
                arrs["grad64:" + k] = v.astype(np.float32)
        yn = y.detach().numpy()
        if yn.size > 600000:  # 'finetune' 8x64^3: keep a strided subsample + fp64 checksums
            arrs["y_sub"] = yn[:, :, ::2, ::2, ::2].copy()
            arrs["y_sum"] = np.array([yn.astype(np.float64).sum(), np.abs(yn).astype(np.float64).sum()])
        else:
            arrs["y"] = yn
        save(tag, **arrs)


def case_losses():
    n, c, s = 2, 8, 12
    logits = torch.from_numpy(formula.uniform("loss.logits", (n, c, s, s, s), -3, 3)).requires_grad_(True)
    lab = (formula.uniform("loss.lab", (n, s, s, s), 0, 1) * c).astype(np.int64).clip(0, c - 1)
    # blocky labels so that Sobel responses of the target are non-trivial
    lab = np.repeat(np.repeat(np.repeat(lab[:, ::3, ::3, ::3], 3, 1), 3, 2), 3, 3)
    onehot = np.zeros((n, c, s, s, s), np.float64)
    for k in range(c):
        onehot[:, k] = (lab == k)
    target = torch.from_numpy(onehot)  # DoubleTensor like model.py:493
    ids = torch.tensor([3, 5])
    ce = ref_model.compute_mrcnn_mask_loss(target, ids, logits)
    ce.backward()
    g_ce = logits.grad.clone(); logits.grad = None
    probs = torch.softmax(logits, dim=1)
    el = ref_model.compute_mrcnn_mask_edge_loss(target, ids, probs)
    el.backward()
    g_edge = logits.grad.clone()
    probs_leaf = probs.detach().clone().requires_grad_(True)
    el2 = ref_model.compute_mrcnn_mask_edge_loss(target, ids, probs_leaf)
    el2.backward()
    save("losses", logits=logits.detach().numpy(), labels=lab.astype(np.uint8), ce=ce.detach().numpy(),
         ce_grad=g_ce.numpy(), edge=el.detach().numpy(), edge_grad_logits=g_edge.numpy(),
         edge_grad_probs=probs_leaf.grad.numpy())


def case_losses_lits():
    """LiTS fork: compute_mrcnn_mask_loss (class weights [1, 1, 100]) and compute_mrcnn_mask_edge_loss (MSE on the raw
    three Sobel responses) of LiTS_2017/model.py:907-979 on 2 RoIs x 3 classes x 10x12x14 voxels."""
    lits_model, _ = import_lits()
    n, c, dhw = 2, 3, (10, 12, 14)
    logits = torch.from_numpy(formula.uniform("lossl.logits", (n, c) + dhw, -3, 3)).requires_grad_(True)
    lab = (formula.uniform("lossl.lab", (n,) + dhw, 0, 1) * c).astype(np.int64).clip(0, c - 1)
    lab = np.repeat(np.repeat(np.repeat(lab[:, ::2, ::3, ::2], 2, 1), 3, 2), 2, 3)[:, :dhw[0], :dhw[1], :dhw[2]]
    onehot = np.zeros((n, c) + dhw, np.float64)
    for k in range(c):
        onehot[:, k] = (lab == k)
    target = torch.from_numpy(onehot)
    ids = torch.tensor([1, 2])
    ce = lits_model.compute_mrcnn_mask_loss(target, ids, logits)
    ce.backward()
    g_ce = logits.grad.clone(); logits.grad = None
    probs = torch.softmax(logits, dim=1)
    el = lits_model.compute_mrcnn_mask_edge_loss(target, ids, probs)
    el.backward()
    save("losses_lits", logits=logits.detach().numpy(), labels=lab.astype(np.uint8), ce=ce.detach().numpy(),
         ce_grad=g_ce.numpy(), edge=el.detach().numpy(), edge_grad_logits=logits.grad.numpy(),
         class_weights=np.array([1.0, 1.0, 100.0], np.float32))


def case_proposal():
    cfg = make_cfg("beginning", 64, 32)
    shapes = ref_model.compute_backbone_shapes(cfg, cfg.IMAGE_SHAPE)
    anchors = ref_utils.generate_pyramid_anchors(cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS, shapes,
                                                 cfg.BACKBONE_STRIDES, cfg.RPN_ANCHOR_STRIDE).astype(np.float32)
    a = anchors.shape[0]
    logits = formula.uniform("prop.logits", (1, a, 2), -3, 3)
    probs = torch.softmax(torch.from_numpy(logits), dim=2)
    bbox = torch.from_numpy(formula.uniform("prop.bbox", (1, a, 6), -2, 2))
    out = {}
    for tag, cnt in (("train", cfg.POST_NMS_ROIS_TRAINING), ("infer", 16)):
        rois = ref_model.proposal_layer([probs.clone(), bbox.clone()], proposal_count=cnt,
                                        nms_threshold=cfg.RPN_NMS_THRESHOLD, anchors=torch.from_numpy(anchors),
                                        config=cfg)
        out["rois_" + tag] = rois.numpy()
        out["count_" + tag] = np.array(cnt)
    save("proposal", anchors=anchors, probs=probs.numpy(), bbox=bbox.numpy(),
         image_dhw=np.array([cfg.IMAGE_SHAPE[2], cfg.IMAGE_SHAPE[0], cfg.IMAGE_SHAPE[1]]), **out)


def case_classifier():
    ch, pool, fc = 8, [4, 4, 4], 16
    net = ref_model.Classifier(ch, pool, None, 2, fc, test_flag=False)
    shapes = load_formula(net)
    net.eval()
    p2 = torch.from_numpy(formula.uniform("cls.p2", (1, ch, 8, 16, 16), -1, 1)).requires_grad_(True)
    p3 = torch.from_numpy(formula.uniform("cls.p3", (1, ch, 4, 8, 8), -1, 1)).requires_grad_(True)
    rois = torch.tensor([[0.1, 0.1, 0.1, 0.4, 0.4, 0.4], [0.0, 0.0, 0.0, 1.0, 1.0, 1.0],
                         [0.3, 0.2, 0.1, 0.9, 0.8, 0.7], [0.5, 0.5, 0.5, 0.7, 0.75, 0.8],
                         [0.2, 0.6, 0.1, 0.5, 0.9, 0.3]], dtype=torch.float32)
    lg, pr, bb = net([p2, p3], rois.clone())
    g1 = torch.from_numpy(formula.uniform("cls.g1", tuple(lg.shape), -1, 1))
    g2 = torch.from_numpy(formula.uniform("cls.g2", tuple(bb.shape), -1, 1))
    ((lg * g1).sum() + (bb * g2).sum()).backward()
    params = dict(net.named_parameters())
    save("classifier", p2=p2.detach().numpy(), p3=p3.detach().numpy(), rois=rois.numpy(), pool=np.array(pool),
         logits=lg.detach().numpy(), probs=pr.detach().numpy(), bbox=bb.detach().numpy(),
         p2_grad=p2.grad.numpy(), p3_grad=p3.grad.numpy(),
         **{"grad:conv1.weight": params["conv1.weight"].grad.numpy(),
            "grad:linear_bbox.weight": params["linear_bbox.weight"].grad.numpy()},
         **shapes_to_npz(shapes))


def case_predict():
    """Full, un-injected reference dataflow at cfg0 (64x64x32, 'beginning'), SURVEY.md 8(d)."""
    stage = "beginning"
    cfg = make_cfg(stage, 64, 32)
    net = ref_model.MaskRCNN(cfg, "/tmp/cfun_logs", test_flag=False)
    shapes = load_formula(net)
    H, W, D = cfg.IMAGE_SHAPE[:3]
    lab = np.zeros((D, H, W), np.int64)
    for k in range(1, 8):  # 7 slabs along x inside the whole-volume GT box
        lab[:, :, (k - 1) * W // 7:k * W // 7] = k
    lab[:2] = 0
    hu = np.where(lab == 0, -1000.0, (lab - 1) * 50.0) + formula.uniform("pred.noise", (D, H, W), -50, 50)
    img = ((hu - hu.mean()) / hu.std()).astype(np.float32)
    image = torch.from_numpy(img)[None, None]
    gt_masks = np.zeros((1, 8, D, H, W), np.float32)
    for k in range(8):
        gt_masks[0, k] = (lab == k)
    gt_boxes = np.tile(np.array([0, 0, 0, D, H, W], np.float32), (7, 1))[None]
    gt_class_ids = np.arange(1, 8, dtype=np.int32)[None]
    anchors = net.anchors.numpy()
    # rpn targets: mark anchors by IoU with the GT box (host-side bookkeeping, inputs to the path)
    ov = ref_utils.compute_overlaps(anchors.astype(np.float64), gt_boxes[0, :1].astype(np.float64))[:, 0]
    rpn_match = np.zeros((1, anchors.shape[0], 1), np.int32)
    rpn_match[0, ov < 0.1, 0] = -1
    pos = np.argsort(-ov)[:6]
    rpn_match[0, pos, 0] = 1
    rpn_bbox_t = np.zeros((1, cfg.RPN_TRAIN_ANCHORS_PER_IMAGE, 6), np.float32)
    rb = ref_utils.box_refinement(torch.from_numpy(anchors[np.sort(pos)]).float(),
                                  torch.from_numpy(np.tile(gt_boxes[0, :1], (6, 1))).float()).numpy()
    rpn_bbox_t[0, :6] = rb / cfg.RPN_BBOX_STD_DEV
    perms = []
    orig_randperm = torch.randperm

    def rec_randperm(n, *a, **k):
        g = torch.Generator().manual_seed(100 + len(perms))
        p = orig_randperm(n, generator=g)
        perms.append(p.numpy())
        return p

    torch.randperm = rec_randperm
    torch.manual_seed(3)
    try:
        with DropRecorder() as rec:
            outs = net.predict([image, None, torch.from_numpy(gt_class_ids), torch.from_numpy(gt_boxes),
                                torch.from_numpy(gt_masks)], "training")
    finally:
        torch.randperm = orig_randperm
    (rpn_class_logits, rpn_pred_bbox, target_class_ids, mrcnn_class_logits, target_deltas, mrcnn_bbox,
     target_mask, mrcnn_mask, mrcnn_mask_logits) = outs
    assert mrcnn_mask_logits.numel() > 0, "heads were skipped -- golden would be void"
    losses = ref_model.compute_losses(torch.from_numpy(rpn_match), torch.from_numpy(rpn_bbox_t), rpn_class_logits,
                                      rpn_pred_bbox, target_class_ids, mrcnn_class_logits, target_deltas,
                                      mrcnn_bbox, target_mask, mrcnn_mask, mrcnn_mask_logits, stage)
    w = cfg.LOSS_WEIGHTS
    total = (w["rpn_class_loss"] * losses[0] + w["rpn_bbox_loss"] * losses[1] + w["mrcnn_class_loss"] * losses[2]
             + w["mrcnn_bbox_loss"] * losses[3] + w["mrcnn_mask_loss"] * losses[4]
             + w["mrcnn_mask_edge_loss"] * losses[5])
    total.backward()
    params = dict(net.named_parameters())
    n_pos = int((target_class_ids > 0).sum())
    n_all = int(target_class_ids.shape[0])
    print("predict: n_pos", n_pos, "n_rois", n_all, "losses", [float(l) for l in losses])
    ml = mrcnn_mask_logits.detach().numpy()
    tm = target_mask.numpy()
    arrs = dict(image=img, gt_masks_labels=lab.astype(np.uint8), gt_boxes=gt_boxes, gt_class_ids=gt_class_ids,
                anchors=anchors, rpn_match=rpn_match, rpn_bbox_t=rpn_bbox_t,
                randperm0=perms[0], randperm1=perms[1],
                rpn_class_logits=rpn_class_logits.detach().numpy(), rpn_pred_bbox=rpn_pred_bbox.detach().numpy(),
                target_class_ids=target_class_ids.numpy(), mrcnn_class_logits=mrcnn_class_logits.detach().numpy(),
                target_deltas=target_deltas.numpy(), mrcnn_bbox=mrcnn_bbox.detach().numpy(),
                target_mask_labels=tm.argmax(1).astype(np.uint8),
                target_mask_is_onehot=np.array(bool(np.all(tm.sum(1) == 1))),
                mask_logits_sub=ml[:, :, ::4, ::4, ::4].copy(),
                mask_logits_sum=np.array([ml.astype(np.float64).sum(), np.abs(ml).astype(np.float64).sum()]),
                losses=np.array([float(l) for l in losses]), total=np.array(float(total)),
                n_pos=np.array(n_pos), n_rois=np.array(n_all), stage=np.array(stage), **shapes_to_npz(shapes))
    for i, m in enumerate(rec.masks):
        arrs["drop%d" % i] = m.numpy()
    for k in ("fpn.C1.0.weight", "fpn.C3.2.conv3.weight", "fpn.P2_conv2.bias", "rpn.conv_class.weight",
              "classifier.linear_class.weight", "classifier.conv2.weight",
              "mask.modified_u_net.conv3d_c1_1.weight", "mask.modified_u_net.norm_lrelu_conv_c3.2.weight",
              "mask.modified_u_net.conv3d_l4.weight"):
        arrs["grad:" + k] = params[k].grad.numpy()
    # the sampled RoI sets themselves (outputs of detection_target_layer) are recoverable from the
    # recorded randperm draws; store them too so the restatement can be checked stage by stage
    save("predict_cfg0", **arrs)


def case_train_epoch():
    """The reference's OWN MaskRCNN.train_epoch (model.py:1574-1676) for 3 optimizer steps at cfg0 (64x64x32, 'beginning')
    on the fixed sample of predict_cfg0.npz, with the optimizer train_model builds (model.py:1538-1545: SGD, weight decay on
    the trainables without 'bn' in their name, momentum) -- the data generator and load_image_gt (augmentation, anchor
    targets: host-side, out of the path) are replaced by that sample.  Recorded: the randperm draws and Dropout3d masks of
    every step, the six losses of every step, the epoch's return value, and what the three steps did to the parameters."""
    import torch.optim as optim
    stage = "beginning"
    cfg = make_cfg(stage, 64, 32)
    cfg.BATCH_SIZE = 1              # = IMAGES_PER_GPU x GPU_COUNT on one GPU (GPU_COUNT = 0 here keeps the tensors on the host)
    net = ref_model.MaskRCNN(cfg, "/tmp/cfun_logs", test_flag=False)
    load_formula(net)
    g = dict(np.load(os.path.join(HERE, "predict_cfg0.npz"), allow_pickle=False))      # inputs only
    img = g["image"]
    lab = g["gt_masks_labels"].astype(np.int64)
    gt_masks = np.stack([(lab == k) for k in range(8)]).astype(np.float32)
    sample = (img[None], g["rpn_match"][0], g["rpn_bbox_t"][0], g["gt_class_ids"][0], g["gt_boxes"][0], gt_masks)
    orig_load, orig_losses, orig_randperm = ref_model.load_image_gt, ref_model.compute_losses, torch.randperm
    perms, step_losses = [], []

    def fake_load_image_gt(image, mask, angle, dataset, config, anchors):
        return sample

    def rec_losses(*a):
        out = orig_losses(*a)
        step_losses.append([float(l.detach()) for l in out])
        return out

    def rec_randperm(n, *a, **k):
        p_ = orig_randperm(n, generator=torch.Generator().manual_seed(100 + len(perms)))
        perms.append(p_.numpy())
        return p_

    # model.py:1534-1545, verbatim in effect
    net.set_trainable(".*")
    trainables_wo_bn = [param for name, param in net.named_parameters() if param.requires_grad and 'bn' not in name]
    trainables_only_bn = [param for name, param in net.named_parameters() if param.requires_grad and 'bn' in name]
    assert not trainables_only_bn          # BatchNorm is frozen on this path (TRAIN_BN = False)
    optimizer = optim.SGD([{'params': trainables_wo_bn, 'weight_decay': cfg.WEIGHT_DECAY}, {'params': trainables_only_bn}],
                          lr=cfg.LEARNING_RATE, momentum=cfg.LEARNING_MOMENTUM)
    before = {k: v.detach().clone() for k, v in net.named_parameters() if v.requires_grad}
    steps = 3
    datagen = [(torch.from_numpy(img)[None], torch.zeros(1, 1), torch.from_numpy(lab)[None]) for _ in range(steps + 1)]
    ref_model.load_image_gt, ref_model.compute_losses, torch.randperm = fake_load_image_gt, rec_losses, rec_randperm
    torch.manual_seed(3)
    try:
        with DropRecorder() as rec:
            ret = net.train_epoch(datagen, optimizer, steps, 0, None)
    finally:
        ref_model.load_image_gt, ref_model.compute_losses, torch.randperm = orig_load, orig_losses, orig_randperm
    assert len(step_losses) == steps and len(perms) == 2 * steps and len(rec.masks) == 5 * steps, (len(perms), len(rec.masks))
    after = dict(net.named_parameters())
    names = sorted(before)
    delta_norm = np.array([float((after[k].detach() - before[k]).double().norm()) for k in names])
    arrs = dict(step_losses=np.array(step_losses), epoch_return=np.array([float(v) for v in ret]), steps=np.array(steps),
                lr=np.array(cfg.LEARNING_RATE), momentum=np.array(cfg.LEARNING_MOMENTUM), weight_decay=np.array(cfg.WEIGHT_DECAY),
                param_names=np.array(names), delta_norm=delta_norm)
    for i, p_ in enumerate(perms):
        arrs["randperm%d" % i] = p_
    for i, m in enumerate(rec.masks):
        arrs["drop%d" % i] = m.numpy()
    for k in ("fpn.C1.0.weight", "fpn.P2_conv2.bias", "rpn.conv_class.weight", "classifier.linear_class.weight",
              "classifier.conv2.weight", "mask.modified_u_net.conv3d_c1_1.weight",
              "mask.modified_u_net.norm_lrelu_conv_c3.2.weight", "mask.modified_u_net.conv3d_l4.weight"):
        arrs["delta:" + k] = (after[k].detach() - before[k]).numpy()
    print("train_epoch: step losses", step_losses, "return", [float(v) for v in ret])
    save("train_epoch_cfg0", **arrs)


def case_refine():
    """model.refine_detections (model.py:584-672): crafted so that several classes pass the 0.7 filter, some boxes
    of one class overlap (per-class NMS at 0.3 suppresses them) and more survive than DETECTION_MAX_INSTANCES."""
    cfg = make_cfg("beginning", 64, 32)
    d, h, w = cfg.IMAGE_SHAPE[2], cfg.IMAGE_SHAPE[0], cfg.IMAGE_SHAPE[1]
    n, ncls = 48, 4
    ctr = formula.uniform("ref.ctr", (n, 3), 0.15, 0.85)
    ctr[8:16] = ctr[0:8] + formula.uniform("ref.jit", (8, 3), -0.02, 0.02)     # near-duplicates of the first 8
    half = formula.uniform("ref.half", (n, 3), 0.05, 0.2)
    rois = np.concatenate([ctr - half, ctr + half], axis=1).astype(np.float32)
    logits = formula.uniform("ref.logits", (n, ncls), -1, 1) * 6.0
    logits[8:16] = logits[0:8] + formula.uniform("ref.ljit", (8, ncls), -0.1, 0.1)  # same classes as their twins
    probs = torch.softmax(torch.from_numpy(logits), dim=1)
    deltas = torch.from_numpy(formula.uniform("ref.deltas", (n, ncls, 6), -1, 1))
    window = np.array([0, 0, 0, d, h, w], dtype=np.float32)
    out = {}
    for tag, max_inst, min_conf in (("a", 100, 0.7), ("b", 6, 0.7), ("c", 100, 0.0)):
        cfg.DETECTION_MAX_INSTANCES, cfg.DETECTION_MIN_CONFIDENCE = max_inst, min_conf
        det = ref_model.refine_detections(torch.from_numpy(rois), probs.clone(), deltas.clone(), window, cfg)
        out["det_" + tag] = det.numpy()
        out["cfg_" + tag] = np.array([max_inst, min_conf, cfg.DETECTION_NMS_THRESHOLD], dtype=np.float64)
    save("refine_detections", rois=rois, probs=probs.numpy(), deltas=deltas.numpy(), window=window,
         image_dhw=np.array([d, h, w]), std_dev=np.asarray(cfg.RPN_BBOX_STD_DEV, dtype=np.float32), **out)


def case_unmold():
    """utils.unmold_mask + MaskRCNN.unmold_detections (utils.py:443-460, model.py:1812-1864): 3 detections (one of
    zero volume, dropped), class probabilities on a 12x10x8 grid, window == whole image."""
    d, h, w, c = 20, 24, 28, 8
    probs = torch.softmax(torch.from_numpy(formula.uniform("unm.logits", (3, 12, 10, 8, c), -3, 3)), dim=-1).numpy()
    det = np.array([[2, 3, 4, 17, 21, 25, 3, 0.95], [5, 5, 5, 5, 9, 9, 2, 0.9], [0, 0, 0, 10, 12, 14, 1, 0.8],
                    [0, 0, 0, 0, 0, 0, 0, 0]], np.float32)
    image_shape = [1, d, h, w]
    window = np.array([0, 0, 0, d, h, w], np.float32)
    full = ref_utils.unmold_mask(probs[0], det[0, :6].astype(np.int32), image_shape)
    pad = np.concatenate([probs, np.zeros((1,) + probs.shape[1:], np.float32)], axis=0)
    boxes, ids, scores, cmap = ref_model.MaskRCNN.unmold_detections(None, det.copy(), pad, image_shape, window)
    save("unmold", probs=probs, detections=det, image_shape=np.array(image_shape), window=window,
         full_mask_sub=full[::3, ::3, ::3].copy(), full_mask_sum=np.array(full.astype(np.float64).sum()),
         boxes=boxes, class_ids=ids, scores=scores, class_map=cmap.astype(np.uint8))


_LITS = {}


def import_lits(full=False):
    """The LiTS fork's model/utils under their own names (its files shadow the heart modules' names).  ``full``: also
    its backbone, mask_branch and LiTS_main (the LiTSConfig) modules."""
    if not _LITS:
        saved = {k: sys.modules.pop(k) for k in ("utils", "model", "backbone", "mask_branch", "config") if k in sys.modules}
        sys.path.insert(0, os.path.join(REF, "LiTS_2017"))
        try:
            import model as lits_model
            import utils as lits_utils
            import backbone as lits_backbone
            import mask_branch as lits_mask_branch
            import LiTS_main as lits_main
        finally:
            sys.path.pop(0)
            for k in ("utils", "model", "backbone", "mask_branch", "config", "LiTS_main"):
                sys.modules.pop(k, None)
            sys.modules.update(saved)
        _LITS.update(model=lits_model, utils=lits_utils, backbone=lits_backbone, mask_branch=lits_mask_branch,
                     main=lits_main)
    if full:
        return _LITS
    return _LITS["model"], _LITS["utils"]


def make_lits_cfg(stage, max_dim=32, min_dim=16, **over):
    """The fork's own LiTSConfig (LiTS_2017/LiTS_main.py:28-175, config.py:196-226) shrunk to fixture size: channel
    counts in the fork's 24:48:160:320 ratios, P3D35, mask crops in a non-cubic ratio."""
    lits = import_lits(full=True)
    ns = dict(GPU_COUNT=0, IMAGE_MAX_DIM=max_dim, IMAGE_MIN_DIM=min_dim, BACKBONE_CHANNELS=[12, 24],
              TOP_DOWN_PYRAMID_SIZE=20, RPN_CONV_CHANNELS=40, FPN_CLASSIFY_FC_LAYERS_SIZE=16, UNET_MASK_BRANCH_CHANNEL=4,
              RPN_ANCHOR_SCALES=(16, 32), PRE_NMS_LIMIT=64, POST_NMS_ROIS_TRAINING=16, POOL_SIZE=[4, 4, 4],
              MASK_POOL_SIZE=[32, 48, 32])
    ns.update(over)
    cfg = type("Cfg", (lits["main"].LiTSConfig,), ns)(stage)
    cfg.MASK_SHAPE = cfg.MINI_MASK_SHAPE = (64, 96, 64) if stage == "finetune" else (32, 48, 32)
    return cfg


def case_fpn_rpn_lits():
    """The fork's detector trunk: P3D35 ([4,5] bottlenecks -> ST pattern A,B,C,A / A,B,C,A,B) with the (5,7,7) stem
    (LiTS_2017/backbone.py:124,172-176) + its FPN and RPN (LiTS_2017/model.py), values and parameter gradients."""
    lits = import_lits(full=True)
    cfg = make_lits_cfg("beginning")
    p3d = lits["backbone"].P3D35(config=cfg)
    c1, c2, c3 = p3d.stages()
    holder = nn.Module()
    holder.fpn = lits["model"].FPN(c1, c2, c3, out_channels=cfg.TOP_DOWN_PYRAMID_SIZE, config=cfg)
    holder.rpn = lits["model"].RPN(len(cfg.RPN_ANCHOR_RATIOS), cfg.RPN_ANCHOR_STRIDE, cfg.TOP_DOWN_PYRAMID_SIZE,
                                   cfg.RPN_CONV_CHANNELS)
    shapes = load_formula(holder)
    holder.eval()
    x = torch.from_numpy(formula.uniform("fpnl.x", (1, 1, 16, 32, 32), -2, 2)).requires_grad_(True)
    feats = {}
    h = holder.fpn.C1(x); feats["c1"] = h
    h = holder.fpn.C2(h); feats["c2"] = h
    h = holder.fpn.C3(h); feats["c3"] = h
    p2, p3 = holder.fpn(x)
    outs = dict(p2=p2, p3=p3)
    for tag, p in (("l2", p2), ("l3", p3)):
        lg, pr, bb = holder.rpn(p)
        outs["rpn_logits_" + tag], outs["rpn_probs_" + tag], outs["rpn_bbox_" + tag] = lg, pr, bb
    loss = 0
    for k in ("p2", "p3", "rpn_logits_l2", "rpn_bbox_l2", "rpn_logits_l3", "rpn_bbox_l3"):
        g = torch.from_numpy(formula.uniform("fpnl.g." + k, tuple(outs[k].shape), -1, 1))
        loss = loss + (outs[k] * g).sum()
    loss.backward()
    grads = {}
    params = dict(holder.named_parameters())
    for k in ("fpn.C1.0.weight", "fpn.C1.0.bias", "fpn.C2.0.conv1.weight", "fpn.C2.3.conv2.weight",
              "fpn.C2.2.conv3.weight", "fpn.C2.0.downsample.0.weight", "fpn.C3.4.conv3.weight", "fpn.C3.3.conv2.weight",
              "fpn.C3.1.conv4.bias", "fpn.P2_conv2.weight", "fpn.P3_conv1.weight", "rpn.conv_shared.weight",
              "rpn.conv_bbox.bias"):
        grads["grad:" + k] = params[k].grad.numpy()
    arrs = dict(x=x.detach().numpy(), x_grad=x.grad.numpy(), **shapes_to_npz(shapes))
    arrs.update({k: v.detach().numpy() for k, v in feats.items()})
    arrs.update({k: v.detach().numpy() for k, v in outs.items()})
    arrs.update(grads)
    save("fpn_rpn_lits", **arrs)


def case_unet_lits():
    """The fork's OWN Modified3DUNet (LiTS_2017/mask_branch.py: no dropout) in train mode on a NON-cubic crop in the
    32x80x80 family (32x48x48), 3 classes, b = 4: logits and gradients."""
    lits = import_lits(full=True)
    tag, stage, b, ncls = "unet_lits_noncubic", "beginning", 4, 3
    net = lits["mask_branch"].Modified3DUNet(1, ncls, stage, b)
    shapes = load_formula(net, gain=1.0)
    net.train(True)
    x = torch.from_numpy(formula.uniform(tag + ".x", (1, 1, 32, 48, 48), -2, 2)).requires_grad_(True)
    y = net(x)
    gy = torch.from_numpy(formula.uniform(tag + ".gy", tuple(y.shape), -1, 1))
    (y * gy).sum().backward()
    params = dict(net.named_parameters())
    arrs = dict(x=x.detach().numpy(), stage=np.array(stage), b=np.array(b), ncls=np.array(ncls), y=y.detach().numpy(),
                x_grad=x.grad.numpy(), no_dropout=np.array(True), **shapes_to_npz(shapes))
    gkeys = ("conv3d_c1_1.weight", "conv3d_c1_2.weight", "lrelu_conv_c1.1.weight", "conv3d_c2.weight",
             "norm_lrelu_conv_c3.2.weight", "norm_lrelu_conv_c5.2.weight",
             "norm_lrelu_upscale_conv_norm_lrelu_l0.3.weight", "conv_norm_lrelu_l2.0.weight", "conv3d_l3.weight",
             "conv_norm_lrelu_l4.0.weight", "conv3d_l4.weight", "ds2_1x1_conv3d.weight", "ds3_1x1_conv3d.weight")
    for k in gkeys:
        arrs["grad:" + k] = params[k].grad.numpy()
    g64 = unet_grads_fp64(net, x.detach(), [], gy, gkeys)
    arrs["x_grad64"] = g64.pop("x").astype(np.float32)
    for k, v in g64.items():
        arrs["grad64:" + k] = v.astype(np.float32)
    save(tag, **arrs)


class PermRecorder:
    """Seeded, recorded stand-in for the torch.randperm draws of detection_target_layer."""

    def __init__(self):
        self.perms = []
        self._orig = torch.randperm

    def __enter__(self):
        rec = self

        def rp(n, *a, **k):
            g = torch.Generator().manual_seed(100 + len(rec.perms))
            p = rec._orig(n, generator=g)
            rec.perms.append(p.numpy())
            return p
        torch.randperm = rp
        return self

    def __exit__(self, *a):
        torch.randperm = self._orig


def case_dtl_lits():
    """The fork's detection_target_layer (LiTS_2017/model.py:405-557): RoI counts by int(round()) at a ratio where
    rounding and truncation differ (15 * 0.37 = 5.55 -> 6; heart: 5), 3-class GT masks, recorded permutations."""
    lits_model, _ = import_lits()
    cfg = make_lits_cfg("beginning")
    cfg.TRAIN_ROIS_PER_IMAGE, cfg.ROI_POSITIVE_RATIO = 15, 0.37
    cfg.MASK_SHAPE = (8, 12, 8)
    D, H, W = 16, 32, 32
    gt = torch.tensor([[0.05, 0.1, 0.1, 0.6, 0.55, 0.5], [0.4, 0.5, 0.45, 0.95, 0.95, 0.9]])
    gt_ids = torch.tensor([1, 2])
    jit = torch.from_numpy(formula.uniform("dtl.jit", (20, 6), -0.1, 0.1))
    props = torch.cat([(gt[i % 2] + jit[i]).clamp(0, 1)[None] for i in range(20)] +
                      [torch.tensor([[0.0, 0.0, 0.6, 0.2, 0.2, 0.9]]), torch.tensor([[0.7, 0.0, 0.0, 1.0, 0.3, 0.3]])] * 6)
    props[:, 3:] = torch.max(props[:, 3:], props[:, :3] + 0.05)
    lab = (formula.uniform("dtl.lab", (D, H, W), 0, 1) * 3).astype(np.int64).clip(0, 2)
    onehot = np.stack([(lab == k) for k in range(3)], axis=0).astype(np.float32)
    with PermRecorder() as rec:
        p_rois, rois, ids, deltas, masks = lits_model.detection_target_layer(
            props[None].clone(), gt_ids[None], gt[None].clone(), torch.from_numpy(onehot)[None], cfg)
    assert p_rois.shape[0] == 6, p_rois.shape
    save("dtl_lits", proposals=props.numpy(), gt_boxes=gt.numpy(), gt_class_ids=gt_ids.numpy(),
         gt_labels=lab.astype(np.uint8), mask_shape=np.array(cfg.MASK_SHAPE), train_rois=np.array(15),
         positive_ratio=np.array(0.37), randperm0=rec.perms[0], randperm1=rec.perms[1], p_rois=p_rois.numpy(),
         rois=rois.numpy(), class_ids=ids.numpy(), deltas=deltas.numpy(),
         mask_labels=masks.numpy().argmax(1).astype(np.uint8),
         masks_onehot=np.array(bool(np.all(masks.numpy().sum(1) == 1))))


def case_predict_lits():
    """The fork's OWN two training phases end to end (LiTS_2017/model.py:1282-1296 build, 1441-1560 predict,
    985-1001 compute_losses) at fixture size: stage 'beginning' = detector only (classifier on 50 RoIs at 33 %, no
    mask head, mask losses 0) and stage 'together' = mask branch only (everything else frozen, 4 positive RoIs, no
    classifier, detection losses 0, class-weighted CE + raw-Sobel edge MSE)."""
    lits_model, lits_utils = import_lits()
    for stage in ("beginning", "together"):
        cfg = make_lits_cfg(stage, POST_NMS_ROIS_TRAINING=64)       # every NMS survivor of the 36 anchors is a proposal
        net = lits_model.MaskRCNN(cfg, "/tmp/cfun_logs", test_flag=False)
        shapes = load_formula(net)
        # closed-form weights give box deltas that throw every proposal off the GT box; damp the RPN's box head so
        # that positives exist and the heads run (the factor is part of the fixture: the test applies it too)
        net.eval()
        with torch.no_grad():
            probe = torch.from_numpy(formula.uniform("predl.noise", (1, 1, 16, 32, 32), -1, 1))
            amax = max(float(net.rpn(p)[2].abs().max()) for p in net.fpn(probe))
        bbox_gain = float("%.0e" % (0.2 / amax))            # one significant digit: deltas of ~0.2 * std_dev
        print("rpn bbox absmax", amax, "-> gain", bbox_gain)
        with torch.no_grad():
            net.rpn.conv_bbox.weight.mul_(bbox_gain)
            net.rpn.conv_bbox.bias.mul_(bbox_gain)
        H, W, D = [int(v) for v in cfg.IMAGE_SHAPE[:3]]
        lab = np.zeros((D, H, W), np.int64)
        lab[2:, :, :W // 2] = 1
        lab[2:, :, W // 2:] = 2
        hu = np.where(lab == 0, -1000.0, (lab - 1) * 100.0) + formula.uniform("predl.noise", (D, H, W), -50, 50)
        img = ((hu - hu.mean()) / hu.std()).astype(np.float32)
        image = torch.from_numpy(img)[None, None]
        gt_masks = np.stack([(lab == k) for k in range(3)], axis=0).astype(np.float32)[None]
        # two GT boxes that clipped anchors reproduce: the whole volume (class 1) and its x < W/2 half (class 2)
        gt_boxes = np.array([[0, 0, 0, D, H, W], [0, 0, 0, D, H, W // 2]], np.float32)[None]
        gt_class_ids = np.arange(1, 3, dtype=np.int32)[None]
        anchors = net.anchors.numpy()
        ov = lits_utils.compute_overlaps(anchors.astype(np.float64), gt_boxes[0, :1].astype(np.float64))[:, 0]
        rpn_match = np.zeros((1, anchors.shape[0], 1), np.int32)
        rpn_match[0, ov < 0.1, 0] = -1
        pos = np.argsort(-ov)[:6]
        rpn_match[0, pos, 0] = 1
        rpn_bbox_t = np.zeros((1, cfg.RPN_TRAIN_ANCHORS_PER_IMAGE, 6), np.float32)
        rb = lits_utils.box_refinement(torch.from_numpy(anchors[np.sort(pos)]).float(),
                                       torch.from_numpy(np.tile(gt_boxes[0, :1], (6, 1))).float()).numpy()
        rpn_bbox_t[0, :6] = rb / cfg.RPN_BBOX_STD_DEV
        with PermRecorder() as rec:
            outs = net.predict([image, None, torch.from_numpy(gt_class_ids), torch.from_numpy(gt_boxes),
                                torch.from_numpy(gt_masks)], "training")
        (rpn_class_logits, rpn_pred_bbox, target_class_ids, mrcnn_class_logits, target_deltas, mrcnn_bbox,
         target_mask, mrcnn_mask, mrcnn_mask_logits) = outs
        losses = lits_model.compute_losses(torch.from_numpy(rpn_match), torch.from_numpy(rpn_bbox_t), rpn_class_logits,
                                           rpn_pred_bbox, target_class_ids, mrcnn_class_logits, target_deltas,
                                           mrcnn_bbox, target_mask, mrcnn_mask, mrcnn_mask_logits, stage)
        w = cfg.LOSS_WEIGHTS
        keys = ("rpn_class_loss", "rpn_bbox_loss", "mrcnn_class_loss", "mrcnn_bbox_loss", "mrcnn_mask_loss",
                "mrcnn_mask_edge_loss")
        total = sum(w[k] * l for k, l in zip(keys, losses))
        total.backward()
        params = dict(net.named_parameters())
        n_pos, n_all = int((target_class_ids > 0).sum()), int(target_class_ids.shape[0])
        print("predict_lits", stage, "n_pos", n_pos, "n_rois", n_all, "losses", [float(l) for l in losses])
        assert n_pos > 0, "heads were skipped -- golden would be void"
        arrs = dict(image=img, gt_labels=lab.astype(np.uint8), gt_boxes=gt_boxes, gt_class_ids=gt_class_ids,
                    rpn_match=rpn_match, rpn_bbox_t=rpn_bbox_t, randperm0=rec.perms[0],
                    randperm1=rec.perms[1] if len(rec.perms) > 1 else np.zeros((0,), np.int64),
                    rpn_class_logits=rpn_class_logits.detach().numpy(), rpn_pred_bbox=rpn_pred_bbox.detach().numpy(),
                    target_class_ids=target_class_ids.numpy(), target_deltas=target_deltas.numpy(),
                    target_mask_labels=target_mask.numpy().argmax(1).astype(np.uint8),
                    losses=np.array([float(l) for l in losses]), total=np.array(float(total)),
                    loss_weights=np.array([w[k] for k in keys]), n_pos=np.array(n_pos), n_rois=np.array(n_all),
                    stage=np.array(stage), rpn_bbox_gain=np.array(bbox_gain), trainable=np.array(sorted(k for k, p in params.items() if p.requires_grad)),
                    with_grad=np.array(sorted(k for k, p in params.items() if p.grad is not None)),
                    **shapes_to_npz(shapes))
        if stage == "beginning":
            assert mrcnn_mask_logits.numel() == 0 and mrcnn_class_logits.numel() > 0
            arrs.update(mrcnn_class_logits=mrcnn_class_logits.detach().numpy(), mrcnn_bbox=mrcnn_bbox.detach().numpy())
            gk = ("fpn.C1.0.weight", "fpn.C3.4.conv3.weight", "fpn.P2_conv2.bias", "rpn.conv_class.weight",
                  "classifier.linear_class.weight", "classifier.conv2.weight", "classifier.conv1.weight")
        else:
            assert mrcnn_class_logits.numel() == 0 and mrcnn_mask_logits.numel() > 0
            ml = mrcnn_mask_logits.detach().numpy()
            arrs.update(mask_logits_sub=ml[:, :, ::2, ::2, ::2].copy(),
                        mask_logits_sum=np.array([ml.astype(np.float64).sum(), np.abs(ml).astype(np.float64).sum()]))
            gk = ("mask.modified_u_net.conv3d_c1_1.weight", "mask.modified_u_net.norm_lrelu_conv_c3.2.weight",
                  "mask.modified_u_net.conv_norm_lrelu_l4.0.weight", "mask.modified_u_net.conv3d_l4.weight",
                  "mask.modified_u_net.ds2_1x1_conv3d.weight")
        for k in gk:
            arrs["grad:" + k] = params[k].grad.numpy()
        save("predict_lits_" + stage, **arrs)



def case_unmold_lits():
    """LiTS overlap-tile utils.unmold_mask + MaskRCNN.unmold_detections (LiTS_2017/utils.py:383-408,
    LiTS_2017/model.py:1777-1835): 5 detections (one of zero volume, dropped; the other four overlap, up to four deep), 3 class probabilities on an 8x10x12 grid, window == whole image."""
    lits_model, lits_utils = import_lits()
    d, h, w, c = 20, 24, 28, 3
    probs = torch.softmax(torch.from_numpy(formula.uniform("unl.logits", (5, 8, 10, 12, c), -3, 3)), dim=-1).numpy()
    det = np.array([[2, 3, 4, 17, 21, 25, 1, 0.95], [5, 5, 5, 5, 9, 9, 2, 0.9], [0, 0, 0, 10, 12, 14, 1, 0.8],
                    [8, 10, 12, 20, 24, 28, 2, 0.75], [6, 2, 9, 12, 20, 16, 2, 0.72], [0, 0, 0, 0, 0, 0, 0, 0]],
                   np.float32)
    image_shape = [1, d, h, w]
    window = np.array([0, 0, 0, d, h, w], np.float32)
    keep = [0, 2, 3, 4]
    full = lits_utils.unmold_mask(probs[keep], det[keep, :6].astype(np.int32), image_shape)
    pad = np.concatenate([probs, np.zeros((1,) + probs.shape[1:], np.float32)], axis=0)
    boxes, ids, scores, cmap = lits_model.MaskRCNN.unmold_detections(None, det.copy(), pad, image_shape, window)
    save("unmold_lits", probs=probs, detections=det, image_shape=np.array(image_shape), window=window,
         full_mask_sub=full[::3, ::3, ::3].copy(), full_mask_sum=np.array(full.astype(np.float64).sum()),
         boxes=boxes, class_ids=ids, scores=scores, class_map=cmap.astype(np.uint8))


CASES = dict(fpn_rpn_lits=case_fpn_rpn_lits, unet_lits=case_unet_lits, dtl_lits=case_dtl_lits, predict_lits=case_predict_lits,
             losses_lits=case_losses_lits, unmold_lits=case_unmold_lits, unmold=case_unmold, refine=case_refine, nms=case_nms, anchors=case_anchors, roi_align=case_roi_align, fpn_rpn=case_fpn_rpn, unet=case_unet,
             losses=case_losses, proposal=case_proposal, classifier=case_classifier, predict=case_predict,
             train_epoch=case_train_epoch)

if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or [k for k in CASES if k not in ("predict", "predict_lits", "train_epoch")]
    for k in which:
        print("== case", k)
        CASES[k]()
