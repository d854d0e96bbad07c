"""CPU: the oracle restatement (oracle/cfun_oracle.py) against the golden vectors generated from
the reference import (tests/golden/gen_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from conftest import golden_state_dict, load_golden
from oracle import cfun_oracle as orc


def t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e", "f", "tie"])
def test_nms_bit_exact(tag):
    g = load_golden("nms")
    thr, mx = g[tag + "_cfg"]
    keep = orc.nms(g[tag + "_boxes"], g[tag + "_scores"], float(thr), int(mx))
    assert keep.dtype == np.int32
    np.testing.assert_array_equal(keep, g[tag + "_keep"])


@pytest.mark.parametrize("tag", ["cfg0", "odd"])
def test_anchors_exact(tag):
    g = load_golden("anchors")
    a = orc.generate_pyramid_anchors((64, 128), g[tag + "_shapes"], (8, 16), 1)
    np.testing.assert_array_equal(a, g[tag + "_anchors"])


def test_roi_align_forward_and_grad():
    g = load_golden("roi_align")
    fm = t(g["fm"]).requires_grad_(True)
    out = orc.roi_align(fm, [int(v) for v in g["pool"]], t(g["boxes"]))
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=0, atol=1e-6)
    assert np.all(g["out"][2] == 0) and np.all(g["out"][5] == 0)  # degenerate boxes -> zeros
    (out * t(g["gy"])).sum().backward()
    np.testing.assert_allclose(fm.grad.numpy(), g["fm_grad"], rtol=0, atol=1e-5)


def test_pyramid_roi_align_levels_and_order():
    g = load_golden("roi_align")
    lv = orc.roi_levels(t(g["pboxes"]))
    np.testing.assert_array_equal(lv.numpy(), g["plevels"])
    assert set(g["plevels"].tolist()) == {2, 3}
    pooled = orc.pyramid_roi_align(t(g["pboxes"]), [t(g["p2"]), t(g["p3"])], [3, 3, 3])
    np.testing.assert_allclose(pooled.numpy(), g["pooled"], rtol=0, atol=1e-6)


def test_fpn_rpn_forward_and_grads():
    g = load_golden("fpn_rpn")
    sd = golden_state_dict(g)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype == torch.float32}
    sd_live = dict(sd)
    sd_live.update(params)
    x = t(g["x"]).requires_grad_(True)
    c1, c2, c3 = orc.p3d_stages(x, sd_live, "fpn.")
    for k, v in (("c1", c1), ("c2", c2), ("c3", c3)):
        np.testing.assert_allclose(v.detach().numpy(), g[k], rtol=1e-5, atol=1e-6)
    p2, p3 = orc.fpn(x, sd_live)
    outs = dict(p2=p2, p3=p3)
    for tag, p in (("l2", p2), ("l3", p3)):
        lg, pr, bb = orc.rpn(p, sd_live)
        outs["rpn_logits_" + tag], outs["rpn_probs_" + tag], outs["rpn_bbox_" + tag] = lg, pr, bb
    for k, v in outs.items():
        np.testing.assert_allclose(v.detach().numpy(), g[k], rtol=1e-5, atol=2e-6, err_msg=k)
    from oracle import formula
    loss = 0
    for k in ("p2", "p3", "rpn_logits_l2", "rpn_bbox_l2", "rpn_logits_l3", "rpn_bbox_l3"):
        loss = loss + (outs[k] * t(formula.uniform("fpn.g." + k, tuple(outs[k].shape), -1, 1))).sum()
    loss.backward()
    np.testing.assert_allclose(x.grad.numpy(), g["x_grad"], rtol=1e-4, atol=1e-5)
    n = 0
    for k in g:
        if k.startswith("grad:"):
            got = params[k[5:]].grad.numpy()
        elif k.startswith("grad4:"):
            got = params[k[6:]].grad.numpy()[::4, ::4]
        else:
            continue
        n += 1
        scale = np.abs(g[k]).max()
        np.testing.assert_allclose(got, g[k], rtol=1e-4, atol=1e-5 * max(scale, 1.0), err_msg=k)
    assert n >= 10


@pytest.mark.parametrize("name", ["unet_beginning_eval", "unet_beginning_train", "unet_finetune_train",
                                  "unet_lits_eval"])
def test_unet(name):
    g = load_golden(name)
    sd = golden_state_dict(g)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = t(g["x"]).requires_grad_(True)
    masks = [t(g["drop%d" % i]) for i in range(5)] if "drop0" in g else None
    y = orc.unet(x, params, "", str(g["stage"]), masks)
    yn = y.detach().numpy()
    if "y" in g:
        np.testing.assert_allclose(yn, g["y"], rtol=1e-4, atol=1e-4)
    else:
        np.testing.assert_allclose(yn[:, :, ::2, ::2, ::2], g["y_sub"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(np.abs(yn).astype(np.float64).sum(), g["y_sum"][1], rtol=1e-5)
    if masks is not None:
        from oracle import formula
        (y * t(formula.uniform(name + ".gy", tuple(y.shape), -1, 1))).sum().backward()
        np.testing.assert_allclose(x.grad.numpy(), g["x_grad"], rtol=1e-3, atol=1e-3 * np.abs(g["x_grad"]).max())
        n = 0
        for k in g:
            if k.startswith("grad:"):
                n += 1
                ref = g[k]
                np.testing.assert_allclose(params[k[5:]].grad.numpy(), ref, rtol=1e-3,
                                           atol=1e-3 * np.abs(ref).max(), err_msg=k)
        assert n >= 10


def test_mask_losses():
    g = load_golden("losses")
    lab = g["labels"].astype(np.int64)
    c = g["logits"].shape[1]
    onehot = torch.stack([t(lab == k) for k in range(c)], dim=1).double()
    logits = t(g["logits"]).requires_grad_(True)
    ce = orc.mask_ce_loss(onehot, logits)
    np.testing.assert_allclose(ce.item(), g["ce"], rtol=1e-6)
    ce.backward()
    np.testing.assert_allclose(logits.grad.numpy(), g["ce_grad"], rtol=1e-5, atol=1e-9)
    logits.grad = None
    el = orc.edge_loss(onehot, torch.softmax(logits, dim=1))
    np.testing.assert_allclose(el.item(), g["edge"].item(), rtol=1e-6)
    el.backward()
    np.testing.assert_allclose(logits.grad.numpy(), g["edge_grad_logits"], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("tag", ["train", "infer"])
def test_proposal_layer(tag):
    g = load_golden("proposal")
    rois, keep, order = orc.proposal_layer(t(g["probs"][0]), t(g["bbox"][0]), t(g["anchors"]),
                                           int(g["count_" + tag]), 0.7, [int(v) for v in g["image_dhw"]])
    np.testing.assert_array_equal(rois.numpy(), g["rois_" + tag][0])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_refine_detections(tag):
    """model.refine_detections (the inference NMS site, A16) vs the reference's own output."""
    g = load_golden("refine_detections")
    max_inst, min_conf, thr = g["cfg_" + tag]
    det = orc.refine_detections(t(g["rois"]), t(g["probs"]), t(g["deltas"]), [float(v) for v in g["window"]],
                                [int(v) for v in g["image_dhw"]], float(min_conf), float(thr), int(max_inst),
                                tuple(float(v) for v in g["std_dev"]))
    np.testing.assert_array_equal(det.numpy(), g["det_" + tag])
    if tag == "a":   # the case exercises suppression: more candidates pass the filter than survive the NMS
        ids = np.argmax(g["probs"], axis=1)
        sc = g["probs"][np.arange(len(ids)), ids]
        assert ((ids > 0) & (sc >= 0.7)).sum() > det.shape[0]
    empty = orc.refine_detections(t(g["rois"]), t(g["probs"]), t(g["deltas"]), [float(v) for v in g["window"]],
                                  [int(v) for v in g["image_dhw"]], 1.5, float(thr), int(max_inst))
    assert tuple(empty.shape) == (0, 8)


def test_classifier():
    g = load_golden("classifier")
    sd = golden_state_dict(g)
    params = {k: v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v
              for k, v in sd.items()}
    p2 = t(g["p2"]).requires_grad_(True)
    p3 = t(g["p3"]).requires_grad_(True)
    lg, pr, bb = orc.classifier([p2[0], p3[0]], t(g["rois"]), params, [int(v) for v in g["pool"]], prefix="")
    np.testing.assert_allclose(lg.detach().numpy(), g["logits"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pr.detach().numpy(), g["probs"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bb.detach().numpy(), g["bbox"], rtol=1e-5, atol=1e-6)
    from oracle import formula
    ((lg * t(formula.uniform("cls.g1", tuple(lg.shape), -1, 1))).sum()
     + (bb * t(formula.uniform("cls.g2", tuple(bb.shape), -1, 1))).sum()).backward()
    np.testing.assert_allclose(p2.grad.numpy(), g["p2_grad"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(p3.grad.numpy(), g["p3_grad"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(params["conv1.weight"].grad.numpy(), g["grad:conv1.weight"], rtol=1e-4, atol=1e-6)


def test_predict_cfg0_full_dataflow():
    """The anchor of the whole path: the reference's OWN un-injected predict('training') + compute_losses +
    backward at BASELINE configs[0] (64x64x32, 'beginning', real channel counts; gen_golden.case_predict) against
    the oracle chain FPN -> RPN -> proposal_layer -> detection_target_layer -> classifier / mask head -> 6 losses."""
    g = load_golden("predict_cfg0")
    sd = golden_state_dict(g)
    params = {k: v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v
              for k, v in sd.items()}
    image = t(g["image"])[None, None]
    D, H, W = [int(v) for v in image.shape[2:]]
    p2, p3 = orc.fpn(image, params)
    l2, pr2, b2 = orc.rpn(p2, params)
    l3, pr3, b3 = orc.rpn(p3, params)
    rpn_logits, rpn_probs, rpn_box = [torch.cat(v, dim=1) for v in ((l2, l3), (pr2, pr3), (b2, b3))]
    np.testing.assert_allclose(rpn_logits.detach().numpy(), g["rpn_class_logits"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rpn_box.detach().numpy(), g["rpn_pred_bbox"], rtol=1e-4, atol=1e-5)
    rois_all, _, _ = orc.proposal_layer(rpn_probs[0].detach(), rpn_box[0].detach(), t(g["anchors"]), 500, 0.7, (D, H, W))
    gt_boxes = t(g["gt_boxes"][0]) / torch.tensor([D, H, W, D, H, W], dtype=torch.float32)
    lab = torch.from_numpy(g["gt_masks_labels"].astype(np.int64))
    onehot = torch.stack([(lab == k) for k in range(8)], dim=0).float()
    p_rois, rois, cls_ids, deltas, masks = orc.detection_target_layer(
        rois_all, torch.from_numpy(g["gt_class_ids"][0].astype(np.int64)), gt_boxes, onehot, (96, 96, 96),
        torch.from_numpy(g["randperm0"]), torch.from_numpy(g["randperm1"]))
    assert p_rois.shape[0] == int(g["n_pos"]) and rois.shape[0] == int(g["n_rois"])
    np.testing.assert_array_equal(cls_ids.numpy(), g["target_class_ids"])
    np.testing.assert_allclose(deltas.numpy(), g["target_deltas"], rtol=1e-4, atol=1e-5)   # deltas = d(box)/0.1
    np.testing.assert_array_equal(masks.argmax(1).numpy().astype(np.uint8), g["target_mask_labels"])
    cls_logits, _, cls_bbox = orc.classifier([p2[0], p3[0]], rois, params, [12, 12, 12])
    np.testing.assert_allclose(cls_logits.detach().numpy(), g["mrcnn_class_logits"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(cls_bbox.detach().numpy(), g["mrcnn_bbox"], rtol=1e-4, atol=1e-5)
    drops = [t(g["drop%d" % i]) for i in range(5)]
    m_logits, m_probs = orc.mask_head(image[0], p_rois, params, [96, 96, 96], "beginning", dropout_masks=drops)
    ml = m_logits.detach().numpy()
    assert np.abs(ml[:, :, ::4, ::4, ::4] - g["mask_logits_sub"]).max() < 1e-3
    assert abs(np.abs(ml).astype(np.float64).sum() - g["mask_logits_sum"][1]) < 1e-5 * g["mask_logits_sum"][1]
    losses = [orc.rpn_class_loss(torch.from_numpy(g["rpn_match"]), rpn_logits),
              orc.rpn_bbox_loss(t(g["rpn_bbox_t"]), torch.from_numpy(g["rpn_match"]), rpn_box),
              orc.mrcnn_class_loss(cls_ids, cls_logits), orc.mrcnn_bbox_loss(deltas, cls_ids, cls_bbox),
              orc.mask_ce_loss(masks.double(), m_logits), torch.zeros(())]
    for i, (a, r) in enumerate(zip(losses, g["losses"])):
        assert abs(float(a.detach()) - float(r)) <= 1e-4 * max(abs(float(r)), 1e-3), "loss %d vs %g" % (i, r)
    total = sum(w * l for w, l in zip(orc.LOSS_WEIGHTS, losses))
    assert abs(float(total) - float(g["total"])) <= 1e-4 * abs(float(g["total"]))
    total.backward()
    for k in [k[5:] for k in g if k.startswith("grad:")]:
        a, r = params[k].grad.numpy(), g["grad:" + k]
        e = np.linalg.norm((a - r).ravel()) / max(np.linalg.norm(r.ravel()), 1e-30)
        assert e < 2e-2, "%s: rel L2 %.3e" % (k, e)   # LeakyReLU mask flips, see tests/module_cases.UNET_GRAD_L2_TOL


def test_train_epoch_golden_is_consistent():
    """tests/golden/train_epoch_cfg0.npz (the reference's own train_epoch, 3 optimizer steps; the product is held to it
    in the GPU tier, module_cases.check_train_epoch_golden): its first step IS the step of predict_cfg0.npz -- same
    sample, same seeds -- which the oracle reproduces above; the epoch's return value is the mean of the recorded steps
    (model.py:1659-1666: weighted total first, then the six unweighted losses); the recorded draws have the sizes the
    replay needs.  Running the oracle through all three steps would take minutes here."""
    g0, g = load_golden("predict_cfg0"), load_golden("train_epoch_cfg0")
    steps = int(g["steps"])
    sl = g["step_losses"]
    assert sl.shape == (steps, 6)
    np.testing.assert_allclose(sl[0], g0["losses"], rtol=1e-6, atol=0)
    np.testing.assert_array_equal(g["randperm0"], g0["randperm0"])
    np.testing.assert_array_equal(g["randperm1"], g0["randperm1"])
    for i in range(5):
        np.testing.assert_array_equal(g["drop%d" % i], g0["drop%d" % i])
    np.testing.assert_allclose(g["epoch_return"][1:], sl.mean(axis=0), rtol=1e-6, atol=1e-9)
    totals = (sl * np.array(orc.LOSS_WEIGHTS, dtype=np.float64)[None]).sum(axis=1)
    np.testing.assert_allclose(g["epoch_return"][0], totals.mean(), rtol=1e-5)
    assert totals[-1] < totals[0]                         # three clipped SGD steps on one sample: the loss goes down
    assert all(("randperm%d" % i) in g for i in range(2 * steps)) and all(("drop%d" % i) in g for i in range(5 * steps))
    assert len(g["param_names"]) == len(g["delta_norm"]) and float(g["delta_norm"].max()) > 0
    # the layers the 'beginning' stage never runs were left alone (torch.optim.SGD skips .grad None): no weight decay either
    untouched = [str(n) for n, d in zip(g["param_names"], g["delta_norm"]) if d == 0.0]
    assert any("out_upscale" in n for n in untouched)


def test_unmold_golden():
    """utils.unmold_mask / MaskRCNN.unmold_detections (inference tail) vs the
    reference's own outputs."""
    g = load_golden("unmold")
    shape = [int(v) for v in g["image_shape"]]
    full = orc.unmold_mask(g["probs"][0], g["detections"][0, :6].astype(np.int32), shape)
    np.testing.assert_array_equal(full[::3, ::3, ::3], g["full_mask_sub"])
    assert abs(full.astype(np.float64).sum() - float(g["full_mask_sum"])) < 1e-9 * float(g["full_mask_sum"])
    pad = np.concatenate([g["probs"], np.zeros((1,) + g["probs"].shape[1:], np.float32)], axis=0)
    boxes, ids, scores, cmap = orc.unmold_detections(g["detections"], pad, shape, g["window"])
    np.testing.assert_array_equal(boxes, g["boxes"])
    np.testing.assert_array_equal(ids, g["class_ids"])
    np.testing.assert_array_equal(scores, g["scores"])
    np.testing.assert_array_equal(cmap.astype(np.uint8), g["class_map"])
    assert boxes.shape[0] == 2                      # the zero-volume detection was dropped


def test_unmold_lits_golden():
    """LiTS fork: overlap-tile utils.unmold_mask / MaskRCNN.unmold_detections (LiTS_2017/utils.py:383-408,
    LiTS_2017/model.py:1777-1835) vs the fork's own outputs."""
    g = load_golden("unmold_lits")
    shape = [int(v) for v in g["image_shape"]]
    keep = [0, 2, 3, 4]
    full = orc.unmold_mask_overlap(g["probs"][keep], g["detections"][keep, :6].astype(np.int32), shape)
    np.testing.assert_array_equal(full[::3, ::3, ::3], g["full_mask_sub"])
    assert abs(full.astype(np.float64).sum() - float(g["full_mask_sum"])) < 1e-9 * float(g["full_mask_sum"])
    pad = np.concatenate([g["probs"], np.zeros((1,) + g["probs"].shape[1:], np.float32)], axis=0)
    boxes, ids, scores, cmap = orc.unmold_detections_overlap(g["detections"], pad, shape, g["window"])
    np.testing.assert_array_equal(boxes, g["boxes"])
    np.testing.assert_array_equal(ids, g["class_ids"])
    np.testing.assert_array_equal(scores, g["scores"])
    np.testing.assert_array_equal(cmap.astype(np.uint8), g["class_map"])
    assert boxes.shape[0] == 4                      # the zero-volume detection was dropped


def test_mask_losses_lits():
    """LiTS fork: weighted CE ([1, 1, 100]) and the raw-Sobel edge MSE (LiTS_2017/model.py:907-979) vs the fork's own
    outputs and gradients."""
    g = load_golden("losses_lits")
    lab = g["labels"].astype(np.int64)
    c = g["logits"].shape[1]
    onehot = torch.stack([t(lab == k) for k in range(c)], dim=1).double()
    logits = t(g["logits"]).requires_grad_(True)
    ce = orc.mask_ce_loss_weighted(onehot, logits, g["class_weights"])
    np.testing.assert_allclose(ce.item(), g["ce"], rtol=1e-6)
    ce.backward()
    np.testing.assert_allclose(logits.grad.numpy(), g["ce_grad"], rtol=1e-5, atol=1e-10)
    logits.grad = None
    el = orc.edge_loss_raw(onehot, torch.softmax(logits, dim=1))
    np.testing.assert_allclose(el.item(), g["edge"].item(), rtol=1e-6)
    el.backward()
    np.testing.assert_allclose(logits.grad.numpy(), g["edge_grad_logits"], rtol=1e-4, atol=1e-8)


# ---------------------------------------------------------------------------------------------------------------
# LiTS fork detector side + two-phase control flow, pinned by goldens from the fork's OWN modules
# (gen_golden.case_fpn_rpn_lits / case_unet_lits / case_dtl_lits / case_predict_lits)
# ---------------------------------------------------------------------------------------------------------------
LITS_LAYERS, LITS_STEM_PAD = (4, 5), (2, 3, 3)     # P3D35, stem k(5,7,7) p(2,3,3): LiTS_2017/backbone.py:124,172-176


def test_fpn_rpn_lits_forward_and_grads():
    g = load_golden("fpn_rpn_lits")
    sd = golden_state_dict(g)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype == torch.float32}
    sd_live = dict(sd)
    sd_live.update(params)
    x = t(g["x"]).requires_grad_(True)
    c1, c2, c3 = orc.p3d_stages(x, sd_live, "fpn.", layers=LITS_LAYERS, stem_pad=LITS_STEM_PAD)
    for k, v in (("c1", c1), ("c2", c2), ("c3", c3)):
        np.testing.assert_allclose(v.detach().numpy(), g[k], rtol=1e-5, atol=2e-6)
    p2, p3 = orc.fpn(x, sd_live, layers=LITS_LAYERS, stem_pad=LITS_STEM_PAD)
    outs = dict(p2=p2, p3=p3)
    for tag, p in (("l2", p2), ("l3", p3)):
        lg, pr, bb = orc.rpn(p, sd_live)
        outs["rpn_logits_" + tag], outs["rpn_probs_" + tag], outs["rpn_bbox_" + tag] = lg, pr, bb
    for k, v in outs.items():
        np.testing.assert_allclose(v.detach().numpy(), g[k], rtol=1e-5, atol=5e-6, err_msg=k)
    from oracle import formula
    loss = 0
    for k in ("p2", "p3", "rpn_logits_l2", "rpn_bbox_l2", "rpn_logits_l3", "rpn_bbox_l3"):
        loss = loss + (outs[k] * t(formula.uniform("fpnl.g." + k, tuple(outs[k].shape), -1, 1))).sum()
    loss.backward()
    np.testing.assert_allclose(x.grad.numpy(), g["x_grad"], rtol=1e-4, atol=1e-5 * np.abs(g["x_grad"]).max())
    n = 0
    for k in [k for k in g if k.startswith("grad:")]:
        n += 1
        np.testing.assert_allclose(params[k[5:]].grad.numpy(), g[k], rtol=1e-4,
                                   atol=1e-5 * max(np.abs(g[k]).max(), 1.0), err_msg=k)
    assert n >= 10


def test_unet_lits_noncubic():
    """The fork's own mask_branch.py (no Dropout3d) on a 32x48x48 crop: logits and gradients."""
    g = load_golden("unet_lits_noncubic")
    sd = golden_state_dict(g)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = t(g["x"]).requires_grad_(True)
    y = orc.unet(x, params, "", str(g["stage"]), None)
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=1e-4, atol=1e-4)
    from oracle import formula
    (y * t(formula.uniform("unet_lits_noncubic.gy", tuple(y.shape), -1, 1))).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["x_grad"], rtol=1e-3, atol=1e-3 * np.abs(g["x_grad"]).max())
    for k in [k for k in g if k.startswith("grad:")]:
        np.testing.assert_allclose(params[k[5:]].grad.numpy(), g[k], rtol=1e-3, atol=1e-3 * np.abs(g[k]).max(), err_msg=k)


def test_detection_target_layer_lits_round():
    """int(round()) RoI counts of the fork (LiTS_2017/model.py:448, 496) vs its own detection_target_layer."""
    g = load_golden("dtl_lits")
    lab = torch.from_numpy(g["gt_labels"].astype(np.int64))
    onehot = torch.stack([(lab == k) for k in range(3)], dim=0).float()
    r = orc.detection_target_layer(t(g["proposals"]), torch.from_numpy(g["gt_class_ids"]), t(g["gt_boxes"]), onehot,
                                   tuple(int(v) for v in g["mask_shape"]), torch.from_numpy(g["randperm0"]),
                                   torch.from_numpy(g["randperm1"]), int(g["train_rois"]), float(g["positive_ratio"]),
                                   count_round=True)
    assert r[0].shape[0] == 6                          # truncation would give 5
    np.testing.assert_array_equal(r[0].numpy(), g["p_rois"])
    np.testing.assert_array_equal(r[1].numpy(), g["rois"])
    np.testing.assert_array_equal(r[2].numpy(), g["class_ids"])
    np.testing.assert_allclose(r[3].numpy(), g["deltas"], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(r[4].argmax(1).numpy().astype(np.uint8), g["mask_labels"])


def lits_predict_oracle(g):
    """The fork's predict('training') + compute_losses dataflow restated with the oracle's functions, for one of the
    two predict_lits_* goldens.  Returns (params, outputs dict, losses, total)."""
    stage = str(g["stage"])
    sd = golden_state_dict(g)
    params = {k: v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v
              for k, v in sd.items()}
    with torch.no_grad():
        for k in ("rpn.conv_bbox.weight", "rpn.conv_bbox.bias"):
            params[k].mul_(float(g["rpn_bbox_gain"]))
    image = t(g["image"])[None, None]
    D, H, W = [int(v) for v in image.shape[2:]]
    p2, p3 = orc.fpn(image, params, layers=LITS_LAYERS, stem_pad=LITS_STEM_PAD)
    l2, pr2, b2 = orc.rpn(p2, params)
    l3, pr3, b3 = orc.rpn(p3, params)
    rpn_logits, rpn_probs, rpn_box = [torch.cat(v, dim=1) for v in ((l2, l3), (pr2, pr3), (b2, b3))]
    anchors = torch.from_numpy(orc.generate_pyramid_anchors((16, 32), np.array([[D // 8, H // 8, W // 8],
                                                                               [D // 16, H // 16, W // 16]]),
                                                           (8, 16), 1)).float()
    rois_all, _, _ = orc.proposal_layer(rpn_probs[0].detach(), rpn_box[0].detach(), anchors, 64, 0.7, (D, H, W), 64)
    gt_boxes = t(g["gt_boxes"][0]) / torch.tensor([D, H, W, D, H, W], dtype=torch.float32)
    lab = torch.from_numpy(g["gt_labels"].astype(np.int64))
    onehot = torch.stack([(lab == k) for k in range(3)], dim=0).float()
    train_rois, ratio = (50, 0.33) if stage == "beginning" else (4, 1.0)        # LiTS_2017/config.py:216-226
    p_rois, rois, cls_ids, deltas, masks = orc.detection_target_layer(
        rois_all, torch.from_numpy(g["gt_class_ids"][0].astype(np.int64)), gt_boxes, onehot, (32, 48, 32),
        torch.from_numpy(g["randperm0"]), torch.from_numpy(g["randperm1"]), train_rois, ratio, count_round=True)
    out = dict(rpn_logits=rpn_logits, rpn_box=rpn_box, p_rois=p_rois, rois=rois, cls_ids=cls_ids, deltas=deltas,
               masks=masks)
    zero = torch.zeros(())
    if stage == "beginning":        # detector phase: classifier on all sampled RoIs, no mask head
        cls_logits, _, cls_bbox = orc.classifier([p2[0], p3[0]], rois, params, [4, 4, 4])
        out.update(cls_logits=cls_logits, cls_bbox=cls_bbox)
        losses = [orc.rpn_class_loss(torch.from_numpy(g["rpn_match"]), rpn_logits),
                  orc.rpn_bbox_loss(t(g["rpn_bbox_t"]), torch.from_numpy(g["rpn_match"]), rpn_box),
                  orc.mrcnn_class_loss(cls_ids, cls_logits), orc.mrcnn_bbox_loss(deltas, cls_ids, cls_bbox), zero, zero]
    else:                           # mask phase: U-Net on the positives, weighted CE + raw-Sobel edge MSE
        m_logits, m_probs = orc.mask_head(image[0], p_rois, params, [32, 48, 32], stage, dropout_masks=None)
        out.update(m_logits=m_logits)
        losses = [zero, zero, zero, zero, orc.mask_ce_loss_weighted(masks.double(), m_logits, [1.0, 1.0, 100.0]),
                  orc.edge_loss_raw(masks.double(), m_probs)]
    total = sum(float(w) * l for w, l in zip(g["loss_weights"], losses))
    return params, out, losses, total


@pytest.mark.parametrize("stage", ["beginning", "together"])
def test_predict_lits_two_phase(stage):
    """The fork's own predict('training') + compute_losses + backward in both of its training phases
    (gen_golden.case_predict_lits) against the oracle chain -- pins the P3D35 detector, the int(round()) sampler, the
    phase control flow (which head runs, which losses are zero, which parameters are frozen) and the fork's losses."""
    g = load_golden("predict_lits_" + stage)
    params, out, losses, total = lits_predict_oracle(g)
    np.testing.assert_allclose(out["rpn_logits"].detach().numpy(), g["rpn_class_logits"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["rpn_box"].detach().numpy(), g["rpn_pred_bbox"], rtol=1e-4, atol=1e-5)
    assert out["p_rois"].shape[0] == int(g["n_pos"]) and out["rois"].shape[0] == int(g["n_rois"])
    np.testing.assert_array_equal(out["cls_ids"].numpy(), g["target_class_ids"])
    np.testing.assert_allclose(out["deltas"].numpy(), g["target_deltas"], rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(out["masks"].argmax(1).numpy().astype(np.uint8), g["target_mask_labels"])
    if stage == "beginning":
        np.testing.assert_allclose(out["cls_logits"].detach().numpy(), g["mrcnn_class_logits"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(out["cls_bbox"].detach().numpy(), g["mrcnn_bbox"], rtol=1e-4, atol=1e-5)
    else:
        ml = out["m_logits"].detach().numpy()
        assert np.abs(ml[:, :, ::2, ::2, ::2] - g["mask_logits_sub"]).max() < 1e-3
    for i, (a, r) in enumerate(zip(losses, g["losses"])):
        assert abs(float(a.detach()) - float(r)) <= 1e-4 * max(abs(float(r)), 1e-3), "loss %d: %g vs %g" % (i, float(a), r)
    assert abs(float(total) - float(g["total"])) <= 1e-4 * abs(float(g["total"]))
    total.backward()
    # which tensors the phase trains (LiTS_2017/model.py:1282-1296): 'together' freezes everything but the mask branch
    trainable = set(str(k) for k in g["trainable"])
    with_grad = set(str(k) for k in g["with_grad"])
    if stage == "together":      # FPN + RPN frozen; the classifier (built after the freeze) keeps requires_grad but gets no gradient
        assert all(k.startswith(("mask.", "classifier.")) for k in trainable)
        assert all(k.startswith("mask.") for k in with_grad)
    got = set(k for k, v in params.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None
              and float(v.grad.abs().max()) > 0)
    assert got & trainable == with_grad & got, sorted((got & trainable) ^ (with_grad & got))[:5]
    for k in [k[5:] for k in g if k.startswith("grad:")]:
        a, r = params[k].grad.numpy(), g["grad:" + k]
        e = np.linalg.norm((a - r).ravel()) / max(np.linalg.norm(r.ravel()), 1e-30)
        assert e < 2e-2, "%s: rel L2 %.3e" % (k, e)


def test_resize_known_answers():
    """f-4: the oracle's restatement of skimage.transform.resize (scipy.ndimage.zoom, grid-constant, grid_mode) against known
    answers derived by hand from skimage's published algorithm (tests/golden/gen_resize_kat.py) -- scikit-image itself is not
    in the image, so this is what pins the restatement."""
    import module_cases as mc
    mc.check_resize_kat_oracle()
