"""The caller harness of the hot path: the dataflow of ``MaskRCNN.predict(mode='training')`` +
``compute_losses`` (model.py:1391-1514, 984-1000) with the module tree -- and therefore the state-dict keys
``fpn.* / rpn.* / classifier.* / mask.modified_u_net.*`` -- of the reference's ``MaskRCNN``
(model.py:1259-1304), so a reference checkpoint loads with ``strict=True``.

Out of scope here: sampling of detection targets.  The head RoI sets (positives
first, then negatives), their class ids / box deltas and the uint8 mask labels are inputs.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import backbone, layers, model, ops, utils

# The mask head runs on its own HIP stream beside FPN / RPN / classifier (see _mask_head); CFUN_OVERLAP_MASK_HEAD=0 puts
# everything on one stream.  Round 1 measured the overlap 1.2 ms per step SLOWER (56.8 vs 55.6 ms) and left it off -- but
# that step still had a dozen host synchronisations inside (pageable uploads, nonzero() in the losses and the RoI level
# split, the NMS count), each of which stalled BOTH streams.  With the step free of host waits (round 4) the few hundred
# small, latency-bound launches of the detector hide beside the U-Net's chip-filling convs: 44.9 -> 42.9 ms per step at
# cfg2 (profiles/round4_*), same bits (test_mask_head_side_stream).
OVERLAP_MASK_HEAD = os.environ.get("CFUN_OVERLAP_MASK_HEAD", "1") == "1"
# Round 6: in the heart 'finetune' step the softmax of Mask.forward (model.py:799), the cross entropy (model.py:909-935) and the
# Sobel edge loss (model.py:938-981) are ONE pass over the logits, their backward ONE pass over the probabilities
# (ops.mask_losses_fused, csrc/loss_fused.hip) instead of three forward passes + a gather over a 1.5 GB coefficient field.
# CFUN_FUSED_MASK_LOSS=0: the separate kernels of rounds 1-5 (A/B runs; same values to fp32 summation order).
FUSED_MASK_LOSS = os.environ.get("CFUN_FUSED_MASK_LOSS", "1") == "1"
MASK_HEAD_BEFORE_PROLOGUE = os.environ.get("CFUN_MASK_HEAD_FIRST", "1") == "1"


class CFUNHotPath(nn.Module):
    def __init__(self, config, test_flag=False):
        super().__init__()
        self.config = config
        d, h, w = config.image_dhw
        if any(v % 16 for v in (d, h, w)):
            raise Exception("Image size must be dividable by 16. Use 256, 320, 512, ... etc.")  # model.py:1263-1265
        layers = tuple(getattr(config, "BACKBONE_LAYERS", (2, 3)))
        net = backbone.P3D(backbone.Bottleneck, list(layers), config=config,
                           stem_kd=getattr(config, "BACKBONE_STEM_KD", 3))
        c1, c2, c3 = net.stages()
        self.fpn = model.FPN(c1, c2, c3, out_channels=config.TOP_DOWN_PYRAMID_SIZE, config=config)
        anchors = utils.generate_pyramid_anchors(config.RPN_ANCHOR_SCALES, config.RPN_ANCHOR_RATIOS,
                                                 utils.compute_backbone_shapes(config, config.IMAGE_SHAPE),
                                                 config.BACKBONE_STRIDES, config.RPN_ANCHOR_STRIDE)
        self.anchors = torch.from_numpy(anchors).float()   # plain attribute, not in the state dict (model.py:1276)
        self.rpn = model.RPN(len(config.RPN_ANCHOR_RATIOS), config.RPN_ANCHOR_STRIDE, config.TOP_DOWN_PYRAMID_SIZE,
                             config.RPN_CONV_CHANNELS)
        if self.mask_phase_only:      # LiTS fork, stage != 'beginning': everything built SO FAR -- FPN and RPN -- is
            for p in self.parameters():   # frozen (LiTS_2017/model.py:1309-1311).  The classifier is built after this
                p.requires_grad = False   # point and keeps requires_grad (it just never runs in this phase, so it
                                          # receives no gradient); pinned by tests/golden/predict_lits_together.npz
        self.classifier = model.Classifier(config.TOP_DOWN_PYRAMID_SIZE, config.POOL_SIZE, config.IMAGE_SHAPE, 2,
                                           config.FPN_CLASSIFY_FC_LAYERS_SIZE, test_flag)
        self.mask = model.Mask(1, config.MASK_POOL_SIZE, config.NUM_CLASSES, config.UNET_MASK_BRANCH_CHANNEL,
                               config.STAGE, test_flag, dropout_p=getattr(config, "UNET_DROPOUT", 0.6))
        if not config.TRAIN_BN:                           # model.py:1297-1304
            for m in self.modules():
                if isinstance(m, nn.BatchNorm3d):
                    for p in m.parameters():
                        p.requires_grad = False
        self.initialize_weights()

    @property
    def detector_phase_only(self):
        """LiTS fork 'beginning': no mask head, mask losses 0 (LiTS_2017/model.py:985-994, 1528-1535)."""
        return bool(getattr(self.config, "STAGE_SPLIT", False)) and self.config.STAGE == "beginning"

    @property
    def mask_phase_only(self):
        """LiTS fork, any other stage: no classifier head, detection losses 0 (LiTS_2017/model.py:995-1001, 1536-1548)."""
        return bool(getattr(self.config, "STAGE_SPLIT", False)) and self.config.STAGE != "beginning"

    def initialize_weights(self):
        """model.py:1306-1319: xavier-uniform convs, zero biases, BN 1/0, Linear N(0, 0.01)."""
        from .layers import Conv3dParams
        for m in self.modules():
            if isinstance(m, Conv3dParams):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.BatchNorm3d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, nn.Linear):
                m.weight.data.normal_(0, 0.01)
                m.bias.data.zero_()

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.anchors = fn(self.anchors)
        return out

    # ------------------------------------------------------------------------------------------
    _loss_weights = {}

    def backbone_rpn(self, image):
        """image [1,1,D,H,W] -> p2, p3 (NDHWC), rpn_class_logits [1,A,2], rpn_probs, rpn_bbox [1,A,6]."""
        p2, p3 = self.fpn.forward_ndhwc(ops.to_ndhwc(image))
        outs = [self.rpn.forward_ndhwc(p) for p in (p2, p3)]           # levels concatenated p2-then-p3
        logits, probs, bbox = [torch.cat([o[i] for o in outs], dim=1) for i in range(3)]
        return p2, p3, logits, probs, bbox

    def proposals(self, rpn_probs, rpn_bbox, mode="training", lazy=False):
        cfg = self.config
        count = cfg.POST_NMS_ROIS_TRAINING if mode == "training" else cfg.POST_NMS_ROIS_INFERENCE
        return model.proposal_layer([rpn_probs, rpn_bbox], proposal_count=count, nms_threshold=cfg.RPN_NMS_THRESHOLD,
                                    anchors=self.anchors, config=cfg, lazy=lazy)

    def fused_mask_losses(self):
        """Does this configuration's step take the one-pass mask losses (heart 'finetune': CE + Sobel-magnitude edge loss)?"""
        cfg = self.config
        return bool(FUSED_MASK_LOSS and cfg.STAGE == "finetune" and getattr(cfg, "MASK_CE_CLASS_WEIGHTS", None) is None
                    and not getattr(cfg, "EDGE_LOSS_RAW_SOBEL", False) and int(cfg.NUM_CLASSES) in (8, 3)
                    and min(int(v) for v in cfg.MASK_SHAPE) >= 3)

    def _mask_head(self, image, p_rois, softmax=True):
        """The mask head (RoIAlign of the raw image + U-Net + softmax) on its own HIP stream: it reads only the image
        and the positive RoIs, so FPN / RPN / proposals / classifier -- a few hundred small, low-occupancy launches --
        run beside the U-Net's chip-filling convs instead of in front of them; autograd replays every node on its
        forward stream, so the two backward passes overlap the same way.  Returns (logits, probs, join): ``join()``
        makes the current stream wait for the head (call it before the outputs are consumed)."""
        img = ops.to_ndhwc(image)[0]
        if not (image.is_cuda and OVERLAP_MASK_HEAD):
            logits, probs = self.mask.forward_ndhwc(img, p_rois, softmax)
            return logits, probs, (lambda: None)
        main = torch.cuda.current_stream(image.device)
        side = ops.side_stream(image.device, "mask_head", priority=int(os.environ.get("CFUN_MASK_STREAM_PRIORITY", "0")))
        side.wait_stream(main)                  # image / RoIs / this step's weights are ready
        with torch.cuda.stream(side):
            logits, probs = self.mask.forward_ndhwc(img, p_rois, softmax)

        def join():
            main.wait_stream(side)
            for t in (logits, probs):           # allocated on `side`, consumed (and possibly freed) on `main`
                if t is not None:
                    t.record_stream(main)
        return logits, probs, join

    def predict_training(self, image, p_rois, n_rois, lazy_rois=False, defer_mask_probs=False):
        """BatchNorm stays in eval mode while the rest trains (model.py:1397-1406): folded BN needs no switch.
        p_rois [n_pos,6] / n_rois [n_neg,6] normalised.  Returns a dict of the path's outputs.  ``lazy_rois``:
        ``rpn_rois`` comes back as a ``model.LazyProposals`` (the NMS keep count is not read yet: no host wait here).
        ``defer_mask_probs`` (training_step): where the configuration takes the one-pass mask losses, ``mrcnn_mask`` is left None
        here and filled in by ``compute_losses`` (the fused loss pass produces the probabilities)."""
        self.train()
        mask_logits = mask_probs = cls_logits = cls_probs = cls_bbox = None
        join = lambda: None
        early = MASK_HEAD_BEFORE_PROLOGUE and not self.detector_phase_only
        if early:
            # Round 6: the mask head is enqueued BEFORE the detector's step prologue (the batched weight preparation of FPN /
            # RPN / classifier, the folded-bias launch, their table uploads): its stream waits for the main stream as it stands
            # HERE -- the previous step's joined backward -- not for prologue work the U-Net does not read (its own weight
            # preparation runs inside Modified3DUNet.forward_ndhwc, on its own stream)
            mask_logits, mask_probs, join = self._mask_head(
                image, p_rois, softmax=not (defer_mask_probs and self.fused_mask_losses()))
        layers.begin_step(self)         # the step's conv-bias folds under frozen BatchNorm: one multi-tensor launch
        ok = False
        try:
            if not self.detector_phase_only and not early:
                mask_logits, mask_probs, join = self._mask_head(        # enqueued first, on its own stream
                    image, p_rois, softmax=not (defer_mask_probs and self.fused_mask_losses()))
            p2, p3, rpn_logits, rpn_probs, rpn_bbox = self.backbone_rpn(image)
            rpn_rois = self.proposals(rpn_probs, rpn_bbox, "training", lazy=lazy_rois)
            if not self.mask_phase_only:
                rois = torch.cat([p_rois, n_rois], dim=0)
                cls_logits, cls_probs, cls_bbox = self.classifier.forward_ndhwc([p2[0], p3[0]], rois)
            ok = True
        finally:
            layers.end_step(self, ok)   # (a pass that raised must not leave its partial record as the net's weight plan)
        join()
        return dict(rpn_class_logits=rpn_logits, rpn_probs=rpn_probs, rpn_bbox=rpn_bbox, rpn_rois=rpn_rois,
                    mrcnn_class_logits=cls_logits, mrcnn_class=cls_probs, mrcnn_bbox=cls_bbox,
                    mrcnn_mask_logits=mask_logits, mrcnn_mask=mask_probs, p2=p2, p3=p3)

    def predict_training_full(self, image, gt_class_ids, gt_boxes, gt_labels, perms=None):
        """The un-injected ``predict(mode='training')`` (model.py:1462-1514): the RoI sets come from
        ``detection_target_layer`` on this step's own proposals.  gt_boxes [G,6] in voxels (z1,y1,x1,z2,y2,x2),
        gt_labels uint8 [D,H,W].  The returned dict also carries the targets; with no positive proposal the head
        outputs are None (the reference returns empty tensors and ``compute_losses`` yields zeros)."""
        self.train()
        cfg = self.config
        height, width, depth = [float(v) for v in cfg.IMAGE_SHAPE[:3]]
        p2, p3, rpn_logits, rpn_probs, rpn_bbox = self.backbone_rpn(image)
        rpn_rois = self.proposals(rpn_probs, rpn_bbox, "training")
        scale = torch.tensor([depth, height, width, depth, height, width], dtype=torch.float32, device=image.device)
        p_rois, rois, tcls, tdeltas, tlabels = model.detection_target_layer(
            rpn_rois, gt_class_ids, gt_boxes.float() / scale, gt_labels, cfg, perms)
        out = dict(rpn_class_logits=rpn_logits, rpn_probs=rpn_probs, rpn_bbox=rpn_bbox, rpn_rois=rpn_rois, p2=p2, p3=p3,
                   p_rois=p_rois, rois=rois, target_class_ids=tcls, target_deltas=tdeltas, mask_labels=tlabels,
                   mrcnn_class_logits=None, mrcnn_class=None, mrcnn_bbox=None, mrcnn_mask_logits=None,
                   mrcnn_mask=None)
        if rois.shape[0]:
            # the mask head waits for this step's targets, then runs beside the classifier head; in backward it
            # overlaps the FPN / RPN gradients (see _mask_head)
            join = lambda: None
            if not self.detector_phase_only and p_rois.shape[0]:
                out["mrcnn_mask_logits"], out["mrcnn_mask"], join = self._mask_head(image, p_rois)
            if not self.mask_phase_only:
                out["mrcnn_class_logits"], out["mrcnn_class"], out["mrcnn_bbox"] = self.classifier.forward_ndhwc(
                    [p2[0], p3[0]], rois)
            join()
        return out

    @torch.no_grad()
    def predict_inference(self, image, window=None):
        """``predict(mode='inference')`` (model.py:1436-1461): proposals (POST_NMS_ROIS_INFERENCE) -> classifier ->
        detection_layer / refine_detections -> mask head on the detected boxes.  Returns
        [detections [1,M,8] in voxels, mrcnn_mask probabilities [1,M,C,d,h,w]]; M may be 0 (the reference crashes
        there)."""
        self.eval()
        cfg = self.config
        height, width, depth = [int(v) for v in cfg.IMAGE_SHAPE[:3]]
        if window is None:
            window = (0, 0, 0, depth, height, width)
        p2, p3, _, rpn_probs, rpn_bbox = self.backbone_rpn(image)
        rpn_rois = self.proposals(rpn_probs, rpn_bbox, "inference")
        _, cls_probs, cls_bbox = self.classifier.forward_ndhwc([p2[0], p3[0]], rpn_rois[0])
        det = model.detection_layer(cfg, rpn_rois, cls_probs, cls_bbox, window)
        n_cls = int(cfg.NUM_CLASSES)
        if det.shape[0] == 0:
            ms = tuple(int(v) for v in cfg.MASK_SHAPE)
            return [det.unsqueeze(0), torch.zeros((1, 0, n_cls) + ms, device=det.device)]
        if self.detector_phase_only:      # LiTS fork 'beginning': no mask branch yet, zeros (LiTS_2017/model.py:1485-1489)
            ms = tuple(int(v) for v in cfg.MINI_MASK_SHAPE)
            return [det.unsqueeze(0), torch.zeros((1, det.shape[0], n_cls) + ms, device=det.device)]
        scale = torch.tensor([depth, height, width, depth, height, width], dtype=torch.float32, device=det.device)
        _, mask_probs = self.mask.forward_ndhwc(ops.to_ndhwc(image)[0], det[:, :6] / scale)
        return [det.unsqueeze(0), ops.to_ncdhw(mask_probs).unsqueeze(0)]

    def detect(self, image, window=None):
        """``MaskRCNN.detect`` for one already molded volume [1,1,D,H,W] (model.py:1341-1389 without the host-side
        resize / z-score of ``mold_inputs``): inference forward, then ``unmold_detections``.  Returns the
        reference's result dict (rois (y1,x1,z1,y2,x2,z2), class_ids, scores, mask [H,W,D])."""
        det, masks = self.predict_inference(image, window)
        if det.shape[1] == 0:
            return dict(rois=np.zeros((0, 6), np.int32), class_ids=np.zeros((0,), np.int32),
                        scores=np.zeros((0,), np.float32), mask=None)
        height, width, depth = [int(v) for v in self.config.IMAGE_SHAPE[:3]]
        win = (0, 0, 0, depth, height, width) if window is None else window
        probs = masks[0].permute(0, 2, 3, 4, 1).contiguous()          # [N, d, h, w, C]
        unmold = model.unmold_detections_overlap if getattr(self.config, "UNMOLD_OVERLAP_TILE", False) \
            else model.unmold_detections
        rois, class_ids, scores, mask = unmold(det[0], probs, [1, depth, height, width], win)
        return dict(rois=rois, class_ids=class_ids, scores=scores, mask=mask)

    def mold_inputs(self, images):
        """``MaskRCNN.mold_inputs`` (model.py:1774-1810; LiTS_2017/model.py:1730-1775 for the fork's configs) on the
        device: raw volumes [H,W,D,1] (heart) / [H,W,D] (LiTS) -> (molded [N,1,D,H,W], image_metas, windows)."""
        dev = next(self.parameters()).device
        if hasattr(self.config, "PAD_IMAGE_SHAPE"):
            return utils.mold_inputs_lits(self.config, images, device=dev)
        return utils.mold_inputs(self.config, images, device=dev)

    def detect_images(self, images):
        """``MaskRCNN.detect`` (model.py:1341-1389) from raw volumes: mold (resize + normalise, on the device), run the
        inference dataflow, un-mold.  One result dict per image, as the reference returns."""
        molded, _, windows = self.mold_inputs(images)
        return [self.detect(molded[i:i + 1], window=tuple(float(v) for v in windows[i])) for i in range(molded.shape[0])]

    def compute_losses(self, out, rpn_match, rpn_bbox_t, target_class_ids, target_deltas, mask_labels):
        """The 6 losses of model.py:984-1000 (mask labels: uint8 [n_pos,d,h,w])."""
        zero = torch.zeros((), device=mask_labels.device)
        if self.mask_phase_only:      # LiTS_2017/model.py:995-999
            losses = [zero, zero, zero, zero]
        else:
            losses = [model.compute_rpn_class_loss(rpn_match, out["rpn_class_logits"]),
                      model.compute_rpn_bbox_loss(rpn_bbox_t, rpn_match, out["rpn_bbox"]),
                      model.compute_mrcnn_class_loss(target_class_ids, out["mrcnn_class_logits"]),
                      model.compute_mrcnn_bbox_loss(target_deltas, target_class_ids, out["mrcnn_bbox"])]
        if self.detector_phase_only:  # LiTS_2017/model.py:993-994
            return losses + [zero, zero]
        cw = getattr(self.config, "MASK_CE_CLASS_WEIGHTS", None)
        raw = getattr(self.config, "EDGE_LOSS_RAW_SOBEL", False)
        if cw is not None or raw:             # LiTS fork: class-weighted CE, edge loss on the raw Sobel responses
            ce = ops.mask_cross_entropy(out["mrcnn_mask_logits"], mask_labels, weight=cw)
            # the fork computes its edge loss in EVERY non-'beginning' stage ('finetune' and 'together',
            # LiTS_2017/model.py:995-1001); the heart code only in 'finetune' (model.py:995-996)
            edge_on = self.config.STAGE != "beginning" if (getattr(self.config, "STAGE_SPLIT", False) or raw) \
                else self.config.STAGE == "finetune"
            if edge_on:
                edge = ops.edge_loss_raw(out["mrcnn_mask"], mask_labels) if raw else \
                    ops.edge_loss(out["mrcnn_mask"], mask_labels)
            else:
                edge = torch.zeros((), device=mask_labels.device)
            losses += [ce, edge]
        elif self.config.STAGE == "finetune" and out["mrcnn_mask"] is None:
            # one pass forward (softmax + CE + Sobel edge loss), one pass backward; the probabilities come out of it
            ce, edge, out["mrcnn_mask"] = ops.mask_losses_fused(out["mrcnn_mask_logits"], mask_labels)
            losses += [ce, edge]
        elif self.config.STAGE == "finetune":   # CE + Sobel edge loss share one fused backward pass
            losses += list(ops.mask_losses(out["mrcnn_mask_logits"], out["mrcnn_mask"], mask_labels))
        else:
            losses += [ops.mask_cross_entropy(out["mrcnn_mask_logits"], mask_labels),
                       torch.zeros((), device=mask_labels.device)]
        return losses

    def total_loss(self, losses):
        w = self.config.LOSS_WEIGHTS
        keys = ("rpn_class_loss", "rpn_bbox_loss", "mrcnn_class_loss", "mrcnn_bbox_loss", "mrcnn_mask_loss",
                "mrcnn_mask_edge_loss")
        vals = tuple(float(w[k]) for k in keys)
        key = (vals, str(losses[0].device))
        wv = self._loss_weights.get(key)
        if wv is None:          # (a constant per weights and device: built once, not one host-to-device copy per step)
            wv = self._loss_weights[key] = torch.tensor(vals, dtype=torch.float32, device=losses[0].device)
        return (torch.stack(list(losses)) * wv).sum()      # 3 launches (and 3 in the backward) instead of 12 (+ 6)


# ---------------------------------------------------------------------------------------------- synthetic step
def synthetic_inputs(config, device, seed=0):
    """The synthetic training sample of SURVEY.md section 8(d): z-scored piecewise-constant 8-class CT,
    GT box = the central half in (y, x), 4 positive RoIs (GT box shifted by (0,+-8,+-8) voxels, IoU 0.78)
    and 8 negative 64^3-style corner boxes, their class ids / deltas, uint8 mask labels at MASK_SHAPE and
    RPN targets.  Host-side bookkeeping (numpy), done once outside the timed region."""
    rng = np.random.default_rng(seed)
    d, h, w = config.image_dhw
    ncls = config.NUM_CLASSES
    y1, y2, x1, x2 = h // 4, 3 * h // 4, w // 4, 3 * w // 4
    lab = np.zeros((d, h, w), np.uint8)
    nfg = ncls - 1
    for k in range(nfg):                                    # equal slabs along x inside the GT box
        lab[:, y1:y2, x1 + (x2 - x1) * k // nfg: x1 + (x2 - x1) * (k + 1) // nfg] = k + 1
    hu = np.where(lab == 0, -1000.0, (lab.astype(np.float32) - 1) * 50.0) + rng.normal(0, 30, lab.shape)
    img = ((hu - hu.mean()) / hu.std()).astype(np.float32)
    gt = np.array([0, y1, x1, d, y2, x2], np.float32)
    norm = np.array([d, h, w, d, h, w], np.float32)
    sh = 8.0 * min(1.0, h / 256.0)
    p_rois = np.stack([gt + np.array([0, sy, sx, 0, sy, sx], np.float32) * sh
                       for sy in (-1, 1) for sx in (-1, 1)]) / norm
    cz, cy, cx = d // 2, h // 4, w // 4
    n_rois = np.array([[z0, y0, x0, z0 + cz, y0 + cy, x0 + cx] for z0 in (0, d - cz) for y0 in (0, h - cy)
                       for x0 in (0, w - cx)], np.float32) / norm
    target_class_ids = np.array([1, 2, 3, 4] + [0] * 8, np.int64)
    p_t, gt_t = torch.from_numpy(p_rois), torch.from_numpy(np.tile(gt / norm, (4, 1)))
    deltas = utils.box_refinement(p_t, gt_t) / torch.from_numpy(config.BBOX_STD_DEV).float()
    target_deltas = torch.cat([deltas, torch.zeros(8, 6)], dim=0)
    # mask labels: crop with int() truncation, nearest resize to MASK_SHAPE (model.py:481-493)
    ms = config.MASK_SHAPE
    labels = np.zeros((4,) + tuple(ms), np.uint8)
    for i in range(4):
        b = p_rois[i]
        z0, z1 = int(d * b[0]), int(d * b[3]); yy0, yy1 = int(h * b[1]), int(h * b[4]); xx0, xx1 = int(w * b[2]), int(w * b[5])
        crop = lab[z0:z1, yy0:yy1, xx0:xx1]
        idx = [np.clip(np.floor((np.arange(o) + 0.5) * (n / o)).astype(np.int64), 0, n - 1)
               for o, n in zip(ms, crop.shape)]
        labels[i] = crop[idx[0][:, None, None], idx[1][None, :, None], idx[2][None, None, :]]
    # RPN targets: best anchors by IoU with the GT box positive, far anchors negative
    anchors = utils.generate_pyramid_anchors(config.RPN_ANCHOR_SCALES, config.RPN_ANCHOR_RATIOS,
                                             utils.compute_backbone_shapes(config, config.IMAGE_SHAPE),
                                             config.BACKBONE_STRIDES, config.RPN_ANCHOR_STRIDE).astype(np.float32)
    vol_a = np.prod(anchors[:, 3:] - anchors[:, :3], axis=1)
    iou = utils.compute_iou(gt, anchors, np.prod(gt[3:] - gt[:3]), vol_a)
    rpn_match = np.zeros((1, anchors.shape[0], 1), np.int32)
    rpn_match[0, iou < 0.1, 0] = -1
    pos = np.sort(np.argsort(-iou)[:8])
    rpn_match[0, pos, 0] = 1
    rpn_bbox_t = np.zeros((1, config.RPN_TRAIN_ANCHORS_PER_IMAGE, 6), np.float32)
    rb = utils.box_refinement(torch.from_numpy(anchors[pos]), torch.from_numpy(np.tile(gt, (len(pos), 1))))
    rpn_bbox_t[0, :len(pos)] = rb.numpy() / config.RPN_BBOX_STD_DEV
    t = lambda a: torch.as_tensor(a).to(device)
    return dict(image=t(img)[None, None], p_rois=t(p_rois), n_rois=t(n_rois), target_class_ids=t(target_class_ids),
                target_deltas=target_deltas.to(device), mask_labels=t(labels), rpn_match=t(rpn_match),
                rpn_bbox_t=t(rpn_bbox_t), labels_volume=lab)


def training_step_full(net, image, gt_class_ids, gt_boxes, gt_labels, rpn_match, rpn_bbox_t, perms=None):
    """Forward + 6 losses + backward of the un-injected dataflow (targets sampled on device from the proposals)."""
    out = net.predict_training_full(image, gt_class_ids, gt_boxes, gt_labels, perms)
    # "no RoIs" = detection_target_layer found no positive proposal and both heads were skipped (model.py:1481-1491).
    # NOT `mrcnn_mask_logits is None`: in the LiTS fork's detector phase the mask head never runs, yet the
    # classifier's two losses are computed (LiTS_2017/model.py:985-994).
    if out["rois"].shape[0] == 0:
        z = torch.zeros((), device=image.device)
        losses = [model.compute_rpn_class_loss(rpn_match, out["rpn_class_logits"]),
                  model.compute_rpn_bbox_loss(rpn_bbox_t, rpn_match, out["rpn_bbox"]), z, z, z, z]
    else:
        losses = net.compute_losses(out, rpn_match, rpn_bbox_t, out["target_class_ids"], out["target_deltas"],
                                    out["mask_labels"])
    total = net.total_loss(losses)
    total.backward()
    return out, losses, total


def training_step(net, s):
    """One forward + 6 losses + backward of the hot path on the sample ``s`` (no optimizer step).  The head RoI sets are
    inputs here, so nothing consumes the proposals inside the step: their NMS keep count is read only after the backward
    pass has been enqueued (model.LazyProposals) and the host never waits for the GPU mid-step."""
    out = net.predict_training(s["image"], s["p_rois"], s["n_rois"], lazy_rois=True, defer_mask_probs=True)
    losses = net.compute_losses(out, s["rpn_match"], s["rpn_bbox_t"], s["target_class_ids"], s["target_deltas"],
                                s["mask_labels"])
    total = net.total_loss(losses)
    total.backward()
    out["rpn_rois"] = out["rpn_rois"].resolve()
    return out, losses, total
