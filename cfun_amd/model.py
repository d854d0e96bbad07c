"""Detection glue of the hot path on HIP kernels -- the ★ symbols of the reference's ``model.py``:
FPN (124-148), RPN (700-743), proposal_layer (199-258), RoI_Align (265-289), pyramid_roi_align (292-370),
Classifier (750-784), Mask (787-801) and the two mask losses (909-981).  Constructors, call signatures,
return shapes and state-dict keys follow the reference; tensors at the surface are NCDHW-shaped views of
NDHWC buffers (channels_last_3d), internally everything stays NDHWC.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import mask_branch, ops, utils
from .layers import Conv3dParams, default_algo, folded_bn, frozen_bn
from .ops import ACT_NONE, ACT_RELU


def _stage_ndhwc(stage, x):
    if hasattr(stage, "forward_ndhwc"):
        return stage.forward_ndhwc(x)
    return ops.to_ndhwc(stage(ops.to_ncdhw(x)))   # foreign (e.g. reference) stage modules


class FPN(nn.Module):
    def __init__(self, C1, C2, C3, out_channels, config):
        super().__init__()
        self.out_channels = out_channels
        self.C1, self.C2, self.C3 = C1, C2, C3
        self.P3_conv1 = Conv3dParams(config.BACKBONE_CHANNELS[1] * 4, out_channels, 1)
        self.P3_conv2 = Conv3dParams(out_channels, out_channels, 3, padding=1)
        self.P2_conv1 = Conv3dParams(config.BACKBONE_CHANNELS[0] * 4, out_channels, 1)
        self.P2_conv2 = Conv3dParams(out_channels, out_channels, 3, padding=1)

    def forward_ndhwc(self, x):
        c2 = _stage_ndhwc(self.C2, _stage_ndhwc(self.C1, x))
        c3 = _stage_ndhwc(self.C3, c2)
        p3 = self.P3_conv1(c3)
        p2 = self.P2_conv1(c2, res=p3, res_up2=True)        # P2_conv1(c2) + nearest_up2(p3), model.py:144
        return self.P2_conv2(p2), self.P3_conv2(p3)

    def forward(self, x):
        p2, p3 = self.forward_ndhwc(ops.to_ndhwc(x))
        return [ops.to_ncdhw(p2), ops.to_ncdhw(p3)]


class RPN(nn.Module):
    def __init__(self, anchors_per_location, anchor_stride, channel, conv_channel):
        super().__init__()
        self.anchors_per_location = anchors_per_location
        self.conv_shared = Conv3dParams(channel, conv_channel, 3, stride=anchor_stride, padding=1)
        self.conv_class = Conv3dParams(conv_channel, 2 * anchors_per_location, 1)
        self.conv_bbox = Conv3dParams(conv_channel, 6 * anchors_per_location, 1)

    def forward_ndhwc(self, p):
        """p [N,D,H,W,C] -> [logits [N,A,2], probs [N,A,2], bbox [N,A,6]], anchors flattened (z,y,x)."""
        n = p.shape[0]
        h = self.conv_shared(p, ACT_RELU)
        # the class (2a) and bbox (6a) 1x1x1 heads run as ONE 8a-channel MFMA GEMM over the shared features
        w = torch.cat([self.conv_class.weight, self.conv_bbox.weight], dim=0)
        b = torch.cat([self.conv_class.bias, self.conv_bbox.bias], dim=0)
        spec = ops.ConvSpec(k=(1, 1, 1), co=w.shape[0], algo=default_algo())
        out = ops.conv3d_w(h, w, spec, shift=b)
        a2 = 2 * self.anchors_per_location
        logits, bbox = ops.split_channels(out, a2)
        logits, bbox = logits.reshape(n, -1, 2), bbox.reshape(n, -1, 6)
        probs = ops.softmax_channels(logits)
        return [logits, probs, bbox]

    def forward(self, x):
        return self.forward_ndhwc(ops.to_ndhwc(x))


# ------------------------------------------------------------------------------------------ proposals
def apply_box_deltas(boxes, deltas):
    """model.py:155-182."""
    size = boxes[:, 3:] - boxes[:, :3]
    center = boxes[:, :3] + 0.5 * size
    center = center + deltas[:, :3] * size
    size = size * torch.exp(deltas[:, 3:])
    lo = center - 0.5 * size
    return torch.cat([lo, lo + size], dim=1)


_CONSTANTS = {}


def _const(values, dtype, device):
    """A small constant tensor on ``device``, built once per (values, dtype, device): torch.tensor(list, device=gpu) is a
    host-to-device copy on every call."""
    key = (tuple(float(v) for v in values), dtype, str(device))
    t = _CONSTANTS.get(key)
    if t is None:
        t = _CONSTANTS[key] = torch.tensor(list(key[0]), dtype=dtype, device=device)
    return t


def clip_boxes(boxes, window):
    """model.py:185-196; window = (z1,y1,x1,z2,y2,x2)."""
    lo = _const([window[0], window[1], window[2]] * 2, boxes.dtype, boxes.device)
    hi = _const([window[3], window[4], window[5]] * 2, boxes.dtype, boxes.device)
    return torch.max(torch.min(boxes, hi), lo)


class LazyProposals:
    """The proposal set with its NMS keep COUNT still on the device: every kernel has been enqueued, only the host read
    of the count -- the one mid-step synchronisation of the path (the reference has it at model.py:244) -- waits until
    ``resolve()``.  ``step.training_step`` (injected RoI sets: nothing downstream consumes the proposals) resolves it after
    the backward pass has been enqueued, so the host never stalls inside the step ("no .item()
    syncs inside the timed region; one sync at the end")."""

    def __init__(self, boxes_norm, keep, count):
        # the count starts its way to the host right behind the NMS kernels; resolve() waits for that copy only
        self.boxes_norm, self.keep, self.count, self._t = boxes_norm, keep, ops.AsyncScalar(count), None

    def resolve(self):
        if self._t is None:
            k = self.keep[:int(self.count.get()[0])].long()
            self._t = self.boxes_norm[k].unsqueeze(0)
        return self._t


def proposal_layer(inputs, proposal_count, nms_threshold, anchors, config=None, lazy=False):
    """model.py:199-258 with the NMS on device (no boxes.cpu().numpy() round trip): scores sorted
    descending, top PRE_NMS_LIMIT, decode, clip, HIP NMS, normalise.  Returns [1, K, 6] (``lazy``: a LazyProposals)."""
    probs, bbox = inputs[0].squeeze(0), inputs[1].squeeze(0)
    scores = probs[:, 1]
    limit = min(config.PRE_NMS_LIMIT, anchors.shape[0])
    scores, order = scores.sort(descending=True)
    order, scores = order[:limit], scores[:limit]
    return proposals_from_candidates(scores, bbox[order.detach()], anchors[order.detach()], proposal_count, nms_threshold,
                                     config, lazy)


def proposals_from_candidates(scores, bbox, anchors, proposal_count, nms_threshold, config, lazy=False):
    """The tail of proposal_layer on an already selected, score-sorted candidate set: scores [K], raw RPN box outputs
    [K,6] and the candidates' anchors [K,6] (voxel units).  Shared with the depth-sharded path, where every rank
    contributes its local top PRE_NMS_LIMIT (cfun_amd.dist.gather_rpn_candidates)."""
    std = _const(np.reshape(config.RPN_BBOX_STD_DEV, [6]), torch.float32, bbox.device).reshape(1, 6)
    boxes = apply_box_deltas(anchors, bbox * std)
    height, width, depth = [float(v) for v in config.IMAGE_SHAPE[:3]]
    boxes = clip_boxes(boxes, (0.0, 0.0, 0.0, depth, height, width))
    norm = _const([depth, height, width, depth, height, width], torch.float32, boxes.device)
    if lazy:
        keep, count = ops.nms3d(boxes, scores, nms_threshold, proposal_count)
        return LazyProposals(boxes / norm, keep, count)
    keep = utils.nms_device(boxes, scores, nms_threshold, proposal_count)
    return (boxes[keep] / norm).unsqueeze(0)


# ------------------------------------------------------------------------------------------ training targets
def bbox_overlaps(boxes1, boxes2):
    """model.py:373-411: IoU matrix [len(boxes1), len(boxes2)] (fp32, no epsilon)."""
    b1, b2 = boxes1[:, None, :], boxes2[None, :, :]
    z1 = torch.max(b1[..., 0], b2[..., 0]); y1 = torch.max(b1[..., 1], b2[..., 1]); x1 = torch.max(b1[..., 2], b2[..., 2])
    z2 = torch.min(b1[..., 3], b2[..., 3]); y2 = torch.min(b1[..., 4], b2[..., 4]); x2 = torch.min(b1[..., 5], b2[..., 5])
    inter = (x2 - x1).clamp(min=0) * (y2 - y1).clamp(min=0) * (z2 - z1).clamp(min=0)
    v1 = (b1[..., 3] - b1[..., 0]) * (b1[..., 4] - b1[..., 1]) * (b1[..., 5] - b1[..., 2])
    v2 = (b2[..., 3] - b2[..., 0]) * (b2[..., 4] - b2[..., 1]) * (b2[..., 5] - b2[..., 2])
    return inter / (v1 + v2 - inter)


def detection_target_layer(proposals, gt_class_ids, gt_boxes, gt_labels, config, perms=None):
    """model.py:414-563 on device: IoU of the proposals with the GT boxes, positive
    (IoU >= 0.5) / negative sampling at ROI_POSITIVE_RATIO, box-refinement targets and the GT mask targets.

    proposals [1,N,6] or [N,6] and gt_boxes [G,6] normalised; gt_class_ids [G]; ``gt_labels`` uint8 [D,H,W] is the
    class id per voxel (the argmax of the reference's one-hot ``gt_masks``), so the mask targets come back as uint8
    labels [n_pos, *MASK_SHAPE] instead of the 8-channel float64 host tensor of model.py:480-493 -- the loss kernels
    take labels.  ``perms`` = (perm_pos, perm_neg) replaces the two torch.randperm draws (model.py:459, 505) for
    reproducible tests.  Returns (positive_rois, rois, target_class_ids, target_deltas, mask_labels), positives
    first; with no positive RoI every entry is empty (the reference then skips the heads, model.py:1481-1491)."""
    proposals = (proposals.squeeze(0) if proposals.dim() == 3 else proposals).detach()   # targets carry no gradient
    dev = proposals.device
    overlaps = bbox_overlaps(proposals, gt_boxes)
    iou_max = overlaps.max(dim=1)[0]
    pos_idx = torch.nonzero(iou_max >= config.DETECTION_TARGET_IOU_THRESHOLD)[:, 0]
    if pos_idx.numel() == 0:
        e = torch.zeros((0, 6), device=dev)
        return e, e, torch.zeros((0,), dtype=torch.long, device=dev), e, torch.zeros(
            (0,) + tuple(config.MASK_SHAPE), dtype=torch.uint8, device=dev)
    # heart: int() truncation (model.py:457, 504); LiTS fork: int(round()) (LiTS_2017/model.py:448, 496)
    count = (lambda v: int(round(v))) if getattr(config, "ROI_COUNT_ROUND", False) else int
    want = count(config.TRAIN_ROIS_PER_IMAGE * config.ROI_POSITIVE_RATIO)
    perm = torch.randperm(pos_idx.numel()) if perms is None else perms[0]
    pos_idx = pos_idx[perm[:want].to(dev)]
    n_pos = pos_idx.numel()
    p_rois = proposals[pos_idx]
    assign = overlaps[pos_idx].max(dim=1)[1]
    std = torch.tensor(np.asarray(config.BBOX_STD_DEV, dtype=np.float32), device=dev)
    deltas = utils.box_refinement(p_rois, gt_boxes[assign]) / std
    class_ids = gt_class_ids[assign].long()
    labels = ops.mask_target_labels(gt_labels, p_rois, config.MASK_SHAPE)
    neg_idx = torch.nonzero(iou_max < config.DETECTION_TARGET_IOU_THRESHOLD)[:, 0]
    rois = p_rois
    if neg_idx.numel() != 0:
        n_neg = count((1.0 / config.ROI_POSITIVE_RATIO) * n_pos - n_pos)
        perm = torch.randperm(neg_idx.numel()) if perms is None else perms[1]
        neg_idx = neg_idx[perm[:n_neg].to(dev)]
        rois = torch.cat([p_rois, proposals[neg_idx]], dim=0)
        class_ids = torch.cat([class_ids, torch.zeros(neg_idx.numel(), dtype=torch.long, device=dev)])
        deltas = torch.cat([deltas, torch.zeros((neg_idx.numel(), 6), device=dev)], dim=0)
    return p_rois, rois, class_ids, deltas, labels


# ------------------------------------------------------------------------------------------ detections
def clip_to_window(window, boxes):
    """model.py:570-581."""
    lo = torch.tensor([window[0], window[1], window[2]] * 2, dtype=boxes.dtype, device=boxes.device)
    hi = torch.tensor([window[3], window[4], window[5]] * 2, dtype=boxes.dtype, device=boxes.device)
    return torch.max(torch.min(boxes, hi), lo)


def refine_detections(rois, probs, deltas, window, config):
    """model.py:584-672 with the per-class NMS (threshold DETECTION_NMS_THRESHOLD, DETECTION_MAX_INSTANCES) on the
    HIP kernel instead of the ``.cpu().numpy()`` round trip of model.py:651.  rois [N,6] normalised, probs [N,K],
    deltas [N,K,6] -> [M, (z1,y1,x1,z2,y2,x2, class_id, score)] in voxel coordinates, best score first.
    Where the reference crashes (no box passes the confidence filter: ``nms_keep`` unbound, model.py:662) this
    returns an empty [0,8] tensor."""
    n = probs.shape[0]
    dev = probs.device
    class_ids = torch.argmax(probs, dim=1)
    idx = torch.arange(n, device=dev)
    class_scores = probs[idx, class_ids]
    std = torch.tensor(np.reshape(config.RPN_BBOX_STD_DEV, [1, 6]), dtype=torch.float32, device=dev)
    refined = apply_box_deltas(rois, deltas[idx, class_ids] * std)
    height, width, depth = [float(v) for v in config.IMAGE_SHAPE[:3]]
    refined = refined * torch.tensor([depth, height, width, depth, height, width], dtype=torch.float32, device=dev)
    refined = torch.round(clip_to_window([float(v) for v in window], refined))
    keep_bool = class_ids > 0
    if config.DETECTION_MIN_CONFIDENCE:
        keep_bool = keep_bool & (class_scores >= config.DETECTION_MIN_CONFIDENCE)
    keep = torch.nonzero(keep_bool)[:, 0]
    if keep.numel() == 0:
        return torch.zeros((0, 8), dtype=torch.float32, device=dev)
    pre_ids, pre_scores, pre_rois = class_ids[keep], class_scores[keep], refined[keep]
    nms_keep = []
    for cid in torch.unique(pre_ids).tolist():
        ixs = torch.nonzero(pre_ids == cid)[:, 0]
        sc, order = pre_scores[ixs].sort(descending=True)
        picked = utils.nms_device(pre_rois[ixs][order].contiguous(), sc.contiguous(),
                                  config.DETECTION_NMS_THRESHOLD, config.DETECTION_MAX_INSTANCES)
        nms_keep.append(keep[ixs[order[picked]]])
    keep = torch.unique(torch.cat(nms_keep))
    count = min(int(config.DETECTION_MAX_INSTANCES), keep.numel())
    keep = keep[class_scores[keep].sort(descending=True)[1][:count]]
    return torch.cat([refined[keep], class_ids[keep].unsqueeze(1).float(), class_scores[keep].unsqueeze(1)], dim=1)


def detection_layer(config, rois, mrcnn_class, mrcnn_bbox, window):
    """model.py:675-689; ``window`` (z1,y1,x1,z2,y2,x2) is what the reference parses out of image_meta."""
    return refine_detections(rois.squeeze(0) if rois.dim() == 3 else rois, mrcnn_class, mrcnn_bbox, window, config)


def unmold_detections(detections, mrcnn_mask, image_shape, window):
    """model.py:1812-1864 + utils.unmold_mask (utils.py:443-460) with the mask resize and the class arg-max fused on
    the device (``cfun_unmold_argmax``).  detections [N,8] (z1,y1,x1,z2,y2,x2,class,score) device tensor or numpy,
    mrcnn_mask [N,d,h,w,C] class probabilities (device tensor, NDHWC per detection), image_shape [c,D,H,W], window
    (z1,y1,x1,z2,y2,x2).  Returns the reference's tuple: boxes (y1,x1,z1,y2,x2,z2) int32 numpy, class ids
    ``np.arange(1, 8)`` (sic, model.py:1864), scores, class map [H,W,D] (int64 numpy, = argmax over classes of the
    FIRST detection's un-molded mask, 0 outside its box)."""
    det = detections.detach().cpu().numpy() if torch.is_tensor(detections) else np.asarray(detections)
    zero_ix = np.where(det[:, 6] == 0)[0]
    n = zero_ix[0] if zero_ix.shape[0] > 0 else det.shape[0]
    boxes = det[:n, :6].astype(np.int32)
    scores = det[:n, 7]
    window = np.asarray(window, dtype=np.float64)
    scales = np.array([image_shape[1] / (window[3] - window[0]), image_shape[2] / (window[4] - window[1]),
                       image_shape[3] / (window[5] - window[2])] * 2)
    shifts = np.array([window[0], window[1], window[2]] * 2)
    boxes = np.multiply(boxes - shifts, scales).astype(np.int32)
    keep = np.where((boxes[:, 3] - boxes[:, 0]) * (boxes[:, 4] - boxes[:, 1]) * (boxes[:, 5] - boxes[:, 2]) > 0)[0]
    if keep.shape[0] == 0:
        raise ValueError("unmold_detections: no detection with a non-empty box (the reference indexes masks[0] here)")
    boxes, scores = boxes[keep], scores[keep]
    cmap = ops.unmold_argmax(mrcnn_mask[int(keep[0])], boxes[0], image_shape[1:4])
    boxes[:, [0, 1, 2, 3, 4, 5]] = boxes[:, [1, 2, 0, 4, 5, 3]]
    return boxes, np.arange(1, 8), scores, cmap.permute(1, 2, 0).cpu().numpy().astype(np.int64)


def unmold_detections_overlap(detections, mrcnn_mask, image_shape, window):
    """LiTS fork (LiTS_2017/model.py:1777-1835 + the overlap-tile utils.unmold_mask, LiTS_2017/utils.py:383-408):
    ALL detections' masks are resized to their boxes, averaged where they overlap and arg-maxed -- one fused pass on
    the device (``cfun_unmold_overlap``).  Arguments as ``unmold_detections``.  Returns the reference's tuple: boxes
    (y1,x1,z1,y2,x2,z2) int32, class ids ``np.arange(1, 3)`` (sic, :1835), scores, class map [H,W,D] int64."""
    det = detections.detach().cpu().numpy() if torch.is_tensor(detections) else np.asarray(detections)
    zero_ix = np.where(det[:, 6] == 0)[0]
    n = zero_ix[0] if zero_ix.shape[0] > 0 else det.shape[0]
    boxes = det[:n, :6].astype(np.int32)
    scores = det[:n, 7]
    window = np.asarray(window, dtype=np.float64)
    scales = np.array([image_shape[1] / (window[3] - window[0]), image_shape[2] / (window[4] - window[1]),
                       image_shape[3] / (window[5] - window[2])] * 2)
    shifts = np.array([window[0], window[1], window[2]] * 2)
    boxes = np.multiply(boxes - shifts, scales).astype(np.int32)
    keep = np.where((boxes[:, 3] - boxes[:, 0]) * (boxes[:, 4] - boxes[:, 1]) * (boxes[:, 5] - boxes[:, 2]) > 0)[0]
    boxes, scores = boxes[keep], scores[keep]
    masks = mrcnn_mask[torch.as_tensor(keep, dtype=torch.long, device=mrcnn_mask.device)]
    cmap = ops.unmold_overlap(masks, boxes, image_shape[1:4])
    boxes[:, [0, 1, 2, 3, 4, 5]] = boxes[:, [1, 2, 0, 4, 5, 3]]
    return boxes, np.arange(1, 3), scores, cmap.permute(1, 2, 0).cpu().numpy().astype(np.int64)


# ------------------------------------------------------------------------------------------ RoIAlign
def roi_levels(boxes):
    """model.py:322-332: clamp(round(4 + log2(h*w*d)/3), 2, 3) on normalised boxes (fp32, half-to-even)."""
    d = boxes[:, 3] - boxes[:, 0]
    h = boxes[:, 4] - boxes[:, 1]
    w = boxes[:, 5] - boxes[:, 2]
    ln2 = _const([float(np.log(np.float32(2.0)))], torch.float32, boxes.device)      # fp32 log(2), as torch.log gives it
    return (4 + (1.0 / 3.0) * (torch.log(h * w * d) / ln2)).round().int().clamp(2, 3)


def pyramid_roi_align_ndhwc(boxes, feature_maps, pool_size, slabs=None):
    """boxes [R,6] normalised; feature_maps = two [D,H,W,C] maps (levels 2, 3) -> [R,pd,ph,pw,C].  ``slabs`` = per level
    (z0, D): the maps are this rank's depth slabs and the result its additive share of the crops (ops.roi_align)."""
    if slabs is None:
        slabs = (None, None)
    if feature_maps[0] is feature_maps[1]:      # mask head: both "levels" are the raw image (model.py:1413)
        return ops.roi_align(feature_maps[0], boxes.detach(), pool_size, slabs[0])[0]
    # Every box is aligned on BOTH levels, with the boxes of the other level collapsed to the empty box (all zeros): an
    # empty crop is zeros (model.py:281-287) and contributes nothing in the backward pass, so the sum of the two results
    # is exactly the reference's gather / concatenate / restore-order (model.py:334-370) -- without the nonzero() calls,
    # whose result sizes the host has to wait for (two synchronisations per head and step).
    lv = roi_levels(boxes.detach())
    b = boxes.detach()
    out = None
    for i, level in enumerate((2, 3)):
        bi = b * (lv == level).to(b.dtype).unsqueeze(1)
        r = ops.roi_align(feature_maps[i], bi, pool_size, slabs[i])[0]
        out = r if out is None else out + r
    return out


def RoI_Align(feature_map, pool_size, boxes):
    """model.py:265-289 drop-in: feature_map [C,D,H,W], boxes [R,6] normalised -> [R,C,pd,ph,pw]."""
    fm = feature_map.permute(1, 2, 3, 0).contiguous()
    return ops.roi_align(fm, boxes, pool_size)[0].permute(0, 4, 1, 2, 3)


def pyramid_roi_align(inputs, pool_size, test_flag=False):
    """model.py:292-370 drop-in: inputs = [boxes, fm_level2, fm_level3] (batch dim 1) -> [R,C,pd,ph,pw]."""
    boxes = inputs[0].squeeze(0) if inputs[0].dim() == 3 else inputs[0]
    fms = [f.squeeze(0) if f.dim() == 5 else f for f in inputs[1:]]
    same = fms[0] is fms[1] or inputs[1] is inputs[2]
    fms = [f.permute(1, 2, 3, 0).contiguous() for f in fms]
    if same:
        fms[1] = fms[0]
    return pyramid_roi_align_ndhwc(boxes, fms, pool_size).permute(0, 4, 1, 2, 3)


# ------------------------------------------------------------------------------------------ heads
class Classifier(nn.Module):
    """model.py:750-784 on HIP kernels end to end: RoIAlign (cfun_roi_align3d), then conv1 -- whose kernel equals the
    pool size, i.e. a [R x C*pd*ph*pw] . [.. x fc] GEMM that streams the model's largest tensor (113 MB) once --
    with bias + folded BN + ReLU in its epilogue, the 1x1x1 conv2 + BN + ReLU and the two linear heads, all on the
    weight-streaming GEMM kernels of csrc/fc.hip (``ops.fc``); only the 2-way softmax of R <= 64 rows is torch."""

    def __init__(self, channel, pool_size, image_shape, num_classes, fc_size, test_flag=False):
        super().__init__()
        self.pool_size, self.image_shape, self.fc_size, self.test_flag = pool_size, image_shape, fc_size, test_flag
        self.conv1 = Conv3dParams(channel, fc_size, tuple(pool_size))
        self.bn1 = frozen_bn(fc_size, eps=0.001, momentum=0.01)
        self.conv2 = Conv3dParams(fc_size, fc_size, 1)
        self.bn2 = frozen_bn(fc_size, eps=0.001, momentum=0.01)
        self.linear_class = nn.Linear(fc_size, num_classes)
        self.linear_bbox = nn.Linear(fc_size, num_classes * 6)

    @staticmethod
    def _conv_bn_relu(x, conv, bn):
        """relu(bn(conv(x))) for a conv that reduces its whole input: one GEMM, bias and the frozen BN in the epilogue."""
        s, t = folded_bn(bn, bn.eps)
        return ops.fc(x, conv.weight.reshape(conv.out_channels, -1), s, ops.fold_bias(conv.bias, s, t), ACT_RELU)

    def forward_ndhwc(self, feature_maps, rois):
        return self.head_ndhwc(pyramid_roi_align_ndhwc(rois, feature_maps, self.pool_size))

    def head_ndhwc(self, x):
        """The head on RoI-aligned crops x [R,pd,ph,pw,C] (split from the pooling for the depth-sharded step, where the
        crops are summed over the ranks' slabs first, cfun_amd.dist)."""
        x = x.permute(0, 4, 1, 2, 3).reshape(x.shape[0], -1)                        # OIDHW flatten order
        outs = []
        for i in range(0, max(x.shape[0], 1), 64):                                   # (64 RoIs per launch)
            h = self._conv_bn_relu(x[i:i + 64], self.conv1, self.bn1)
            h = self._conv_bn_relu(h, self.conv2, self.bn2)
            outs.append((ops.fc(h, self.linear_class.weight, None, self.linear_class.bias),
                         ops.fc(h, self.linear_bbox.weight, None, self.linear_bbox.bias)))
        logits = torch.cat([o[0] for o in outs], dim=0) if len(outs) > 1 else outs[0][0]
        bbox = torch.cat([o[1] for o in outs], dim=0) if len(outs) > 1 else outs[0][1]
        return [logits, F.softmax(logits, dim=1), bbox.view(bbox.shape[0], -1, 6)]

    def forward(self, x, rois):
        rois = rois.squeeze(0) if rois.dim() == 3 else rois
        fms = [(f.squeeze(0) if f.dim() == 5 else f).permute(1, 2, 3, 0).contiguous() for f in x]
        return self.forward_ndhwc(fms, rois)


class Mask(nn.Module):
    """model.py:787-801: RoIAlign of the RAW image -> U-Net -> softmax over classes."""

    def __init__(self, channel, pool_size, num_classes, conv_channel, stage, test_flag=False, dropout_p=0.6):
        super().__init__()
        self.pool_size, self.test_flag = pool_size, test_flag
        self.modified_u_net = mask_branch.Modified3DUNet(channel, num_classes, stage, conv_channel, dropout_p)

    def forward_ndhwc(self, image, rois, softmax=True):
        """image [D,H,W,C]; rois [R,6] -> (logits, probs) both [R,d,h,w,classes].  ``softmax=False``: (logits, None) -- the
        training step's fused loss pass (ops.mask_losses_fused) produces the probabilities together with both mask losses."""
        x = ops.roi_align(image, rois.detach(), self.pool_size)[0]
        logits = self.modified_u_net.forward_ndhwc(x)
        return logits, (ops.softmax_channels(logits) if softmax else None)

    def forward(self, x, rois):
        rois = rois.squeeze(0) if rois.dim() == 3 else rois
        img = x[0].squeeze(0) if x[0].dim() == 5 else x[0]
        logits, probs = self.forward_ndhwc(img.permute(1, 2, 3, 0).contiguous(), rois)
        return ops.to_ncdhw(logits), ops.to_ncdhw(probs)


# ------------------------------------------------------------------------------------------ losses
def _as_ndhwc(t):
    """[n,C,D,H,W] (any memory format) -> contiguous [n,D,H,W,C]; free for the views our heads return."""
    return t.permute(0, 2, 3, 4, 1).contiguous()


def mask_labels(target_masks):
    """One-hot GT masks [n,C,D,H,W] (float/double, model.py:493) -> uint8 class labels [n,D,H,W]
    (the argmax of model.py:925).  uint8 label tensors pass through."""
    if target_masks.dtype == torch.uint8:
        return target_masks
    return torch.argmax(target_masks, dim=1).to(torch.uint8)


def compute_mrcnn_mask_loss(target_masks, target_class_ids, pred_masks):
    """model.py:909-935 for positive-first RoI sets: CrossEntropyLoss(logits, argmax(one-hot target))."""
    n_pos = int((target_class_ids > 0).sum()) if target_class_ids.numel() else 0
    if n_pos == 0:
        return torch.zeros((), device=pred_masks.device)
    labels = mask_labels(target_masks)[:n_pos].to(pred_masks.device)
    return ops.mask_cross_entropy(_as_ndhwc(pred_masks[:n_pos]), labels.contiguous())


def compute_mrcnn_mask_edge_loss(target_masks, target_class_ids, pred_masks):
    """model.py:938-981 (Sobel channels 0,1,0 -- reproduced as is) on softmax probabilities."""
    n_pos = int((target_class_ids > 0).sum()) if target_class_ids.numel() else 0
    if n_pos == 0:
        return torch.zeros((), device=pred_masks.device)
    labels = mask_labels(target_masks)[:n_pos].to(pred_masks.device)
    return ops.edge_loss(_as_ndhwc(pred_masks[:n_pos]), labels.contiguous())


def compute_rpn_class_loss(rpn_match, rpn_class_logits):
    """model.py:808-832 (<= a few hundred anchors: left to torch)."""
    # mean over the non-neutral anchors as a masked sum: torch.nonzero would make the host wait for its result size
    # (rows the reference never gathers -- neutral anchors -- are SELECTED away before any arithmetic, torch.where on the
    # inputs: a NaN / Inf there must neither reach the loss nor, as 0 * NaN, the gradient)
    m = rpn_match.squeeze(2).reshape(-1)
    usedb = m != 0
    logits = rpn_class_logits.reshape(-1, rpn_class_logits.shape[-1])
    logits = torch.where(usedb.unsqueeze(1), logits, torch.zeros_like(logits))
    ce = F.cross_entropy(logits, (m == 1).long(), reduction="none")
    used = usedb.to(ce.dtype)
    return (ce * used).sum() / used.sum()


def compute_rpn_bbox_loss(target_bbox, rpn_match, rpn_bbox):
    """model.py:835-860."""
    # the k-th positive anchor pairs with target row k (model.py:851-857): row index = running count of positives, gathered
    # per anchor and masked -- no nonzero(), no host wait
    pos = (rpn_match.squeeze(2).reshape(-1) == 1)
    row = (torch.cumsum(pos.long(), 0) - 1).clamp(min=0, max=target_bbox.shape[1] - 1)
    pred, tgt = rpn_bbox.reshape(-1, 6), target_bbox[0][row]
    sel = pos.unsqueeze(1)                  # non-positive rows: both operands selected to 0 (no 0 * NaN, see above)
    l1 = F.smooth_l1_loss(torch.where(sel, pred, torch.zeros_like(pred)), torch.where(sel, tgt, torch.zeros_like(tgt)),
                          reduction="none")
    return l1.sum() / (pos.to(l1.dtype).sum() * 6)


def compute_mrcnn_class_loss(target_class_ids, pred_class_logits):
    """model.py:863-878 with the binarisation of model.py:989."""
    if target_class_ids.numel() == 0:
        return torch.zeros((), device=pred_class_logits.device)
    return F.cross_entropy(pred_class_logits, (target_class_ids > 0).long())


def compute_mrcnn_bbox_loss(target_bbox, target_class_ids, pred_bbox):
    """model.py:881-906 with the binarised ids (class column 1)."""
    if target_class_ids.numel() == 0:
        return torch.zeros((), device=pred_bbox.device)
    posb = target_class_ids > 0                               # masked mean: no nonzero(), no host wait
    sel = posb.unsqueeze(1)                 # negative RoIs' rows (padded / uninitialised targets) are selected to 0 first
    pred = pred_bbox[:, 1, :]
    l1 = F.smooth_l1_loss(torch.where(sel, pred, torch.zeros_like(pred)),
                          torch.where(sel, target_bbox, torch.zeros_like(target_bbox)), reduction="none")
    cnt = posb.to(pred_bbox.dtype).sum() * 6
    return torch.where(cnt > 0, l1.sum() / cnt.clamp(min=1), torch.zeros_like(cnt))
