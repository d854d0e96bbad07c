"""TEST INFRASTRUCTURE -- functional CPU restatement of CFUN's volumetric hot path.

Every function restates (does not import) the reference algorithm and cites the
reference file:line it follows (paths relative to /root/reference).  Arithmetic
is plain fp32 torch-CPU / numpy: those are the reference's own arithmetic
libraries, so equality with the reference is exact up
to op ordering.  Pinned by tests/test_oracle_golden.py against
tests/golden/*.npz (generated from the reference import by
tests/golden/gen_golden.py).

All functions take a flat ``sd`` = {state-dict key: tensor} in the reference's
checkpoint naming plus a key prefix, NCDHW fp32 activations.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# backbone.py
# --------------------------------------------------------------------------------------


def _conv(x, sd, key, stride=1, padding=0):
    return F.conv3d(x, sd[key + ".weight"], sd.get(key + ".bias"), stride=stride, padding=padding)


def _bn_eval(x, sd, key, eps=1e-5):
    """BatchNorm3d frozen in eval mode (model.py:1397-1406): constant affine."""
    shp = (1, -1, 1, 1, 1)
    rstd = torch.rsqrt(sd[key + ".running_var"] + eps)
    return (x - sd[key + ".running_mean"].view(shp)) * (rstd * sd[key + ".weight"]).view(shp) \
        + sd[key + ".bias"].view(shp)


def bottleneck(x, sd, p, block_idx, expand, stride):
    """backbone.py:26-114.  ST pattern cycles A,B,C with (block_idx-1)%3 (backbone.py:41)."""
    st = "ABC"[(block_idx - 1) % 3]
    out = F.relu(_bn_eval(_conv(x, sd, p + "conv1", stride=stride), sd, p + "bn1"))

    def s(v):  # conv_S 1x3x3, backbone.py:14-17,42
        return F.relu(_bn_eval(_conv(v, sd, p + "conv2", padding=(0, 1, 1)), sd, p + "bn2"))

    def t(v):  # conv_T 3x1x1, backbone.py:20-23,44
        return F.relu(_bn_eval(_conv(v, sd, p + "conv3", padding=(1, 0, 0)), sd, p + "bn3"))

    if st == "A":      # backbone.py:58-67
        out = t(s(out))
    elif st == "B":    # backbone.py:69-78
        out = t(out) + s(out)
    else:              # backbone.py:80-89
        y = s(out)
        out = y + t(y)
    out = _bn_eval(_conv(out, sd, p + "conv4"), sd, p + "bn4")
    res = x
    if expand:         # backbone.py:49-52,107-108
        res = _bn_eval(_conv(x, sd, p + "downsample.0", stride=2), sd, p + "downsample.1")
    return F.relu(out + res)


def p3d_stages(x, sd, prefix="", layers=(2, 3), stem_pad=(1, 3, 3)):
    """backbone.py:117-158: C1 (conv k(3,7,7) s2 + BN + ReLU + MaxPool 2/2), C2, C3."""
    c1 = F.relu(_bn_eval(_conv(x, sd, prefix + "C1.0", stride=2, padding=stem_pad), sd, prefix + "C1.1"))
    c1 = F.max_pool3d(c1, kernel_size=2, stride=2)
    feats = [c1]
    h = c1
    for si, nblk in enumerate(layers):
        name = "C%d." % (si + 2)
        for b in range(nblk):
            h = bottleneck(h, sd, prefix + name + "%d." % b, b + 1, expand=(b == 0), stride=2 if b == 0 else 1)
        feats.append(h)
    return feats


def fpn(x, sd, prefix="fpn.", layers=(2, 3), stem_pad=(1, 3, 3)):
    """model.py:124-148."""
    _, c2, c3 = p3d_stages(x, sd, prefix, layers, stem_pad)
    p3 = _conv(c3, sd, prefix + "P3_conv1")
    p2 = _conv(c2, sd, prefix + "P2_conv1") + F.interpolate(p3, scale_factor=2)  # nearest
    p3 = _conv(p3, sd, prefix + "P3_conv2", padding=1)
    p2 = _conv(p2, sd, prefix + "P2_conv2", padding=1)
    return p2, p3


def rpn(p, sd, prefix="rpn."):
    """model.py:700-743 -> logits [1,A,2], probs [1,A,2], bbox [1,A,6]; flatten order (z,y,x)."""
    h = F.relu(_conv(p, sd, prefix + "conv_shared", padding=1))
    logits = _conv(h, sd, prefix + "conv_class").permute(0, 2, 3, 4, 1).contiguous().view(p.shape[0], -1, 2)
    probs = F.softmax(logits, dim=2)
    bbox = _conv(h, sd, prefix + "conv_bbox").permute(0, 2, 3, 4, 1).contiguous().view(p.shape[0], -1, 6)
    return logits, probs, bbox


# --------------------------------------------------------------------------------------
# utils.py: anchors, NMS
# --------------------------------------------------------------------------------------


def generate_pyramid_anchors(scales, feature_shapes, feature_strides, anchor_stride=1):
    """utils.py:467-528 with one ratio ([1]).  np.meshgrid default 'xy' indexing makes the
    enumeration y-slowest, then z, then x."""
    out = []
    for scale, (d, h, w), fs in zip(scales, feature_shapes, feature_strides):
        z = np.arange(0, d, anchor_stride) * fs
        y = np.arange(0, h, anchor_stride) * fs
        x = np.arange(0, w, anchor_stride) * fs
        # flat index = iy*(nz*nx) + iz*nx + ix
        yy, zz, xx = np.meshgrid(y, z, x, indexing="ij")
        c = np.stack([zz.reshape(-1), yy.reshape(-1), xx.reshape(-1)], axis=1).astype(np.float64)
        half = 0.5 * float(scale)
        out.append(np.concatenate([c - half, c + half], axis=1))
    return np.concatenate(out, axis=0)


def nms(boxes, scores, threshold, max_num):
    """utils.py:122-157 + compute_iou utils.py:50-70, float32 numpy, same op order.
    Written as an O(N*K) mask loop instead of np.delete; identical pick list."""
    boxes = np.asarray(boxes, dtype=np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    n = boxes.shape[0]
    z1, y1, x1, z2, y2, x2 = [boxes[:, i] for i in range(6)]
    volume = (z2 - z1) * (y2 - y1) * (x2 - x1)
    order = scores.argsort()[::-1]
    alive = np.ones(n, dtype=bool)
    pick = []
    for pos in range(n):
        i = order[pos]
        if not alive[i]:
            continue
        pick.append(i)
        if len(pick) >= max_num:
            break
        rest = order[pos + 1:]
        iz1 = np.maximum(z1[i], z1[rest]); iz2 = np.minimum(z2[i], z2[rest])
        iy1 = np.maximum(y1[i], y1[rest]); iy2 = np.minimum(y2[i], y2[rest])
        ix1 = np.maximum(x1[i], x1[rest]); ix2 = np.minimum(x2[i], x2[rest])
        inter = np.maximum(ix2 - ix1, 0) * np.maximum(iy2 - iy1, 0) * np.maximum(iz2 - iz1, 0)
        union = volume[i] + volume[rest] - inter
        iou = inter / (union + np.float32(1e-6))
        alive[rest[iou > np.float32(threshold)]] = False
    return np.array(pick, dtype=np.int32)


# --------------------------------------------------------------------------------------
# model.py: proposals, RoIAlign
# --------------------------------------------------------------------------------------


def apply_box_deltas(boxes, deltas):
    """model.py:155-182."""
    d = boxes[:, 3] - boxes[:, 0]
    h = boxes[:, 4] - boxes[:, 1]
    w = boxes[:, 5] - boxes[:, 2]
    cz = boxes[:, 0] + 0.5 * d
    cy = boxes[:, 1] + 0.5 * h
    cx = boxes[:, 2] + 0.5 * w
    cz = cz + deltas[:, 0] * d
    cy = cy + deltas[:, 1] * h
    cx = cx + deltas[:, 2] * w
    d = d * torch.exp(deltas[:, 3])
    h = h * torch.exp(deltas[:, 4])
    w = w * torch.exp(deltas[:, 5])
    z1 = cz - 0.5 * d
    y1 = cy - 0.5 * h
    x1 = cx - 0.5 * w
    return torch.stack([z1, y1, x1, z1 + d, y1 + h, x1 + w], dim=1)


def clip_boxes(boxes, window):
    """model.py:185-196."""
    lo = torch.tensor([window[0], window[1], window[2]] * 2, dtype=boxes.dtype)
    hi = torch.tensor([window[3], window[4], window[5]] * 2, dtype=boxes.dtype)
    return torch.max(torch.min(boxes, hi), lo)


def proposal_layer(rpn_probs, rpn_bbox, anchors, proposal_count, nms_threshold, image_dhw,
                   pre_nms_limit=1000, std_dev=(0.1, 0.1, 0.1, 0.2, 0.2, 0.2)):
    """model.py:199-258.  rpn_probs [A,2], rpn_bbox [A,6], anchors [A,6] -> normalised [K,6]."""
    scores = rpn_probs[:, 1]
    deltas = rpn_bbox * torch.tensor(std_dev, dtype=torch.float32).view(1, 6)
    limit = min(pre_nms_limit, anchors.shape[0])
    scores, order = scores.sort(descending=True)
    order = order[:limit]
    scores = scores[:limit]
    boxes = apply_box_deltas(anchors[order], deltas[order])
    depth, height, width = [float(v) for v in image_dhw]
    boxes = clip_boxes(boxes, (0.0, 0.0, 0.0, depth, height, width))
    keep = nms(boxes.detach().numpy(), scores.detach().numpy(), nms_threshold, proposal_count)
    boxes = boxes[torch.from_numpy(keep).long()]
    norm = torch.tensor([depth, height, width, depth, height, width], dtype=torch.float32)
    return boxes / norm, keep, order


def clip_to_window(window, boxes):
    """model.py:570-581: clamp (z, y, x) pairs to the image window."""
    lo = torch.tensor([window[0], window[1], window[2]] * 2, dtype=boxes.dtype)
    hi = torch.tensor([window[3], window[4], window[5]] * 2, dtype=boxes.dtype)
    return torch.max(torch.min(boxes, hi), lo)


def refine_detections(rois, probs, deltas, window, image_dhw, min_confidence=0.7, nms_threshold=0.3,
                      max_instances=32, std_dev=(0.1, 0.1, 0.1, 0.2, 0.2, 0.2)):
    """model.py:584-672 (the inference NMS site).  rois [N,6] normalised, probs [N,K],
    deltas [N,K,6] -> detections [M,8] = (z1,y1,x1,z2,y2,x2 in voxels, class id, score), sorted by score.

    Per predicted class: boxes sorted by score, NMS(nms_threshold, max_instances) (utils.py:122-157); the kept
    indices of all classes are united, intersected with the confidence filter and the top max_instances by score
    survive.  The reference leaves ``nms_keep`` unbound when nothing passes the filter (model.py:662, App. A-16);
    this restatement returns an empty [0,8] tensor for that input instead of raising."""
    n = probs.shape[0]
    class_ids = torch.argmax(probs, dim=1)
    idx = torch.arange(n)
    class_scores = probs[idx, class_ids]
    deltas_specific = deltas[idx, class_ids]
    refined = apply_box_deltas(rois, deltas_specific * torch.tensor(std_dev, dtype=torch.float32).view(1, 6))
    depth, height, width = [float(v) for v in image_dhw]
    refined = refined * torch.tensor([depth, height, width, depth, height, width], dtype=torch.float32)
    refined = torch.round(clip_to_window(window, refined))
    keep_bool = class_ids > 0
    if min_confidence:
        keep_bool = keep_bool & (class_scores >= min_confidence)
    keep = torch.nonzero(keep_bool)[:, 0]
    if keep.numel() == 0:
        return torch.zeros((0, 8), dtype=torch.float32)
    pre_ids, pre_scores, pre_rois = class_ids[keep], class_scores[keep], refined[keep]
    nms_keep = []
    for cid in torch.unique(pre_ids).tolist():
        ixs = torch.nonzero(pre_ids == cid)[:, 0]
        sc, order = pre_scores[ixs].sort(descending=True)
        picked = nms(pre_rois[ixs][order].detach().numpy(), sc.detach().numpy(), nms_threshold, max_instances)
        nms_keep.append(keep[ixs[order[torch.from_numpy(picked).long()]]])
    keep = torch.unique(torch.cat(nms_keep))           # == intersect1d(keep, unique1d(cat)) : a subset of keep, sorted
    count = min(max_instances, keep.numel())
    top = class_scores[keep].sort(descending=True)[1][:count]
    keep = keep[top]
    return torch.cat([refined[keep], class_ids[keep].unsqueeze(1).float(), class_scores[keep].unsqueeze(1)], dim=1)


def unmold_mask(mask, bbox, image_shape):
    """utils.py:443-460: mask [d,h,w,C] (one detection's class probabilities) -> float32 [D,H,W,C], resized to the
    box with F.interpolate(mode='trilinear', align_corners=False) (the reference's own call) and pasted into zeros.
    image_shape = [channels, depth, height, width]."""
    z1, y1, x1, z2, y2, x2 = [int(v) for v in bbox]
    m = torch.as_tensor(mask).float().permute(3, 0, 1, 2).unsqueeze(0)
    m = F.interpolate(m, size=(z2 - z1, y2 - y1, x2 - x1), mode="trilinear", align_corners=False)
    m = m.squeeze(0).permute(1, 2, 3, 0).numpy()
    full = np.zeros((image_shape[1], image_shape[2], image_shape[3], m.shape[-1]), dtype=np.float32)
    full[z1:z2, y1:y2, x1:x2, :] = m
    return full


def unmold_detections(detections, mrcnn_mask, image_shape, window):
    """model.py:1812-1864: detections [N,8] (numpy), mrcnn_mask [N,d,h,w,C], image_shape [c,D,H,W], window
    (z1,y1,x1,z2,y2,x2) -> (boxes (y1,x1,z1,y2,x2,z2) int32, class ids = arange(1,8) (sic), scores, class map
    [H,W,D] = argmax of the FIRST detection's un-molded mask)."""
    detections = np.asarray(detections)
    zero_ix = np.where(detections[:, 6] == 0)[0]
    n = zero_ix[0] if zero_ix.shape[0] > 0 else detections.shape[0]
    boxes = detections[:n, :6].astype(np.int32)
    scores = detections[:n, 7]
    masks = np.asarray(mrcnn_mask)[np.arange(n)]
    window = np.asarray(window, dtype=np.float64)
    scales = np.array([image_shape[1] / (window[3] - window[0]), image_shape[2] / (window[4] - window[1]),
                       image_shape[3] / (window[5] - window[2])] * 2)
    shifts = np.array([window[0], window[1], window[2]] * 2)
    boxes = np.multiply(boxes - shifts, scales).astype(np.int32)
    keep = np.where((boxes[:, 3] - boxes[:, 0]) * (boxes[:, 4] - boxes[:, 1]) * (boxes[:, 5] - boxes[:, 2]) > 0)[0]
    boxes, scores, masks = boxes[keep], scores[keep], masks[keep]
    full = np.argmax(unmold_mask(masks[0], boxes[0], image_shape), axis=3)
    boxes[:, [0, 1, 2, 3, 4, 5]] = boxes[:, [1, 2, 0, 4, 5, 3]]
    return boxes, np.arange(1, 8), scores, full.transpose((1, 2, 0))


def unmold_mask_overlap(masks, bboxes, image_shape):
    """LiTS_2017/utils.py:383-408 (overlap-tile): masks [n,d,h,w,C], bboxes [n,6] ints, image_shape [c,D,H,W] ->
    float32 [D,H,W,C]: per-detection trilinear (align_corners=False) resize, fp32 add + count in detection order,
    add / (count + 1e-6), clip to [0,1]."""
    masks = np.asarray(masks)
    assert masks.shape[0] == len(bboxes)
    shape = (image_shape[1], image_shape[2], image_shape[3], masks.shape[-1])
    add = np.zeros(shape, dtype=np.float32)
    cnt = np.zeros(shape, dtype=np.float32)
    for i in range(masks.shape[0]):
        z1, y1, x1, z2, y2, x2 = [int(v) for v in bboxes[i]]
        m = torch.from_numpy(masks[i]).float().permute(3, 0, 1, 2).unsqueeze(0)
        m = F.interpolate(m, size=(z2 - z1, y2 - y1, x2 - x1), mode="trilinear", align_corners=False)
        add[z1:z2, y1:y2, x1:x2, :] += m.squeeze(0).numpy().transpose(1, 2, 3, 0)
        cnt[z1:z2, y1:y2, x1:x2, :] += 1.0
    return (add / (cnt + np.float32(1e-6))).clip(min=0.0, max=1.0)


def unmold_detections_overlap(detections, mrcnn_mask, image_shape, window):
    """LiTS_2017/model.py:1777-1835: as ``unmold_detections`` but ALL kept detections are un-molded together
    (overlap-tile) before the arg-max, and the class ids are ``arange(1, 3)`` (sic)."""
    detections = np.asarray(detections)
    zero_ix = np.where(detections[:, 6] == 0)[0]
    n = zero_ix[0] if zero_ix.shape[0] > 0 else detections.shape[0]
    boxes = detections[:n, :6].astype(np.int32)
    scores = detections[:n, 7]
    masks = np.asarray(mrcnn_mask)[np.arange(n)]
    window = np.asarray(window, dtype=np.float64)
    scales = np.array([image_shape[1] / (window[3] - window[0]), image_shape[2] / (window[4] - window[1]),
                       image_shape[3] / (window[5] - window[2])] * 2)
    shifts = np.array([window[0], window[1], window[2]] * 2)
    boxes = np.multiply(boxes - shifts, scales).astype(np.int32)
    keep = np.where((boxes[:, 3] - boxes[:, 0]) * (boxes[:, 4] - boxes[:, 1]) * (boxes[:, 5] - boxes[:, 2]) > 0)[0]
    boxes, scores, masks = boxes[keep], scores[keep], masks[keep]
    full = np.argmax(unmold_mask_overlap(masks, boxes, image_shape), axis=3)
    boxes[:, [0, 1, 2, 3, 4, 5]] = boxes[:, [1, 2, 0, 4, 5, 3]]
    return boxes, np.arange(1, 3), scores, full.transpose((1, 2, 0))


def roi_bounds(boxes, dhw):
    """model.py:271-278 / utils.py:160-174: fp32 product, floor lo / ceil hi, int64."""
    scale = torch.tensor([dhw[0], dhw[1], dhw[2]] * 2, dtype=torch.float32)
    b = torch.mul(boxes.float(), scale)
    lo = b[:, :3].floor()
    hi = b[:, 3:].ceil()
    return torch.cat([lo, hi], dim=1).long()


def roi_align(feature_map, pool_size, boxes):
    """model.py:265-289: crop [lo:hi] (python slicing semantics: negative indices wrap,
    upper bounds clamp) + trilinear align_corners=True resize; any failure -> zeros."""
    c, d, h, w = feature_map.shape
    ib = roi_bounds(boxes, (d, h, w))
    out = torch.zeros((boxes.shape[0], c, pool_size[0], pool_size[1], pool_size[2]), dtype=feature_map.dtype)
    for i in range(boxes.shape[0]):
        z1, y1, x1, z2, y2, x2 = [int(v) for v in ib[i]]
        crop = feature_map[:, z1:z2, y1:y2, x1:x2]
        if crop.numel() == 0:
            continue  # F.interpolate raises on an empty crop -> caught -> zeros (model.py:281-287)
        out[i] = F.interpolate(crop.unsqueeze(0), size=tuple(pool_size), mode="trilinear", align_corners=True)[0]
    return out


def roi_levels(boxes):
    """model.py:322-332: level = clamp(round(4 + log2(h*w*d)/3), 2, 3), fp32, half-to-even."""
    d = boxes[:, 3] - boxes[:, 0]
    h = boxes[:, 4] - boxes[:, 1]
    w = boxes[:, 5] - boxes[:, 2]
    ln2 = torch.log(torch.tensor([2.0], dtype=torch.float32))
    lvl = 4 + (1.0 / 3.0) * (torch.log(h * w * d) / ln2)
    return lvl.round().int().clamp(2, 3)


def pyramid_roi_align(boxes, feature_maps, pool_size):
    """model.py:292-370.  boxes [N,6] normalised; feature_maps = [level2, level3], each [C,D,H,W]."""
    lv = roi_levels(boxes)
    pooled, idx = [], []
    for i, level in enumerate((2, 3)):
        ix = torch.nonzero(lv == level)[:, 0]
        if ix.numel() == 0:
            continue
        idx.append(ix)
        pooled.append(roi_align(feature_maps[i], pool_size, boxes[ix].detach()))
    pooled = torch.cat(pooled, dim=0)
    _, back = torch.sort(torch.cat(idx, dim=0))
    return pooled[back]


# --------------------------------------------------------------------------------------
# mask_branch.py
# --------------------------------------------------------------------------------------


def _inorm(x):
    return F.instance_norm(x, eps=1e-5)


# Test instrumentation (tests/module_cases.flip_fit): when a list, every LeakyReLU site appends (pre-activation, output)
# -- both still attached to the autograd graph, the output with retain_grad() -- so that a test can ask "which voxels
# sit within rounding of the kink at 0, and how would the gradient change if their slope flipped".
KINK_TAPS = None


def _lrelu(x):
    y = F.leaky_relu(x, 0.01)
    if KINK_TAPS is not None and y.requires_grad:
        y.retain_grad()
        KINK_TAPS.append((x, y))
    return y


def unet(x, sd, prefix="", stage="beginning", dropout_masks=None):
    """mask_branch.py:124-220.  dropout_masks: None (eval) or list of 5 [N,C] multipliers
    (Dropout3d keep/(1-p) per (n,c)) applied at the 5 call sites."""
    def w(name):
        return sd[prefix + name + ".weight"]

    def c3(v, name, stride=1):
        return F.conv3d(v, w(name), None, stride=stride, padding=1)

    def c1(v, name):
        return F.conv3d(v, w(name), None)

    def drop(v, i):
        if dropout_masks is None:
            return v
        m = dropout_masks[i]
        return v * m.view(m.shape[0], m.shape[1], 1, 1, 1)

    def up(v):
        return F.interpolate(v, scale_factor=2, mode="nearest")

    def nluc(v, name):  # norm_lrelu_upscale_conv_norm_lrelu, mask_branch.py:108-116
        return _lrelu(_inorm(c3(up(_lrelu(_inorm(v))), name + ".3")))

    # level 1 (mask_branch.py:126-136; note residual is pre-activation, context_1 is pre-norm)
    out = c3(x, "conv3d_c1_1")
    res = out
    out = c3(_lrelu(out), "conv3d_c1_2")
    out = drop(out, 0)
    out = c3(_lrelu(out), "lrelu_conv_c1.1")
    out = out + res
    ctx1 = _lrelu(out)
    out = _lrelu(_inorm(out))
    ctx = [ctx1]
    # levels 2..5 (mask_branch.py:138-177); the norm_lrelu_conv weight is applied twice
    for lvl in (2, 3, 4, 5):
        out = c3(out, "conv3d_c%d" % lvl, stride=2)
        res = out
        name = "norm_lrelu_conv_c%d.2" % lvl
        out = c3(_lrelu(_inorm(out)), name)
        out = drop(out, lvl - 1)
        out = c3(_lrelu(_inorm(out)), name)
        out = out + res
        if lvl < 5:
            out = _lrelu(_inorm(out))
            ctx.append(out)
    out = nluc(out, "norm_lrelu_upscale_conv_norm_lrelu_l0")
    out = _lrelu(_inorm(c1(out, "conv3d_l0")))
    # localisation path (mask_branch.py:183-207)
    out = torch.cat([out, ctx[3]], dim=1)
    out = _lrelu(_inorm(c3(out, "conv_norm_lrelu_l1.0")))
    out = c1(out, "conv3d_l1")
    out = nluc(out, "norm_lrelu_upscale_conv_norm_lrelu_l1")
    out = torch.cat([out, ctx[2]], dim=1)
    out = _lrelu(_inorm(c3(out, "conv_norm_lrelu_l2.0")))
    ds2 = out
    out = c1(out, "conv3d_l2")
    out = nluc(out, "norm_lrelu_upscale_conv_norm_lrelu_l2")
    out = torch.cat([out, ctx[1]], dim=1)
    out = _lrelu(_inorm(c3(out, "conv_norm_lrelu_l3.0")))
    ds3 = out
    out = c1(out, "conv3d_l3")
    out = nluc(out, "norm_lrelu_upscale_conv_norm_lrelu_l3")
    out = torch.cat([out, ctx[0]], dim=1)
    out = _lrelu(_inorm(c3(out, "conv_norm_lrelu_l4.0")))
    out_pred = c1(out, "conv3d_l4")
    # deep supervision (mask_branch.py:209-215)
    s = up(c1(ds2, "ds2_1x1_conv3d")) + c1(ds3, "ds3_1x1_conv3d")
    out = out_pred + up(s)
    if stage == "finetune":  # mask_branch.py:216-218
        out = up(out) + F.conv3d(up(out), w("out_upscale_conv.1"), None, padding=2)
    return out


# --------------------------------------------------------------------------------------
# model.py heads + losses
# --------------------------------------------------------------------------------------


def classifier(feature_maps, rois, sd, pool_size, prefix="classifier."):
    """model.py:750-784 (BN eps 1e-3, eval).  feature_maps [p2,p3] each [C,D,H,W]; rois [N,6]."""
    x = pyramid_roi_align(rois, feature_maps, pool_size)
    x = F.relu(_bn_eval(_conv(x, sd, prefix + "conv1"), sd, prefix + "bn1", eps=1e-3))
    x = F.relu(_bn_eval(_conv(x, sd, prefix + "conv2"), sd, prefix + "bn2", eps=1e-3))
    x = x.view(x.shape[0], -1)
    logits = F.linear(x, sd[prefix + "linear_class.weight"], sd[prefix + "linear_class.bias"])
    probs = F.softmax(logits, dim=1)
    bbox = F.linear(x, sd[prefix + "linear_bbox.weight"], sd[prefix + "linear_bbox.bias"])
    return logits, probs, bbox.view(bbox.shape[0], -1, 6)


def mask_head(image, rois, sd, pool_size, stage, prefix="mask.modified_u_net.", dropout_masks=None):
    """model.py:787-801: RoIAlign of the RAW image -> U-Net -> softmax(dim=1)."""
    x = pyramid_roi_align(rois, [image, image], pool_size)
    logits = unet(x, sd, prefix, stage, dropout_masks)
    return logits, F.softmax(logits, dim=1)


def mask_ce_loss(target_onehot, logits):
    """model.py:909-935 for all-positive inputs: target = argmax over one-hot channels,
    CrossEntropyLoss mean over n*voxels.  target_onehot [n,C,D,H,W], logits [n,C,D,H,W]."""
    y = torch.argmax(target_onehot.long(), dim=1)
    return F.cross_entropy(logits, y)


def mask_ce_loss_weighted(target_onehot, logits, weight):
    """LiTS_2017/model.py:907-933: as mask_ce_loss with nn.CrossEntropyLoss(weight=[1, 1, 100])."""
    y = torch.argmax(target_onehot.long(), dim=1)
    return F.cross_entropy(logits, y, weight=torch.as_tensor(weight, dtype=logits.dtype))


def edge_loss_raw(target_onehot, probs):
    """LiTS_2017/model.py:936-979: per (roi, class 1..C-1) valid Sobel conv of target and prediction, MSE over the raw
    [1,3,D-2,H-2,W-2] responses (the magnitude is commented out in the fork), summed, / n_pos."""
    k = sobel_stack().to(probs.dtype)      # (fp32 as the reference; fp64 when the caller runs the whole oracle in fp64)
    n, c = probs.shape[:2]
    loss = torch.zeros(1, dtype=probs.dtype)
    for i in range(n):
        for j in range(1, c):
            t = F.conv3d(target_onehot[i, j][None, None].to(probs.dtype), k)
            p = F.conv3d(probs[i, j][None, None], k)
            loss = loss + F.mse_loss(p, t)
    return loss / n


def sobel_stack():
    """model.py:947-952."""
    kx = np.array([[[1, 2, 1], [0, 0, 0], [-1, -2, -1]],
                   [[2, 4, 2], [0, 0, 0], [-2, -4, -2]],
                   [[1, 2, 1], [0, 0, 0], [-1, -2, -1]]])
    ky = kx.transpose((1, 0, 2))
    kz = kx.transpose((0, 2, 1))
    return torch.from_numpy(np.array([kx, ky, kz]).reshape((3, 1, 3, 3, 3))).float()


def edge_loss(target_onehot, probs):
    """model.py:938-981: per (roi, class 1..C-1) valid Sobel conv, magnitude sqrt(c0^2+c1^2+c0^2)
    (channel 0 twice, channel 2 unused -- reproduced as is), MSE mean; sum / n_pos."""
    k = sobel_stack().to(probs.dtype)
    n, c = probs.shape[:2]
    loss = torch.zeros(1, dtype=probs.dtype)
    for i in range(n):
        for j in range(1, c):
            t = F.conv3d(target_onehot[i, j][None, None].to(probs.dtype), k)
            p = F.conv3d(probs[i, j][None, None], k)
            tm = torch.sqrt(t[:, 0] ** 2 + t[:, 1] ** 2 + t[:, 0] ** 2)
            pm = torch.sqrt(p[:, 0] ** 2 + p[:, 1] ** 2 + p[:, 0] ** 2)
            loss = loss + F.mse_loss(pm, tm)
    return loss / n


def rpn_class_loss(rpn_match, rpn_class_logits):
    """model.py:808-832.  rpn_match [1,A,1] in {-1,0,1}."""
    m = rpn_match.squeeze(2)
    idx = torch.nonzero(m != 0)
    return F.cross_entropy(rpn_class_logits[idx[:, 0], idx[:, 1], :], (m == 1).long()[idx[:, 0], idx[:, 1]])


def rpn_bbox_loss(target_bbox, rpn_match, rpn_bbox):
    """model.py:835-860."""
    m = rpn_match.squeeze(2)
    idx = torch.nonzero(m == 1)
    pred = rpn_bbox[idx[:, 0], idx[:, 1]]
    return F.smooth_l1_loss(pred, target_bbox[0, :pred.shape[0], :])


def mrcnn_class_loss(target_class_ids, logits):
    """model.py:863-878 with the binarised ids of model.py:989."""
    return F.cross_entropy(logits, (target_class_ids > 0).long())


def mrcnn_bbox_loss(target_deltas, target_class_ids, pred_bbox):
    """model.py:881-906 with the binarised ids of model.py:991-992 (class column = 1)."""
    pos = torch.nonzero(target_class_ids > 0)[:, 0]
    return F.smooth_l1_loss(pred_bbox[pos, 1, :], target_deltas[pos, :])


def box_refinement(box, gt_box):
    """utils.py:92-119."""
    d = box[:, 3] - box[:, 0]; h = box[:, 4] - box[:, 1]; w = box[:, 5] - box[:, 2]
    cz = box[:, 0] + 0.5 * d; cy = box[:, 1] + 0.5 * h; cx = box[:, 2] + 0.5 * w
    gd = gt_box[:, 3] - gt_box[:, 0]; gh = gt_box[:, 4] - gt_box[:, 1]; gw = gt_box[:, 5] - gt_box[:, 2]
    gz = gt_box[:, 0] + 0.5 * gd; gy = gt_box[:, 1] + 0.5 * gh; gx = gt_box[:, 2] + 0.5 * gw
    return torch.stack([(gz - cz) / d, (gy - cy) / h, (gx - cx) / w,
                        torch.log(gd / d), torch.log(gh / h), torch.log(gw / w)], dim=1)


def nearest_resize(vol, out_shape):
    """Restatement of skimage.transform.resize(order=0) as used at model.py:490 via
    utils.py:318-339 (flagged restatement): output index o maps to input
    index floor((o + 0.5) * in / out).  vol [..., d, h, w] -> [..., D, H, W]."""
    idx = []
    for ax, o in zip((-3, -2, -1), out_shape):
        n = vol.shape[ax]
        i = torch.floor((torch.arange(o, dtype=torch.float64) + 0.5) * (n / o)).long().clamp(0, n - 1)
        idx.append(i)
    return vol[..., idx[0][:, None, None], idx[1][None, :, None], idx[2][None, None, :]]


def mask_targets(p_rois, gt_masks, mask_shape):
    """model.py:481-493: crop with int() truncation of shape*coord, nearest resize.
    gt_masks [C,D,H,W] one-hot; p_rois [n,6] normalised.  Returns [n,C,*mask_shape]."""
    out = []
    _, D, H, W = gt_masks.shape
    for i in range(p_rois.shape[0]):
        z1 = int(D * p_rois[i, 0]); z2 = int(D * p_rois[i, 3])
        y1 = int(H * p_rois[i, 1]); y2 = int(H * p_rois[i, 4])
        x1 = int(W * p_rois[i, 2]); x2 = int(W * p_rois[i, 5])
        out.append(nearest_resize(gt_masks[:, z1:z2, y1:y2, x1:x2], mask_shape))
    return torch.stack(out, dim=0)


def bbox_overlaps(boxes1, boxes2):
    """model.py:373-411: IoU matrix [len(boxes1), len(boxes2)], fp32, no epsilon (x, y, z factor order)."""
    b1 = boxes1[:, None, :]
    b2 = boxes2[None, :, :]
    z1 = torch.max(b1[..., 0], b2[..., 0]); y1 = torch.max(b1[..., 1], b2[..., 1]); x1 = torch.max(b1[..., 2], b2[..., 2])
    z2 = torch.min(b1[..., 3], b2[..., 3]); y2 = torch.min(b1[..., 4], b2[..., 4]); x2 = torch.min(b1[..., 5], b2[..., 5])
    zero = torch.zeros(())
    inter = torch.max(x2 - x1, zero) * torch.max(y2 - y1, zero) * torch.max(z2 - z1, zero)
    v1 = (b1[..., 3] - b1[..., 0]) * (b1[..., 4] - b1[..., 1]) * (b1[..., 5] - b1[..., 2])
    v2 = (b2[..., 3] - b2[..., 0]) * (b2[..., 4] - b2[..., 1]) * (b2[..., 5] - b2[..., 2])
    return inter / (v1 + v2 - inter)


def detection_target_layer(proposals, gt_class_ids, gt_boxes, gt_masks, mask_shape, perm_pos, perm_neg,
                           train_rois=15, positive_ratio=0.33, iou_threshold=0.5,
                           std_dev=(0.1, 0.1, 0.1, 0.2, 0.2, 0.2), count_round=False):
    """model.py:414-563 with the two torch.randperm draws (459, 505) injected: proposals [N,6] and gt_boxes [G,6]
    normalised, gt_class_ids [G], gt_masks one-hot [C,D,H,W].  Returns (positive_rois, rois, class_ids, deltas,
    masks [n_pos,C,*mask_shape]) -- positives first.  Only the 'positives and negatives' / 'positives only' branches
    are restated (the others return empty sets or hit the NameError of App. A-15)."""
    overlaps = bbox_overlaps(proposals, gt_boxes)
    iou_max = overlaps.max(dim=1)[0]
    pos_idx = torch.nonzero(iou_max >= iou_threshold)[:, 0]
    if pos_idx.numel() == 0:
        raise ValueError("no positive RoI: the reference skips the heads for this sample")
    count = (lambda v: int(round(v))) if count_round else int     # LiTS_2017/model.py:448, 496 round; heart truncates
    pos_idx = pos_idx[perm_pos[:count(train_rois * positive_ratio)]]
    n_pos = pos_idx.numel()
    p_rois = proposals[pos_idx]
    assign = overlaps[pos_idx].max(dim=1)[1]
    deltas = box_refinement(p_rois, gt_boxes[assign]) / torch.tensor(std_dev, dtype=torch.float32)
    class_ids = gt_class_ids[assign].long()
    masks = mask_targets(p_rois, gt_masks, mask_shape)
    neg_idx = torch.nonzero(iou_max < iou_threshold)[:, 0]
    rois = p_rois
    if neg_idx.numel() != 0:
        n_neg = count((1.0 / positive_ratio) * n_pos - n_pos)
        neg_idx = neg_idx[perm_neg[:n_neg]]
        rois = torch.cat([p_rois, proposals[neg_idx]], dim=0)
        class_ids = torch.cat([class_ids, torch.zeros(neg_idx.numel(), dtype=torch.long)])
        deltas = torch.cat([deltas, torch.zeros(neg_idx.numel(), 6)], dim=0)
    return p_rois, rois, class_ids, deltas, masks


# --------------------------------------------------------------------------------------
# one training step with injected RoI sets; the P row of 8(a)
# --------------------------------------------------------------------------------------

LOSS_WEIGHTS = (100.0, 50.0, 1.0, 20.0, 1.0, 1.0)  # heart_main.py:161-168


def training_step(sd, image, anchors, rpn_match, rpn_bbox_t, p_rois, n_rois, target_class_ids,
                  target_deltas, target_mask, stage, pool_size, mask_pool_size, dropout_masks=None,
                  proposal_count=500, nms_threshold=0.7, pre_nms_limit=1000, layers=(2, 3), stem_pad=(1, 3, 3),
                  ce_class_weights=None, edge_raw=False, stage_split=False, loss_weights=None):
    """predict('training') dataflow (model.py:1391-1514) + compute_losses (984-1000) with the
    head RoIs injected (p_rois positives first, then n_rois).  image [1,1,D,H,W].
    ce_class_weights / edge_raw: the LiTS fork's mask losses (LiTS_2017/model.py:926, 959-972); stage_split: its two
    training phases ('beginning': no mask head, mask losses 0; otherwise: no classifier head, detection losses 0;
    LiTS_2017/model.py:985-1001, 1528-1548).
    Returns dict of outputs and the 6 losses."""
    p2, p3 = fpn(image, sd, layers=layers, stem_pad=stem_pad)
    l2, pr2, b2 = rpn(p2, sd)
    l3, pr3, b3 = rpn(p3, sd)
    rpn_logits = torch.cat([l2, l3], dim=1)
    rpn_probs = torch.cat([pr2, pr3], dim=1)
    rpn_box = torch.cat([b2, b3], dim=1)
    D, H, W = image.shape[2:]
    rpn_rois, keep, order = proposal_layer(rpn_probs[0], rpn_box[0], anchors, proposal_count, nms_threshold,
                                           (D, H, W), pre_nms_limit)
    rois = torch.cat([p_rois, n_rois], dim=0)
    det_only, mask_only = stage_split and stage == "beginning", stage_split and stage != "beginning"
    cls_logits = cls_probs = cls_bbox = m_logits = m_probs = None
    zero = torch.zeros(())
    if not mask_only:
        cls_logits, cls_probs, cls_bbox = classifier([p2[0], p3[0]], rois, sd, pool_size)
        losses = [rpn_class_loss(rpn_match, rpn_logits),
                  rpn_bbox_loss(rpn_bbox_t, rpn_match, rpn_box),
                  mrcnn_class_loss(target_class_ids, cls_logits),
                  mrcnn_bbox_loss(target_deltas, target_class_ids, cls_bbox)]
    else:
        losses = [zero, zero, zero, zero]
    if not det_only:
        m_logits, m_probs = mask_head(image[0], p_rois, sd, mask_pool_size, stage, dropout_masks=dropout_masks)
        losses += [mask_ce_loss(target_mask, m_logits) if ce_class_weights is None
                   else mask_ce_loss_weighted(target_mask, m_logits, ce_class_weights),
                   ((edge_loss_raw if edge_raw else edge_loss)(target_mask, m_probs)[0])
                   # heart: 'finetune' only (model.py:995-996); the fork: every non-'beginning' stage, incl. 'together'
                   # (LiTS_2017/model.py:995-1001)
                   if (stage != "beginning" if (stage_split or edge_raw) else stage == "finetune") else zero]
    else:
        losses += [zero, zero]
    total = sum(wt * l for wt, l in zip(LOSS_WEIGHTS if loss_weights is None else loss_weights, losses))
    return dict(p2=p2, p3=p3, rpn_logits=rpn_logits, rpn_probs=rpn_probs, rpn_bbox=rpn_box, rpn_rois=rpn_rois,
                nms_keep=keep, cls_logits=cls_logits, cls_bbox=cls_bbox, mask_logits=m_logits, mask_probs=m_probs,
                losses=losses, total=total)


def inference_step(sd, image, anchors, stage, pool_size, mask_pool_size, window=None, proposal_count=64,
                   nms_threshold=0.7, pre_nms_limit=1000, min_confidence=0.7, detection_nms_threshold=0.3,
                   max_instances=32, layers=(2, 3), stem_pad=(1, 3, 3)):
    """predict(mode='inference') dataflow (model.py:1408-1461): FPN -> RPN -> proposal_layer
    (POST_NMS_ROIS_INFERENCE) -> classifier -> detection_layer / refine_detections -> mask head (eval: no dropout)
    on the detected boxes.  Returns detections [M,8] (voxels) and mask probabilities [M,C,d,h,w]."""
    with torch.no_grad():
        p2, p3 = fpn(image, sd, layers=layers, stem_pad=stem_pad)
        _, pr2, b2 = rpn(p2, sd)
        _, pr3, b3 = rpn(p3, sd)
        rpn_probs = torch.cat([pr2, pr3], dim=1)
        rpn_box = torch.cat([b2, b3], dim=1)
        D, H, W = [int(v) for v in image.shape[2:]]
        rpn_rois, _, _ = proposal_layer(rpn_probs[0], rpn_box[0], anchors, proposal_count, nms_threshold, (D, H, W),
                                        pre_nms_limit)
        _, cls_probs, cls_bbox = classifier([p2[0], p3[0]], rpn_rois, sd, pool_size)
        if window is None:
            window = (0.0, 0.0, 0.0, float(D), float(H), float(W))
        det = refine_detections(rpn_rois, cls_probs, cls_bbox, window, (D, H, W), min_confidence,
                                detection_nms_threshold, max_instances)
        if det.shape[0] == 0:
            return dict(rpn_rois=rpn_rois, cls_probs=cls_probs, detections=det, mask_probs=None)
        boxes = det[:, :6] / torch.tensor([D, H, W, D, H, W], dtype=torch.float32)
        _, m_probs = mask_head(image[0], boxes, sd, mask_pool_size, stage, dropout_masks=None)
    return dict(rpn_rois=rpn_rois, cls_probs=cls_probs, detections=det, mask_probs=m_probs)


# --------------------------------------------------------------------------------------
# input pipeline.  PARITY UNPINNED by the reference: scikit-image is not in this image, so
# no golden can be produced from utils.resize_image itself.  skimage.transform.resize (>= 0.19, _warps.py) evaluates an
# n-D (n > 2) resize without anti-aliasing as scipy.ndimage.zoom(image, out/in, order, mode = 'grid-constant' for
# mode = 'constant', cval, grid_mode = True) followed by a clip to the input's range; scipy 1.15.3 IS here and is what
# this restatement calls.
# --------------------------------------------------------------------------------------
def skimage_resize(image, output_shape, order):
    import scipy.ndimage as ndi
    image = np.asarray(image)
    zoom = [o / float(i) for o, i in zip(output_shape, image.shape)]
    out = ndi.zoom(image.astype(np.float64) if order else image, zoom, order=order, mode="grid-constant", cval=0.0, grid_mode=True)
    if order:
        out = np.clip(out, image.min(), image.max())
    return out


def resize_image_self(image, min_dim, max_dim):
    """utils.resize_image(mode='self') (utils.py:389-393): [H,W,D,1] -> [max,max,min,1], cast back to the input dtype."""
    out = skimage_resize(image, (max_dim, max_dim, min_dim, 1), 1)
    return out.astype(image.dtype), (0, 0, 0, min_dim, max_dim, max_dim), -1, [(0, 0)] * 4, None


def mold_inputs(images, min_dim, max_dim, num_classes):
    """MaskRCNN.mold_inputs (model.py:1774-1810): resize 'self', mold_image (z-score, population std), [C,D,H,W]."""
    molded, metas, windows = [], [], []
    for image in images:
        m, window, _, _, _ = resize_image_self(image, min_dim, max_dim)
        m = (m - m.mean()) / m.std()
        molded.append(m.transpose((3, 2, 0, 1)))
        metas.append(np.array([0] + list(image.shape) + list(window) + [0] * num_classes))
        windows.append(window)
    return np.stack(molded), np.stack(metas), np.stack(windows)


def mold_inputs_lits(images, pad_shape, image_shape, min_dim, max_dim, num_classes):
    """LiTS_2017/model.py:1730-1775: preprocess_image ((x - 300) / -600 clamped to [0,1], :1875-1883), centre in a zero
    PAD_IMAGE_SHAPE frame, resize(order = 0) to IMAGE_SHAPE, [1,D,H,W], fractional window."""
    molded, metas, windows = [], [], []
    for image in images:
        img = (np.asarray(image, dtype=np.float64) - 300.0) / (-600.0)
        img[img > 1.0] = 1.0
        img[img < 0.0] = 0.0
        whole = np.zeros(pad_shape)
        sx, sy, sz = [int((p - i) / 2.0) for p, i in zip(pad_shape, img.shape)]
        whole[sx:sx + img.shape[0], sy:sy + img.shape[1], sz:sz + img.shape[2]] = img
        out = skimage_resize(whole, tuple(image_shape[:3]), 0)
        window = (sz * image_shape[2] / pad_shape[2], sx * image_shape[0] / pad_shape[0], sy * image_shape[1] / pad_shape[1],
                  min_dim - sz * image_shape[2] / pad_shape[2], max_dim - sx * image_shape[0] / pad_shape[0],
                  max_dim - sy * image_shape[1] / pad_shape[1])
        molded.append(np.expand_dims(out.transpose((2, 0, 1)), axis=0))
        metas.append(np.array([0] + list(out.shape) + list(window) + [0] * num_classes))
        windows.append(window)
    return np.stack(molded), np.stack(metas), np.stack(windows)
