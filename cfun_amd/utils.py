"""Box utilities of the hot path -- drop-in surface of the reference's ``utils`` for NMS, IoU and anchors.

``non_max_suppression`` keeps the reference signature (numpy in, int32 numpy out, utils.py:122-157) but
runs the HIP NMS kernel; ``nms_device`` is the sync-free variant used inside ``proposal_layer``.
"""
import numpy as np
import torch

from . import ops


def _device():
    from . import _lib
    return torch.device("cpu") if _lib.is_emulator() else torch.device("cuda", torch.cuda.current_device())


def nms_device(boxes, scores, threshold, max_num):
    """boxes [n,6] / scores [n] device tensors -> int64 indices of the kept boxes in pick order.
    One host sync (reading the count), where the reference synchronises too (model.py:244)."""
    keep, count = ops.nms3d(boxes, scores, threshold, max_num)
    return keep[:int(count.item())].long()


def non_max_suppression(boxes, scores, threshold, max_num):
    """utils.py:122-157 drop-in: numpy [N,6] (z1,y1,x1,z2,y2,x2), numpy [N] -> np.int32 pick list."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 6)
    scores = np.ascontiguousarray(scores, dtype=np.float32).reshape(-1)
    if boxes.shape[0] == 0:
        return np.zeros((0,), dtype=np.int32)
    dev = _device()
    keep, count = ops.nms3d(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), threshold, max_num)
    return keep[:int(count.item())].cpu().numpy().astype(np.int32)


def compute_iou(box, boxes, box_volume, boxes_volume):
    """utils.py:50-70 (host helper kept for callers outside the hot path)."""
    z1 = np.maximum(box[0], boxes[:, 0]); z2 = np.minimum(box[3], boxes[:, 3])
    y1 = np.maximum(box[1], boxes[:, 1]); y2 = np.minimum(box[4], boxes[:, 4])
    x1 = np.maximum(box[2], boxes[:, 2]); x2 = np.minimum(box[5], boxes[:, 5])
    inter = np.maximum(x2 - x1, 0) * np.maximum(y2 - y1, 0) * np.maximum(z2 - z1, 0)
    return inter / (box_volume + boxes_volume - inter + 1e-6)


def generate_anchors(scale, shape, feature_stride, anchor_stride=1):
    """Cubic anchors of side ``scale`` centred at index*stride, enumerated y-slowest, then z, then x --
    the order np.meshgrid(z, y, x) (default 'xy' indexing) gives the reference (utils.py:467-507, App. A-7)."""
    z = np.arange(0, shape[0], anchor_stride) * feature_stride
    y = np.arange(0, shape[1], anchor_stride) * feature_stride
    x = np.arange(0, shape[2], anchor_stride) * feature_stride
    yy, zz, xx = np.meshgrid(y, z, x, indexing="ij")
    centers = np.stack([zz.ravel(), yy.ravel(), xx.ravel()], axis=1).astype(np.float64)
    half = 0.5 * float(scale)
    return np.concatenate([centers - half, centers + half], axis=1)


def generate_pyramid_anchors(scales, ratios, feature_shapes, feature_strides, anchor_stride):
    """utils.py:510-528 (one scale per pyramid level, ratios == [1])."""
    if list(ratios) != [1]:
        raise NotImplementedError("the reference only uses RPN_ANCHOR_RATIOS == [1]")
    return np.concatenate([generate_anchors(s, sh, st, anchor_stride)
                           for s, sh, st in zip(scales, feature_shapes, feature_strides)], axis=0)


def compute_backbone_shapes(config, image_shape):
    """model.py:91-101: [[D/s, H/s, W/s] for each backbone stride], image_shape = [H, W, D, C]."""
    h, w, d = [int(v) for v in image_shape[:3]]
    return np.array([[-(-d // s), -(-h // s), -(-w // s)] for s in config.BACKBONE_STRIDES])


def denorm_boxes_graph(boxes, size):
    """utils.py:160-174: normalised -> pixel coordinates (fp32 torch.mul)."""
    d, h, w = size
    return torch.mul(boxes, torch.tensor([d, h, w, d, h, w], dtype=torch.float32, device=boxes.device))


def box_refinement(box, gt_box):
    """utils.py:92-119."""
    d = box[:, 3] - box[:, 0]; h = box[:, 4] - box[:, 1]; w = box[:, 5] - box[:, 2]
    cz = box[:, 0] + 0.5 * d; cy = box[:, 1] + 0.5 * h; cx = box[:, 2] + 0.5 * w
    gd = gt_box[:, 3] - gt_box[:, 0]; gh = gt_box[:, 4] - gt_box[:, 1]; gw = gt_box[:, 5] - gt_box[:, 2]
    gz = gt_box[:, 0] + 0.5 * gd; gy = gt_box[:, 1] + 0.5 * gh; gx = gt_box[:, 2] + 0.5 * gw
    return torch.stack([(gz - cz) / d, (gy - cy) / h, (gx - cx) / w,
                        torch.log(gd / d), torch.log(gh / h), torch.log(gw / w)], dim=1)


# ---- input formatting (model.py:1870-1904) --------------------------------------------------------------------
def mold_image(images):
    """model.py:1902-1904: z-score with the whole-volume mean and POPULATION std (numpy's default ddof = 0); works
    on numpy arrays and on (device) tensors."""
    if torch.is_tensor(images):
        return (images - images.mean()) / images.std(unbiased=False)
    return (images - images.mean()) / images.std()


def compose_image_meta(image_id, image_shape, window, active_class_ids):
    """model.py:1870-1887."""
    return np.array([image_id] + list(image_shape) + list(window) + list(active_class_ids))


def parse_image_meta(meta):
    """model.py:1890-1899."""
    return meta[:, 0], meta[:, 1:5], meta[:, 5:11], meta[:, 11:]


# ---- input pipeline ---------------------------------------------------------------------
def resize_image(image, min_dim=None, max_dim=None, min_scale=None, mode="square", device=None):
    """utils.py:342-393 drop-in: image [H,W,D,C=1] (numpy or tensor).  The reference implements two modes: 'none' and
    'self' -- resize to [max_dim, max_dim, min_dim, 1] with skimage.transform.resize(order=1, mode='constant',
    preserve_range=True) -- and falls off the end (returns None) for the documented 'square' / 'pad64' / 'crop'; those
    raise here.  Returns the reference's tuple (image [H',W',D',1] in the input's dtype, window (z1,y1,x1,z2,y2,x2),
    scale, padding, crop); the resize runs on the device (cfun_resize3d), numpy in -> numpy out."""
    h, w, d = image.shape[:3]
    if mode == "none":
        return image, (0, 0, 0, d, h, w), 1, [(0, 0), (0, 0), (0, 0), (0, 0)], None
    if mode != "self":
        raise NotImplementedError("resize_image: mode %r is not implemented by the reference either (utils.py:380-393)" % (mode,))
    was_numpy = not torch.is_tensor(image)
    t = torch.as_tensor(np.ascontiguousarray(image) if was_numpy else image)
    dtype = t.dtype
    if t.dim() != 4 or t.shape[3] != 1:
        raise ValueError("resize_image: [H,W,D,1] volume expected, got %s" % (tuple(t.shape),))
    dev = torch.device(device) if device is not None else (t.device if t.is_cuda else _device())
    vol = t[..., 0].to(device=dev, dtype=torch.float32)
    out = ops.resize3d(vol, (max_dim, max_dim, min_dim), order=1, clip=True)[..., None].to(dtype)
    out = out.cpu().numpy() if was_numpy else out
    return out, (0, 0, 0, min_dim, max_dim, max_dim), -1, [(0, 0), (0, 0), (0, 0), (0, 0)], None


def resize_mask(mask, scale, padding, max_dim=0, min_dim=0, crop=None, mode="square", device=None):
    """utils.py:396-408 drop-in, mode 'self': nearest resize of a [H,W,D] label volume to [max_dim, max_dim, min_dim],
    rounded to int32."""
    if mode != "self":
        raise NotImplementedError("resize_mask: only mode 'self' exists in the reference (utils.py:404-408)")
    was_numpy = not torch.is_tensor(mask)
    t = torch.as_tensor(np.ascontiguousarray(mask) if was_numpy else mask)
    dev = torch.device(device) if device is not None else (t.device if t.is_cuda else _device())
    out = torch.round(ops.resize3d(t.to(device=dev, dtype=torch.float32), (max_dim, max_dim, min_dim), order=0)).to(torch.int32)
    return out.cpu().numpy() if was_numpy else out


def mold_inputs(config, images, device=None):
    """MaskRCNN.mold_inputs (model.py:1774-1810): every image [H,W,D,1] is resized to the network size (mode 'self'),
    z-scored (mold_image) and laid out [1,D,H,W]; resize, statistics and normalisation run on the device and the source
    array is read in place through its strides (no host transpose).  Returns (molded_images float32 tensor [N,1,D,H,W]
    on the device, image_metas numpy [N, 1+4+6+NUM_CLASSES], windows numpy [N,6])."""
    dev = torch.device(device) if device is not None else _device()
    molded, metas, windows = [], [], []
    mx, mn = int(config.IMAGE_MAX_DIM), int(config.IMAGE_MIN_DIM)
    for image in images:
        if config.IMAGE_RESIZE_MODE != "self":
            raise NotImplementedError("mold_inputs: IMAGE_RESIZE_MODE %r" % (config.IMAGE_RESIZE_MODE,))
        t = torch.as_tensor(np.ascontiguousarray(image) if not torch.is_tensor(image) else image)
        dtype = t.dtype
        vol = t[..., 0].to(device=dev, dtype=torch.float32).permute(2, 0, 1)        # [D,H,W] view of the [H,W,D] array
        out = ops.resize3d(vol, (mn, mx, mx), order=1, clip=True)
        if not dtype.is_floating_point:       # the reference casts the resized image back to the loader's dtype
            out = out.to(dtype).to(torch.float32)
        molded.append(mold_image(out)[None])
        window = (0, 0, 0, mn, mx, mx)
        metas.append(compose_image_meta(0, tuple(image.shape), window, np.zeros([config.NUM_CLASSES], dtype=np.int32)))
        windows.append(window)
    return torch.stack(molded), np.stack(metas), np.stack(windows)


def preprocess_image_lits(image):
    """LiTS_2017/model.py:1875-1883 (sic: MIN_BOUND = 300, MAX_BOUND = -300): (x - 300) / (-600), clamped to [0, 1]."""
    return ((image - 300.0) / (-600.0)).clamp(0.0, 1.0)


def mold_inputs_lits(config, images, device=None):
    """The LiTS fork's MaskRCNN.mold_inputs (LiTS_2017/model.py:1730-1775): clamp-normalise, centre in a zero
    PAD_IMAGE_SHAPE frame, nearest-resize the frame to IMAGE_SHAPE, fractional window.  The frame is virtual
    (cfun_resize3d's ``frame`` / ``offset``): the 646x646x536 host array of the reference is never built.  Returns
    (molded [N,1,D,H,W] device tensor, image_metas, windows) with the fork's 3-entry image shape in the meta."""
    dev = torch.device(device) if device is not None else _device()
    ph, pw, pd = [int(v) for v in config.PAD_IMAGE_SHAPE]
    h, w, d = [int(v) for v in config.IMAGE_SHAPE[:3]]
    molded, metas, windows = [], [], []
    for image in images:
        t = torch.as_tensor(np.ascontiguousarray(image) if not torch.is_tensor(image) else image)
        vol = preprocess_image_lits(t.to(device=dev, dtype=torch.float32))          # [H,W,D]
        ih, iw, idp = [int(v) for v in vol.shape]
        sx, sy, sz = int((ph - ih) / 2.0), int((pw - iw) / 2.0), int((pd - idp) / 2.0)
        out = ops.resize3d(vol.permute(2, 0, 1), (d, h, w), order=0, frame=(pd, ph, pw), offset=(sz, sx, sy))
        window = (sz * d / pd, sx * h / ph, sy * w / pw, config.IMAGE_MIN_DIM - sz * d / pd,
                  config.IMAGE_MAX_DIM - sx * h / ph, config.IMAGE_MAX_DIM - sy * w / pw)
        molded.append(out[None])
        metas.append(compose_image_meta(0, (h, w, d), window, np.zeros([config.NUM_CLASSES], dtype=np.int32)))
        windows.append(window)
    return torch.stack(molded), np.stack(metas), np.stack(windows)
