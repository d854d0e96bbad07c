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
