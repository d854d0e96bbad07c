"""The reference's training driver around the hot path (model.py:1516-1676), on the flat-arena optimizer:

    make_optimizer   the SGD train_model builds (model.py:1534-1545): every trainable parameter, weight decay WEIGHT_DECAY
                     on the ones without 'bn' in their name (BatchNorm is frozen on this path, so that is all of them),
                     momentum LEARNING_MOMENTUM
    train_epoch      MaskRCNN.train_epoch (model.py:1574-1676): per sample forward, six losses, weighted total, backward,
                     clip_grad_norm_(5.0); optimizer step + zero_grad every BATCH_SIZE samples; the epoch's mean losses

Out of this path: the DataLoader, load_image_gt's augmentation and anchor targets -- a sample here is
what they hand to ``predict``: image, GT class ids / boxes / label volume, rpn_match, rpn_bbox.
"""
import torch

from . import optim, step


def make_optimizer(net, config, learning_rate=None, group=None, bucket_bytes=64 << 20):
    lr = config.LEARNING_RATE if learning_rate is None else learning_rate
    return optim.FlatSGD(net.named_parameters(), lr=lr, momentum=config.LEARNING_MOMENTUM, weight_decay=config.WEIGHT_DECAY,
                         clip_norm=5.0, group=group, bucket_bytes=bucket_bytes)


def train_epoch(net, samples, optimizer, steps, config, perms=None):
    """``samples``: an iterable of dicts with ``image`` [1,1,D,H,W], ``gt_class_ids`` [G], ``gt_boxes`` [G,6],
    ``gt_labels`` [D,H,W] (uint8 label volume: the one-hot ``gt_masks`` of the reference in 1/8 of the bytes),
    ``rpn_match`` [1,A,1], ``rpn_bbox_t`` [1,R,6].  ``perms``: optional iterable of (positive, negative) randperm draws
    per step for detection_target_layer (tests replay the reference's).  Returns (loss, rpn_class, rpn_bbox, mrcnn_class,
    mrcnn_bbox, mrcnn_mask, mrcnn_mask_edge) averaged over ``steps`` like the reference: the first is the WEIGHTED total,
    the others the unweighted losses."""
    batch_size = max(1, int(getattr(config, "BATCH_SIZE", 1)))
    sums = torch.zeros(7, dtype=torch.float64)
    acc = None
    perms = iter(perms) if perms is not None else None
    batch_count = 0
    optimizer.zero_grad()
    for i, s in enumerate(samples):
        batch_count += 1
        pm = next(perms) if perms is not None else None
        # data-parallel ranks: only the last backward of the batch all-reduces (the accumulated sums); the clip after an
        # accumulating pass acts on the rank's own gradient, the step's fused clip on the mean over the ranks
        optimizer.begin_backward(last=(batch_count % batch_size == 0))
        _, losses, total = step.training_step_full(net, s["image"], s["gt_class_ids"], s["gt_boxes"], s["gt_labels"],
                                                   s["rpn_match"], s["rpn_bbox_t"], perms=pm)
        if batch_count % batch_size == 0:
            optimizer.step()          # (the clip of this backward is fused into the update)
            optimizer.zero_grad()
            batch_count = 0
        else:
            optimizer.clip_()
        vals = torch.stack([total.detach()] + [l.detach() for l in losses]).double()      # stays on the device: no sync per step
        acc = vals if acc is None else acc + vals
        if i == steps - 1:
            break
    if acc is not None:
        sums = (acc / steps).cpu()
    return tuple(float(v) for v in sums)
