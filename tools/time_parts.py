#!/usr/bin/env python3
"""How the cfg2 step splits between the RoI mask head (U-Net + mask losses) and everything else (FPN, RPN, proposals,
classifier head and their losses): each part's forward + backward timed alone (tools only)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfun_amd import config as C, model, ops, step as S  # noqa: E402

cfg = C.heart_config("finetune", 256, 256, 128)
dev = torch.device("cuda")
net = S.CFUNHotPath(cfg).to(dev)
s = S.synthetic_inputs(cfg, dev)


def full():
    S.training_step(net, s)


def mask_only():
    net.train()
    img = ops.to_ndhwc(s["image"])[0]
    if S.FUSED_MASK_LOSS:
        logits, _ = net.mask.forward_ndhwc(img, s["p_rois"], softmax=False)
        ce, edge, probs = ops.mask_losses_fused(logits, s["mask_labels"])
    else:
        logits, probs = net.mask.forward_ndhwc(img, s["p_rois"])
        ce, edge = ops.mask_losses(logits, probs, s["mask_labels"])
    (ce + edge).backward()


def rest_only():
    net.train()
    p2, p3, rpn_logits, rpn_probs, rpn_bbox = net.backbone_rpn(s["image"])
    net.proposals(rpn_probs, rpn_bbox, "training")
    rois = torch.cat([s["p_rois"], s["n_rois"]], dim=0)
    cls_logits, cls_probs, cls_bbox = net.classifier.forward_ndhwc([p2[0], p3[0]], rois)
    losses = [model.compute_rpn_class_loss(s["rpn_match"], rpn_logits),
              model.compute_rpn_bbox_loss(s["rpn_bbox_t"], s["rpn_match"], rpn_bbox),
              model.compute_mrcnn_class_loss(s["target_class_ids"], cls_logits),
              model.compute_mrcnn_bbox_loss(s["target_deltas"], s["target_class_ids"], cls_bbox)]
    sum(losses).backward()


for name, fn in (("full step", full), ("mask head only", mask_only), ("FPN+RPN+proposals+classifier only", rest_only)):
    for _ in range(3):
        net.zero_grad(set_to_none=True)
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        net.zero_grad(set_to_none=True)
        fn()
    torch.cuda.synchronize()
    print("%-36s %.2f ms" % (name, (time.perf_counter() - t0) * 100.0))
