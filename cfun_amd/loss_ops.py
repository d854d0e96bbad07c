"""torch.autograd bindings of the mask-head loss kernels (csrc/loss.hip): channel softmax, cross-entropy (heart + the fork's
weighted form), the 3-D Sobel edge losses and the fused mask-loss pair of the training step (reference model.py:794-981,
LiTS_2017/model.py:907-979).  Split out of ops.py in round 4; ``cfun_amd.ops`` re-exports every name."""
import torch

from . import _lib
from ._lib import check, ptr, stream, workspace


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _Softmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits):
        lib = _lib.load()
        logits = _c(logits)
        c = logits.shape[-1]
        probs = torch.empty_like(logits)
        check(lib.cfun_softmax_fwd(ptr(logits), ptr(probs), logits.numel() // c, c, stream(logits)), "softmax_fwd")
        ctx.save_for_backward(probs)
        return probs

    @staticmethod
    def backward(ctx, dp):
        lib = _lib.load()
        (probs,) = ctx.saved_tensors
        dp = _c(dp)
        c = probs.shape[-1]
        dl = torch.empty_like(probs)
        check(lib.cfun_softmax_bwd(ptr(probs), ptr(dp), ptr(dl), probs.numel() // c, c, stream(probs)), "softmax_bwd")
        return dl


def softmax_channels(logits):
    """softmax over the last (channel) axis of an NDHWC tensor (model.py:799)."""
    return _Softmax.apply(logits)


class _MaskCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        lib = _lib.load()
        logits = _c(logits)
        labels = _c(labels)
        c = logits.shape[-1]
        nvox = logits.numel() // c
        if labels.dtype != torch.uint8 or labels.numel() != nvox:
            raise RuntimeError("mask_cross_entropy: labels must be uint8 [n,D,H,W]")
        loss = torch.empty((1,), dtype=torch.float32, device=logits.device)
        ws = workspace(lib.cfun_loss_workspace_bytes(nvox), logits)
        check(lib.cfun_softmax_ce_fwd(ptr(logits), ptr(labels), ptr(loss), nvox, c, ptr(ws), ws.numel(),
                                      stream(logits)), "softmax_ce_fwd")
        ctx.save_for_backward(logits, labels)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        logits, labels = ctx.saved_tensors
        c = logits.shape[-1]
        gs = _c(g.reshape(1).float())
        dl = torch.empty_like(logits)
        check(lib.cfun_softmax_ce_bwd(ptr(logits), ptr(labels), ptr(gs), ptr(dl), logits.numel() // c, c,
                                      stream(logits)), "softmax_ce_bwd")
        return dl, None


class _MaskCEWeighted(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, weight):
        lib = _lib.load()
        logits, labels = _c(logits), _c(labels)
        weight = _c(weight.detach().to(device=logits.device, dtype=torch.float32))
        c = logits.shape[-1]
        nvox = logits.numel() // c
        if labels.dtype != torch.uint8 or labels.numel() != nvox or weight.numel() != c:
            raise RuntimeError("mask_cross_entropy: labels must be uint8 [n,D,H,W], weight [C]")
        loss = torch.empty((1,), dtype=torch.float32, device=logits.device)
        wsum = torch.empty((1,), dtype=torch.float32, device=logits.device)
        ws = workspace(lib.cfun_ce_weighted_workspace_bytes(), logits)
        check(lib.cfun_softmax_ce_weighted_fwd(ptr(logits), ptr(labels), ptr(weight), ptr(loss), ptr(wsum), nvox, c, ptr(ws),
                                               ws.numel(), stream(logits)), "softmax_ce_weighted_fwd")
        ctx.save_for_backward(logits, labels, weight, wsum)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        logits, labels, weight, wsum = ctx.saved_tensors
        c = logits.shape[-1]
        gs = _c(g.reshape(1).float())
        dl = torch.empty_like(logits)
        check(lib.cfun_softmax_ce_weighted_bwd(ptr(logits), ptr(labels), ptr(weight), ptr(gs), ptr(wsum), ptr(dl),
                                               logits.numel() // c, c, stream(logits)), "softmax_ce_weighted_bwd")
        return dl, None, None


def mask_cross_entropy(logits, labels, weight=None):
    """CrossEntropyLoss(logits [n,D,H,W,C], labels uint8 [n,D,H,W]) -- model.py:909-935; ``weight`` [C]: the LiTS
    fork's class weights (LiTS_2017/model.py:926)."""
    if weight is None:
        return _MaskCE.apply(logits, labels)
    return _MaskCEWeighted.apply(logits, labels, torch.as_tensor(weight, dtype=torch.float32))


class _EdgeRaw(torch.autograd.Function):
    @staticmethod
    def forward(ctx, probs, labels):
        lib = _lib.load()
        probs, labels = _c(probs), _c(labels)
        n, d, h, w, c = probs.shape
        loss = torch.empty((1,), dtype=torch.float32, device=probs.device)
        keep = ctx.needs_input_grad[0]
        dc = torch.empty(lib.cfun_edge_raw_dc_bytes(n, d, h, w, c) // 4, dtype=torch.float32, device=probs.device) \
            if keep else None
        ws = workspace(lib.cfun_loss_workspace_bytes(n * d * h * w), probs)
        check(lib.cfun_edge_raw_fwd(ptr(probs), ptr(labels), ptr(loss), ptr(dc), n, d, h, w, c, ptr(ws), ws.numel(),
                                    stream(probs)), "edge_raw_fwd")
        ctx.shape = tuple(probs.shape)
        ctx.save_for_backward(dc)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (dc,) = ctx.saved_tensors
        n, d, h, w, c = ctx.shape
        gs = _c(g.reshape(1).float())
        dp = torch.empty(ctx.shape, dtype=torch.float32, device=dc.device)
        check(lib.cfun_edge_raw_bwd(ptr(dc), ptr(gs), ptr(dp), n, d, h, w, c, stream(dc)), "edge_raw_bwd")
        return dp, None


def edge_loss_raw(probs, labels):
    """The LiTS fork's edge loss (LiTS_2017/model.py:936-979): MSE on the raw three Sobel responses of NDHWC probabilities
    vs uint8 labels, foreground classes only; differentiable w.r.t. ``probs``."""
    return _EdgeRaw.apply(probs, labels)


class _EdgeLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, probs, labels):
        lib = _lib.load()
        probs = _c(probs)
        labels = _c(labels)
        n, d, h, w, c = probs.shape
        loss = torch.empty((1,), dtype=torch.float32, device=probs.device)
        ws = workspace(lib.cfun_loss_workspace_bytes(n * d * h * w), probs)
        check(lib.cfun_edge_loss_fwd(ptr(probs), ptr(labels), ptr(loss), n, d, h, w, c, ptr(ws), ws.numel(),
                                     stream(probs)), "edge_loss_fwd")
        ctx.save_for_backward(probs, labels)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        probs, labels = ctx.saved_tensors
        n, d, h, w, c = probs.shape
        gs = _c(g.reshape(1).float())
        dp = torch.empty_like(probs)
        ws = workspace(lib.cfun_edge_loss_bwd_workspace_bytes(n, d, h, w, c), probs)
        check(lib.cfun_edge_loss_bwd(ptr(probs), ptr(labels), ptr(gs), ptr(dp), n, d, h, w, c, ptr(ws), ws.numel(),
                                     stream(probs)), "edge_loss_bwd")
        return dp, None


def edge_loss(probs, labels):
    """3-D Sobel edge-agreement loss (model.py:938-981) on NDHWC probabilities and uint8 labels."""
    return _EdgeLoss.apply(probs, labels)


class _MaskLosses(torch.autograd.Function):
    """(CE(logits, labels), edge(probs, labels)) with ONE fused backward pass; probs = softmax(logits) comes from
    the Mask module (model.py:799) and is taken as a constant here -- its dependence on logits is folded into the
    backward (softmax_bwd inside cfun_mask_losses_bwd), so no gradient flows through the softmax node."""

    @staticmethod
    def forward(ctx, logits, probs, labels):
        lib = _lib.load()
        logits, probs, labels = _c(logits), _c(probs), _c(labels)
        n, d, h, w, c = logits.shape
        nvox = n * d * h * w
        if labels.dtype != torch.uint8 or labels.numel() != nvox:
            raise RuntimeError("mask_losses: labels must be uint8 [n,D,H,W]")
        out = torch.empty((2,), dtype=torch.float32, device=logits.device)
        ws = workspace(lib.cfun_loss_workspace_bytes(nvox), logits)
        st = stream(logits)
        check(lib.cfun_softmax_ce_fwd(ptr(logits), ptr(labels), ptr(out), nvox, c, ptr(ws), ws.numel(), st),
              "softmax_ce_fwd")
        dc = None
        if ctx.needs_input_grad[0] and min(d, h, w) >= 3:   # training: keep the edge coefficients, one-pass backward
            dc = torch.empty(lib.cfun_edge_loss_bwd_workspace_bytes(n, d, h, w, c) // 4, dtype=torch.float32,
                             device=logits.device)
            check(lib.cfun_edge_loss_fwd_save(ptr(probs), ptr(labels), ptr(out[1:]), ptr(dc), n, d, h, w, c, ptr(ws),
                                              ws.numel(), st), "edge_loss_fwd_save")
        else:
            check(lib.cfun_edge_loss_fwd(ptr(probs), ptr(labels), ptr(out[1:]), n, d, h, w, c, ptr(ws), ws.numel(), st),
                  "edge_loss_fwd")
        ctx.save_for_backward(probs, labels, dc)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_ce, g_edge):
        lib = _lib.load()
        probs, labels, dc = ctx.saved_tensors
        n, d, h, w, c = probs.shape
        g = torch.stack([g_ce.reshape(()).float(), g_edge.reshape(()).float()])
        dl = torch.empty_like(probs)
        if dc is not None:
            check(lib.cfun_mask_losses_bwd_saved(ptr(probs), ptr(labels), ptr(g), ptr(g[1:]), ptr(dc), ptr(dl), n, d, h,
                                                 w, c, stream(probs)), "mask_losses_bwd_saved")
        else:
            ws = workspace(lib.cfun_edge_loss_bwd_workspace_bytes(n, d, h, w, c), probs)
            check(lib.cfun_mask_losses_bwd(ptr(probs), ptr(labels), ptr(g), ptr(g[1:]), ptr(dl), n, d, h, w, c, ptr(ws),
                                           ws.numel(), stream(probs)), "mask_losses_bwd")
        return dl, None, None


def mask_losses(logits, probs, labels):
    """Both 'finetune' mask losses (model.py:909-981) of logits [n,D,H,W,C] with probs = softmax(logits):
    returns (cross entropy, Sobel edge loss); the backward is one fused pass (cfun_mask_losses_bwd)."""
    return _MaskLosses.apply(logits, probs.detach(), labels)


class _MaskLossesFused(torch.autograd.Function):
    """(CE, edge, probs) from the logits in ONE forward pass (cfun_mask_fused_fwd: softmax + cross entropy + Sobel edge loss,
    model.py:799 / 909-935 / 938-981) and ONE backward pass (cfun_mask_fused_bwd: a 2-D stencil per plane over the field the
    training forward leaves behind + the softmax backward).  ``probs`` is returned as a constant (the reference's
    Mask.forward output); the gradient of both losses through the softmax reaches ``logits`` in the backward."""

    @staticmethod
    def forward(ctx, logits, labels):
        # (undefined output gradients arrive as None: without this the engine materialises a ZERO tensor of the probabilities'
        # shape for the non-differentiable third output on every backward -- a 0.9 GB fill, 127 us on the step's critical path)
        ctx.set_materialize_grads(False)
        lib = _lib.load()
        logits, labels = _c(logits), _c(labels)
        n, d, h, w, c = logits.shape
        if labels.dtype != torch.uint8 or labels.numel() != n * d * h * w:
            raise RuntimeError("mask_losses_fused: labels must be uint8 [n,D,H,W]")
        probs = torch.empty_like(logits)
        out = torch.empty((2,), dtype=torch.float32, device=logits.device)
        ws = workspace(lib.cfun_mask_fused_workspace_bytes(), logits)
        u = None
        if ctx.needs_input_grad[0]:
            u = torch.empty(lib.cfun_mask_fused_u_bytes(n, d, h, w, c) // 4, dtype=torch.float32, device=logits.device)
        check(lib.cfun_mask_fused_fwd(ptr(logits), ptr(labels), ptr(probs), ptr(out), ptr(u), n, d, h, w, c, ptr(ws),
                                      ws.numel(), stream(logits)), "mask_fused_fwd")
        ctx.save_for_backward(probs, labels, u)
        ctx.mark_non_differentiable(probs)
        return out[0], out[1], probs

    @staticmethod
    def backward(ctx, g_ce, g_edge, _g_probs):
        lib = _lib.load()
        probs, labels, u = ctx.saved_tensors
        n, d, h, w, c = probs.shape
        if g_ce is None and g_edge is None:
            return None, None
        zero = torch.zeros((), dtype=torch.float32, device=probs.device)
        g = torch.stack([(zero if g_ce is None else g_ce.reshape(()).float()),
                         (zero if g_edge is None else g_edge.reshape(()).float())])
        dl = torch.empty_like(probs)
        check(lib.cfun_mask_fused_bwd(ptr(u), ptr(probs), ptr(labels), ptr(g), ptr(dl), n, d, h, w, c, stream(probs)),
              "mask_fused_bwd")
        return dl, None


def mask_losses_fused_supported(logits):
    n, d, h, w, c = logits.shape
    return bool(_lib.load().cfun_mask_fused_supported(n, d, h, w, c))


def mask_losses_fused(logits, labels):
    """Both 'finetune' mask losses AND the mask probabilities from the logits [n,D,H,W,C] in one pass each way: returns
    (cross entropy, Sobel edge loss, probs = softmax(logits)).  Shapes the fused kernels do not take (C not 8 / 3, an axis
    shorter than 3) go through the separate kernels."""
    if not mask_losses_fused_supported(logits):
        probs = softmax_channels(logits.detach())
        ce, edge = mask_losses(logits, probs, labels)
        return ce, edge, probs
    return _MaskLossesFused.apply(logits, labels)
