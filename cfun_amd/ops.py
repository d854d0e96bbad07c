"""torch.autograd bindings of the HIP kernels (libcfun_hip.so).

All activations are fp32 tensors shaped [N, D, H, W, C] and contiguous (NDHWC).  Weight packing
(OIDHW -> [tap][ci][CoP]) is a differentiable one-launch op (cfun_weight_pack / cfun_weight_unpack), so parameter
gradients arrive in the reference's OIDHW layout and shared weights (mask_branch.py:141/143 ...) are summed by
autograd.

This module holds the convolution / normalisation / buffer / RoI / resize bindings; the weight operands (pack, folds,
``WeightScope``) live in ``weights.py``, the mask-head losses in ``loss_ops.py``, the non-blocking host plumbing in
``hostio.py`` -- all re-exported here, so callers keep writing ``ops.<name>``.
"""
import ctypes as C
import os
from dataclasses import dataclass

import torch

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ALGO_AUTO, ConvParams, check, ptr, stream, workspace

LRELU_SLOPE = 0.01  # nn.LeakyReLU() default, mask_branch.py:18


from .weights import (  # noqa: E402,F401  (the weight operands live in weights.py; re-exported)
    _round16, _PackWeight, _pack, pack_weight, _FoldBias, fold_bias, _FoldBiasMany, fold_bias_many,
    _FoldUp2, fold_up2_weight, _SplitChannels, split_channels, _GatherSlices, gather_slices,
    materialize_weight, WeightScope, fold_up2_weight_eager, _transpose_pack)
from .hostio import _UploadRing, AsyncScalar, upload, side_stream, side_streams, _UPLOADERS, _SIDE_STREAMS  # noqa: E402,F401
from .loss_ops import (  # noqa: E402,F401
    _Softmax, softmax_channels, _MaskCE, _MaskCEWeighted, mask_cross_entropy, _EdgeRaw, edge_loss_raw, _EdgeLoss,
    edge_loss, _MaskLosses, mask_losses, _MaskLossesFused, mask_losses_fused, mask_losses_fused_supported)


@dataclass(frozen=True)
class ConvSpec:
    k: tuple            # (kd, kh, kw)
    co: int
    stride: int = 1
    pad: tuple = (0, 0, 0)
    up2: bool = False   # conv input = nearest x2 upsample of x
    act: int = ACT_NONE
    res_up2: bool = False
    scale_per_n: bool = False   # scale is [N, Co] (Dropout3d mask) instead of [Co]
    d2s: bool = False           # depth-to-space x2 epilogue (co = 8*CqP), see include/cfun_hip.h
    d2s_cq: int = 0             # valid channels per parity (0: co/8)
    tap_skip: bool = False      # parity-folded 'nearest x2 -> 3x3x3' weights: skip the folded-zero taps
    algo: int = ALGO_AUTO


def _out_dim(i, k, s, p):
    return (i + 2 * p - k) // s + 1


def _params(spec, x_shape, has_scale, has_shift, has_res):
    n, d, h, w, ci = x_shape
    sh = 1 if spec.up2 else 0
    p = ConvParams()
    p.N, p.Di, p.Hi, p.Wi, p.Ci = n, d, h, w, ci
    p.kd, p.kh, p.kw = spec.k
    p.stride = spec.stride
    p.pd, p.ph, p.pw = spec.pad
    p.Do = _out_dim(d << sh, p.kd, p.stride, p.pd)
    p.Ho = _out_dim(h << sh, p.kh, p.stride, p.ph)
    p.Wo = _out_dim(w << sh, p.kw, p.stride, p.pw)
    p.Co = spec.co
    p.CoP, p.CiP = _round16(spec.co), _round16(ci)
    p.up2 = int(spec.up2)
    p.act = spec.act
    p.slope = LRELU_SLOPE
    p.scale_mode = 0 if not has_scale else (2 if spec.scale_per_n else 1)
    p.has_shift = int(has_shift)
    p.res_mode = int(has_res)
    p.res_up2 = int(spec.res_up2 and has_res)
    p.d2s = int(spec.d2s)
    p.d2s_cq = int(spec.d2s_cq)
    p.tap_skip = int(spec.tap_skip)
    p.algo = spec.algo
    return p


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class LaunchTimer:
    """Times conv launches with HIP events on the launch stream (bench.py roofline legs).
    ``match(p)`` returns a key (or None) for a launch's CfunConv3dParams; durations are read per key after a
    synchronise with ``durations_ms(key)`` -- the forward under ``key``, the data- and weight-gradient calls of the same
    conv under ``key + "_dgrad"`` / ``key + "_wgrad"``."""

    def __init__(self, match):
        self.match = match
        self.events = {}

    def add(self, key, ev0, ev1):
        self.events.setdefault(key, []).append((ev0, ev1))

    def durations_ms(self, key):
        return [a.elapsed_time(b) for a, b in self.events.get(key, [])]


_TIMER = None
# The weight gradient of a conv is off the backward chain's critical path -- dx feeds the next layer, dw only the optimizer --
# so conv3d_w's backward enqueues it on a HIP stream of its own BEFORE the data gradient: it runs beside the chain's HBM- and
# latency-bound stretches (norm / activation gradients, the small launches of the deep levels): 42.7 -> 40.8 ms per step at
# cfg2 (round 5, same-box A/B).  CFUN_WGRAD_STREAM=0: everything on the chain's stream.
WGRAD_STREAM = os.environ.get("CFUN_WGRAD_STREAM", "1") == "1"


def wgrad_stream(device, chain_stream):
    """The weight-gradient stream that belongs to the stream a conv runs on (one per chain: the mask head's and the
    detector's weight gradients do not queue behind each other)."""
    return side_stream(device, "wgrad%d" % chain_stream.cuda_stream)


class _OnWgradStream(torch.autograd.Function):
    """Identity on a weight whose autograd NODE lives on the weight-gradient stream.  conv3d_w's backward produces dw on that
    stream; the engine attributes a gradient to the stream of the node that returned it (the conv's forward stream) and orders
    consumers only against that.  Routed through this node -- created under the weight-gradient stream, so the engine runs its
    backward there and records ITS event there -- dw reaches every consumer (AccumulateGrad, the gather / fold backward of a
    sliced or folded weight, a reducer hook) ordered behind the kernels that write it, with no hand-placed waits; the leaf's
    accumulation runs on the same stream and ``backward()`` joins it with the caller's stream at its end."""

    @staticmethod
    def forward(ctx, w):
        return w.view_as(w)

    @staticmethod
    def backward(ctx, g):
        return g


def _tag_wgrad_stream(w, x):
    if not (WGRAD_STREAM and x.is_cuda and w.requires_grad and torch.is_grad_enabled()):
        return w
    cur = torch.cuda.current_stream(x.device)
    with torch.cuda.stream(wgrad_stream(x.device, cur)):
        wj = _OnWgradStream.apply(w)
    desc = WeightScope._desc(w)
    if desc is not None:
        wj._cfun_src = desc
    lz = getattr(w, "_cfun_lazy", None)
    if lz is not None:
        wj._cfun_lazy = lz
    wj._cfun_wstream = True
    return wj


def set_launch_timer(timer):
    global _TIMER
    _TIMER = timer


class StatsSlot:
    """Receives InstanceNorm statistics (mean, rstd) [N,C,2] of a conv's OUTPUT from the epilogue that writes it
    (cfun_conv3d_fwd_fused, CfunConvFusion.out_stats) -- ``conv3d_w(..., stats=slot)`` fills it, ``instnorm_lrelu(y,
    stats=slot)`` then skips its own statistics pass over y.  Per-sample convs of one batch fill one row each:
    ``stats=(slot, i)``.  ``stats`` stays None when the kernel that ran has no statistics epilogue (the norm then runs
    its own pass): nothing downstream depends on whether the fusion happened."""

    def __init__(self, n=1, eps=1e-5):
        self.n, self.eps, self.stats, self.filled = n, eps, None, 0

    def row(self, i, c, like):
        if self.stats is None:
            self.stats = torch.empty((self.n, c, 2), dtype=torch.float32, device=like.device)
        if self.stats.shape[1] != c:
            raise RuntimeError("StatsSlot: %d channels, the slot holds %d" % (c, self.stats.shape[1]))
        return self.stats[i:i + 1]

    def get(self, n, c):
        """The finished [n,c,2] tensor, or None if not every row was filled by a fused epilogue."""
        if self.stats is None or self.filled != self.n or tuple(self.stats.shape) != (n, c, 2):
            return None
        return self.stats


class _Conv3d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, wp, scale, shift, res, spec, out=None, dx_slot=None, w_src=None, stats=None, pro=None,
                shift_scaled=False):
        # pro = (stats [N,Ci,2] or None, act, slope): the conv reads act((x - mean) * rstd) in place of x (NormedInput)
        lib = _lib.load()
        x = _c(x)
        wpT = None
        scale = None if scale is None else _c(scale)
        shift = None if shift is None else _c(shift)
        res = None if res is None else _c(res)
        p = _params(spec, x.shape, scale is not None, shift is not None, res is not None)
        if w_src is not None:      # OIDHW weight: packed here, its gradient comes back in OIDHW (one fused pass)
            scope = WeightScope.current()
            got = scope.lookup(w_src, p, ctx.needs_input_grad[0]) if scope is not None else None
            if got is not None:          # operands from the pass's batched preparation (the Winograd U etc., not plain packs)
                wp, wpT, p.w_prepared = got
            else:
                materialize_weight(w_src)
                if ctx.needs_input_grad[0]:      # the data gradient's layout in the same launch, kept for backward
                    wp, wpT = _pack(w_src, both=True)
                else:
                    wp = _pack(w_src)
        wp = _c(wp)
        if not (p.w_prepared & 1) and wp.shape != (p.kd * p.kh * p.kw, p.Ci, p.CoP):
            raise RuntimeError("packed weight %s does not match conv %s" % (tuple(wp.shape), spec))
        if spec.d2s:
            cq = spec.d2s_cq or p.Co // 8
            y = torch.empty((p.N, 2 * p.Do, 2 * p.Ho, 2 * p.Wo, cq), dtype=torch.float32, device=x.device)
        elif torch.is_tensor(out):   # write into a caller-provided dense region (a depth range of a larger buffer)
            y = out.view(out.shape)
            if tuple(y.shape) != (p.N, p.Do, p.Ho, p.Wo, p.Co) or not y.is_contiguous():
                raise RuntimeError("conv3d: out %s / contiguous=%s does not fit the result %s"
                                   % (tuple(y.shape), y.is_contiguous(), (p.N, p.Do, p.Ho, p.Wo, p.Co)))
        elif out is not None:        # write into sample `i` of a BatchBuffer (zero-copy batch join)
            y = out[0].sample(out[1], (p.N, p.Do, p.Ho, p.Wo, p.Co), x)
        else:
            y = torch.empty((p.N, p.Do, p.Ho, p.Wo, p.Co), dtype=torch.float32, device=x.device)
        timed = _TIMER.match(p) if (_TIMER is not None and x.is_cuda) else None
        if timed:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        if pro is not None or (stats is not None and (lib.cfun_conv3d_fused_support(C.byref(p)) & _lib.FUSE_OUT_STATS)):
            # InstanceNorm statistics of y from this epilogue (the norm that follows skips its pass over y) and / or the
            # norm + activation in front of this conv applied while x is staged
            dst = None
            if stats is not None and (lib.cfun_conv3d_fused_support(C.byref(p)) & _lib.FUSE_OUT_STATS):
                slot, i = stats if isinstance(stats, tuple) else (stats, None)
                if (i is None and slot.n != p.N) or (i is not None and p.N != 1):
                    raise RuntimeError("conv3d: stats slot of %d samples does not fit a conv over %d" % (slot.n, p.N))
                if i is None:
                    slot.stats = torch.empty((p.N, y.shape[-1], 2), dtype=torch.float32, device=x.device)
                    dst, slot.filled = slot.stats, slot.n
                else:
                    dst = slot.row(i, y.shape[-1], x)
                    slot.filled += 1
            fz = _lib.ConvFusion(None, 0, 0.0, None if dst is None else dst.data_ptr(), float(slot.eps) if dst is not None else 0.0)
            if pro is not None:
                pst = None if pro[0] is None else _c(pro[0])
                if pst is not None and tuple(pst.shape) != (p.N, p.Ci, 2):
                    raise RuntimeError("conv3d: input statistics %s do not fit x %s" % (tuple(pst.shape), tuple(x.shape)))
                fz.in_stats, fz.in_act, fz.in_slope = (None if pst is None else pst.data_ptr()), int(pro[1]), float(pro[2])
            ws = workspace(lib.cfun_conv3d_fwd_fused_workspace_bytes(C.byref(p), C.byref(fz)), x)
            check(lib.cfun_conv3d_fwd_fused(ptr(x), ptr(wp), ptr(scale), ptr(shift), ptr(res), ptr(y), C.byref(p),
                                            C.byref(fz), ptr(ws), ws.numel(), stream(x)), "conv3d_fwd_fused")
        else:
            ws = workspace(lib.cfun_conv3d_fwd_workspace_bytes(C.byref(p)), x)
            check(lib.cfun_conv3d_fwd(ptr(x), ptr(wp), ptr(scale), ptr(shift), ptr(res), ptr(y), C.byref(p), ptr(ws),
                                      ws.numel(), stream(x)), "conv3d_fwd")
        if timed:
            ev1.record()
            _TIMER.add(timed, ev0, ev1)
        ctx.spec = spec
        ctx.p = p
        ctx.res_shape = None if res is None else res.shape
        ctx.dx_slot = dx_slot
        ctx.wshape = None if w_src is None else tuple(w_src.shape)
        ctx.pro = None if pro is None else (int(pro[1]), float(pro[2]))
        ctx.shift_scaled = bool(shift_scaled)
        ctx.w_on_stream = bool(w_src is not None and getattr(w_src, "_cfun_wstream", False))
        ctx.save_for_backward(x, wp if not (p.w_prepared & 1) else None, scale, y if spec.act != ACT_NONE else None, wpT,
                              None if pro is None or pro[0] is None else _c(pro[0]))
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, wp, scale, y, wpT, pst = ctx.saved_tensors
        spec, p = ctx.spec, ctx.p
        fz = None
        if ctx.pro is not None:      # the weight gradient stages x through the same prologue as the forward did
            fz = _lib.ConvFusion(None if pst is None else pst.data_ptr(), ctx.pro[0], ctx.pro[1], None, 0.0)
        need_x, need_w, need_scale, need_shift, need_res = ctx.needs_input_grad[:5]
        need_wsrc = ctx.needs_input_grad[8]
        if need_scale:
            raise RuntimeError("cfun_amd conv3d: gradient w.r.t. the epilogue scale is not implemented "
                               "(BatchNorm is frozen on this path, model.py:1297-1304)")
        dy = _c(dy)
        st = stream(dy)
        nvox = p.N * p.Do * p.Ho * p.Wo
        # gp = dL/d(pre-activation); g = gp * scale = dL/d(conv sum)
        # shift_scaled (the shift is t + b * scale of a folded BatchNorm, ops.fold_bias(..., pre=True)): db = scale * sum(gp)
        # = sum(g), so the pre-activation gradient is never needed by itself (unless a residual wants it) and g comes
        # out of ONE pass: g = dy * act'(y) * scale
        one_pass = ctx.shift_scaled and scale is not None and not spec.d2s and not need_res
        gp = dy
        if one_pass:
            g = torch.empty_like(dy)
            check(lib.cfun_act_bwd(ptr(y) if spec.act != ACT_NONE else None, ptr(dy), ptr(scale), ptr(g), nvox, p.Co,
                                   p.Do * p.Ho * p.Wo, spec.act, LRELU_SLOPE, p.scale_mode, st), "act_bwd(act, scale)")
            gp = gp_out = g
        else:
            if spec.act != ACT_NONE:
                gp = torch.empty_like(dy)
                check(lib.cfun_act_bwd(ptr(y), ptr(dy), None, ptr(gp), dy.numel() // dy.shape[-1], dy.shape[-1],
                                       p.Do * p.Ho * p.Wo * (8 if spec.d2s else 1), spec.act, LRELU_SLOPE, 0, st), "act_bwd")
            gp_out = gp     # for d2s convs the kernels take the gradient in y's hi-res layout and gather the parities
            g = gp
            if scale is not None:
                g = torch.empty_like(dy)
                check(lib.cfun_act_bwd(None, ptr(gp), ptr(scale), ptr(g), nvox, p.Co, p.Do * p.Ho * p.Wo, ACT_NONE,
                                       LRELU_SLOPE, p.scale_mode, st), "act_bwd(scale)")
        def run_wgrad(st):
            dw = torch.empty(ctx.wshape, dtype=torch.float32, device=dy.device)
            if fz is not None:
                ws = workspace(lib.cfun_conv3d_bwd_weight_workspace_bytes(C.byref(p)), x)
                check(lib.cfun_conv3d_bwd_weight_fused(ptr(x), ptr(g), ptr(dw), 1, C.byref(p), C.byref(fz), ptr(ws),
                                                       ws.numel(), st), "conv3d_bwd_weight_fused(oidhw)")
            else:
                nb = lib.cfun_conv3d_bwd_weight_workspace_bytes(C.byref(p))
                ws = workspace(nb, x)
                tk = _TIMER.match(p) if (_TIMER is not None and x.is_cuda) else None
                if tk:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                check(lib.cfun_conv3d_bwd_weight_oidhw(ptr(x), ptr(g), ptr(dw), C.byref(p), ptr(ws), ws.numel(), st),
                      "conv3d_bwd_weight_oidhw")
                if tk:
                    e1.record()
                    _TIMER.add(tk + "_wgrad", e0, e1)
            return dw
        dx = dwp = dshift = dres = dw = None
        # the weight gradient on its own stream, enqueued BEFORE the data gradient (see WGRAD_STREAM / _OnWgradStream)
        side_w = None
        if need_wsrc and ctx.w_on_stream:
            cur_s = torch.cuda.current_stream(dy.device)
            side_w = wgrad_stream(dy.device, cur_s)
            side_w.wait_stream(cur_s)            # g (and x) are complete
            with torch.cuda.stream(side_w):
                dw = run_wgrad(side_w.cuda_stream)
            for t in (x, g, pst):
                if t is not None:
                    t.record_stream(side_w)
        if need_x:
            # dx of a per-sample conv goes straight into its sample of the batch's gradient (zero-copy batch split)
            dx = torch.empty_like(x) if ctx.dx_slot is None else ctx.dx_slot[0].sample(ctx.dx_slot[1], x.shape, x)
            if wpT is None:
                if wp is None:
                    raise RuntimeError("conv3d backward: no weight operand was kept for the data gradient")
                wpT = _transpose_pack(wp, p.Co)
            nb = lib.cfun_conv3d_bwd_data_workspace_bytes(C.byref(p))
            ws = workspace(nb, x)
            tk = _TIMER.match(p) if (_TIMER is not None and x.is_cuda) else None
            if tk:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            check(lib.cfun_conv3d_bwd_data(ptr(g), ptr(wpT), ptr(dx), C.byref(p), ptr(ws), ws.numel(), st),
                  "conv3d_bwd_data")
            if tk:
                e1.record()
                _TIMER.add(tk + "_dgrad", e0, e1)
        if need_w:
            dwp = torch.empty_like(wp)
            nb = lib.cfun_conv3d_bwd_weight_workspace_bytes(C.byref(p))
            ws = workspace(nb, x)
            if fz is not None:
                check(lib.cfun_conv3d_bwd_weight_fused(ptr(x), ptr(g), ptr(dwp), 0, C.byref(p), C.byref(fz), ptr(ws),
                                                       ws.numel(), st), "conv3d_bwd_weight_fused")
            else:
                check(lib.cfun_conv3d_bwd_weight(ptr(x), ptr(g), ptr(dwp), C.byref(p), ptr(ws), ws.numel(), st),
                      "conv3d_bwd_weight")
        if need_wsrc and side_w is None:
            dw = run_wgrad(st)
        if need_shift:      # (shift_scaled: db = sum(g) whichever way g was formed)
            dshift = channel_sum((g if ctx.shift_scaled and scale is not None else gp).view(-1, p.Co))
        if need_res:
            if spec.d2s:
                dres = torch.empty(ctx.res_shape, dtype=torch.float32, device=dy.device)
                check(lib.cfun_upsample2_bwd(ptr(gp_out), ptr(dres), p.N, p.Do, p.Ho, p.Wo, gp_out.shape[-1], st),
                      "upsample2_bwd")
            elif p.res_up2:
                dres = torch.empty(ctx.res_shape, dtype=torch.float32, device=dy.device)
                check(lib.cfun_upsample2_bwd(ptr(gp), ptr(dres), p.N, p.Do // 2, p.Ho // 2, p.Wo // 2, p.Co, st),
                      "upsample2_bwd")
            else:
                dres = gp
        return dx, dwp, None, dshift, dres, None, None, None, dw, None, None, None


class NormedInput:
    """A tensor that exists only as "act((raw - mean) * rstd)": the InstanceNorm3d + LeakyReLU (or the plain LeakyReLU)
    between a producer and the conv that consumes it (mask_branch.py:23-25,91-116), not written to memory.  ``token``
    aliases the RAW tensor and carries the norm's backward in the autograd graph (``instnorm_lrelu(..., lazy=True)``,
    ``lrelu(..., lazy=True)``); a conv whose kernel has the input prologue (cfun_conv3d_fused_support) stages the raw
    values through it, every other consumer gets ``materialize()`` -- the stand-alone apply pass, run at most once.
    Either way the gradient reaches the producer through the token's one backward node."""

    def __init__(self, token, stats, act, slope):
        self.token, self.stats, self.act, self.slope = token, stats, act, slope
        self._mat = None

    @property
    def shape(self):
        return self.token.shape

    @property
    def device(self):
        return self.token.device

    def pro(self):
        return (self.stats, self.act, self.slope)

    def materialize(self):
        if self._mat is None:
            self._mat = _Materialize.apply(self.token, self.stats, self.act, self.slope)
        return self._mat

    def samples(self):
        """Per-sample NormedInputs [1,...] (zero-copy split, gradients rejoin without a copy: split_batch)."""
        parts = split_batch(self.token)
        return [NormedInput(t, None if self.stats is None else self.stats[i:i + 1], self.act, self.slope)
                for i, t in enumerate(parts)]


class _Materialize(torch.autograd.Function):
    """token -> the normalised + activated tensor (the stand-alone apply pass); the gradient passes through unchanged to
    the token, whose backward node is the norm's."""

    @staticmethod
    def forward(ctx, token, stats, act, slope):
        lib = _lib.load()
        x = _c(token)
        y = torch.empty_like(x)
        if stats is not None:
            if act != ACT_LRELU:
                raise RuntimeError("materialize: InstanceNorm is fused with LeakyReLU only")
            n, c = x.shape[0], x.shape[-1]
            check(lib.cfun_instnorm_lrelu_fwd(ptr(x), ptr(_c(stats)), ptr(y), n, x.numel() // (n * c), c, slope, stream(x)),
                  "instnorm_lrelu_fwd")
        else:
            check(lib.cfun_lrelu_fwd(ptr(x), ptr(y), x.numel(), slope, stream(x)), "lrelu_fwd")
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, None, None, None


def _fusable_input(x, spec, scale, shift, res, need_wgrad):
    """Can the conv (spec) stage the NormedInput x through its prologue -- forward kernel and, when the weight needs a
    gradient, the weight-gradient kernel?"""
    if spec.up2:
        return False
    lib = _lib.load()
    p = _params(spec, x.shape, scale is not None, shift is not None, res is not None)
    have = lib.cfun_conv3d_fused_support(C.byref(p))
    want = _lib.FUSE_IN_NORM | (_lib.FUSE_IN_NORM_WGRAD if need_wgrad else 0)
    return (have & want) == want


def conv3d(x, wp, spec, scale=None, shift=None, res=None, out=None, dx_slot=None, stats=None):
    """y = act(scale * conv(x) + shift + res); see include/cfun_hip.h (cfun_conv3d_fwd).  ``out`` / ``dx_slot``:
    (BatchBuffer, i) -- write y / the input gradient into sample i of a shared batch buffer (per-sample convs).
    ``stats``: a ``StatsSlot`` (or (slot, i) for per-sample convs) that receives y's InstanceNorm statistics.
    x may be a ``NormedInput``."""
    pro = None
    if isinstance(x, NormedInput):
        if _fusable_input(x, spec, scale, shift, res, wp.requires_grad):
            x, pro = x.token, x.pro()
        else:
            x = x.materialize()
    return _Conv3d.apply(x, wp, scale, shift, res, spec, out, dx_slot, None, stats, pro)


def conv3d_w(x, w, spec, scale=None, shift=None, res=None, out=None, dx_slot=None, stats=None, shift_scaled=False):
    """``conv3d`` on an OIDHW weight [Co,Ci,kd,kh,kw] (a parameter, a gathered slice of one, a folded up-conv
    weight): packed inside the op, and the weight gradient is produced directly in OIDHW -- the reduction of the
    wgrad kernel's per-chunk partial sums and the un-packing are one kernel (cfun_conv3d_bwd_weight_oidhw) instead
    of conv3d(x, pack_weight(w))'s reduce + un-pack launches.  Same values bit for bit."""
    pro = None
    if isinstance(x, NormedInput):
        if _fusable_input(x, spec, scale, shift, res, w.requires_grad):
            x, pro = x.token, x.pro()
        else:
            x = x.materialize()
    return _Conv3d.apply(x, None, scale, shift, res, spec, out, dx_slot, _tag_wgrad_stream(w, x), stats, pro, shift_scaled)


# ---- zero-copy batch split / join (per-sample convs inside a batched graph) ------------------------------------
class BatchBuffer:
    """[N, ...] storage allocated on first use whose samples are written in place by per-sample ops."""

    def __init__(self, n):
        self.n, self.data = n, None

    def sample(self, i, shape1, like):
        if self.data is None:
            self.data = torch.empty((self.n,) + tuple(shape1[1:]), dtype=torch.float32, device=like.device)
        if tuple(self.data.shape[1:]) != tuple(shape1[1:]) or shape1[0] != 1:
            raise RuntimeError("BatchBuffer: sample shape %s does not fit %s" % (tuple(shape1), tuple(self.data.shape)))
        return self.data[i:i + 1]


def _consecutive_samples(parts):
    """The [N,...] tensor whose samples the contiguous [1,...] tensors ``parts`` are, if they lie back to back in one
    storage (then no copy is needed to stack them); else None."""
    p0 = parts[0]
    if p0 is None:
        return None
    step = p0.numel()
    for i, t in enumerate(parts):
        if (t is None or not t.is_contiguous() or t.shape != p0.shape or t.dtype != p0.dtype
                or t.untyped_storage().data_ptr() != p0.untyped_storage().data_ptr()
                or t.storage_offset() != p0.storage_offset() + i * step):
            return None
    return torch.as_strided(p0, (len(parts),) + tuple(p0.shape[1:]), (step,) + tuple(p0.stride()[1:]), p0.storage_offset())


class _SplitBatch(torch.autograd.Function):
    """x [N,...] -> N views [1,...].  Backward: if the N gradients already are the samples of one buffer (the per-sample
    convs wrote them there, ``dx_slot``; or they are slices of one upstream gradient) it is returned as is -- the
    plain ``unbind`` would stack them (a full-size copy), ``x[i:i+1]`` would zero-fill N full-size tensors."""

    @staticmethod
    def forward(ctx, x):
        return tuple(x.narrow(0, i, 1) for i in range(x.shape[0]))

    @staticmethod
    def backward(ctx, *grads):
        full = _consecutive_samples(grads)
        if full is not None:
            return full
        ref = next(g for g in grads if g is not None)
        return torch.cat([g if g is not None else torch.zeros_like(ref) for g in grads], dim=0)


def split_batch(x):
    return _SplitBatch.apply(x)


class _JoinBatch(torch.autograd.Function):
    """The per-sample results written into ``buf`` (conv3d(out=(buf, i))) as one [N,...] tensor, without a copy."""

    @staticmethod
    def forward(ctx, buf, *parts):
        if _consecutive_samples(parts) is None:
            raise RuntimeError("join_batch: the parts are not the samples of the BatchBuffer")
        return buf.data.view(buf.data.shape)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        return (None,) + tuple(g[i:i + 1] for i in range(g.shape[0]))


def join_batch(buf, parts):
    return _JoinBatch.apply(buf, *parts)


class _JoinDepth(torch.autograd.Function):
    """Results that convs wrote into consecutive depth ranges of ``buf`` [1,D,...] (``conv3d(out=buf[:, a:b])``) as one
    tensor, without a copy; the backward hands each producer its depth range of the gradient (dense views: N == 1)."""

    @staticmethod
    def forward(ctx, buf, *parts):
        ctx.sizes = [p.shape[1] for p in parts]
        if buf.shape[0] != 1 or sum(ctx.sizes) != buf.shape[1]:
            raise RuntimeError("join_depth: the parts do not tile the buffer")
        return buf.view(buf.shape)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        outs, z = [], 0
        for n in ctx.sizes:
            outs.append(g[:, z:z + n])
            z += n
        return (None,) + tuple(outs)


def join_depth(buf, parts):
    return _JoinDepth.apply(buf, *parts)


def channel_sum(g2d):
    """[V, C] -> [C] sum over rows (bias gradients)."""
    lib = _lib.load()
    g2d = _c(g2d)
    v, c = g2d.shape
    out = torch.empty((c,), dtype=torch.float32, device=g2d.device)
    ws = workspace(lib.cfun_channel_sum_workspace_bytes(v, c), g2d)
    check(lib.cfun_channel_sum(ptr(g2d), ptr(out), v, c, ptr(ws), ws.numel(), stream(g2d)), "channel_sum")
    return out


# ---- classifier head GEMMs (fc.hip) --------------------------------------------------------------------------------
class _FC(torch.autograd.Function):
    """y = act(scale * (x . w^T) + shift): x [R,K], w [O,K] (a Conv3d weight whose kernel covers its whole input, a
    1x1x1 conv or an nn.Linear weight, viewed 2-D), scale / shift [O] or None.  The weight streams through
    cfun_fc_fwd / cfun_fc_bwd_* in its own layout; gradients for x, w and shift (scale is a frozen-BN fold)."""

    @staticmethod
    def forward(ctx, x, w, scale, shift, act):
        lib = _lib.load()
        x, w = _c(x), _c(w)
        scale = None if scale is None else _c(scale)
        shift = None if shift is None else _c(shift)
        r, k = x.shape
        o = w.shape[0]
        if w.shape[1] != k:
            raise RuntimeError("fc: x %s does not match w %s" % (tuple(x.shape), tuple(w.shape)))
        if r > 64:
            raise RuntimeError("fc: at most 64 rows per call (got %d)" % r)
        y = torch.empty((r, o), dtype=torch.float32, device=x.device)
        ws = workspace(lib.cfun_fc_workspace_bytes(r, k, o), x)
        check(lib.cfun_fc_fwd(ptr(x), ptr(w), ptr(scale), ptr(shift), ptr(y), r, k, o, act, ptr(ws), ws.numel(), stream(x)),
              "fc_fwd")
        ctx.act = act
        ctx.save_for_backward(x, w, scale, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w, scale, y = ctx.saved_tensors
        need_x, need_w, need_scale, need_shift = ctx.needs_input_grad[:4]
        if need_scale:
            raise RuntimeError("cfun_amd fc: gradient w.r.t. the epilogue scale is not implemented (frozen BatchNorm)")
        r, k = x.shape
        o = w.shape[0]
        dy = _c(dy)
        st = stream(dy)
        gp = dy
        if ctx.act != ACT_NONE:
            gp = torch.empty_like(dy)
            check(lib.cfun_act_bwd(ptr(y), ptr(dy), None, ptr(gp), r, o, 1, ctx.act, LRELU_SLOPE, 0, st), "act_bwd")
        g = gp
        if scale is not None:
            g = torch.empty_like(dy)
            check(lib.cfun_act_bwd(None, ptr(gp), ptr(scale), ptr(g), r, o, 1, ACT_NONE, LRELU_SLOPE, 1, st), "act_bwd(scale)")
        dx = dw = dshift = None
        if need_x:
            dx = torch.empty_like(x)
            check(lib.cfun_fc_bwd_data(ptr(g), ptr(w), ptr(dx), r, k, o, st), "fc_bwd_data")
        if need_w:
            dw = torch.empty_like(w)
            rmax = max(1, int(lib.cfun_fc_bwd_weight_max_rows(o)))      # rows whose g + x slice fit the kernel's LDS
            if r <= rmax:
                check(lib.cfun_fc_bwd_weight(ptr(x), ptr(g), ptr(dw), r, k, o, st), "fc_bwd_weight")
            else:                       # (wide FC layers at large R: row chunks, partial gradients added)
                part = torch.empty_like(w)
                for i, r0 in enumerate(range(0, r, rmax)):
                    xs, gs = _c(x[r0:r0 + rmax]), _c(g[r0:r0 + rmax])
                    check(lib.cfun_fc_bwd_weight(ptr(xs), ptr(gs), ptr(dw if i == 0 else part), xs.shape[0], k, o, st),
                          "fc_bwd_weight")
                    if i:
                        dw += part
        if need_shift:
            dshift = channel_sum(gp) if r else torch.zeros((o,), dtype=torch.float32, device=dy.device)
        return dx, dw, None, dshift, None


def fc(x, w, scale=None, shift=None, act=ACT_NONE):
    """act(scale * (x [R,K] . w [O,K]^T) + shift) on the weight-streaming GEMM kernels (classifier head, model.py:750-784)."""
    if x.shape[0] == 0:
        return x.new_zeros((0, w.shape[0])) + 0.0 * w.sum()
    return _FC.apply(x, w, scale, shift, act)


# ---- zero-copy channel concatenation ------------------------------------------------------------------------------
class ConcatBuffer:
    """A [N,D,H,W,C_total] NDHWC buffer whose channel ranges are filled IN PLACE by the ops that produce the operands
    of a ``torch.cat(..., dim=-1)`` (``lrelu(x, out=buf.slot(c0, c1))``, ``instnorm_lrelu(x, out=...)``) and handed to
    the consumer by ``join()`` without a copy; the backward hands the gradient's channel ranges back as strided views
    that the strided backward kernels read in place.  All ranges and C_total must be multiples of 4."""

    def __init__(self, like, c_total):
        self.data = torch.empty(tuple(like.shape[:-1]) + (c_total,), dtype=torch.float32, device=like.device)
        self.c_total = c_total

    def slot(self, c0, c1):
        if c0 % 4 or c1 % 4 or self.c_total % 4:
            raise ValueError("ConcatBuffer slots must be multiples of 4 channels")
        return (self, c0, c1)

    def join(self, *parts):
        """parts: the tensors returned by the ops that filled the slots, in channel order."""
        return _JoinChannels.apply(self, *parts)


def _slot_view(out):
    buf, c0, c1 = out
    return buf.data[..., c0:c1]


def _row_stride(t):
    """Channel-row stride (floats) of an NDHWC tensor that is dense or a channel slice of a dense buffer; None if the
    layout is anything else (then the caller makes it contiguous)."""
    c = t.shape[-1]
    if t.is_contiguous():
        return c
    st, shp = t.stride(), t.shape
    if st[-1] != 1:
        return None
    rs = st[-2]
    exp = rs
    for d in range(t.dim() - 2, -1, -1):          # every outer stride must be the dense one for row stride rs
        if st[d] != exp and shp[d] != 1:
            return None
        exp *= shp[d]
    return rs if rs % 4 == 0 and c % 4 == 0 and t.data_ptr() % 16 == 0 else None


def ptr_raw(t):
    """Device pointer of a tensor that may be a channel slice (checked by the caller with _row_stride)."""
    if not t.is_cuda and not _lib.is_emulator():
        raise RuntimeError("cfun_amd: CPU tensor passed to a HIP kernel (there is no CPU fallback)")
    return t.data_ptr()


class _JoinChannels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, buf, *parts):
        ctx.widths = [p.shape[-1] for p in parts]
        if sum(ctx.widths) != buf.c_total:
            raise RuntimeError("ConcatBuffer.join: the parts do not cover the buffer")
        return buf.data.view(buf.data.shape)        # a fresh alias of the filled buffer: no copy

    @staticmethod
    def backward(ctx, g):
        outs, c0 = [], 0
        for w in ctx.widths:
            outs.append(g[..., c0:c0 + w])            # strided views; the strided backward kernels read them in place
            c0 += w
        return (None,) + tuple(outs)


def _combine_stats_over_ranks(stats, eps, shard):
    """(mean, rstd) of every rank's depth slab [n,c,2] -> the statistics of the whole volume, identical on every rank of
    ``shard`` (equal slabs): E[x] and E[x^2] are averaged with ONE all-reduce of 2*C values per sample (SURVEY.md section
    8(e): "all_reduce of 2*C floats per InstanceNorm"); fp64 on these few values keeps the variance cancellation exact."""
    import torch.distributed as dist
    m = stats[..., 0].double()
    ex2 = (1.0 / stats[..., 1].double() ** 2 - eps).clamp_(min=0.0) + m * m
    both = torch.stack([m, ex2], dim=-1)
    dist.all_reduce(both, group=shard.group)
    both /= shard.world
    mean = both[..., 0]
    var = (both[..., 1] - mean * mean).clamp_(min=0.0)
    return torch.stack([mean, torch.rsqrt(var + eps)], dim=-1).float().contiguous()


class _InstNormLReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps, out=None, shard=None, pre=None, lazy=None, passthrough=False):
        # lazy: a list that receives the statistics -- no apply pass, the result aliases x (see NormedInput)
        # passthrough: also return an alias of x for x's OTHER consumer (the residual add that follows, mask_branch.py:
        # 131-176): its gradient arrives here as a second argument and is summed inside the backward kernel
        ctx.set_materialize_grads(False)      # (an unused output's gradient arrives as None, not as a zero tensor)
        lib = _lib.load()
        x = _c(x)
        n, c = x.shape[0], x.shape[-1]
        v = x.numel() // (n * c)
        zs = shard is not None and shard.world > 1       # x is this rank's depth slab of a z-sharded volume
        if v * (shard.world if zs else 1) <= 1:
            raise ValueError("Expected more than 1 spatial element when training, got input size %s"
                             % (tuple(x.shape),))  # InstanceNorm3d behaviour, SURVEY.md App. A-3
        st = stream(x)
        stats = pre.get(n, c) if (pre is not None and pre.eps == eps) else None      # from the producer conv's epilogue
        if stats is None:
            stats = torch.empty((n, c, 2), dtype=torch.float32, device=x.device)
            ws = workspace(lib.cfun_instnorm_workspace_bytes(n, v, c), x)
            check(lib.cfun_instnorm_stats(ptr(x), ptr(stats), n, v, c, eps, ptr(ws), ws.numel(), st), "instnorm_stats")
        if zs:
            stats = _combine_stats_over_ranks(stats, eps, shard)
        if lazy is not None:
            if out is not None:
                raise RuntimeError("instnorm_lrelu: lazy and out= are exclusive")
            lazy.append(stats)
            y = x.view(x.shape)
        elif out is None:
            y = torch.empty_like(x)
            check(lib.cfun_instnorm_lrelu_fwd(ptr(x), ptr(stats), ptr(y), n, v, c, LRELU_SLOPE, st), "instnorm_lrelu_fwd")
        else:                                        # write into a channel range of a ConcatBuffer
            y = _slot_view(out)
            if y.shape != x.shape:
                raise RuntimeError("instnorm_lrelu: out slot %s does not match %s" % (tuple(y.shape), tuple(x.shape)))
            check(lib.cfun_instnorm_lrelu_fwd_strided(ptr(x), ptr(stats), ptr_raw(y), n, v, c, out[0].c_total,
                                                      LRELU_SLOPE, st), "instnorm_lrelu_fwd_strided")
        ctx.shard = shard if zs else None
        ctx.save_for_backward(x, stats)
        if passthrough:
            return y, x.view(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy, dx_other=None):
        lib = _lib.load()
        x, stats = ctx.saved_tensors
        if dy is None:                      # only the passthrough branch was used
            return dx_other, None, None, None, None, None, None
        n, c = x.shape[0], x.shape[-1]
        v = x.numel() // (n * c)
        rs = _row_stride(dy)
        if rs is None:
            dy, rs = dy.contiguous(), c
        dx = torch.empty_like(x)
        ws = workspace(lib.cfun_instnorm_workspace_bytes(n, v, c), x)
        if ctx.shard is None:
            add = None if dx_other is None else _c(dx_other)
            check(lib.cfun_instnorm_lrelu_bwd_add(ptr(x), ptr(stats), ptr_raw(dy), ptr(add), ptr(dx), n, v, c, rs, LRELU_SLOPE,
                                                  ptr(ws), ws.numel(), stream(x)), "instnorm_lrelu_bwd")
            dx_other = None
        else:       # z-sharded volume: the two means are over ALL slabs -- one all-reduce between the two kernel halves
            import torch.distributed as dist
            means = torch.empty((n, c, 2), dtype=torch.float32, device=x.device)
            check(lib.cfun_instnorm_bwd_means(ptr(x), ptr(stats), ptr_raw(dy), ptr(means), n, v, c, rs, LRELU_SLOPE, ptr(ws),
                                              ws.numel(), stream(x)), "instnorm_bwd_means")
            dist.all_reduce(means, group=ctx.shard.group)
            means /= ctx.shard.world
            check(lib.cfun_instnorm_lrelu_bwd_apply(ptr(x), ptr(stats), ptr(means), ptr_raw(dy), ptr(dx), n, v, c, rs,
                                                    LRELU_SLOPE, stream(x)), "instnorm_lrelu_bwd_apply")
        if dx_other is not None:
            dx = dx + dx_other
        return dx, None, None, None, None, None, None


def instnorm_lrelu(x, eps=1e-5, out=None, shard=None, stats=None, lazy=False, passthrough=False):
    """LeakyReLU(InstanceNorm3d(x)) (affine=False, biased variance), mask_branch.py:28-116.  ``out``: a
    ``ConcatBuffer.slot`` to write the result into (zero-copy concat).  ``shard``: x is this rank's equal depth slab of
    a volume z-sharded over ``shard``'s ranks -- the statistics (forward) and the two gradient means (backward) are
    combined with one small all-reduce each.  ``stats``: a ``StatsSlot`` the conv that produced x filled from its
    epilogue (this rank's slab for a sharded x) -- the statistics pass over x is then skipped.  ``lazy``: return a
    ``NormedInput`` instead of running the apply pass -- the consumer convs stage x through the norm themselves.
    ``passthrough``: return (result, x') where x' aliases x and stands for it at x's other consumer (a residual add):
    the two gradients of x are then summed inside the norm's backward kernel instead of by a pass of autograd's own."""
    if not lazy:
        return _InstNormLReLU.apply(x, eps, out, shard, stats, None, passthrough)
    got = []
    r = _InstNormLReLU.apply(x, eps, None, shard, stats, got, passthrough)
    if passthrough:
        return NormedInput(r[0], got[0], ACT_LRELU, LRELU_SLOPE), r[1]
    return NormedInput(r, got[0], ACT_LRELU, LRELU_SLOPE)


class _LReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out=None, lazy=False, passthrough=False):
        ctx.set_materialize_grads(False)
        lib = _lib.load()
        x = _c(x)
        if lazy:                      # no pass: the result aliases x, the consumer conv applies the activation (NormedInput)
            y = x.view(x.shape)
        elif out is None:
            y = torch.empty_like(x)
            check(lib.cfun_lrelu_fwd(ptr(x), ptr(y), x.numel(), LRELU_SLOPE, stream(x)), "lrelu_fwd")
        else:
            y = _slot_view(out)
            c = x.shape[-1]
            if y.shape != x.shape:
                raise RuntimeError("lrelu: out slot %s does not match %s" % (tuple(y.shape), tuple(x.shape)))
            check(lib.cfun_lrelu_fwd_strided(ptr(x), ptr_raw(y), x.numel() // c, c, c, out[0].c_total, LRELU_SLOPE,
                                             stream(x)), "lrelu_fwd_strided")
        ctx.save_for_backward(x)
        if passthrough:               # (see _InstNormLReLU: x' for x's other consumer, gradients summed in the backward kernel)
            return y, x.view(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy, dx_other=None):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        if dy is None:
            return dx_other, None, None, None
        c = x.shape[-1]
        rs = _row_stride(dy)
        dx = torch.empty_like(x)
        if dx_other is not None and c % 4 == 0 and x.data_ptr() % 16 == 0:
            if rs is None:
                dy, rs = _c(dy), c
            add = _c(dx_other)
            check(lib.cfun_lrelu_bwd_add(ptr(x), ptr_raw(dy), ptr(add), ptr(dx), x.numel() // c, c, rs, LRELU_SLOPE, stream(x)),
                  "lrelu_bwd_add")
            return dx, None, None, None
        if rs is None or rs == c or c % 4:          # dense (or odd) gradient: the flat kernel
            dy = _c(dy)
            check(lib.cfun_lrelu_bwd(ptr(x), ptr(dy), ptr(dx), x.numel(), LRELU_SLOPE, stream(x)), "lrelu_bwd")
        else:
            check(lib.cfun_lrelu_bwd_strided(ptr(x), ptr_raw(dy), ptr(dx), x.numel() // c, c, rs, LRELU_SLOPE,
                                             stream(x)), "lrelu_bwd_strided")
        if dx_other is not None:
            dx = dx + dx_other
        return dx, None, None, None


def lrelu(x, out=None, lazy=False, passthrough=False):
    """LeakyReLU (mask_branch.py:18).  out / lazy / passthrough: see ``instnorm_lrelu``."""
    r = _LReLU.apply(x, None if lazy else out, lazy, passthrough)
    if not lazy:
        return r
    if passthrough:
        return NormedInput(r[0], None, ACT_LRELU, LRELU_SLOPE), r[1]
    return NormedInput(r, None, ACT_LRELU, LRELU_SLOPE)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        lib = _lib.load()
        a, b = _c(a), _c(b)
        out = torch.empty_like(a)
        check(lib.cfun_add(ptr(a), ptr(b), ptr(out), a.numel(), stream(a)), "add")
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    if a.shape != b.shape:
        raise RuntimeError("add: shape mismatch %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    return _Add.apply(a, b)


class _MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _c(x)
        n, d, h, w, c = x.shape
        do, ho, wo = d // 2, h // 2, w // 2
        if (d | h | w) & 1:
            raise RuntimeError("maxpool2: odd input size %s" % (tuple(x.shape),))
        y = torch.empty((n, do, ho, wo, c), dtype=torch.float32, device=x.device)
        idx = torch.empty((n, do, ho, wo, c), dtype=torch.uint8, device=x.device)
        check(lib.cfun_maxpool2_fwd(ptr(x), ptr(y), ptr(idx), n, do, ho, wo, c, stream(x)), "maxpool2_fwd")
        ctx.save_for_backward(idx)
        ctx.mark_non_differentiable(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        dy = _c(dy)
        n, do, ho, wo, c = dy.shape
        dx = torch.empty((n, 2 * do, 2 * ho, 2 * wo, c), dtype=torch.float32, device=dy.device)
        check(lib.cfun_maxpool2_bwd(ptr(dy), ptr(idx), ptr(dx), n, do, ho, wo, c, stream(dy)), "maxpool2_bwd")
        return dx


def maxpool2(x):
    """MaxPool3d(kernel_size=2, stride=2), backbone.py:127."""
    return _MaxPool2.apply(x)


class _RoIAlign(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fm, boxes, pool, slab=None):
        # slab = (z0, D): fm holds depth planes [z0, z0 + fm.shape[0]) of a [D,H,W,C] map (see roi_align)
        lib = _lib.load()
        fm = _c(fm)
        boxes = _c(boxes.detach().float())
        dl, h, w, c = fm.shape
        z0, d = (0, dl) if slab is None else (int(slab[0]), int(slab[1]))
        r = boxes.shape[0]
        pd, ph, pw = [int(v) for v in pool]
        out = torch.empty((r, pd, ph, pw, c), dtype=torch.float32, device=fm.device)
        bounds = torch.empty((max(r, 1), 6), dtype=torch.int32, device=fm.device)
        check(lib.cfun_roi_align3d_slab_fwd(ptr(fm), ptr(boxes), ptr(out), ptr(bounds), r, d, h, w, c, z0, dl, pd, ph, pw,
                                            stream(fm)), "roi_align3d_fwd")
        ctx.save_for_backward(bounds)
        ctx.dims = (r, d, h, w, c, pd, ph, pw, z0, dl)
        ctx.mark_non_differentiable(bounds)
        return out, bounds

    @staticmethod
    def backward(ctx, dout, _dbounds):
        lib = _lib.load()
        (bounds,) = ctx.saved_tensors
        r, d, h, w, c, pd, ph, pw, z0, dl = ctx.dims
        dout = _c(dout)
        dfm = torch.empty((dl, h, w, c), dtype=torch.float32, device=dout.device)      # (every element is written)
        check(lib.cfun_roi_align3d_slab_bwd(ptr(dout), ptr(bounds), ptr(dfm), r, d, h, w, c, z0, dl, pd, ph, pw,
                                            stream(dout)), "roi_align3d_bwd")
        return dfm, None, None, None


def roi_align(fm, boxes, pool, slab=None):
    """fm [D,H,W,C], boxes [R,6] normalised -> ([R,pd,ph,pw,C], int32 crop bounds [R,6]).  ``slab`` = (z0, D): fm is the
    depth slab [z0, z0 + fm.shape[0]) of a [D,H,W,C] map and the result is this slab's additive share of the crops
    (planes outside count as zeros; RoIAlign is linear in the map)."""
    return _RoIAlign.apply(fm, boxes, tuple(pool), slab)


def mask_target_labels(labels, rois, mask_shape):
    """GT mask targets of model.py:481-493 as uint8 labels: labels [D,H,W] uint8, rois [R,6] normalised ->
    [R, *mask_shape] uint8 (crop with int(shape*coord) truncation in fp32, nearest resize)."""
    lib = _lib.load()
    labels = _c(labels)
    if labels.dtype != torch.uint8 or labels.dim() != 3:
        raise RuntimeError("mask_target_labels: labels must be uint8 [D,H,W]")
    d, h, w = labels.shape
    scale = torch.tensor([d, h, w, d, h, w], dtype=torch.float32, device=rois.device)
    bounds = _c((rois.detach().float() * scale).to(torch.int32))          # fp32 product, truncation toward zero
    r = bounds.shape[0]
    md, mh, mw = [int(v) for v in mask_shape]
    out = torch.empty((r, md, mh, mw), dtype=torch.uint8, device=labels.device)
    check(lib.cfun_mask_target_labels(ptr(labels), ptr(bounds), ptr(out), r, d, h, w, md, mh, mw, stream(labels)),
          "mask_target_labels")
    return out


def unmold_argmax(probs, box, image_dhw):
    """utils.unmold_mask + argmax (utils.py:443-460, model.py:1853-1858): probs [md,mh,mw,C] (NDHWC of ONE
    detection), box (z1,y1,x1,z2,y2,x2) voxel ints inside the volume -> uint8 class map [D,H,W]."""
    lib = _lib.load()
    probs = _c(probs.detach().float())
    md, mh, mw, c = probs.shape
    d, h, w = [int(v) for v in image_dhw]
    out = torch.empty((d, h, w), dtype=torch.uint8, device=probs.device)
    hbox = (C.c_int32 * 6)(*[int(v) for v in box])
    check(lib.cfun_unmold_argmax(ptr(probs), ptr(out), d, h, w, md, mh, mw, c, hbox, stream(probs)), "unmold_argmax")
    return out


def unmold_overlap(probs, boxes, image_dhw, want_full=False):
    """LiTS overlap-tile utils.unmold_mask + argmax (LiTS_2017/utils.py:383-408, LiTS_2017/model.py:1828-1829):
    probs [n,md,mh,mw,C] (NDHWC per detection), boxes [n,6] voxel ints inside the volume -> uint8 class map [D,H,W]
    (and, with ``want_full``, the averaged + clipped probabilities [D,H,W,C] the reference's function returns)."""
    lib = _lib.load()
    probs = _c(probs.detach().float())
    n, md, mh, mw, c = probs.shape
    d, h, w = [int(v) for v in image_dhw]
    boxes = [int(v) for row in boxes for v in row]
    if len(boxes) != 6 * n:
        raise ValueError("unmold_overlap: %d masks but %d box coordinates" % (n, len(boxes)))
    labels = torch.empty((d, h, w), dtype=torch.uint8, device=probs.device)
    full = torch.empty((d, h, w, c), dtype=torch.float32, device=probs.device) if want_full else None
    hbox = (C.c_int32 * max(6 * n, 1))(*boxes)
    check(lib.cfun_unmold_overlap(ptr(probs) if n else None, hbox, n, ptr(labels), ptr(full) if want_full else None,
                                  d, h, w, md, mh, mw, c, stream(probs)), "unmold_overlap")
    return (labels, full) if want_full else labels


def nms3d(boxes, scores, threshold, max_num):
    """Greedy 3-D NMS on device; returns (keep int32 [n], count int32 [1]) -- keep[:count] is the pick order."""
    lib = _lib.load()
    boxes = _c(boxes.detach().float())
    scores = _c(scores.detach().float())
    n = boxes.shape[0]
    keep = torch.empty((max(n, 1),), dtype=torch.int32, device=boxes.device)
    count = torch.zeros((1,), dtype=torch.int32, device=boxes.device)
    ws = workspace(lib.cfun_nms3d_workspace_bytes(n), boxes)
    check(lib.cfun_nms3d(ptr(boxes), ptr(scores), n, float(threshold), int(max_num), ptr(keep), ptr(count), ptr(ws),
                         ws.numel(), stream(boxes)), "nms3d")
    return keep, count




def resize3d(vol, out_dims, order=1, frame=None, offset=None, clip=False):
    """Resize a 3-D float volume (any strides: a permuted view of the loader's [H,W,D] array is read in place) to the
    dense ``out_dims`` with cfun_resize3d: order 1 = skimage.transform.resize(order=1, mode='constant') as evaluated
    for 3-D inputs (scipy zoom, grid-constant, grid_mode), order 0 = nearest; ``frame`` / ``offset``: the volume sits
    at ``offset`` inside a virtual zero frame of extent ``frame`` and that frame is what gets resized (LiTS
    pad-and-resize); ``clip``: skimage's clip=True (to the source's range)."""
    lib = _lib.load()
    if vol.dim() != 3 or vol.dtype != torch.float32:
        raise RuntimeError("resize3d: float32 [a,b,c] volume expected, got %s %s" % (vol.dtype, tuple(vol.shape)))
    out = torch.empty(tuple(int(v) for v in out_dims), dtype=torch.float32, device=vol.device)
    i64, i32 = C.c_int64 * 3, C.c_int32 * 3
    mm = None
    if clip:
        lo, hi = torch.aminmax(vol)
        mm = torch.stack([lo, hi]).contiguous()
    check(lib.cfun_resize3d(ptr_raw(vol), i64(*vol.stride()), i32(*vol.shape),
                            i32(*[int(v) for v in frame]) if frame is not None else None,
                            i32(*[int(v) for v in offset]) if offset is not None else None,
                            ptr(out), i32(*out.shape), int(order), ptr(mm), stream(vol)), "resize3d")
    return out


def halo_pack(x, z0, planes):
    lib = _lib.load()
    x = _c(x)
    n, d, h, w, c = x.shape
    buf = torch.empty((n, planes, h, w, c), dtype=torch.float32, device=x.device)
    check(lib.cfun_halo_pack(ptr(x), ptr(buf), n, d, h, w, c, z0, planes, stream(x)), "halo_pack")
    return buf


def halo_unpack(buf, x, z0):
    lib = _lib.load()
    n, d, h, w, c = x.shape
    planes = buf.shape[1]
    buf = _c(buf)                  # held across the launch (a temporary could be recycled before the kernel reads it)
    check(lib.cfun_halo_unpack(ptr(buf), ptr(x), n, d, h, w, c, z0, planes, stream(x)), "halo_unpack")
    return x




def to_ndhwc(x):
    """[N,C,D,H,W] (any memory format) -> contiguous [N,D,H,W,C]; free for channels_last_3d / C == 1."""
    return x.permute(0, 2, 3, 4, 1).contiguous()


def to_ncdhw(y):
    """[N,D,H,W,C] -> [N,C,D,H,W] view (channels_last_3d memory format, no copy)."""
    return y.permute(0, 4, 1, 2, 3)
