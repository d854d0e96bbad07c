"""Parameter holders shared by the drop-in modules.

The state-dict contract is the reference's: every conv keeps ``weight`` in OIDHW (and
``bias``), BatchNorm3d keeps its five entries.  These holders never call torch convolutions; the forward
passes in backbone.py / mask_branch.py / model.py feed them to the HIP kernels through cfun_amd.ops.
"""
import math
import os

import torch
import torch.nn as nn

from . import dist, ops
from ._lib import ALGO_AUTO, ALGO_DIRECT, ALGO_MFMA


def default_algo():
    """CFUN_CONV_ALGO=auto|direct|mfma selects the HIP conv kernel family (tests; default auto)."""
    return {"auto": ALGO_AUTO, "direct": ALGO_DIRECT, "mfma": ALGO_MFMA}[
        os.environ.get("CFUN_CONV_ALGO", "auto")]


class Conv3dParams(nn.Module):
    """weight [Co,Ci,kd,kh,kw] (+ bias [Co]) with nn.Conv3d's default initialisation."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        k = (kernel_size,) * 3 if isinstance(kernel_size, int) else tuple(kernel_size)
        p = (padding,) * 3 if isinstance(padding, int) else tuple(padding)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = k, int(stride), p
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *k))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1.0 / math.sqrt(in_channels * k[0] * k[1] * k[2])
            nn.init.uniform_(self.bias, -bound, bound)

    def spec(self, act=ops.ACT_NONE, up2=False, res_up2=False, scale_per_n=False, pad=None):
        return ops.ConvSpec(k=self.kernel_size, co=self.out_channels, stride=self.stride,
                            pad=self.padding if pad is None else pad, up2=up2, act=act, res_up2=res_up2,
                            scale_per_n=scale_per_n, algo=default_algo())

    def packed(self):
        return ops.pack_weight(self.weight)

    def forward(self, x, act=ops.ACT_NONE, bn=None, bn_eps=None, res=None, up2=False, res_up2=False, scale=None, stats=None):
        """NDHWC in / out (x may be an ``ops.NormedInput``).  bn: a frozen nn.BatchNorm3d folded into the epilogue;
        scale: per-(n, channel) multiplier (Dropout3d mask) -- mutually exclusive with bn; stats: an ``ops.StatsSlot`` that
        receives the output's InstanceNorm statistics from the conv's epilogue (left empty on depth slabs)."""
        shift = self.bias
        per_n = False
        pad = None
        shard = dist.current()
        sharded = shard is not None and shard.world > 1
        pre = False                 # shift = t + b * scale with the "conv hands back db" contract (ops.fold_bias, pre)
        if bn is not None:
            eps = bn.eps if bn_eps is None else bn_eps
            scale, t = folded_bn(bn, eps)
            if self.bias is None:
                shift = t
            elif sharded:           # (the split / halo launches go through the plain contract)
                shift = ops.fold_bias(self.bias, scale, t)                      # (b - mean) * s + beta
            else:
                shift, pre = _step_shift(self, bn, eps, scale, t), True
        elif scale is not None:
            per_n = True
        if shard is not None and shard.world > 1 and isinstance(x, ops.NormedInput):
            x = x.materialize()          # depth slabs: the split / padded launches take a real tensor
        if shard is not None and shard.world > 1 and self.kernel_size[0] > 1:
            # depth-sharded volume: the conv runs depth-VALID with the halo planes of the ring neighbours
            if up2:
                raise NotImplementedError("depth sharding of unfolded up-sampling convs: use mask_branch._up_conv")
            return sharded_conv(x, self.weight, self.spec(act, False, res_up2, per_n, (0, self.padding[1], self.padding[2])),
                                self.kernel_size[0], self.stride, self.padding[0], scale, shift, res, shard)
        return ops.conv3d_w(x, self.weight, self.spec(act, up2, res_up2, per_n, pad), scale=scale, shift=shift, res=res,
                            stats=stats, shift_scaled=pre)

    def extra_repr(self):
        return "%d, %d, kernel_size=%s, stride=%d, padding=%s, bias=%s" % (
            self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding, self.bias is not None)


def sharded_conv(x, w, spec, kd, stride, pd, scale, shift, res, shard):
    """conv3d_w on this rank's depth slab of a z-sharded tensor (``spec`` already has depth padding 0; kd / stride / pd
    are the conv's depth geometry): interior output planes straight from the slab while the halo transfer is in flight,
    boundary planes from the edge tensors (``dist.halo_conv``); thin slabs fall back to the padded-slab exchange.  A
    residual ``res`` (same resolution, or low resolution with res_up2 / a depth-to-space conv's parity layout) is sliced
    to each region's output planes."""
    co = spec.d2s_cq or (spec.co // 8) if spec.d2s else spec.co
    sc = 2 if spec.d2s else 1                              # output planes per conv output plane (depth-to-space)
    ho = (x.shape[2] + 2 * spec.pad[1] - spec.k[1]) // spec.stride + 1
    wo = (x.shape[3] + 2 * spec.pad[2] - spec.k[2]) // spec.stride + 1

    def res_rows(z0, z1):
        if res is None:
            return None
        if spec.d2s:                                       # residual at the conv's (low) resolution, d2s epilogue
            return res[:, z0:z1]
        if spec.res_up2:                                   # residual at half the output resolution
            return res[:, z0 // 2:(z1 + 1) // 2]
        return res[:, z0:z1]

    def run(inp, out, z0, z1):
        return ops.conv3d_w(inp, w, spec, scale=scale, shift=shift, res=res_rows(z0, z1),
                            out=None if out is None else out)

    if not spec.d2s and not (spec.res_up2 and res is not None):
        y = dist.halo_conv(x, run, kd, stride, pd, (ho, wo, co), shard)
        if y is not None:
            return y
    lo, hi = dist.conv_depth_halo(kd, stride, pd)
    return ops.conv3d_w(dist.halo_exchange(x, lo, hi, shard), w, spec, scale=scale, shift=shift, res=res)


# ---- the bias folds of a step, batched.  Which (conv, bn) pairs a network runs is only known at the call sites, so the first
# step between begin_step / end_step records them; later steps fold all of them in ONE multi-tensor launch up front
# (ops.fold_bias_many) and the convs pick their shift up here.  Entries live only inside that window and carry the
# identity of (s, t), so a call outside a step, another pairing or a re-folded BatchNorm falls back to the per-conv fold.
_STEP = {"shifts": None, "record": None}


BATCH_BIAS_FOLD = os.environ.get("CFUN_BATCH_BIAS_FOLD", "1") != "0"


def begin_step(net):
    # the weight operands of the step's convs outside the U-Net (FPN, RPN, classifier): one batched preparation launch
    scope = ops.WeightScope(net)
    scope.__enter__()
    _STEP["wscope"] = scope
    if not BATCH_BIAS_FOLD:
        return
    pairs = getattr(net, "_cfun_fold_pairs", None)
    if pairs is None:
        _STEP["shifts"], _STEP["record"] = None, []
        return
    _STEP["record"] = None
    live = [(c, bn, eps) + folded_bn(bn, eps) for c, bn, eps in pairs if c.bias is not None and c.bias.requires_grad]
    if not live:
        _STEP["shifts"] = None
        return
    outs = ops.fold_bias_many([c.bias for c, _, _, _, _ in live], [s for _, _, _, s, _ in live], [t for _, _, _, _, t in live],
                              pre=True)
    _STEP["shifts"] = {id(c): (sh, id(bn), eps, s, t) for (c, bn, eps, s, t), sh in zip(live, outs)}


def end_step(net, ok=True):
    scope = _STEP.pop("wscope", None)
    if scope is not None:
        scope.__exit__(None if ok else RuntimeError, None, None)
    rec = _STEP["record"]
    if rec is not None and ok and getattr(net, "_cfun_fold_pairs", None) is None:
        seen, pairs = set(), []
        for c, bn, eps in rec:
            if id(c) not in seen:
                seen.add(id(c))
                pairs.append((c, bn, eps))
        net._cfun_fold_pairs = pairs
    _STEP["shifts"], _STEP["record"] = None, None


def _step_shift(conv, bn, eps, s, t):
    shifts = _STEP["shifts"]
    if shifts is not None:
        e = shifts.get(id(conv))
        if e is not None and e[1] == id(bn) and e[2] == eps and e[3] is s and e[4] is t:
            return e[0]
    if _STEP["record"] is not None:
        _STEP["record"].append((conv, bn, eps))
    return ops.fold_bias(conv.bias, s, t, pre=True)


def folded_bn(bn, eps):
    """The constant affine of a frozen BatchNorm3d: s = gamma / sqrt(var + eps),
    t = beta - mean * s, so that bn(conv + b) = conv * s + (b * s + t).  The four tensors never receive gradients
    on this path (model.py:1297-1304), so (s, t) are cached and recomputed only when one of them is written
    (load_state_dict, .to(), in-place edits bump ``_version``)."""
    src = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
    key = (float(eps),) + tuple((t.data_ptr(), t._version, t.device) for t in src)
    cached = getattr(bn, "_cfun_fold", None)
    if cached is None or cached[0] != key:
        with torch.no_grad():
            s = bn.weight * torch.rsqrt(bn.running_var + eps)
            t = bn.bias - bn.running_mean * s
        cached = (key, s, t)
        bn._cfun_fold = cached
    return cached[1], cached[2]


def frozen_bn(channels, eps=1e-5, momentum=0.1):
    """BatchNorm3d used as a parameter/buffer holder only: always evaluated in eval mode as a constant
    per-channel affine folded into the preceding conv (model.py:1397-1406)."""
    return nn.BatchNorm3d(channels, eps=eps, momentum=momentum)
