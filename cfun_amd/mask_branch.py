"""3-D U-Net mask head on HIP kernels -- drop-in for the reference's ``mask_branch`` module.

``Modified3DUNet(in_channels, n_classes, stage, base_n_filter)`` keeps the constructor, the attribute /
state-dict names (27 bias-free convs) and the forward semantics of
mask_branch.py:11-220, but runs as NDHWC HIP kernels:

  * every conv is one launch of the MFMA implicit-GEMM kernel (direct VALU kernel for the C_in = 1 stem);
  * nearest x2 up-sampling is never materialised (the consumer conv reads (z>>1, y>>1, x>>1));
  * residual / deep-supervision adds, the Dropout3d channel mask and the 'finetune' skip are conv epilogues;
  * InstanceNorm3d + LeakyReLU is one fused statistics pass + one apply pass.
"""
import os

import torch
import torch.nn as nn

from . import dist, ops
from .layers import Conv3dParams, default_algo, sharded_conv


def _holder(index, length, conv):
    """nn.Sequential whose child ``index`` is the conv: reproduces names such as 'norm_lrelu_conv_c2.2.weight'."""
    mods = [nn.Identity() for _ in range(length)]
    mods[index] = conv
    return nn.Sequential(*mods)


class Modified3DUNet(nn.Module):
    def __init__(self, in_channels, n_classes, stage, base_n_filter=32, dropout_p=0.6):
        super().__init__()
        self.in_channels, self.n_classes, self.stage = in_channels, n_classes, stage
        self.base_n_filter = b = base_n_filter
        self.dropout_p = dropout_p
        self.dropout_masks = None        # tests: list of 5 [N,C] multipliers injected instead of torch's RNG

        def c3(ci, co, stride=1):
            return Conv3dParams(ci, co, 3, stride=stride, padding=1, bias=False)

        def c1(ci, co):
            return Conv3dParams(ci, co, 1, bias=False)

        # context pathway (mask_branch.py:22-50)
        self.conv3d_c1_1 = c3(in_channels, b)
        self.conv3d_c1_2 = c3(b, b)
        self.lrelu_conv_c1 = _holder(1, 2, c3(b, b))
        for lvl, (ci, co) in enumerate(((b, 2 * b), (2 * b, 4 * b), (4 * b, 8 * b), (8 * b, 16 * b)), start=2):
            setattr(self, "conv3d_c%d" % lvl, c3(ci, co, stride=2))
            setattr(self, "norm_lrelu_conv_c%d" % lvl, _holder(2, 3, c3(co, co)))
        # localisation pathway (mask_branch.py:51-88)
        self.norm_lrelu_upscale_conv_norm_lrelu_l0 = _holder(3, 6, c3(16 * b, 8 * b))
        self.conv3d_l0 = c1(8 * b, 8 * b)
        self.conv_norm_lrelu_l1 = _holder(0, 3, c3(16 * b, 16 * b))
        self.conv3d_l1 = c1(16 * b, 8 * b)
        self.norm_lrelu_upscale_conv_norm_lrelu_l1 = _holder(3, 6, c3(8 * b, 4 * b))
        self.conv_norm_lrelu_l2 = _holder(0, 3, c3(8 * b, 8 * b))
        self.conv3d_l2 = c1(8 * b, 4 * b)
        self.norm_lrelu_upscale_conv_norm_lrelu_l2 = _holder(3, 6, c3(4 * b, 2 * b))
        self.conv_norm_lrelu_l3 = _holder(0, 3, c3(4 * b, 4 * b))
        self.conv3d_l3 = c1(4 * b, 2 * b)
        self.norm_lrelu_upscale_conv_norm_lrelu_l3 = _holder(3, 6, c3(2 * b, b))
        self.conv_norm_lrelu_l4 = _holder(0, 3, c3(2 * b, 2 * b))
        self.conv3d_l4 = c1(2 * b, n_classes)
        self.ds2_1x1_conv3d = c1(8 * b, n_classes)
        self.ds3_1x1_conv3d = c1(4 * b, n_classes)
        self.out_upscale_conv = _holder(1, 2, Conv3dParams(n_classes, n_classes, 5, padding=2, bias=False))

    # ------------------------------------------------------------------------------------------
    def _drop_masks(self, n, device):
        """Five Dropout3d(p) channel masks (keep / (1-p)); None in eval mode (mask_branch.py:19,130-175)."""
        if not self.training or self.dropout_p <= 0:
            return [None] * 5
        if self.dropout_masks is not None:
            return [m.detach().to(device="cpu", dtype=torch.float32).contiguous() for m in self.dropout_masks]
        b, keep = self.base_n_filter, 1.0 - self.dropout_p
        # drawn on the host: the kept-channel index lists of _dropout_pair must not cost a device sync
        return [torch.empty((n, c)).bernoulli_(keep).div_(keep) for c in (b, 2 * b, 4 * b, 8 * b, 16 * b)]

    # Dropout3d zeroes whole channels per sample, and InstanceNorm + LeakyReLU keep an all-zero channel at exactly
    # zero.  So in "conv1 -> Dropout3d -> norm/act -> conv2" (every level's pair, mask_branch.py:129-176) conv1 only
    # has to produce the kept channels and conv2 only has to read them: with p = 0.6 that is 40 % of conv1's output
    # rows and 40 % of conv2's reduction -- identical values and gradients (the dropped channels contribute exact
    # zeros), ~55 % of the pair's FLOPs gone in forward, data-gradient and weight-gradient.  The kept sets differ per
    # RoI, so the pair runs per sample on weights gathered by ``index_select`` (differentiable: the weight gradient
    # scatters back); only where a single RoI still fills the GPU (>= SPARSE_MIN_VOX voxels, levels 1-3 at 96^3).
    SPARSE_MIN_VOX = 24 * 24 * 24

    @staticmethod
    def _kept_channels(drop_cpu):
        """Per sample: the kept channel indices of a host mask [N,C], padded with dropped ones (scale 0) to a multiple
        of 4 (host lists)."""
        lists = []
        for row in drop_cpu.tolist():                      # plain Python lists: per-element tensor indexing is ~10 us each
            kept = [j for j, v in enumerate(row) if v > 0]
            dropped = [j for j, v in enumerate(row) if v <= 0]
            kept += dropped[:(-len(kept)) % 4 or (4 if not kept else 0)]
            lists.append(sorted(kept))
        return lists

    def _upload_dropout(self, drops, device):
        """All five masks and all kept-channel lists go to the device in TWO transfers at the start of the forward
        pass, asynchronously through persistent pinned staging buffers (ops.upload): a pageable host-to-device copy waits
        for the stream and leaves the GPU idle until the host has caught up (10 such stalls in the middle of the pass cost
        ~1 ms per step in round 1; pinning fresh staging buffers per step cost more than it saved -- the ring is allocated
        once)."""
        if drops[0] is None:
            return [None] * 5
        lists = [self._kept_channels(m) for m in drops]
        flat_idx = ops.upload(torch.tensor([j for lv in lists for l in lv for j in l], dtype=torch.long), device)
        pad = [(-m.numel()) % 4 for m in drops]                       # every mask starts 16-byte aligned
        flat_m = ops.upload(torch.cat([torch.nn.functional.pad(m.reshape(-1), (0, p)) for m, p in zip(drops, pad)]), device)
        out, oi, om = [], 0, 0
        for m, lv, p in zip(drops, lists, pad):
            idxs = []
            for l in lv:
                idxs.append(flat_idx[oi:oi + len(l)])
                oi += len(l)
            out.append((flat_m[om:om + m.numel()].view(m.shape), idxs))
            om += m.numel() + p
        return out

    def _dropout_pair(self, src, head, conv1, conv2, drop_cpu, pre2, out_stats=None, level=None):
        """One level's ``(a, res) = head(src); out = conv2(pre2(Dropout3d(conv1(a)))) + res``: conv1 / conv2 are the
        level's 3x3x3 Conv3dParams (conv1: Ci -> C, conv2: C -> Co), drop_cpu = (device mask [N,C] (keep / (1-p)),
        per-sample kept-channel index tensors) from ``_upload_dropout`` or None, ``head`` the ops that produce the
        pair's input and residual from ``src`` [N,...].  ``pre2(t, stats)`` gets the ``ops.StatsSlot`` conv1's epilogue
        filled with t's InstanceNorm statistics; ``out_stats`` (a slot of N samples) receives those of the result."""
        n = src.shape[0]
        c = conv1.out_channels
        if drop_cpu is None:
            a, res = head(src)
            s1 = ops.StatsSlot(n)
            return conv2(pre2(conv1(a, stats=s1), s1), res=res, stats=out_stats)
        drop, idxs = drop_cpu
        s2 = getattr(head, "stride", 1)
        vox = (src.shape[1] // s2) * (src.shape[2] // s2) * (src.shape[3] // s2)
        sparse = (vox >= self.SPARSE_MIN_VOX and c % 4 == 0 and conv2.out_channels % 4 == 0
                  and conv1.in_channels % 4 == 0 and conv1.bias is None and conv2.bias is None)
        if not sparse:
            a, res = head(src)
            s1 = ops.StatsSlot(n)
            return conv2(pre2(conv1(a, scale=drop, stats=s1), s1), res=res, stats=out_stats)
        algo = default_algo()
        # the level's head ops stay batched (per-sample heads measured no better); batched <-> per-sample hand-overs are
        # zero-copy in both directions (ops.split_batch / join_batch: results and gradients are written in place)
        a_all, res_all = head(src)
        a_parts = a_all.samples() if isinstance(a_all, ops.NormedInput) else ops.split_batch(a_all)
        res_parts = ops.split_batch(res_all)
        ybuf, gbuf = ops.BatchBuffer(n), ops.BatchBuffer(n)     # outputs / conv1 input gradients, written in place
        outs = []
        zs = dist.current()
        zs = zs if zs is not None and zs.world > 1 else None      # z-sharded RoI: slabs, halos inside the convs
        # (level: the key of this level's index lists in the pass's ops.WeightScope -- the slices' conv operands are then
        # gathered by the batched weight preparation and the slices themselves never materialised)
        gk = None if (zs is not None or level is None) else level
        w1s, w2s = ops.gather_slices(conv1.weight, 0, idxs, key=gk), ops.gather_slices(conv2.weight, 1, idxs, key=gk)
        for i in range(n):
            a, res = a_parts[i], res_parts[i]
            idx = idxs[i]
            w1 = w1s[i]
            sc1 = drop[i:i + 1].index_select(1, idx).contiguous()
            if zs is not None:
                p1, p2 = (0,) + tuple(conv1.padding[1:]), (0,) + tuple(conv2.padding[1:])
                spec1 = ops.ConvSpec(k=conv1.kernel_size, co=idx.numel(), pad=p1, scale_per_n=True, algo=algo)
                t = sharded_conv(a, w1, spec1, conv1.kernel_size[0], 1, conv1.padding[0], sc1, None, None, zs)
                spec2 = ops.ConvSpec(k=conv2.kernel_size, co=conv2.out_channels, pad=p2, algo=algo)
                outs.append(sharded_conv(pre2(t, None), w2s[i], spec2, conv2.kernel_size[0], 1, conv2.padding[0], None, None, res, zs))
                continue
            spec1 = ops.ConvSpec(k=conv1.kernel_size, co=idx.numel(), pad=conv1.padding, scale_per_n=True, algo=algo)
            s1 = ops.StatsSlot(1)
            t = ops.conv3d_w(a, w1, spec1, scale=sc1, dx_slot=(gbuf, i), stats=s1)
            w2 = w2s[i]
            spec2 = ops.ConvSpec(k=conv2.kernel_size, co=conv2.out_channels, pad=conv2.padding, algo=algo)
            outs.append(ops.conv3d_w(pre2(t, s1), w2, spec2, res=res, out=(ybuf, i),
                                     stats=None if out_stats is None else (out_stats, i)))
        if zs is not None:
            return outs[0] if n == 1 else torch.cat(outs, dim=0)
        return ops.join_batch(ybuf, outs)

    @staticmethod
    def _up_conv(h, conv, depth_padded=False, stats=None):
        """conv3x3x3(nearest_up2(h)).  On the up-sampled grid each output parity only sees 2x2x2 distinct
        low-resolution voxels, so the weights are folded per parity (ops.fold_up2_weight, differentiable) and the
        conv runs on the LOW-resolution tensor with a depth-to-space epilogue, skipping the 19 folded-zero taps:
        8/27 of the FLOPs of the straightforward "read (z>>1,y>>1,x>>1)" form, for forward, dgrad and wgrad."""
        ci, co = conv.in_channels, conv.out_channels
        if ci % 4 or co % 4:
            if depth_padded:
                raise NotImplementedError("z-sharded up-conv needs C % 4 == 0 (the folded form)")
            return conv(h, up2=True, stats=stats)
        cqp = (co + 15) // 16 * 16
        # depth_padded: h already carries one low-resolution halo plane on each side (z-sharded RoI) -> depth-VALID
        spec = ops.ConvSpec(k=(3, 3, 3), co=8 * cqp, pad=(0 if depth_padded else 1, 1, 1), d2s=True, d2s_cq=co,
                            tap_skip=True, algo=default_algo())
        return ops.conv3d_w(h, ops.fold_up2_weight(conv.weight, cqp), spec, stats=None if depth_padded else stats)

    def forward_ndhwc(self, x, zshard=None):
        """x [N,D,H,W,in]: the RoI crops.  ``zshard`` (a ``dist.ShardContext`` over the RoI's sub-group of ranks, N == 1):
        x is this rank's depth slab of ONE crop and the result is its slab of the logits -- the two high-resolution levels
        run on slabs (halo planes inside the 3x3x3 convs, InstanceNorm statistics all-reduced), the levels at 1/4
        resolution and below are folded onto every rank (``dist.gather_replicated`` / ``dist.enter_slab``)."""
        zs = zshard if zshard is not None and zshard.world > 1 else None
        if zs is not None and x.shape[0] != 1:
            raise ValueError("a z-sharded U-Net handles one RoI per sub-group")
        drop = self._upload_dropout(self._drop_masks(x.shape[0], x.device), x.device)
        # every conv weight of the pass in the layout its kernels read, by ONE launch (ops.WeightScope: recorded on the
        # first pass; the per-RoI Dropout3d slices are gathered by that launch from this pass's index lists)
        with ops.WeightScope(self, dyn={lv: d[1] for lv, d in enumerate(drop) if d is not None}):
            return self._forward_body(x, zs, drop)

    def _forward_body(self, x, zs, drop):
        sharded = (lambda: dist.depth_sharded_as(zs)) if zs is not None else dist.nullcontext
        folded = dist.slab_local if zs is not None else dist.nullcontext

        def nl(t, out=None, rep=False, stats=None, lazy=False, passthrough=False):      # rep: t is a replicated (folded) tensor -> plain local statistics
            # stats: the StatsSlot t's producer conv filled from its epilogue (empty on depth slabs: own pass then)
            # lazy: every consumer is a conv -> no apply pass, they stage t through the norm (ops.NormedInput)
            # passthrough: also return t' (= t) for t's other consumer, the residual: gradients summed in the norm's backward
            return ops.instnorm_lrelu(t, out=out, shard=None if rep else zs, stats=stats, lazy=lazy and zs is None and LAZY,
                                      passthrough=passthrough)

        LAZY = os.environ.get("CFUN_FUSE_NORM", "1") != "0"      # (A/B switch: 0 = every norm / activation as its own pass)
        lz = zs is None and LAZY

        def slot(t_or_n):
            return ops.StatsSlot(t_or_n if isinstance(t_or_n, int) else t_or_n.shape[0])

        nb = x.shape[0]

        def nluc(h, holder, out=None, src="same", stats=None, lazy_out=False):
            """norm -> lrelu -> nearest x2 -> 3x3x3 conv -> norm -> lrelu (mask_branch.py:108-116).  src: where h lives --
            'same' (no sharding, or sharded in and out), 'rep' (replicated in and out) or 'enter' (replicated in, sharded out).
            stats: the slot h's producer filled; lazy_out: the result feeds convs only."""
            a = nl(h, rep=src != "same", stats=stats, lazy=True)       # -> the folded up-conv
            if zs is None or src == "rep":
                su = slot(nb)
                return nl(self._up_conv(a, holder[3], stats=su), out=out, rep=src == "rep", stats=su, lazy=lazy_out)
            a = dist.enter_slab(a, zs, 1, 1) if src == "enter" else dist.halo_exchange(a, 1, 1, zs)
            return nl(self._up_conv(a, holder[3], depth_padded=True), out=out)

        with sharded():
            # level 1: residual is the pre-activation stem output, context_1 is taken before the norm
            def head1(xp):
                res = self.conv3d_c1_1(xp)
                return ops.lrelu(res, lazy=lz, passthrough=True)       # (activation -> conv1, res' -> the residual add)
            s_out = slot(nb)
            out = self._dropout_pair(x, head1, self.conv3d_c1_2, self.lrelu_conv_c1[1], drop[0],
                                     lambda t, st: ops.lrelu(t, lazy=lz), out_stats=s_out, level=0)
            # context_1 is only ever read by the level-1 concat (mask_branch.py:203): it is written straight into the
            # second half of that concat's buffer, the decoder later writes the first half -- no torch.cat, no copies
            b1 = self.base_n_filter
            cat1 = ops.ConcatBuffer(out, 2 * b1) if b1 % 4 == 0 else None
            # out feeds the level-1 concat (LeakyReLU) and the norm in front of conv3d_c2: the norm reads out' so that the two
            # gradients meet inside the LeakyReLU's backward kernel
            c0, out_ = ops.lrelu(out, out=None if cat1 is None else cat1.slot(b1, 2 * b1), passthrough=True)
            ctx = [c0]
            h = nl(out_, stats=s_out, lazy=True)      # -> conv3d_c2 only
        # levels 2..5: stride-2 conv, then the SAME norm_lrelu_conv weights twice around the dropout
        for lvl in (2, 3, 4, 5):
            down = getattr(self, "conv3d_c%d" % lvl)
            rep = zs is not None and lvl >= 3          # folded levels

            def head(hp, down=down, lvl=lvl, rep=rep):
                sd = None
                if zs is not None and lvl == 3:        # the fold: slabs of the 1/4-resolution tensor -> every rank
                    with dist.depth_sharded_as(zs):
                        res = down(hp)
                    res = dist.gather_replicated(res, zs)
                else:
                    sd = slot(nb)
                    res = down(hp, stats=sd)
                return nl(res, rep=rep, stats=sd, lazy=True, passthrough=True)      # (normed -> conv1, res' -> the residual)
            head.stride = 2
            conv = getattr(self, "norm_lrelu_conv_c%d" % lvl)[2]
            with (folded() if rep else sharded()):
                s_out = slot(nb)
                out = self._dropout_pair(h, head, conv, conv, drop[lvl - 1],
                                         lambda t, st, rep=rep: nl(t, rep=rep, stats=st, lazy=True), out_stats=s_out,
                                         level=lvl - 1)
                if lvl < 5:
                    h = nl(out, rep=rep, stats=s_out)
                    ctx.append(h)
        R = zs is not None       # decoder: replicated up to 1/4 resolution, sharded from 1/2
        with folded():
            h = nluc(out, self.norm_lrelu_upscale_conv_norm_lrelu_l0, src="rep" if R else "same", stats=s_out, lazy_out=True)
            sa, sb, sc, sd_ = slot(nb), slot(nb), slot(nb), slot(nb)
            h = nl(self.conv3d_l0(h, stats=sa), rep=R, stats=sa)
            h = nl(self.conv_norm_lrelu_l1[0](torch.cat([h, ctx[3]], dim=-1), stats=sb), rep=R, stats=sb, lazy=True)
            h = nluc(self.conv3d_l1(h, stats=sc), self.norm_lrelu_upscale_conv_norm_lrelu_l1, src="rep" if R else "same",
                     stats=sc)
            ds2 = nl(self.conv_norm_lrelu_l2[0](torch.cat([h, ctx[2]], dim=-1), stats=sd_), rep=R, stats=sd_, lazy=True)
            s_l2 = slot(nb)
            h_l2 = self.conv3d_l2(ds2, stats=s_l2)
            ds2_out = self.ds2_1x1_conv3d(ds2)
        h = nluc(h_l2, self.norm_lrelu_upscale_conv_norm_lrelu_l2, src="enter" if R else "same", stats=s_l2)
        if R:
            ds2_out = dist.enter_slab(ds2_out, zs)
        with sharded():
            s3, s_l3, s4 = slot(nb), slot(nb), slot(nb)
            ds3 = nl(self.conv_norm_lrelu_l3[0](torch.cat([h, ctx[1]], dim=-1), stats=s3), stats=s3, lazy=True)
            if cat1 is None:
                h = nluc(self.conv3d_l3(ds3, stats=s_l3), self.norm_lrelu_upscale_conv_norm_lrelu_l3, stats=s_l3)
                joined = torch.cat([h, ctx[0]], dim=-1)
            else:
                h = nluc(self.conv3d_l3(ds3, stats=s_l3), self.norm_lrelu_upscale_conv_norm_lrelu_l3, out=cat1.slot(0, b1),
                         stats=s_l3)
                joined = cat1.join(h, ctx[0])
            h = nl(self.conv_norm_lrelu_l4[0](joined, stats=s4), stats=s4, lazy=True)      # -> conv3d_l4 only
            # deep supervision: up(up(ds2_1x1) + ds3_1x1) + out_pred, each add is a conv epilogue
            s = self.ds3_1x1_conv3d(ds3, res=ds2_out, res_up2=True)
            out = self.conv3d_l4(h, res=s, res_up2=True)
        if self.stage == "finetune":   # up(out) + conv5(up(out)), mask_branch.py:216-218
            # The 5x5x5 conv reads a nearest-x2 up-sampled tensor, so its 125 taps collapse onto 27 low-resolution
            # taps per output parity: run it as ONE 3x3x3 conv (n_classes -> 8 parities x n_classes channels) on
            # the low-res logits with a depth-to-space epilogue -- 27/125 of the FLOPs, identical arithmetic up
            # to the fp32 pre-summation of the folded weights.  The up(out) skip is the epilogue residual.
            conv = self.out_upscale_conv[1]
            wf = ops.fold_up2_weight(conv.weight)
            spec = ops.ConvSpec(k=(3, 3, 3), co=8 * self.n_classes, pad=(0 if R else 1, 1, 1), d2s=True, res_up2=True,
                                algo=default_algo())
            out = ops.conv3d_w(dist.halo_exchange(out, 1, 1, zs) if R else out, wf, spec, res=out)
        return out

    def forward(self, x):
        """x [N,in,D,H,W] -> logits [N,n_classes,D,H,W] (x2 spatial in stage 'finetune'); the result is a
        channels_last_3d view of the NDHWC buffer."""
        return ops.to_ncdhw(self.forward_ndhwc(ops.to_ndhwc(x)))
