"""Depth-wise (z) sharding of one CT volume across the GPUs of a node.

One process per GPU (``torch.distributed``; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the
CPU test tier).  The volume is split into R contiguous depth slabs; the only ops on the backbone / FPN / RPN
path that couple neighbouring slabs are the convolutions with a depth kernel of 3 (the stem, every conv_T,
the FPN / RPN 3x3x3 convs).  Inside ``depth_sharded(group)`` every ``Conv3dParams`` with kD > 1 first
exchanges its depth halo with the ring neighbours (point-to-point send/recv of packed planes -- xGMI is
point-to-point, messages are 0.13-2.1 MB) and then runs as a depth-VALID convolution on the padded slab;
all other ops are slab-local.  The halo exchange is an autograd Function: its backward returns the halo
planes' gradients to their owners, so the same context also serves training.

RPN outputs are flattened (z, y, x) (model.py:727-729), i.e. every rank owns a contiguous anchor range per
pyramid level; ``gather_rpn_outputs`` is the single all-gather after which every rank runs the identical,
deterministic proposal_layer / NMS -- no broadcast needed.
"""
import contextlib

import torch
import torch.distributed as dist

from . import ops

_CTX = None


class ShardContext:
    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    @property
    def prev(self):
        return self.rank - 1 if self.rank > 0 else None

    @property
    def next(self):
        return self.rank + 1 if self.rank < self.world - 1 else None


def current():
    return _CTX


@contextlib.contextmanager
def depth_sharded(group=None):
    """Activate depth sharding for the convolutions executed inside the block."""
    global _CTX
    old, _CTX = _CTX, ShardContext(group)
    try:
        yield _CTX
    finally:
        _CTX = old


@contextlib.contextmanager
def depth_sharded_as(ctx):
    """``depth_sharded`` with an existing ShardContext (e.g. a RoI's sub-group)."""
    global _CTX
    old, _CTX = _CTX, ctx
    try:
        yield ctx
    finally:
        _CTX = old


nullcontext = contextlib.nullcontext


@contextlib.contextmanager
def slab_local():
    """Suspend depth sharding inside a ``depth_sharded`` block: the convolutions executed here work on whole,
    rank-private tensors (the per-RoI heads: a RoI crop is not a slab of anything)."""
    global _CTX
    old, _CTX = _CTX, None
    try:
        yield
    finally:
        _CTX = old


def _host_staged(ctx, like):
    """gloo moves host memory and is not stream-ordered: handed a device tensor, its send/recv read and write the raw
    pointer from the host at once, whatever the stream still has queued (the 2-rank GPU-tier test caught exactly that: stale
    halo planes once the GPU ran behind the host).  Device tensors on a gloo group -- only the single-GPU test set-up,
    tests/test_dist_gpu.py and CFUN_BENCH_BACKEND=gloo -- are therefore staged through host copies, which ARE ordered on
    the current stream.  RCCL ("nccl") enqueues on the current stream and takes the device buffers as they are."""
    return like.is_cuda and dist.get_backend(ctx.group) == "gloo"


def _exchange(ctx, send_prev, send_next, recv_prev_shape, recv_next_shape, like):
    """Ring-neighbour exchange of packed plane buffers.  Returns (from_prev, from_next); None at the volume
    boundary.  Global ranks are resolved through the group so sub-groups work."""
    reqs, from_prev, from_next = [], None, None
    staged = _host_staged(ctx, like)
    rdev = torch.device("cpu") if staged else like.device

    def peer(r):
        return dist.get_global_rank(ctx.group, r) if ctx.group is not None else r

    def out(t):
        t = t.contiguous()
        return t.cpu() if staged else t      # (.cpu() waits for the current stream: the planes are complete)

    opsl = []
    if ctx.prev is not None:
        if recv_prev_shape is not None:
            from_prev = torch.empty(recv_prev_shape, dtype=like.dtype, device=rdev)
            opsl.append(dist.P2POp(dist.irecv, from_prev, peer(ctx.prev), ctx.group))
        if send_prev is not None:
            opsl.append(dist.P2POp(dist.isend, out(send_prev), peer(ctx.prev), ctx.group))
    if ctx.next is not None:
        if send_next is not None:
            opsl.append(dist.P2POp(dist.isend, out(send_next), peer(ctx.next), ctx.group))
        if recv_next_shape is not None:
            from_next = torch.empty(recv_next_shape, dtype=like.dtype, device=rdev)
            opsl.append(dist.P2POp(dist.irecv, from_next, peer(ctx.next), ctx.group))
    if opsl:
        reqs = dist.batch_isend_irecv(opsl)
        for r in reqs:
            r.wait()
    if staged:
        from_prev = None if from_prev is None else from_prev.to(like.device)
        from_next = None if from_next is None else from_next.to(like.device)
    return from_prev, from_next


class _HaloExchange(torch.autograd.Function):
    """x [N,Dl,H,W,C] -> [N,lo+Dl+hi,H,W,C]: ``lo`` planes from the previous rank's top, ``hi`` planes from the next
    rank's bottom, zeros at the volume boundary (= the conv's zero padding)."""

    @staticmethod
    def forward(ctx, x, lo, hi, shard):
        x = x.contiguous()
        n, d, h, w, c = x.shape
        send_next = ops.halo_pack(x, d - lo, lo) if lo > 0 else None      # my top planes are next's low halo
        send_prev = ops.halo_pack(x, 0, hi) if hi > 0 else None           # my bottom planes are prev's high halo
        from_prev, from_next = _exchange(shard, send_prev, send_next, (n, lo, h, w, c) if lo > 0 else None,
                                         (n, hi, h, w, c) if hi > 0 else None, x)
        out = torch.zeros((n, lo + d + hi, h, w, c), dtype=x.dtype, device=x.device)
        ops.halo_unpack(x, out, lo)
        if from_prev is not None:
            ops.halo_unpack(from_prev, out, 0)
        if from_next is not None:
            ops.halo_unpack(from_next, out, lo + d)
        ctx.shard, ctx.lo, ctx.hi, ctx.d = shard, lo, hi, d
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        lo, hi, d, shard = ctx.lo, ctx.hi, ctx.d, ctx.shard
        n, _, h, w, c = g.shape
        # gradients of the halo planes go back to the rank that owns them
        send_prev = ops.halo_pack(g, 0, lo) if lo > 0 else None
        send_next = ops.halo_pack(g, lo + d, hi) if hi > 0 else None
        from_prev, from_next = _exchange(shard, send_prev, send_next, (n, hi, h, w, c) if hi > 0 else None,
                                         (n, lo, h, w, c) if lo > 0 else None, g)
        gx = ops.halo_pack(g, lo, d)
        if from_prev is not None:      # prev used my bottom `hi` planes as its high halo
            gx[:, :hi] += from_prev
        if from_next is not None:      # next used my top `lo` planes as its low halo
            gx[:, d - lo:] += from_next
        return gx, None, None, None


class _HaloStart(torch.autograd.Function):
    """First half of the edge exchange of ``halo_conv``: pack this slab's boundary planes and START their transfer on the
    side HIP stream; nothing here waits for it.  Returns an alias of x that every consumer (the interior conv, the edge
    tensors of ``_HaloFinish``) must read, so that this node's backward runs LAST: it waits for the gradient exchange that
    ``_HaloFinish.backward`` started and adds the neighbours' contributions for the planes they used as halos.
    ``state`` (a dict shared by the two halves) carries the receive buffers and the stream hand-over."""

    @staticmethod
    def forward(ctx, x, lo, hi, shard, state):
        n, d, h, w, c = x.shape
        send_next = ops.halo_pack(x, d - lo, lo) if lo > 0 else None      # my top planes are next's low halo
        send_prev = ops.halo_pack(x, 0, hi) if hi > 0 else None           # my bottom planes are prev's high halo
        state["fp"], state["fn"], state["wait"] = _exchange_async(
            shard, send_prev, send_next, (n, lo, h, w, c) if lo > 0 else None, (n, hi, h, w, c) if hi > 0 else None, x)
        ctx.state, ctx.dims = state, (lo, hi, d)
        return x.view(x.shape)

    @staticmethod
    def backward(ctx, gx_in):
        lo, hi, d = ctx.dims
        st = ctx.state
        if "bwait" not in st:           # autograd must have run _HaloFinish.backward first (it is the later forward call)
            raise RuntimeError("halo_conv: the edge gradients were not exchanged before the slab's gradient was finalised")
        st["bwait"]()
        gx = gx_in.clone()
        if st["bfp"] is not None:      # prev used my bottom `hi` planes as its high halo
            gx[:, :hi] += st["bfp"]
        if st["bfn"] is not None:      # next used my top `lo` planes as its low halo
            gx[:, d - lo:] += st["bfn"]
        return gx, None, None, None, None


class _HaloFinish(torch.autograd.Function):
    """Second half: called AFTER the interior conv has been enqueued.  Makes the current stream wait for the transfer and
    builds the two EDGE inputs of the depth-coupled conv: (lo halo planes from the previous rank ++ the slab's first
    ``n_lo`` planes, the slab's last ``n_hi`` planes ++ hi halo planes from the next rank); zeros stand in for a
    neighbour at the volume boundary (= the conv's zero padding).  Backward (it runs BEFORE the interior conv's backward:
    later forward call, higher autograd sequence number): the halo planes' gradients start travelling back to the ranks
    that own the planes on the side stream, the wait is left to ``_HaloStart.backward``."""

    @staticmethod
    def forward(ctx, x, lo, hi, n_lo, n_hi, shard, state):
        n, d, h, w, c = x.shape
        ctx.shard, ctx.dims, ctx.xshape, ctx.state = shard, (lo, hi, n_lo, n_hi), tuple(x.shape), state
        head, tail = x[:, :n_lo], x[:, d - n_hi:]
        state["wait"]()
        from_prev, from_next = state.pop("fp"), state.pop("fn")
        zl = from_prev if from_prev is not None else x.new_zeros((n, lo, h, w, c))
        zh = from_next if from_next is not None else x.new_zeros((n, hi, h, w, c))
        return torch.cat([zl, head], dim=1), torch.cat([tail, zh], dim=1)

    @staticmethod
    def backward(ctx, g_lo, g_hi):
        lo, hi, n_lo, n_hi = ctx.dims
        shard, st = ctx.shard, ctx.state
        n, d, h, w, c = ctx.xshape
        g_lo, g_hi = g_lo.contiguous(), g_hi.contiguous()
        send_prev = g_lo[:, :lo].contiguous() if lo > 0 else None         # gradient of prev's top planes
        send_next = g_hi[:, n_hi:].contiguous() if hi > 0 else None       # gradient of next's bottom planes
        st["bfp"], st["bfn"], st["bwait"] = _exchange_async(
            shard, send_prev, send_next, (n, hi, h, w, c) if hi > 0 else None, (n, lo, h, w, c) if lo > 0 else None, g_lo)
        gx = g_lo.new_zeros(ctx.xshape)
        gx[:, :n_lo] += g_lo[:, lo:]
        gx[:, d - n_hi:] += g_hi[:, :n_hi]
        return gx, None, None, None, None, None, None


def _exchange_async(ctx, send_prev, send_next, recv_prev_shape, recv_next_shape, like):
    """``_exchange`` issued on a side HIP stream (GPU tensors): returns (from_prev, from_next, wait) immediately; ``wait()``
    makes the CURRENT stream wait for the transfers.  Everything enqueued on the current stream between the call and
    ``wait()`` overlaps the xGMI transfer.  CPU tensors (gloo tier): plain blocking exchange."""
    if not like.is_cuda:
        fp, fn = _exchange(ctx, send_prev, send_next, recv_prev_shape, recv_next_shape, like)
        return fp, fn, (lambda: None)
    main = torch.cuda.current_stream(like.device)
    side = ops.side_stream(like.device, "halo")
    side.wait_stream(main)                       # the packed planes are complete
    with torch.cuda.stream(side):
        fp, fn = _exchange(ctx, send_prev, send_next, recv_prev_shape, recv_next_shape, like)
    for t in (send_prev, send_next, fp, fn):     # allocated on `main`, used on `side` (and back)
        if t is not None:
            t.record_stream(side)

    def wait():
        main.wait_stream(side)
    return fp, fn, wait


def halo_conv(x, run, kd, stride, pd, out_tail, shard=None):
    """A depth-coupled conv on this rank's slab x [1,d,H,W,C] WITHOUT building a padded copy of the slab: the outputs
    whose receptive field is local are computed straight from x while the halo planes are in flight on a side stream;
    the few boundary output planes are then computed from two small edge tensors (halo ++ the slab's first / last
    planes).  ``run(inp, out, z0, z1)`` runs the conv depth-VALID on ``inp``, writing output planes [z0, z1) into
    ``out`` (that dense depth range of the result buffer [1, d/stride, *out_tail]) and returns the tensor; the three
    results are joined without a copy.  Output plane j reads input planes [j*stride - pd, j*stride - pd + kd).
    Returns None when the split does not apply (N > 1, odd slab) -- the caller then uses the padded-slab path."""
    shard = shard or _CTX
    lo, hi = conv_depth_halo(kd, stride, pd)
    n, d = x.shape[0], x.shape[1]
    if shard is None or shard.world == 1 or n != 1 or d % stride or (lo == 0 and hi == 0):
        return None
    do = d // stride
    j_lo = -(-pd // stride)                                           # first output plane that is fully local
    j_hi = (d - kd + pd) // stride                                    # last one
    if j_hi < j_lo:                                                   # slab thinner than the kernel: edges only
        return None
    n_head = (j_lo - 1) * stride - pd + kd if j_lo > 0 else 0         # local planes the low edge reads
    first_hi = (j_hi + 1) * stride - pd                               # first local plane the high edge reads
    n_tail = d - first_hi if j_hi + 1 < do else 0
    if n_head > d or n_tail > d or first_hi < 0:
        return None
    state = {}
    x = _HaloStart.apply(x.contiguous(), lo, hi, shard, state)        # planes packed, transfer started on the side stream
    buf = x.new_empty((1, do) + tuple(out_tail))
    parts = []
    # interior first: it needs no halo, so it is enqueued on the main stream while the transfer is in flight ...
    mid = run(x[:, j_lo * stride - pd:j_hi * stride - pd + kd], buf[:, j_lo:j_hi + 1], j_lo, j_hi + 1)
    # ... and only now does the main stream wait for the halo planes (backward mirrors this: the gradient planes are sent
    # before the interior conv's backward is enqueued and awaited after it)
    edge_lo, edge_hi = _HaloFinish.apply(x, lo, hi, max(n_head, 0), max(n_tail, 0), shard, state)
    if j_lo > 0:
        parts.append(run(edge_lo, buf[:, 0:j_lo], 0, j_lo))
    parts.append(mid)
    if j_hi + 1 < do:
        parts.append(run(edge_hi, buf[:, j_hi + 1:do], j_hi + 1, do))
    return ops.join_depth(buf, parts)


def halo_exchange(x, lo, hi, shard=None):
    shard = shard or _CTX
    if shard is None or shard.world == 1 or (lo == 0 and hi == 0):
        return torch.nn.functional.pad(x, (0, 0, 0, 0, 0, 0, lo, hi)) if (lo or hi) else x
    return _HaloExchange.apply(x, lo, hi, shard)


def conv_depth_halo(kd, stride, pd):
    """(lo, hi) halo planes a slab needs for a conv with depth kernel kd / stride / padding pd, for slabs whose
    first plane is a multiple of the stride (the slabs of an evenly split, 16-divisible volume)."""
    if kd == 1:
        return 0, 0
    lo = pd
    hi = kd - 1 - pd - (stride - 1)
    return lo, max(hi, 0)


def slab(t, dim=2, shard=None):
    """This rank's depth slab of a full tensor (NCDHW: dim 2; NDHWC: dim 1)."""
    shard = shard or _CTX
    d = t.shape[dim]
    if d % shard.world:
        raise ValueError("depth %d does not split evenly over %d ranks" % (d, shard.world))
    s = d // shard.world
    return t.narrow(dim, shard.rank * s, s)


def gather_rpn_outputs(level_outputs, shard=None):
    """level_outputs: per pyramid level [logits [1,Al,2], probs [1,Al,2], bbox [1,Al,6]] of THIS rank's slab.
    One all-gather per tensor kind; returns the global tensors in the reference's order: all of level 2
    (slabs in rank order = (z,y,x) order), then level 3 (model.py:1424-1426)."""
    shard = shard or _CTX
    outs = []
    for kind in range(3):
        per_level = []
        for lv in level_outputs:
            t = lv[kind].contiguous()
            parts = [torch.empty_like(t) for _ in range(shard.world)]
            dist.all_gather(parts, t, group=shard.group)
            per_level.append(torch.cat(parts, dim=1))
        outs.append(torch.cat(per_level, dim=1))
    return outs


def gather_rpn_candidates(level_outputs, net, mode, shard=None):
    """The proposal set of a depth-sharded volume from ONE small all-gather: every rank
    takes the top PRE_NMS_LIMIT of its own anchors by foreground score and contributes (score, 6 raw box outputs, global
    anchor index) -- K x 8 floats, 188 KB at K = 6000 -- instead of its whole logits / probs / bbox tensors (one
    collective per tensor kind and level before).  The global top PRE_NMS_LIMIT is a subset of the union of the local
    ones, so after the merge (score descending, ties by anchor index) every rank runs the identical, deterministic
    decode / clip / NMS of ``model.proposal_layer``.  Returns rois [1, K', 6]."""
    from . import model as M
    shard = shard or _CTX
    cfg = net.config
    probs = torch.cat([lv[1] for lv in level_outputs], dim=1)[0].detach()          # [A/R, 2] local order: level 2, level 3
    bbox = torch.cat([lv[2] for lv in level_outputs], dim=1)[0].detach()           # [A/R, 6]
    counts = [lv[0].shape[1] * shard.world for lv in level_outputs]
    gidx = local_anchor_index(counts, shard).to(probs.device)
    n_local = probs.shape[0]
    k = min(int(cfg.PRE_NMS_LIMIT), n_local)
    sc, li = probs[:, 1].topk(k, sorted=True)
    pack = torch.empty((k, 8), dtype=torch.float32, device=probs.device)
    pack[:, 0] = sc
    pack[:, 1:7] = bbox[li]
    pack[:, 7] = gidx[li].to(torch.float32)         # (anchor counts are far below 2^24: exact in fp32)
    parts = [torch.empty_like(pack) for _ in range(shard.world)]
    dist.all_gather(parts, pack, group=shard.group)
    allc = torch.cat(parts, dim=0)
    idx = allc[:, 7].to(torch.long)
    # score descending, ties by ascending anchor index: sort by index first, then a stable sort by score
    o1 = torch.argsort(idx, stable=True)
    o2 = torch.argsort(allc[o1, 0], descending=True, stable=True)
    order = o1[o2][:min(int(cfg.PRE_NMS_LIMIT), sum(counts))]
    count = cfg.POST_NMS_ROIS_TRAINING if mode == "training" else cfg.POST_NMS_ROIS_INFERENCE
    return M.proposals_from_candidates(allc[order, 0], allc[order, 1:7], net.anchors[idx[order]], count, cfg.RPN_NMS_THRESHOLD, cfg)


def sharded_backbone_rpn(net, image_slab):
    """Depth-sharded FPN -> RPN -> proposals of ``cfun_amd.step.CFUNHotPath`` on this rank's slab
    [1,1,D/R,H,W].  Returns (p2_slab, p3_slab, rpn_logits, rpn_probs, rpn_bbox, rpn_rois) with the RPN tensors and
    the proposals global and identical on every rank."""
    p2, p3 = net.fpn.forward_ndhwc(ops.to_ndhwc(image_slab))
    local = [net.rpn.forward_ndhwc(p) for p in (p2, p3)]
    logits, probs, bbox = gather_rpn_outputs(local)         # (the full RPN tensors: this entry point returns them)
    rois = gather_rpn_candidates(local, net, "inference" if not net.training else "training")
    return p2, p3, logits, probs, bbox, rois


# ---------------------------------------------------------------------------------------------------------------
# One volume over R ranks: the whole training step (BASELINE.json configs[3])
# ---------------------------------------------------------------------------------------------------------------
class _AllReduceSum(torch.autograd.Function):
    """sum over the ranks, forward and backward (the ranks' additive shares of the RoI-aligned crops -> the crops; the
    ranks' gradients of the crops, each non-zero only for the RoIs that rank classifies -> the full gradient)."""

    @staticmethod
    def forward(ctx, x, shard):
        ctx.shard = shard
        y = x.contiguous().clone()
        dist.all_reduce(y, group=shard.group)
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        dist.all_reduce(g, group=ctx.shard.group)
        return g, None


def all_reduce_sum(x, shard=None):
    shard = shard or _CTX
    if shard is None or shard.world == 1:
        return x
    return _AllReduceSum.apply(x, shard)


# ---------------------------------------------------------------------------------------------------------------
# z-sharded per-RoI U-Net: with 4 positive RoIs only 4 ranks would run a mask head.  A
# RoI's U-Net is therefore split along depth over a SUB-GROUP of ranks at its two high-resolution levels (96^3 and
# 48^3: 84 % of its FLOPs) -- 1-plane halos per 3x3x3 conv, one all-reduce of 2*C floats per InstanceNorm -- while the
# levels at 24^3 and below, where a slab would be thinner than a conv's halo, are "folded": the 24^3 tensor is
# all-gathered once and those levels are computed redundantly on every rank of the sub-group (no communication).
#
#   sharded (slabs)  --GatherReplicated-->  replicated (whole tensor on every rank)  --EnterSlab-->  sharded
#
# Gradient bookkeeping (every rank finally SUMS its parameter gradients with the other ranks'): the replicated section
# is evaluated R times with identical values, so its gradient is scaled by 1/R where it enters (EnterSlab.backward:
# all-reduce of the slabs' contributions, divided by R) and scaled back by R where it leaves towards the sharded
# encoder (GatherReplicated.backward: own depth range times R).  Sum over ranks = the single-GPU gradient.
# ---------------------------------------------------------------------------------------------------------------
class _GatherReplicated(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, shard):
        ctx.shard = shard
        x = x.contiguous()
        parts = [torch.empty_like(x) for _ in range(shard.world)]
        dist.all_gather(parts, x, group=shard.group)
        return torch.cat(parts, dim=1)

    @staticmethod
    def backward(ctx, g):
        shard = ctx.shard
        d = g.shape[1] // shard.world
        return g.narrow(1, shard.rank * d, d) * float(shard.world), None


def gather_replicated(x, shard):
    """Depth slabs [N,d,H,W,C] -> the whole tensor on every rank of ``shard``, entering a replicated section."""
    return _GatherReplicated.apply(x, shard)


class _EnterSlab(torch.autograd.Function):
    @staticmethod
    def forward(ctx, full, lo, hi, shard):
        d = full.shape[1] // shard.world
        z0 = shard.rank * d
        ctx.shard, ctx.geom, ctx.shape = shard, (z0, d, lo, hi), tuple(full.shape)
        a, b = max(z0 - lo, 0), min(z0 + d + hi, full.shape[1])
        out = full[:, a:b]
        pad_lo, pad_hi = a - (z0 - lo), (z0 + d + hi) - b
        if pad_lo or pad_hi:                                   # volume boundary: the conv's zero padding
            out = torch.nn.functional.pad(out, (0, 0, 0, 0, 0, 0, pad_lo, pad_hi))
        return out.contiguous()

    @staticmethod
    def backward(ctx, g):
        shard = ctx.shard
        z0, d, lo, hi = ctx.geom
        full = g.new_zeros(ctx.shape)
        a, b = max(z0 - lo, 0), min(z0 + d + hi, ctx.shape[1])
        full[:, a:b] = g[:, a - (z0 - lo):a - (z0 - lo) + (b - a)]
        dist.all_reduce(full, group=shard.group)
        return full / float(shard.world), None, None, None


def enter_slab(full, shard, lo=0, hi=0):
    """Replicated tensor [N,D,H,W,C] -> this rank's depth slab plus ``lo`` / ``hi`` halo planes (zeros beyond the
    volume), leaving a replicated section for a sharded one; no communication in forward (every rank holds the planes)."""
    return _EnterSlab.apply(full, lo, hi, shard)


def local_anchor_index(level_counts, shard=None):
    """Global flat indices of the anchors this rank's slabs produce, in local order (its level-2 block, then its
    level-3 block): level l's A_l anchors are flattened (z, y, x) (model.py:727-729), so a depth slab owns the
    contiguous range [off_l + r*A_l/R, off_l + (r+1)*A_l/R)."""
    shard = shard or _CTX
    idx, off = [], 0
    for a in level_counts:
        per = a // shard.world
        idx.append(torch.arange(off + shard.rank * per, off + (shard.rank + 1) * per))
        off += a
    return torch.cat(idx)


_ZPLANS = {}


def prepare_zshard_groups(group=None, device=None):
    """Create, ONCE and for every possible number of positive RoIs, the sub-groups ``zshard_plan`` hands out: for each
    divisor n of the group's size R (n < R) the n groups of R / n consecutive ranks.  ``dist.new_group`` is collective
    over the DEFAULT process group -- every rank of the job must call this (same ``group`` argument), at set-up time;
    creating groups lazily inside a step, keyed on the data-dependent RoI count, stalls all ranks mid-step and hangs
    when ``group`` is a strict sub-group (hybrid data-parallel x depth-sharded layouts) because the outside ranks never
    reach the call.  ``device``: this rank's device -- every sub-group it belongs to then runs one tiny all-reduce
    here, so that RCCL builds the communicators at set-up (with ALL members present) and not inside the first step, where
    a sub-group's first operation would be the batched halo send/recv of whichever ranks get there first."""
    world = dist.get_world_size(group)
    base = dist.get_process_group_ranks(group) if group is not None else list(range(world))
    me = dist.get_rank()
    plans = {}
    for n_pos in range(1, world):
        if world % n_pos:
            continue
        rs = world // n_pos
        groups = []
        for i in range(n_pos):
            ranks = base[i * rs:(i + 1) * rs]
            g = dist.new_group(ranks=ranks)
            groups.append(g)
            if device is not None and me in ranks:
                warm = torch.zeros(1, dtype=torch.float32, device=device)
                dist.all_reduce(warm, group=g)
        plans[n_pos] = (rs, groups)
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.current_stream(device).synchronize()
    _ZPLANS[(id(group), world)] = plans
    return plans


def zshard_plan(shard, n_pos):
    """Sub-groups for z-sharded mask heads: with more ranks than positive RoIs (8 GPUs, 4 RoIs) every RoI's U-Net is
    split over world / n_pos consecutive ranks.  Returns (ranks per RoI, [process group per RoI]) or None when the ranks
    do not divide evenly (then RoIs are dealt round-robin, one whole U-Net per rank).  The groups come from
    ``prepare_zshard_groups``; when the job has not called it, they are created on first use -- allowed only while
    ``shard.group`` spans the whole job (then every rank is here and the collective ``new_group`` calls line up)."""
    if n_pos <= 0 or shard.world <= n_pos or shard.world % n_pos:
        return None
    key = (id(shard.group), shard.world)
    if key not in _ZPLANS:
        if shard.group is not None and shard.world != dist.get_world_size():
            raise RuntimeError("zshard_plan: call cfun_amd.dist.prepare_zshard_groups(group) on EVERY rank of the job at "
                               "set-up -- new_group is collective over the default group and cannot be created from "
                               "inside a sub-group's step")
        prepare_zshard_groups(shard.group)
    return _ZPLANS[key].get(n_pos)


def rank_dropout_masks(masks, n_pos, shard=None, zshard_unet=True):
    """This rank's rows of preset Dropout3d masks (five [n_pos, C] tensors, one row per positive RoI) for
    ``sharded_training_step``: the row of the RoI its z-shard sub-group shares, or the rows of its round-robin RoIs."""
    shard = shard or _CTX or ShardContext()
    if masks is None:
        return None
    plan = zshard_plan(shard, n_pos) if zshard_unet else None
    if plan is not None:
        roi = shard.rank // plan[0]
        return [m[roi:roi + 1] for m in masks]
    return [m[:n_pos][shard.rank::shard.world] for m in masks]


def sharded_training_step(net, s, shard=None, zshard_unet=True, dropout_seed=0):
    """ONE volume on R ranks: forward + the 6 losses + backward of ``cfun_amd.step.training_step`` with

    * FPN / RPN depth-sharded (halo exchange inside the depth-coupled convs), proposals from one all-gather;
    * RPN losses on the rank's own anchors with the global normalisation (so local gradients are exact);
    * head RoIs dealt round-robin: rank r classifies rois[r::R] -- on RoI-aligned crops every rank computes on its own
      p2 / p3 slabs and ONE all-reduce sums (``ops.roi_align(slab=...)``) -- and runs the mask U-Net on p_rois[r::R]
      (crops of the raw image, which every rank holds);
    * with more ranks than positive RoIs (``zshard_unet``; 8 GPUs, 4 RoIs): every RoI's U-Net z-sharded over
      world / n_pos ranks (``zshard_plan``, ``Modified3DUNet.forward_ndhwc(zshard=...)``), the mask losses evaluated on
      the slabs (cross entropy voxel-local, the Sobel edge loss with one halo plane of probabilities from each neighbour);
    * every loss returned as THIS rank's additive share: sum over ranks = the single-GPU loss, and the sum over
      ranks of the parameter gradients = the single-GPU gradient (all-reduce them with op=SUM).

    ``s`` is the replicated sample of ``step.synthetic_inputs`` (every rank holds the full image and targets).
    Returns (local losses list, local total)."""
    import torch.nn.functional as F
    from . import model as M
    shard = shard or _CTX
    R, r = shard.world, shard.rank
    cfg = net.config
    net.train()
    image = s["image"]
    p2s, p3s = net.fpn.forward_ndhwc(ops.to_ndhwc(slab(image, dim=2, shard=shard)))
    local = [net.rpn.forward_ndhwc(p) for p in (p2s, p3s)]
    with torch.no_grad():      # one 188 KB all-gather of the ranks' top candidates; identical proposals on every rank
        rpn_rois = gather_rpn_candidates(local, net, "training", shard)

    # The LiTS fork trains in two phases (LiTSConfig.STAGE_SPLIT; LiTS_2017/model.py:985-1001, 1518-1548), as step.compute_losses
    # does: 'beginning' = detector only (no mask head, both mask losses 0), any other stage = mask branch only (FPN / RPN frozen,
    # no classifier head, the four detector losses 0).  ADVICE round 5: this function used to run every head in every phase.
    det_on, mask_on = not net.mask_phase_only, not net.detector_phase_only
    rois = torch.cat([s["p_rois"], s["n_rois"]], dim=0)
    n_all, n_pos = rois.shape[0], s["p_rois"].shape[0]
    zero = torch.zeros((), dtype=torch.float32, device=image.device)
    l_rpn_cls = l_rpn_box = l_cls = l_box = zero
    if det_on:
        # ---- RPN losses on the local anchors (model.py:808-860), global normalisation
        counts = [lv[0].shape[1] * R for lv in local]
        lidx = local_anchor_index(counts, shard).to(image.device)
        m = s["rpn_match"].squeeze(2)[0]                                  # [A] global
        logits_l = torch.cat([lv[0] for lv in local], dim=1)[0]           # [A/R, 2]
        bbox_l = torch.cat([lv[2] for lv in local], dim=1)[0]             # [A/R, 6]
        m_l = m[lidx]
        nz, npos = int((m != 0).sum()), int((m == 1).sum())
        sel = torch.nonzero(m_l != 0)[:, 0]
        l_rpn_cls = F.cross_entropy(logits_l[sel], (m_l[sel] == 1).long(), reduction="sum") / max(nz, 1) if sel.numel() \
            else logits_l.sum() * 0.0
        pos_rank = torch.cumsum((m == 1).long(), 0) - 1                    # k-th positive <-> target row k
        psel = torch.nonzero(m_l == 1)[:, 0]
        l_rpn_box = F.smooth_l1_loss(bbox_l[psel], s["rpn_bbox_t"][0, pos_rank[lidx[psel]]], reduction="sum") / max(npos * 6, 1) \
            if psel.numel() else bbox_l.sum() * 0.0

        # ---- heads on this rank's share of the RoIs.  The classifier's RoIAlign crosses slabs: every rank aligns ALL RoIs on
        # its own p2 / p3 slabs (planes it does not hold count as zeros -- RoIAlign is linear in the map), ONE all-reduce adds
        # the shares up (12 crops: 10.6 MB, instead of all-gathering p2 + p3: 66 MB forward, 2 x 66 MB backward at
        # 512x512x256), and the backward all-reduces the crops' gradients the same way before each rank scatters into its slabs.
        mine = torch.arange(n_all, device=rois.device)[r::R]         # (empty on a rank beyond the RoI count: world-4 test)
        slabs = tuple((r * t.shape[1], R * t.shape[1]) for t in (p2s, p3s))
        crops = all_reduce_sum(M.pyramid_roi_align_ndhwc(rois.detach(), [p2s[0], p3s[0]], net.classifier.pool_size, slabs), shard)
        # Every rank must run the crops' backward (an all-reduce): a rank that holds no RoI at all when R > number of RoIs
        # would otherwise skip a collective its peers issue.  `zero` touches the crops and is added to the total unconditionally.
        zero = crops.sum() * 0.0
        l_cls = l_box = zero
        if mine.numel():
            cls_logits, _, cls_bbox = net.classifier.head_ndhwc(crops[mine])
            tcls = s["target_class_ids"][mine]
            l_cls = F.cross_entropy(cls_logits, (tcls > 0).long(), reduction="sum") / n_all
            pos = torch.nonzero(tcls > 0)[:, 0]
            npos_all = int((s["target_class_ids"] > 0).sum())
            if pos.numel():
                l_box = F.smooth_l1_loss(cls_bbox[pos, 1, :], s["target_deltas"][mine][pos], reduction="sum") / (npos_all * 6)
    pmine = torch.arange(n_pos if mask_on else 0, device=rois.device)[r::R]
    l_mask = l_edge = zero
    plan = zshard_plan(shard, n_pos) if (zshard_unet and mask_on) else None
    unet = net.mask.modified_u_net
    # the mask losses of the configuration (step.CFUNHotPath._mask_losses): heart -- CE, and in 'finetune' the Sobel-magnitude
    # edge loss; LiTS fork -- class-weighted CE, raw-Sobel edge loss in every non-'beginning' stage (LiTS_2017/model.py:907-1001)
    cw = getattr(cfg, "MASK_CE_CLASS_WEIGHTS", None)
    raw = bool(getattr(cfg, "EDGE_LOSS_RAW_SOBEL", False))
    edge_on = (cfg.STAGE != "beginning") if (getattr(cfg, "STAGE_SPLIT", False) or raw) else cfg.STAGE == "finetune"
    edge_fn = ops.edge_loss_raw if raw else ops.edge_loss
    if cw is not None:
        cw_t = torch.tensor([float(v) for v in cw], dtype=torch.float32, device=image.device)
        cw_den = cw_t[s["mask_labels"][:n_pos].long()].sum()       # the labels are replicated: no collective needed
    if plan is not None:        # this rank is one of `rs` ranks that share RoI `roi`
        rs, groups = plan
        roi = r // rs
        zs = ShardContext(groups[roi])
        preset = unet.dropout_masks
        if unet.training and unet.dropout_p > 0 and preset is None:
            # Dropout3d masks must agree inside the sub-group: drawn for ALL RoIs from a generator every rank seeds alike
            b, keep = unet.base_n_filter, 1.0 - unet.dropout_p
            gen = torch.Generator().manual_seed(int(dropout_seed))
            unet.dropout_masks = [torch.empty((n_pos, c)).bernoulli_(keep, generator=gen).div_(keep)[roi:roi + 1]
                                  for c in (b, 2 * b, 4 * b, 8 * b, 16 * b)]
        try:
            crop = ops.roi_align(ops.to_ndhwc(image)[0], s["p_rois"][roi:roi + 1].detach(), net.mask.pool_size)[0]
            with slab_local():
                mslab = unet.forward_ndhwc(slab(crop, dim=1, shard=zs).contiguous(), zshard=zs)
        finally:
            unet.dropout_masks = preset
        # The mask losses on the rank's own slab (no 113 MB logits all-gather): the cross entropy is voxel-local; the Sobel
        # edge loss (a valid 3x3x3 stencil on the probabilities) takes ONE halo plane from each neighbour -- this rank
        # evaluates the output planes whose centre plane it owns -- and both come back weighted by the rank's share of
        # the voxels / output planes, so that the plain sum over the ranks is the whole RoI's loss.
        dl = mslab.shape[1]
        Dm = dl * rs
        z0 = zs.rank * dl
        lab_full = s["mask_labels"][roi]                                   # [D,H,W] uint8, replicated
        share = 1.0 / float(n_pos)
        lab_slab = lab_full[z0:z0 + dl].unsqueeze(0).contiguous()
        if cw is None:
            l_mask = ops.mask_cross_entropy(mslab, lab_slab) * (share * dl / Dm)
        else:       # class-weighted CE = sum(w * nll) / sum(w) over ALL RoIs' voxels: this slab's numerator over the global denominator
            l_mask = ops.mask_cross_entropy(mslab, lab_slab, weight=cw) * (cw_t[lab_slab.long()].sum() / cw_den)
        pmine = torch.tensor([roi], device=rois.device)
        if edge_on:
            mprob = ops.softmax_channels(mslab)
            ph = halo_exchange(mprob, 1, 1, zs)                            # [1, dl+2, ...]; zeros beyond the volume
            a, b = max(z0 - 1, 0), min(z0 + dl - 1, Dm - 2)                # output planes [a, b): inputs [a, b + 2)
            if b > a:
                psl = ph[:, a - (z0 - 1):b + 2 - (z0 - 1)].contiguous()
                l_edge = edge_fn(psl, lab_full[a:b + 2].unsqueeze(0).contiguous()) * (share * (b - a) / (Dm - 2))
            else:
                l_edge = ph.sum() * 0.0
    elif pmine.numel():
        fused = net.fused_mask_losses() and edge_on       # heart 'finetune': one loss pass each way (ops.mask_losses_fused)
        with slab_local():      # the U-Net's 3x3x3 convs see whole RoI crops, not depth slabs
            mlog, mprob = net.mask.forward_ndhwc(ops.to_ndhwc(image)[0], s["p_rois"][pmine], softmax=not fused)
        labels = s["mask_labels"][pmine].contiguous()
        share = pmine.numel() / float(n_pos)
        if fused:
            ce, edge, mprob = ops.mask_losses_fused(mlog, labels)
            l_mask, l_edge = ce * share, edge * share
        elif cw is None and not raw and edge_on:    # ... or CE + edge with the separate forward kernels, one fused backward
            ce, edge = ops.mask_losses(mlog, mprob, labels)
            l_mask, l_edge = ce * share, edge * share
        else:
            if cw is None:
                l_mask = ops.mask_cross_entropy(mlog, labels) * share
            else:
                l_mask = ops.mask_cross_entropy(mlog, labels, weight=cw) * (cw_t[labels.long()].sum() / cw_den)
            if edge_on:
                l_edge = edge_fn(mprob, labels) * share
    losses = [l_rpn_cls, l_rpn_box, l_cls, l_box, l_mask, l_edge]
    total = net.total_loss(losses) + zero
    if total.requires_grad:         # (mask phase on a rank that holds no positive RoI: nothing to back-propagate)
        total.backward()
    return losses, total, rpn_rois


# ---------------------------------------------------------------------------------------------------------------
# Data-parallel gradient reduction (SURVEY.md section 8(e) "Gradients" / "Whole-step alternative")
# ---------------------------------------------------------------------------------------------------------------
class GradientReducer:
    """Average the parameter gradients of data-parallel replicas (one volume per GPU), overlapped with backward.

    MI355X-first choices: the gradients live in a few large flat buckets (``p.grad`` are views, so there is no
    gather copy and each collective moves tens of MB -- xGMI rings are per-link bound, small messages waste them);
    a bucket's all-reduce is issued on a side HIP stream the moment autograd has accumulated its last gradient
    (post-accumulate hooks), so RCCL runs under the rest of the backward pass.  The 113 MB classifier weight is
    reduced while the U-Net and the backbone are still back-propagating.

        red = GradientReducer(net.parameters())
        per step:  red.zero_grad(); loss.backward(); red.finish()      # then p.grad holds the mean over ranks
                                                                       # (average=False: the sum, for sharded_training_step)

    Buckets are filled in reverse parameter order (heads first), the order autograd produces them in.  A parameter
    that gets no gradient in a step still takes part in its bucket's reduction (its slice stays zero); ``finish``
    launches whatever bucket did not complete through the hooks.  With world size 1 it is a no-op container
    (``always_reduce`` keeps the collectives for single-rank tests of the stream logic)."""

    def __init__(self, params, bucket_bytes=64 << 20, group=None, always_reduce=False, average=True, own_group=True):
        self.average = average      # False: plain sum (the ranks hold additive shares of ONE sample's gradient)
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.active = self.world > 1 or (always_reduce and dist.is_available() and dist.is_initialized())
        # The bucket all-reduces are launched from autograd hooks, i.e. interleaved with whatever collectives the
        # backward pass itself issues (sharded_training_step: halo send/recv, the p2/p3 all-reduces).  Ranks whose
        # graphs differ (no positive RoI on one rank) reach the hooks at different points of that sequence, and
        # collectives of ONE communicator must be issued in the same order everywhere -- so the reducer gets a
        # communicator of its own (same ranks), on which only its own, index-ordered launches run.
        if self.active and own_group and self.world > 1:
            ranks = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
            group = dist.new_group(ranks=ranks)       # collective: every rank of `group` constructs the reducer
        self.group = group
        self.hold = False           # True: backward passes only accumulate (no collective is launched from the hooks)
        self._next = 0              # buckets are launched strictly in index order (see _on_grad)
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []       # dicts: flat, params, pending, ready, work
        order = list(reversed(self.params))
        cur, cur_bytes = [], 0
        for p in order:
            nbytes = self._padded(p.numel()) * p.element_size()
            if cur and cur_bytes + nbytes > bucket_bytes:
                self._add_bucket(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._add_bucket(cur)
        self._where = {}        # id(param) -> (bucket, slot)
        for b, bucket in enumerate(self.buckets):
            for i, p in enumerate(bucket["params"]):
                self._where[id(p)] = (b, i)
        self.comm_stream = None
        self._main_stream = None    # the stream the step is enqueued on (zero_grad / arm record it)
        if self.params and self.params[0].is_cuda:
            self.comm_stream = torch.cuda.Stream(device=self.params[0].device)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params] if self.active else []
        if self.active and self.buckets:
            # create the communicator here, on the caller's thread, not inside the first backward's autograd hook
            warm = torch.zeros(1, dtype=self.buckets[0]["flat"].dtype, device=self.buckets[0]["flat"].device)
            dist.all_reduce(warm, group=self.group)
            if warm.is_cuda:
                torch.cuda.current_stream(warm.device).synchronize()

    ALIGN = 64   # elements: every view starts on a 256-byte boundary (the kernels use 16-byte vector accesses)

    @classmethod
    def _padded(cls, n):
        return (n + cls.ALIGN - 1) // cls.ALIGN * cls.ALIGN

    def _add_bucket(self, plist):
        n = sum(self._padded(p.numel()) for p in plist)
        flat = torch.zeros(n, dtype=plist[0].dtype, device=plist[0].device)
        off = 0
        views, offsets = [], []
        for p in plist:
            views.append(flat[off:off + p.numel()].view_as(p))
            offsets.append(off)
            off += self._padded(p.numel())      # the padding stays zero: neutral for sums, norms and updates
        self.buckets.append(dict(flat=flat, params=plist, views=views, offsets=offsets, pending=len(plist), work=None,
                                 launched=False))

    def zero_grad(self):
        """Zero the buckets and (re)attach ``p.grad`` to its bucket view; call instead of ``net.zero_grad()``."""
        self._next = 0
        self.hold = False
        self._record_main()
        for bucket in self.buckets:
            bucket["flat"].zero_()
            bucket["pending"], bucket["work"], bucket["launched"] = len(bucket["params"]), None, False
            for p, v in zip(bucket["params"], bucket["views"]):
                p.grad = v

    def arm(self, sync=True):
        """Before a backward pass that ACCUMULATES into buckets an earlier pass already filled (gradient accumulation over
        BATCH_SIZE samples, model.py:1640-1645): ``sync=False`` -- this pass launches no collective (the hooks stay
        quiet); ``sync=True`` -- the last pass of the batch: the hooks count this pass's gradients afresh and launch each
        bucket, holding the accumulated sums, once it is complete.  ``zero_grad`` arms for a single synchronising pass."""
        if self.in_flight():
            raise RuntimeError("GradientReducer.arm: a reduction is in flight (finish() the previous batch first)")
        self.hold = not sync
        self._next = 0
        self._record_main()
        for bucket in self.buckets:
            bucket["pending"], bucket["work"], bucket["launched"] = len(bucket["params"]), None, False

    def _record_main(self):
        if self.comm_stream is not None:
            self._main_stream = torch.cuda.current_stream(self.params[0].device)

    def in_flight(self):
        """Has any bucket's all-reduce been launched since the last zero_grad() / arm() (and not been finished)?"""
        return bool(self.active and any(b["launched"] for b in self.buckets))

    def _launch(self, bucket):
        bucket["launched"] = True
        if not self.active:
            return
        flat = bucket["flat"]
        if self.comm_stream is not None:
            # The gradients are complete once EVERY stream that produced one of the bucket's slices has got this far: the
            # hook of the bucket's last gradient may fire under a side stream (autograd replays the mask head's nodes on
            # the stream of their forward pass) while other slices of the same bucket came from the step's main stream --
            # so wait for the current stream, for the stream the step runs on (recorded by zero_grad / arm) and for every
            # side stream (without the main stream only timing ordered the all-reduce behind those kernels)
            cur = torch.cuda.current_stream(flat.device)
            self.comm_stream.wait_stream(cur)
            for s in [self._main_stream] + list(ops.side_streams(flat.device)):
                if s is not None and s != cur:
                    self.comm_stream.wait_stream(s)
            with torch.cuda.stream(self.comm_stream):
                bucket["work"] = dist.all_reduce(flat, group=self.group, async_op=True)
        else:
            bucket["work"] = dist.all_reduce(flat, group=self.group, async_op=True)

    def _on_grad(self, p):
        b, slot = self._where[id(p)]
        bucket = self.buckets[b]
        if p.grad is None or p.grad.data_ptr() != bucket["views"][slot].data_ptr():
            raise RuntimeError("GradientReducer: call zero_grad() of the reducer before backward()")
        if self.hold:
            return
        if bucket["launched"]:
            raise RuntimeError("GradientReducer: a gradient arrived for a bucket whose all-reduce is already in flight -- "
                               "a second backward() without arm() / zero_grad() (gradient accumulation: arm(sync=False) "
                               "for the passes that only accumulate, arm(sync=True) for the last)")
        bucket["pending"] -= 1
        # Fixed launch order, as DDP does: bucket i goes out only once buckets 0..i-1 have.  A rank on which some
        # bucket never completes through the hooks (its heads were skipped: no positive proposal / no mask RoI on
        # this rank) defers that bucket and all later ones to finish(); the sequence of collectives is the same on
        # every rank either way, only the overlap with backward is lost behind the gap.
        while self._next < len(self.buckets) and self.buckets[self._next]["pending"] == 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def finish(self):
        """Complete every bucket's reduction and scale to the mean; the compute stream waits for the comm stream."""
        if not self.active:
            return
        self.hold = False
        for bucket in self.buckets[self._next:]:          # the rest, still in index order
            self._launch(bucket)
        self._next = len(self.buckets)
        for bucket in self.buckets:
            bucket["work"].wait()
        if self.comm_stream is not None:
            torch.cuda.current_stream(self.params[0].device).wait_stream(self.comm_stream)
        if self.average:
            for bucket in self.buckets:
                bucket["flat"].div_(self.world)
        for bucket in self.buckets:                       # reduced: nothing is in flight any more
            bucket["launched"] = False

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
