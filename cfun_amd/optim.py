"""The optimizer tail of the reference's ``train_epoch`` (model.py:1538-1545, 1641-1645) as flat-arena HIP kernels:

    torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
    optim.SGD([{params: trainables without 'bn' in the name, weight_decay: WEIGHT_DECAY},
               {params: trainables with 'bn' in the name}], lr, momentum=LEARNING_MOMENTUM).step()

MI355X-first layout: the trainable parameters, their gradients and the momentum live in a few large flat fp32
arenas (the gradient arenas are the buckets of ``cfun_amd.dist.GradientReducer``, so data-parallel averaging,
clipping and the update all work on the same memory); ``p.data`` / ``p.grad`` are views, the ``nn.Module`` and its
state dict are unchanged.  A step is: one sum-of-squares launch per arena, one finalize (the global gradient norm
stays on the device -- no ``.item()`` sync as in clip_grad_norm_), one fused clip + weight-decay + momentum +
update launch per arena, instead of ~8 tiny launches per parameter tensor.

On this path every BatchNorm parameter is frozen (model.py:1297-1304), so the 'bn' group of the reference is empty
and the weight decay is uniform over the arena; a trainable parameter with 'bn' in its name is rejected rather
than silently decayed.

torch.optim.SGD skips a parameter whose ``.grad`` is None -- no weight decay, no momentum update: the layers a stage
does not run (the 'finetune'-only up-convs during 'beginning', a skipped head) stay exactly as they are.  Here every
``p.grad`` is a view of an arena and never None, so the optimizer records which parameters a backward pass reached
(post-accumulate hooks) and puts the untouched ones -- and their momentum -- back after the fused launch.  With
data-parallel ranks a parameter untouched here may have a gradient elsewhere: the flags are combined (MAX) over the
ranks, a parameter is left alone only where no rank reached it.
"""
import torch

from . import _lib, ops
from . import dist as cdist
from ._lib import check, ptr, stream


class FlatSGD:
    def __init__(self, named_params, lr, momentum=0.9, weight_decay=1e-4, clip_norm=5.0, bucket_bytes=64 << 20,
                 group=None):
        named = [(n, p) for n, p in named_params if p.requires_grad]
        for n, _ in named:
            if "bn" in n:
                raise ValueError("FlatSGD: trainable BatchNorm parameter %r (the reference gives it no weight decay; "
                                 "this path keeps BatchNorm frozen)" % n)
        self.lr, self.momentum, self.weight_decay, self.clip_norm = float(lr), float(momentum), float(weight_decay), clip_norm
        self.reducer = cdist.GradientReducer([p for _, p in named], bucket_bytes=bucket_bytes, group=group)
        self.param_arenas, self.momentum_arenas = [], []
        self._slots = []                             # (parameter, arena index, offset) for the untouched-parameter restore
        for a, bucket in enumerate(self.reducer.buckets):          # parameters move into arenas laid out like the gradient buckets
            flat = torch.zeros_like(bucket["flat"])
            with torch.no_grad():
                for p, off in zip(bucket["params"], bucket["offsets"]):
                    view = flat[off:off + p.numel()].view_as(p)
                    view.copy_(p.data)
                    p.data = view
                    self._slots.append((p, a, off))
            self.param_arenas.append(flat)
            self.momentum_arenas.append(torch.zeros_like(flat))
        self._touched = [False] * len(self._slots)
        self._hooks = [p.register_post_accumulate_grad_hook(lambda _p, i=i: self._touched.__setitem__(i, True))
                       for i, (p, _, _) in enumerate(self._slots)]
        dev = self.param_arenas[0].device if self.param_arenas else torch.device("cpu")
        lib = _lib.load()
        self._npart = int(lib.cfun_sumsq_partials_count())
        self._partials = torch.zeros(max(1, len(self.param_arenas)) * self._npart, dtype=torch.float64, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)     # total norm of the last step (device)
        self.steps = 0

    def zero_grad(self):
        self.reducer.zero_grad()
        self._touched = [False] * len(self._slots)

    def begin_backward(self, last=True):
        """Gradient accumulation over several backward passes per step (BATCH_SIZE > 1, model.py:1640-1645) with
        data-parallel ranks: the passes that only accumulate launch no all-reduce (``last=False``); the last one
        (``last=True``) reduces each bucket -- the accumulated sums -- as soon as ITS gradients have arrived.  Without
        ranks to reduce over this is a no-op."""
        if self.reducer.active:
            self.reducer.arm(sync=last)

    @torch.no_grad()
    def clip_(self):
        """torch.nn.utils.clip_grad_norm_(parameters, clip_norm) on the gradient arenas, in place -- what the reference does
        after EVERY backward (model.py:1641), also the ones that only accumulate (BATCH_SIZE > 1); ``step`` has the clip
        of the backward it follows fused in.  The norm stays on the device (``grad_norm``)."""
        clip = float(self.clip_norm) if self.clip_norm else 0.0
        if clip <= 0.0:
            return
        if self.reducer.in_flight():      # (the arenas are being all-reduced on the communication stream)
            raise RuntimeError("FlatSGD.clip_: a gradient all-reduce is in flight -- with data-parallel ranks the passes that "
                               "only accumulate must not synchronise: begin_backward(last=False) before them "
                               "(cfun_amd.train.train_epoch does); the clip then acts on the rank's own accumulated gradient")
        lib = _lib.load()
        for i, bucket in enumerate(self.reducer.buckets):
            g = bucket["flat"]
            check(lib.cfun_sumsq_partials(ptr(g), g.numel(), ptr(self._partials[i * self._npart:]), stream(g)), "sumsq_partials")
        check(lib.cfun_norm_finalize(ptr(self._partials), self._partials.numel(), ptr(self.grad_norm), stream(self.grad_norm)),
              "norm_finalize")
        coef = (clip / (self.grad_norm + 1e-6)).clamp(max=1.0)
        for bucket in self.reducer.buckets:
            bucket["flat"].mul_(coef)

    @torch.no_grad()
    def step(self):
        """Average over data-parallel ranks (if any), clip to ``clip_norm`` by the global L2 norm, SGD update."""
        self.reducer.finish()
        # the gradients were produced on several streams (mask head, weight gradients): the update waits for all of them
        # explicitly instead of relying on the autograd engine's join at the end of backward()
        if self.param_arenas and self.param_arenas[0].is_cuda:
            dev = self.param_arenas[0].device
            cur = torch.cuda.current_stream(dev)
            for st in ops.side_streams(dev):
                if st != cur:
                    cur.wait_stream(st)
        lib = _lib.load()
        clip = float(self.clip_norm) if self.clip_norm else 0.0
        if clip > 0.0:
            for i, bucket in enumerate(self.reducer.buckets):
                g = bucket["flat"]
                check(lib.cfun_sumsq_partials(ptr(g), g.numel(), ptr(self._partials[i * self._npart:]), stream(g)),
                      "sumsq_partials")
            check(lib.cfun_norm_finalize(ptr(self._partials), self._partials.numel(), ptr(self.grad_norm),
                                         stream(self.grad_norm)), "norm_finalize")
        # Parameters no backward pass reached since zero_grad(): torch.optim.SGD leaves them (and their momentum) alone.  With
        # data-parallel ranks a parameter is left alone only if NO rank reached it -- which implies this rank did not, so the
        # candidates are known on the host; whether a candidate was reached elsewhere stays on the DEVICE (flags all-reduced
        # with MAX, then torch.where): no host read between the bucket reductions and the update.
        cand = [i for i, hit in enumerate(self._touched) if not hit]
        flags = None
        if self.reducer.active and self._slots:
            import torch.distributed as tdist
            from . import hostio
            flags = hostio.upload(torch.tensor([1 if t else 0 for t in self._touched], dtype=torch.int32),
                                  self.param_arenas[0].device)
            tdist.all_reduce(flags, op=tdist.ReduceOp.MAX, group=self.reducer.group)
        keep = []
        for i in cand:
            p, a, off = self._slots[i]
            m = self.momentum_arenas[a][off:off + p.numel()]
            keep.append((i, p, m, p.data.clone(), m.clone()))
        for bucket, p, m in zip(self.reducer.buckets, self.param_arenas, self.momentum_arenas):
            g = bucket["flat"]
            check(lib.cfun_sgd_momentum_step(ptr(p), ptr(g), ptr(m), p.numel(), self.lr, self.momentum,
                                             self.weight_decay, clip, ptr(self.grad_norm) if clip > 0.0 else None,
                                             1 if self.steps == 0 else 0, stream(p)), "sgd_momentum_step")
        for i, p, m, p0, m0 in keep:
            if flags is None:
                p.data.copy_(p0)
                m.copy_(m0)
            else:                   # reached on another rank: keep the update
                hit = flags[i] > 0
                p.data.copy_(torch.where(hit, p.data, p0))
                m.copy_(torch.where(hit, m, m0.view_as(m)))
        self.steps += 1
