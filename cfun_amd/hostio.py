"""Host-side plumbing of the step that never blocks the host: pinned asynchronous uploads, scalars read back late, the side
streams (split out of ops.py in round 4; ``cfun_amd.ops`` re-exports every name)."""
import torch


# ---- host -> device uploads that never block the host -----------------------------------------------------------------
class _UploadRing:
    """Small per-step host tensors (Dropout3d masks and kept-channel lists, weight-preparation tables) go to the device
    through a ring of PERSISTENT pinned staging buffers with an asynchronous copy.  A pageable ``tensor.to(device)`` is a
    blocking copy in stream order: the host stops until the GPU has drained everything queued before it -- the previous
    step's backward -- and the GPU then idles until the host has caught up (tools/gap_report.py: ~2.7 ms of gaps per step
    before).  A slot is reused only after the copy that read it has completed (its event; by then long done)."""

    SLOTS = 8

    def __init__(self, device):
        self.device, self.slots, self.i = device, [[None, None] for _ in range(self.SLOTS)], 0

    def upload(self, host):
        host = host.contiguous()
        nbytes = host.numel() * host.element_size()
        out = torch.empty(host.shape, dtype=host.dtype, device=self.device)
        if nbytes == 0:
            return out
        slot = self.slots[self.i]
        self.i = (self.i + 1) % self.SLOTS
        if slot[1] is not None:
            slot[1].synchronize()
        if slot[0] is None or slot[0].numel() < nbytes:
            slot[0] = torch.empty(max(2 * nbytes, 1 << 16), dtype=torch.uint8).pin_memory()
        stage = slot[0][:nbytes].view(host.dtype).view(host.shape)
        stage.copy_(host)
        out.copy_(stage, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        slot[1] = ev
        return out


class AsyncScalar:
    """A small device tensor on its way to the host: the copy into a persistent pinned buffer is enqueued now, ``get()``
    waits only for THAT copy (an event), not for whatever was enqueued after it.  The 16 pinned buffers form a ring; an
    instance whose buffer is about to be handed to a newer one first moves its value into private memory (``_evict``), so a
    caller may defer any number of scalars (per-step loss logging) and still read each one's own value."""

    _ring, _i = [], 0

    def __init__(self, t):
        cls = AsyncScalar
        if t.is_cuda:
            import weakref
            if len(cls._ring) < 16:
                cls._ring.append([torch.empty(64, dtype=torch.int64).pin_memory(), None])
            slot = cls._ring[cls._i % len(cls._ring)]
            cls._i += 1
            if slot[1] is not None:
                owner = slot[1]()
                if owner is not None:
                    owner._evict()
            n = t.numel() * t.element_size()
            if n > slot[0].numel() * 8:
                raise RuntimeError("AsyncScalar: tensor of %d bytes" % n)
            self.host = slot[0].view(torch.uint8)[:n].view(t.dtype).view(t.shape)
            self.host.copy_(t, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record(torch.cuda.current_stream(t.device))
            slot[1] = weakref.ref(self)
        else:
            self.host, self.event = t.detach().clone(), None

    def _evict(self):
        """The ring slot is needed by a newer instance: finish the copy and keep the value privately."""
        if self.event is not None:
            self.event.synchronize()
            self.host, self.event = self.host.clone(), None

    def get(self):
        self._evict()
        return self.host

    def __del__(self):          # a dropped instance's copy may still be in flight into the slot the ring hands out next
        try:
            if self.event is not None:
                self.event.synchronize()
        except Exception:
            pass


_UPLOADERS = {}


def upload(host, device):
    """``host`` (a CPU tensor) on ``device`` without blocking the host (see _UploadRing); plain copy on a CPU device."""
    device = torch.device(device)
    if device.type != "cuda":
        return host.to(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    ring = _UPLOADERS.get(idx)
    if ring is None:
        ring = _UPLOADERS[idx] = _UploadRing(torch.device("cuda", idx))
    return ring.upload(host)


# ---- layout helpers (module boundary only; NCDHW <-> NDHWC) -------------------------------------------
# ---- side streams (independent branches of one step on concurrent HIP streams) ---------------------------------------
_SIDE_STREAMS = {}


def side_stream(device, name, priority=0):
    """The process-wide side HIP stream ``name`` of ``device`` (created on first use; ``priority`` < 0: high)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, name)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=idx, priority=priority)
    return _SIDE_STREAMS[key]


def side_streams(device):
    """Every side stream handed out for ``device`` -- whoever consumes results off-stream (the gradient reducer's
    communication stream) has to wait for all of them, not only for the current stream."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return [s for (d, _), s in _SIDE_STREAMS.items() if d == idx]
