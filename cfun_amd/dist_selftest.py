"""Pre-flight of the multi-GPU path, run INSIDE an initialised process group before anything is timed (bench.py, N > 1).

The sharded layout of ``cfun_amd.dist`` (the dataflow it shards is model.py:1391-1514) uses four
kinds of traffic: ring-neighbour send/recv of packed halo planes (``_exchange`` / ``halo_conv``), one all-gather of the
ranks' RPN candidates, all-reduces of small sums, and the bucketed gradient all-reduce of ``GradientReducer`` on its own
communicator and stream.  Each is exercised here once at a size that takes milliseconds and is compared with what a single
process computes from the same seeded tensors (every rank builds the FULL tensors locally, so no reference has to travel):

    halo      ``halo_exchange`` forward / backward == zero padding of the full tensor, slab by slab
    conv      a 3x3x3 conv (16 -> 32 channels, interior / edge split with the transfer on the side stream) and the stride-2
              down-conv trained through the exchange == the un-sharded conv: outputs, input gradients, summed weight gradients
    gather    all-gather of rank-coded candidate packs: every rank holds every rank's rows, in rank order
    reducer   ``GradientReducer`` with several buckets: mean over ranks of rank-coded gradients

``preflight`` returns a dict for the bench line (``ok``, the worst error per section, ``rccl_ranks_seen`` = the number of
distinct ranks whose tensors arrived through the collectives, the devices they sat on); it never raises on a numerical
mismatch -- the caller decides -- but lets communication errors propagate."""
import time

import torch
import torch.distributed as dist


def _rel(a, b):
    den = float(b.abs().max())
    return float((a - b).abs().max()) / max(den, 1e-30)


def _device_id(dev):
    """(index, an identifier that differs between physical GPUs) of this rank's device; (-1, host pid) on CPU."""
    import os
    if dev.type != "cuda":
        return -1, os.getpid()
    props = torch.cuda.get_device_properties(dev)
    ident = 0
    for name in ("pci_domain_id", "pci_bus_id", "pci_device_id"):
        ident = ident * 65536 + int(getattr(props, name, 0) or 0)
    if ident == 0:                      # (a torch build without the PCI fields: fall back on the index)
        ident = dev.index if dev.index is not None else torch.cuda.current_device()
    return (dev.index if dev.index is not None else torch.cuda.current_device()), ident


def preflight(dev, tol=2e-4):
    from . import dist as cdist
    from .layers import Conv3dParams
    t0 = time.perf_counter()
    dev = torch.device(dev)
    rank, world = dist.get_rank(), dist.get_world_size()
    backend = dist.get_backend()
    err = {}
    g = torch.Generator().manual_seed(1234)

    # ---- who is there: one all-gather of (rank, device index, device identifier)
    idx, ident = _device_id(dev)
    me = torch.tensor([rank, idx, ident], dtype=torch.int64, device=dev)
    parts = [torch.zeros_like(me) for _ in range(world)]
    dist.all_gather(parts, me)
    table = torch.stack(parts).cpu().tolist()
    ranks_seen = len({int(r[0]) for r in table})
    devices_seen = len({int(r[2]) for r in table})

    # ---- halo exchange against plain zero padding (2 planes per rank, 1-plane halos; forward and backward)
    full = torch.randn(1, 2 * world, 3, 4, 4, generator=g).to(dev)
    gy_full = torch.randn(1, 4 * world, 3, 4, 4, generator=g).to(dev)            # per-rank padded slabs, concatenated
    with cdist.depth_sharded():
        x = cdist.slab(full, dim=1).clone().requires_grad_(True)
        y = cdist.halo_exchange(x, 1, 1)
        gy = gy_full[:, 4 * rank:4 * rank + 4]
        (y * gy).sum().backward()
    pad = torch.nn.functional.pad(full, (0, 0, 0, 0, 0, 0, 1, 1))
    err["halo_fwd"] = _rel(y.detach(), pad[:, 2 * rank:2 * rank + 4])
    # d/dx of sum_r <pad slab_r, gy_r>: plane z of the full tensor appears in slab floor(z/2) and, as a halo, in a neighbour's
    gx_full = torch.zeros_like(full)
    for r in range(world):
        for j in range(4):
            z = 2 * r - 1 + j
            if 0 <= z < 2 * world:
                gx_full[:, z] += gy_full[:, 4 * r + j]
    err["halo_bwd"] = _rel(x.grad, gx_full[:, 2 * rank:2 * rank + 2])

    # ---- depth-coupled convs trained through the exchange (4 planes per rank: interior + edges, the overlapped split)
    for tag, ci, co, stride in (("conv", 16, 32, 1), ("conv_s2", 4, 8, 2)):
        torch.manual_seed(11)
        conv = Conv3dParams(ci, co, 3, stride=stride, padding=1).to(dev)
        D = 4 * world
        xs = torch.randn(1, D, 8, 16, ci, generator=g).to(dev)
        gys = torch.randn(1, D // stride, 8 // stride, 16 // stride, co, generator=g).to(dev)
        with cdist.depth_sharded():
            xl = cdist.slab(xs, dim=1).clone().requires_grad_(True)
            yl = conv(xl)
            (yl * cdist.slab(gys, dim=1)).sum().backward()
            wg = torch.cat([conv.weight.grad.reshape(-1), conv.bias.grad.reshape(-1)]).clone()
            dist.all_reduce(wg)
        conv.weight.grad = conv.bias.grad = None
        xr = xs.clone().requires_grad_(True)
        yr = conv(xr)
        (yr * gys).sum().backward()
        wr = torch.cat([conv.weight.grad.reshape(-1), conv.bias.grad.reshape(-1)])
        dl, do = D // world, D // stride // world
        err[tag + "_y"] = _rel(yl.detach(), yr.detach()[:, rank * do:(rank + 1) * do])
        err[tag + "_gx"] = _rel(xl.grad, xr.grad[:, rank * dl:(rank + 1) * dl])
        err[tag + "_gw"] = _rel(wg, wr)

    # ---- all-gather of candidate packs (the proposal set's one collective)
    pack = (torch.arange(64 * 8, dtype=torch.float32).reshape(64, 8) + 1000.0 * rank).to(dev)
    parts = [torch.empty_like(pack) for _ in range(world)]
    dist.all_gather(parts, pack)
    want = torch.cat([torch.arange(64 * 8, dtype=torch.float32).reshape(64, 8) + 1000.0 * r for r in range(world)]).to(dev)
    err["gather"] = _rel(torch.cat(parts), want)

    # ---- bucketed gradient all-reduce on the reducer's own communicator / stream (3 buckets of rank-coded gradients)
    ps = [torch.nn.Parameter(torch.zeros(n, device=dev)) for n in (1000, 70, 513, 4096)]
    red = cdist.GradientReducer(ps, bucket_bytes=4096, average=True)
    n_buckets = len(red.buckets)
    red.zero_grad()
    loss = sum((p * (torch.arange(p.numel(), device=dev, dtype=torch.float32) % 7 + 1.0) * float(rank + 1)).sum() for p in ps)
    loss.backward()
    red.finish()
    mean_factor = sum(r + 1 for r in range(world)) / float(world)
    err["reducer"] = max(_rel(p.grad, (torch.arange(p.numel(), device=dev, dtype=torch.float32) % 7 + 1.0) * mean_factor)
                         for p in ps)
    red.remove()

    # the worst error of any rank, on every rank
    keys = sorted(err)
    ev = torch.tensor([err[k] for k in keys], dtype=torch.float64, device=dev)
    dist.all_reduce(ev, op=dist.ReduceOp.MAX)
    err = {k: float(v) for k, v in zip(keys, ev.cpu().tolist())}
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    ok = all(v <= tol for v in err.values()) and ranks_seen == world
    return {"ok": bool(ok), "backend": backend, "world": world, "rccl_ranks_seen": ranks_seen if backend == "nccl" else 0,
            "ranks_seen": ranks_seen, "devices_seen": devices_seen,
            "rank_device_table": [[int(v) for v in row[:2]] for row in table], "reducer_buckets": n_buckets,
            "max_rel_err": err, "tolerance_rel": tol, "seconds": time.perf_counter() - t0,
            "what": "halo send/recv (forward + backward) vs zero padding, 3x3x3 and stride-2 convs through the overlapped "
                    "halo exchange vs the un-sharded conv (y, dx, summed dw), candidate all-gather, bucketed gradient "
                    "all-reduce vs the analytic mean -- before anything is timed"}
