#!/usr/bin/env python3
"""Headline benchmark of the CFUN hot path on MI355X (BASELINE.json: volumes/sec, forward+backward).

    python bench.py --gpus N --steps K --warmup W

N > 1: either under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / WORLD_SIZE from
the environment; WORLD_SIZE must equal N) or as plain python -- the script then re-executes itself under that launcher,
so `--gpus 8` can never print a one-rank number.  N ranks need N GPUs (CFUN_BENCH_BACKEND=gloo: smoke test on fewer).

A "step" = one synthetic 256x256x128 8-class CT volume through FPN -> RPN -> proposals/NMS -> classifier head
(12 RoIs) -> U-Net mask head (4 positive RoIs, 96^3 in, 192^3 out, stage 'finetune') -> 6 losses incl. the
3-D Sobel edge loss -> backward (cfun_amd.step.training_step); inputs are resident in HBM before the timed
region, no optimizer step, no host sync inside a step except the NMS count read the reference also has.
N > 1: one process per GPU; first a communication pre-flight (cfun_amd.dist_selftest: halo send/recv, candidate all-gather,
bucketed all-reduce, sharded convs vs the un-sharded ones -> `preflight`, `rccl_ranks_seen`), then every rank trains its own
volume (whole volumes are independent units) and the replicated weights' gradients are all-reduced over RCCL during each
step's backward -- weak scaling, `value`.  The same invocation then times ONE volume per step over all ranks (depth-sharded,
strong scaling) and reports it as `sharded_one_volume` with its parity against the single-process step (`sharded_parity`);
--sharded makes that figure `value` instead.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel, timed live with HIP
events on the launch stream) and, at N = 1, `cpu_baseline` (the oracle on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL between processes needs it on this driver

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0           # HBM3E spec (MI355X_MICROARCH.md; ~6.3 TB/s achievable)
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 matrix peak (= fp32 vector peak)

WORKLOADS = {
    # name: (stage, H, W, D)
    "cfg2": ("finetune", 256, 256, 128),   # the configuration BASELINE.json's metric is quoted on
    "cfg1": ("beginning", 128, 128, 64),
    "cfg0": ("beginning", 64, 64, 32),
    "cfg3": ("finetune", 512, 512, 256),   # configs[3]'s volume on however many GPUs are given (--sharded: one volume)
    "cfg4": ("beginning", 320, 320, 256),  # the LiTS fork's shapes (P3D35, 5x7x7 stem, b = 32, 3 classes, 32x80x80 crops)
}


def dominant_kernel_match(cfg):
    """conv_norm_lrelu_l4.0: 3x3x3, 2b -> 2b channels on the full-resolution crops -- the single largest
    launch of the step (76.4 GFLOP per RoI)."""
    b = cfg.UNET_MASK_BRANCH_CHANNEL
    side = cfg.MASK_POOL_SIZE

    def match(p):
        if (p.kd, p.kh, p.kw, p.stride, p.up2) != (3, 3, 3, 1, 0) or (p.Do, p.Ho, p.Wo) != tuple(side):
            return None
        if p.Ci == 2 * b and p.Co == 2 * b:
            return "mfma"            # conv_norm_lrelu_l4.0: the dominant, MFMA-bound launch
        if p.Ci == 1 and p.Co == b:
            return "hbm"             # conv3d_c1_1: the HBM-bound 3x3x3 conv of the path (C_in = 1, AI 13 flop/B)
        return None
    return match


PMC_TRAFFIC_JSON = "profiles/round5_pmc_wino_l4_0.json"
PMC_MFMA_JSON = "profiles/round5_pmc_mfma_wino_l4_0.json"


def dominant_kernel_info(cfg, n_roi):
    """Which kernels the library runs for conv_norm_lrelu_l4.0 (cfun_conv3d_fwd_kernel / cfun_conv3d_wino_plan) and the
    MFMA flops they actually issue per call: the Winograd kernels run 9 (dz,dy) x 4 points (1-D, x 2/3) or 3 (dz) x 16
    points per 2x2 outputs (2-D, x 4/9) instead of 27 taps, on output-channel columns padded to whole 16-wide subtiles
    (40 -> 48).  Returns (kernel code, executed flops, label)."""
    import ctypes as C
    from cfun_amd import _lib, ops
    b, side = cfg.UNET_MASK_BRANCH_CHANNEL, tuple(cfg.MASK_POOL_SIZE)
    spec = ops.ConvSpec(k=(3, 3, 3), co=2 * b, pad=(1, 1, 1))
    p = ops._params(spec, (n_roi,) + side + (2 * b,), False, False, False)
    lib = _lib.load()
    kern = int(lib.cfun_conv3d_fwd_kernel(C.byref(p)))
    tiles = n_roi * -(-side[0] // 4) * -(-side[1] // 4) * -(-side[2] // 16)
    if kern != 2:
        co_pad = -(-2 * b // 16) * 16
        return kern, 2.0 * tiles * 256 * 27 * (2 * b) * co_pad, "k_conv_mfma<3,3,3,1,3>"
    plan = (C.c_int32 * 4)()
    _lib.check(lib.cfun_conv3d_wino_plan(C.byref(p), plan), "conv3d_wino_plan")
    twod, nsub, sb, cols = [int(v) for v in plan]
    executed = 2.0 * tiles * 256 * 27 * ((4.0 / 9.0) if twod else (2.0 / 3.0)) * (2 * b) * cols
    if twod:
        label = ("k_conv_wino<%d, 2-D%s> (x and y in the Winograd F(2x2,3x3) domain: 4/9 of the direct MACs on the MFMA pipe; "
                 "weights transformed by the step's batched cfun_weight_prepare launch)" % (nsub, ", two waves per SIMD" if sb else ""))
    else:
        label = ("k_conv_wino<%d> (x axis in the Winograd F(2,3) domain: 2/3 of the direct MACs on the MFMA pipe; weights "
                 "transformed by the step's batched cfun_weight_prepare launch, outside the timed call)" % nsub)
    return kern, executed, label


def dominant_back_to_back(cfg, n_roi, dev, launches=8):
    """conv_norm_lrelu_l4.0's forward through the C ABI, `launches` times back to back between one pair of HIP events, two
    ways: (a) KERNEL ONLY -- the operand the step's batched weight preparation hands it (w_prepared; for timing any values
    of the right size do), plain epilogue: one launch per call, the figure rocprofv3's per-kernel average reproduces;
    (b) from the plain packed weight: the call first launches its own weight transform (what a caller outside a
    WeightScope pays).  Returns (seconds per call (a), seconds per call (b))."""
    import ctypes as C
    from cfun_amd import _lib, ops
    b, side = cfg.UNET_MASK_BRANCH_CHANNEL, tuple(cfg.MASK_POOL_SIZE)
    lib = _lib.load()
    spec = ops.ConvSpec(k=(3, 3, 3), co=2 * b, pad=(1, 1, 1))
    with torch.no_grad():
        x = torch.randn((n_roi,) + side + (2 * b,), device=dev)
        y = torch.empty((n_roi,) + side + (2 * b,), device=dev)
        out = []
        for prepared in (True, False):
            p = ops._params(spec, x.shape, False, False, False)
            if prepared:
                kinds, nbytes = (C.c_int32 * 2)(), (C.c_size_t * 2)()
                _lib.check(lib.cfun_weight_prepare_kinds(C.byref(p), kinds, nbytes), "weight_prepare_kinds")
                wp = torch.randn(max(1, int(nbytes[0]) // 4), device=dev) * 0.05
                p.w_prepared = 1
            else:
                wp = ops.pack_weight(torch.randn(2 * b, 2 * b, 3, 3, 3, device=dev) * 0.05)
            ws = _lib.workspace(lib.cfun_conv3d_fwd_workspace_bytes(C.byref(p)), x)
            args = (_lib.ptr(x), _lib.ptr(wp), None, None, None, _lib.ptr(y), C.byref(p), _lib.ptr(ws), ws.numel(), _lib.stream(x))
            for _ in range(2):
                _lib.check(lib.cfun_conv3d_fwd(*args), "conv3d_fwd")
            out.append(back_to_back(lambda: lib.cfun_conv3d_fwd(*args), launches))
    return out[0], out[1]


B2B_REPS = []      # every repetition of the last back_to_back() call, ms per launch (reported next to the figure used)


def back_to_back(launch, launches):
    """Seconds per launch of `launches` back-to-back calls between ONE pair of HIP events: four repetitions in a row, the first
    dropped, the median of the other three.  (One repetition right after an idle queue reads 8 - 15 % slow on an 80 us kernel --
    92 -> 87 -> 82 us over three repetitions, profiles/round4_pmc_stem.txt -- the third equals the median of per-launch event
    pairs: the clocks ramp for the first ~10 ms of work.)"""
    reps = []
    for _ in range(4 if launches > 1 else 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(launches):
            launch()
        e1.record()
        torch.cuda.synchronize()
        reps.append(e0.elapsed_time(e1) * 1e-3 / launches)
    B2B_REPS[:] = [r * 1e3 for r in reps]
    tail = sorted(reps[1:]) if len(reps) > 1 else reps
    return tail[len(tail) // 2]


def stem_back_to_back(cfg, net, n_roi, dev, launches=64):
    """conv3d_c1_1 (the HBM-bound 3x3x3 conv of the path: C_in = 1) exactly as the step calls it -- same entry point,
    shapes, weight -- `launches` times back to back between one pair of HIP events.  Returns (seconds per launch, launches,
    the name of the kernel the library runs for it)."""
    import ctypes as C
    from cfun_amd import _lib, ops
    conv = net.mask.modified_u_net.conv3d_c1_1
    side = tuple(cfg.MASK_POOL_SIZE)
    x = torch.randn((n_roi,) + side + (1,), device=dev)
    spec = ops.ConvSpec(k=tuple(conv.kernel_size), co=conv.out_channels, pad=tuple(conv.padding))
    p = ops._params(spec, x.shape, False, False, False)
    kern = int(_lib.load().cfun_conv3d_fwd_kernel(C.byref(p)))
    zpt = os.environ.get("CFUN_STEM_ZPT", "2")
    name = {3: "k_conv_stem333z<%d, %s>" % (conv.out_channels, zpt if zpt in ("2", "4") else "-"),
            1: "k_conv_mfma", 0: "k_conv_fwd_direct"}.get(kern, "kernel code %d" % kern)
    if kern == 3 and zpt not in ("2", "4"):
        name = "k_conv_stem<3,3,3,1,%d>" % conv.out_channels
    # straight through the C ABI (packed weight, output and workspace allocated once): ~5 us of host time per call, so the
    # launches queue up and the event pair measures the kernel's back-to-back rate, not Python's call overhead
    lib = _lib.load()
    with torch.no_grad():
        wp = ops.pack_weight(conv.weight)
        y = torch.empty((n_roi,) + side + (conv.out_channels,), device=dev)
        ws = _lib.workspace(lib.cfun_conv3d_fwd_workspace_bytes(C.byref(p)), x)
        st = _lib.stream(x)
        args = (_lib.ptr(x), _lib.ptr(wp), None, None, None, _lib.ptr(y), C.byref(p), _lib.ptr(ws), ws.numel(), st)
        for _ in range(4):
            _lib.check(lib.cfun_conv3d_fwd(*args), "conv3d_fwd")
        t_launch = back_to_back(lambda: lib.cfun_conv3d_fwd(*args), launches)
    return t_launch, launches, name


def p3d_stem_back_to_back(cfg, net, dev, launches=16):
    """The OTHER C_in = 1 conv of the path: the P3D stem C1 = Conv3d(1 -> 16, k(3,7,7), s2) + BN + ReLU + MaxPool3d(2,2)
    (backbone.py:123-128) on the whole volume, through the drop-in module exactly as FPN calls it (the stem kernel with the
    folded BN + ReLU epilogue, then the pool kernel).  Returns (seconds per call, algorithmic bytes = volume in + pooled map
    out, each once, algorithmic flops of the conv)."""
    d, h, w = cfg.image_dhw
    stem = net.fpn.C1
    conv = stem[0]
    with torch.no_grad():
        x = torch.randn((1, d, h, w, 1), device=dev)
        y = stem.forward_ndhwc(x)
        for _ in range(3):
            stem.forward_ndhwc(x)
        t = back_to_back(lambda: stem.forward_ndhwc(x), launches)
    kd, kh, kw = conv.kernel_size
    do, ho, wo = d // 2, h // 2, w // 2
    flops = 2.0 * conv.out_channels * kd * kh * kw * do * ho * wo
    return t, 4.0 * (x.numel() + y.numel()), flops


def mask_losses_back_to_back(cfg, n_roi, dev, reps=5):
    """The one-pass mask-loss kernels of the step (csrc/loss_fused.hip) at the benchmarked shape, alone: forward (softmax + cross
    entropy + Sobel edge loss + the backward's operand field) and backward, each between its own HIP-event pair, median of
    `reps`.  Algorithmic bytes: logits in, probabilities + field out / field, probabilities in, dlogits out (+ the labels)."""
    from cfun_amd import ops
    c = int(cfg.NUM_CLASSES)
    dd, hh, ww = [int(v) for v in cfg.MASK_SHAPE]
    logits = torch.randn((n_roi, dd, hh, ww, c), device=dev).requires_grad_(True)
    labels = torch.randint(0, c, (n_roi, dd, hh, ww), device=dev, dtype=torch.uint8)
    tf, tb = [], []
    for it in range(reps + 2):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        logits.grad = None
        e[0].record()
        ce, edge, _ = ops.mask_losses_fused(logits, labels)
        e[1].record()
        tot = ce + edge
        e[2].record()
        tot.backward()
        e[3].record()
        torch.cuda.synchronize()
        if it >= 2:
            tf.append(e[0].elapsed_time(e[1]) * 1e-3)
            tb.append(e[2].elapsed_time(e[3]) * 1e-3)
    vox = float(n_roi * dd * hh * ww)
    field = 8.0 * n_roi * dd * (hh - 2) * (ww - 2) * (c - 1)
    nbytes = 4.0 * vox * c * 2 + vox + field
    return sorted(tf)[len(tf) // 2], sorted(tb)[len(tb) // 2], nbytes


def pointwise_back_to_back(cfg, net, n_roi, dev, launches=64):
    """conv3d_l4 (1x1x1, 2b -> n_classes on the 96^3 crops: the other HBM-bound conv of the U-Net, pure streaming) through the
    same C-ABI entry point, timed like stem_back_to_back.  Returns (seconds per launch, algorithmic bytes per launch)."""
    import ctypes as C
    from cfun_amd import _lib, ops
    conv = net.mask.modified_u_net.conv3d_l4
    side = tuple(cfg.MASK_POOL_SIZE)
    ci, co = conv.in_channels, conv.out_channels
    x = torch.randn((n_roi,) + side + (ci,), device=dev)
    spec = ops.ConvSpec(k=(1, 1, 1), co=co, pad=(0, 0, 0))
    p = ops._params(spec, x.shape, False, False, False)
    lib = _lib.load()
    with torch.no_grad():
        wp = ops.pack_weight(conv.weight)
        y = torch.empty((n_roi,) + side + (co,), device=dev)
        ws = _lib.workspace(lib.cfun_conv3d_fwd_workspace_bytes(C.byref(p)), x)
        args = (_lib.ptr(x), _lib.ptr(wp), None, None, None, _lib.ptr(y), C.byref(p), _lib.ptr(ws), ws.numel(), _lib.stream(x))
        for _ in range(4):
            _lib.check(lib.cfun_conv3d_fwd(*args), "conv3d_fwd")
        t_launch = back_to_back(lambda: lib.cfun_conv3d_fwd(*args), launches)
    vox = n_roi * side[0] * side[1] * side[2]
    return t_launch, 4.0 * (vox * ci + vox * co + ci * co)


def git_blob_sha1(path):
    """= `git hash-object path` (works without a .git directory, as on the GPU box)."""
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def pmc_record(rel_path, field):
    """A counter figure of the dominant kernel from a rocprofv3 --pmc pass committed under profiles/ (tools/pmc_*.sh;
    FETCH_SIZE / WRITE_SIZE in separate passes, corrected as MI355X_MICROARCH.md prescribes).  It is NOT measured by this
    run, so it is reported with its source, and only while the kernel source it was collected on (git blob of
    conv3d_mfma.h, recorded in the JSON) is still the one in the tree -- otherwise null."""
    src = {"file": rel_path, "collected_by": "rocprofv3 --pmc, separate run (tools/pmc_traffic.sh / tools/pmc_mfma.sh)"}
    try:
        with open(os.path.join(ROOT, rel_path)) as f:
            rec = json.load(f)
        src["kernel_src_blob"] = rec.get("kernel_src_blob")
        cur = git_blob_sha1(os.path.join(ROOT, rec.get("kernel_src", "cfun_amd/csrc/conv3d_mfma.h")))
        src["current"] = bool(rec.get("kernel_src_blob") == cur)
        return (float(rec[field]) if src["current"] else None), src
    except Exception as e:       # no counter pass for this tree
        src["current"] = False
        src["error"] = str(e)[:80]
        return None, src


def physical_cores():
    """(physical cores, logical CPUs) of the host from /proc/cpuinfo (unique (physical id, core id) pairs)."""
    logical = os.cpu_count() or 1
    try:
        pairs, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if core is not None:
                    pairs.add((phys, core))
                phys = core = None
        if core is not None:
            pairs.add((phys, core))
        return (len(pairs) or logical), logical
    except OSError:
        return logical, logical


def parity_dropout_masks(cfg, n_pos):
    """The Dropout3d masks BOTH legs of the loss-parity step use (generator seed 1; None when the config has no dropout)."""
    if getattr(cfg, "UNET_DROPOUT", 0.6) <= 0:
        return None
    b = cfg.UNET_MASK_BRANCH_CHANNEL
    gen = torch.Generator().manual_seed(1)
    return [torch.empty(n_pos, ch).bernoulli_(0.4, generator=gen) / 0.4 for ch in (b, 2 * b, 4 * b, 8 * b, 16 * b)]


# parameter gradients compared between the extra GPU step and the oracle's CPU step at full size (`grad_parity`): the
# first, the dominant and the last conv of the U-Net, its deepest level, and one tensor of each detector part.
#   * U-Net tensors: held to a MEASURED bound since round 5 -- tests/golden/grad_fp64_cfg2.npz carries the fp64 gradients of
#     exactly this step's mask head (same weights, volume, RoIs, Dropout3d masks; gen_grad_fp64_cfg2.py, computed on the GPU box's
#     host) and, per tensor, the deviation of the reference's own fp32 arithmetic from them (floor: 4.2e-3 / 1.8e-4 / 9.8e-3 /
#     2.1e-6 for the four below -- InstanceNorm over 6^3 voxels amplifies fp32 rounding in the deep levels):
#     relL2(GPU, fp64) <= GRAD_FP64_FACTOR * floor + GRAD_FP64_FLOOR, the rule of tests/module_cases.py.  The second column is
#     the fall-back bound on relL2(GPU, CPU fp32) when the fixture does not belong to the run's weights (another torch build).
#   * detector tensors (no normalisation over tiny volumes on their path): relL2(GPU, CPU fp32) <= 1e-3 as before.
GRAD_FP64_FACTOR, GRAD_FP64_FLOOR = 3.0, 2e-5
GRAD_PARITY_KEYS = {
    "mask.modified_u_net.conv3d_c1_1.weight": 2e-2,
    "mask.modified_u_net.conv_norm_lrelu_l4.0.weight": 1e-3,
    "mask.modified_u_net.norm_lrelu_conv_c5.2.weight": 4e-2,
    "mask.modified_u_net.out_upscale_conv.1.weight": 1e-4,
    "fpn.C1.0.weight": 1e-3,
    "fpn.P2_conv2.weight": 1e-3,
    "rpn.conv_shared.weight": 1e-3,
    "classifier.conv1.weight": 1e-3,
}
GRAD_FP64_FIXTURE = "tests/golden/grad_fp64_cfg2.npz"
UNET_FALLBACK_BOUND = 4e-2      # U-Net tensors not named above, when the fp64 fixture does not belong to the run's weights


def grad_parity_keys(net):
    """The tensors of `grad_parity`: EVERY U-Net conv weight (27; the fp64 fixture covers them all since round 6) and one
    tensor of each detector part.  {name: fall-back bound on relL2(GPU, CPU fp32)}."""
    keys = dict(GRAD_PARITY_KEYS)
    for k, p in net.named_parameters():
        if k.startswith("mask.modified_u_net.") and k.endswith(".weight") and p.requires_grad:
            keys.setdefault(k, UNET_FALLBACK_BOUND)
    return keys


def load_grad_fp64(net, cfg, workload):
    """The fp64 reference gradients of the cfg2 bench step's mask head -- every U-Net conv weight since round 6 -- or None when
    they do not apply to this run (another workload, or weights that are not the ones the fixture was generated from: checked
    through |weight| checksums).  {name: (g64 sample, flat stride, floor)}: the sample is tensor.reshape(-1)[::stride]."""
    import numpy as np
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), GRAD_FP64_FIXTURE)
    if workload != "cfg2" or not os.path.exists(path):
        return None
    d = np.load(path)
    names = sorted(k[4:] for k in d.files if k.startswith("g64_"))
    chk_names = [str(k) for k in d["w_check_names"]]
    sd = net.state_dict()
    chk = np.array([float(sd[k].detach().double().abs().sum().cpu()) for k in chk_names])
    if chk.shape != d["w_check"].shape or np.abs(chk - d["w_check"]).max() > 1e-6 * np.abs(d["w_check"]).max():
        return None
    return {k: (d["g64_" + k].astype(np.float64), int(d["fstride_" + k]), float(d["floor_" + k])) for k in names}


def grad_fp64_error(grad, ref):
    """relL2 of a parameter gradient (torch tensor) against its fixture entry (g64 sample, flat stride, floor)."""
    g64, stride, _ = ref
    g64 = torch.from_numpy(g64)
    a = grad.detach().cpu().double().reshape(-1)[::stride]
    return float((a - g64).norm() / g64.norm().clamp(min=1e-300))


def cpu_baseline(cfg, net, sample, threads, iters=1, small_iters=3, grad_keys=GRAD_PARITY_KEYS):
    """The reference's CPU path timed beside the GPU run: the oracle (oracle/cfun_oracle.py -- the plain fp32 torch-CPU
    restatement of the reference, pinned to it by tests/golden) runs THE SAME training step -- this run's weights, image,
    4 + 8 injected RoIs, targets, all six losses incl. the 3-D Sobel edge loss, forward + backward -- on the host cores.
    Nothing is extrapolated: `value` = 1 / (median wall time of `iters` full iterations).  One untimed warm-up iteration
    of the same step at 64x64x32 / 1 RoI primes the thread pool and oneDNN's primitive caches; the default is ONE timed
    iteration (the step takes the better part of a minute on CPU), --cpu-baseline-iters 3 gives the median of three."""
    from oracle import cfun_oracle as orc
    from cfun_amd import config as ccfg, step
    torch.set_num_threads(threads)
    keys = ("rpn_class_loss", "rpn_bbox_loss", "mrcnn_class_loss", "mrcnn_bbox_loss", "mrcnn_mask_loss",
            "mrcnn_mask_edge_loss")

    def one(cfg_i, net_i, s, n_pos):
        sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype == torch.float32 and "running" not in k)
              for k, v in net_i.state_dict().items()}
        c = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in s.items()}
        masks = parity_dropout_masks(cfg_i, n_pos)
        keep = list(range(n_pos)) + list(range(4, 4 + 2 * n_pos))
        onehot = torch.stack([(c["mask_labels"][:n_pos] == k) for k in range(cfg_i.NUM_CLASSES)], dim=1).double()
        t0 = time.perf_counter()
        ref = orc.training_step(sd, c["image"], net_i.anchors.cpu(), c["rpn_match"], c["rpn_bbox_t"], c["p_rois"][:n_pos],
                                c["n_rois"][:2 * n_pos], c["target_class_ids"][keep], c["target_deltas"][keep], onehot,
                                cfg_i.STAGE, cfg_i.POOL_SIZE, cfg_i.MASK_POOL_SIZE, dropout_masks=masks,
                                proposal_count=cfg_i.POST_NMS_ROIS_TRAINING, nms_threshold=cfg_i.RPN_NMS_THRESHOLD,
                                pre_nms_limit=cfg_i.PRE_NMS_LIMIT,
                                layers=tuple(getattr(cfg_i, "BACKBONE_LAYERS", (2, 3))),
                                stem_pad=(getattr(cfg_i, "BACKBONE_STEM_KD", 3) // 2, 3, 3),
                                ce_class_weights=getattr(cfg_i, "MASK_CE_CLASS_WEIGHTS", None),
                                edge_raw=getattr(cfg_i, "EDGE_LOSS_RAW_SOBEL", False),
                                stage_split=getattr(cfg_i, "STAGE_SPLIT", False),
                                loss_weights=[float(cfg_i.LOSS_WEIGHTS[k]) for k in keys])
        ref["total"].backward()
        dt = time.perf_counter() - t0
        grads = {k: sd[k].grad.detach().clone() for k in grad_keys if k in sd and sd[k].grad is not None}
        return dt, [float(l) for l in ref["losses"]], grads

    # warm-up: the same code path at the smallest configuration (BASELINE configs[0]'s 64x64x32 volume, 1 + 2 RoIs), then
    # `small_iters` timed iterations of it: the cfg0 figure SURVEY.md section 8(d) asks for beside cfg2's
    wcfg = ccfg.heart_config("beginning", 64, 64, 32) if not isinstance(cfg, ccfg.LiTSConfig) else None   # BASELINE configs[0]
    small = None
    if wcfg is not None:
        wnet = step.CFUNHotPath(wcfg)
        ws_ = step.synthetic_inputs(wcfg, torch.device("cpu"), 0)
        one(wcfg, wnet, ws_, 1)
        st = sorted(one(wcfg, wnet, ws_, 1)[0] for _ in range(max(1, small_iters)))
        small = {"workload": "BASELINE configs[0]: 64x64x32 volume, stage 'beginning', 1 positive + 2 negative RoIs, forward + backward",
                 "value": 1.0 / st[len(st) // 2], "unit": "volumes/s", "iters": len(st), "median_s": st[len(st) // 2]}
    times, losses, grads = [], None, None
    want = iters if iters > 0 else 3
    for i in range(want):
        t, losses, grads = one(cfg, net, sample, 4)
        times.append(t)
        if iters <= 0 and i == 0 and t >= 60.0:      # auto: a slow host times one iteration only
            break
    med = sorted(times)[len(times) // 2]
    d, h, w = cfg.image_dhw
    phys, logical = physical_cores()
    return dict(value=1.0 / med, unit="volumes/s", cores=threads, kind="port", physical_cores=phys, logical_cpus=logical,
                sample="oracle (torch %s CPU fp32, torch.get_num_threads() = %d threads on a host with %d physical cores / "
                       "%d logical CPUs -- one thread per physical core, capped at 64: past that torch's CPU conv3d / "
                       "instance-norm kernels stop scaling): the identical %dx%dx%d '%s' training step -- same "
                       "weights, image, 4 + 8 RoIs, targets, Dropout3d masks (seed 1), six losses, forward + backward; %d "
                       "timed full iteration(s) after a warm-up at 64x64x32: %s s (median %.1f s); nothing extrapolated"
                       % (torch.__version__, torch.get_num_threads(), phys, logical, h, w, d, cfg.STAGE, len(times),
                          ", ".join("%.1f" % t for t in times), med),
                losses=losses, small_config=small, _grads=grads)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(argv, gpus, port=None):
    """The command `python bench.py --gpus N ...` turns itself into when it was started as plain python with N > 1: N ranks
    of this script on one node, one per GPU, rendezvous on the loopback address (the contract's launch line)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr",
            "127.0.0.1", "--master-port", str(port or free_port()), os.path.abspath(__file__)] + list(argv)


def check_gpu_count(gpus, backend):
    """N ranks need N GPUs: RCCL refuses two ranks on one device, and a silent `rank % device_count` would print an
    `n_gpus: N` line measured on fewer.  CFUN_BENCH_BACKEND=gloo is the one exception (the path's smoke test on a 1-GPU box:
    several ranks share the device, collectives on the host -- the line says so in `backend`)."""
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if gpus > have and backend != "gloo":
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (set CFUN_BENCH_BACKEND=gloo to smoke-test the N-rank "
                         "path on fewer devices; such a line is labelled and is not a scaling measurement)" % (gpus, have))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-iters", type=int, default=0,
                    help="timed full iterations of the oracle's CPU step (median reported).  0 (default): 3 -- the median of 3 "
                         "warm iterations SURVEY.md section 8(d) asks for -- when the first one takes < 60 s on this host, "
                         "else that one only (the whole run has to finish within a few minutes)")
    ap.add_argument("--no-hbm-loop", action="store_true",
                    help="skip the 64 back-to-back launches of the HBM-bound stem conv behind `roofline_hbm` (profiling "
                         "runs: the loop would show up in the per-step kernel statistics); the in-step timing is reported")
    ap.add_argument("--sharded", action="store_true",
                    help="N > 1: `value` = ONE volume per step over all ranks (depth-sharded FPN/RPN with halo exchange, head "
                         "RoIs dealt round-robin / z-sharded; strong scaling) instead of one volume per rank")
    ap.add_argument("--no-sharded-leg", action="store_true",
                    help="N > 1 without --sharded: skip the extra one-volume-over-all-ranks leg reported as `sharded_one_volume`")
    ap.add_argument("--preflight-only", action="store_true",
                    help="N > 1: launch the ranks, run the communication pre-flight (cfun_amd.dist_selftest) and print its "
                         "report as the JSON line; nothing is timed")
    return ap.parse_args(argv)


def sharded_parity_check(cdist, step, dist, net, cfg, sample, reducer, one_sharded_step, rank, world):
    """The ranks hold additive SHARES of the one volume's losses.  One more (untimed) sharded step with fixed Dropout3d masks
    and, on rank 0, the single-process step of the same sample with the same masks: the summed loss shares must be the
    single-GPU losses."""
    unet = net.mask.modified_u_net
    pm = parity_dropout_masks(cfg, 4)
    unet.dropout_masks = cdist.rank_dropout_masks(pm, 4, cdist.ShardContext())
    lp = torch.stack([l.detach().float() for l in one_sharded_step()])
    dist.all_reduce(lp)
    out = None
    if rank == 0:
        unet.dropout_masks = pm
        reducer.zero_grad()
        reducer.arm(sync=False)                 # single-process step: accumulate into the buckets, no collective
        _, ls, _ = step.training_step(net, sample)
        ls = [float(l.detach()) for l in ls]
        rel = [abs(a - b) / max(abs(b), 1e-12) for a, b in zip(lp.tolist(), ls)]
        out = {"what": "sum over the %d ranks' loss shares of one extra, untimed sharded step vs the "
                       "single-process step on rank 0 (same weights, sample, Dropout3d masks)" % world,
               "sharded_sum": lp.tolist(), "single": ls, "rel_diff": rel, "tolerance_rel": 5e-4,
               "ok": bool(max(rel) <= 5e-4)}
    unet.dropout_masks = None
    return out


def main():
    args = parse_args()
    backend = os.environ.get("CFUN_BENCH_BACKEND", "nccl")     # "nccl" is RCCL on ROCm; gloo: smoke test on fewer GPUs
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: become the N-rank job (one process per GPU) instead of printing an
        # N = 1 number under the wrong label
        if not args.preflight_only or torch.cuda.is_available():
            check_gpu_count(args.gpus, backend)
        sys.stdout.flush()
        os.execve(sys.executable, launch_command(sys.argv[1:], args.gpus), dict(os.environ))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE = %d (launch with --nproc-per-node %d, or plain `python "
                         "bench.py --gpus %d` which launches the ranks itself)" % (args.gpus, world, args.gpus, args.gpus))
    # (a GPU-less host runs nothing of the product; --preflight-only on CPU tensors exists for the CPU test tier, which
    # points CFUN_LIB_PATH at the HIP emulator build of the same kernel sources)
    cpu_preflight = args.preflight_only and not torch.cuda.is_available() and backend == "gloo" and os.environ.get("CFUN_LIB_PATH")
    if not torch.cuda.is_available() and not cpu_preflight:
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if cpu_preflight:
        dev = torch.device("cpu")
    else:
        check_gpu_count(world, backend)
        local %= torch.cuda.device_count()    # (several ranks on one GPU only under CFUN_BENCH_BACKEND=gloo)
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    dist = None
    preflight = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
        from cfun_amd import dist as cdist, dist_selftest
        # every communication pattern of the path once, against single-process results, BEFORE anything is timed
        preflight = dist_selftest.preflight(dev)
        if rank == 0 and not preflight["ok"]:
            sys.stderr.write("communication pre-flight FAILED: %s\n" % json.dumps(preflight))
    elif args.preflight_only:
        raise SystemExit("--preflight-only needs --gpus N > 1")
    if args.preflight_only:
        if rank == 0:
            print(json.dumps({"n_gpus": world, "backend": backend, "preflight": preflight}), flush=True)
        dist.destroy_process_group()
        sys.exit(0 if preflight["ok"] else 4)

    from cfun_amd import config, ops, step
    stage, h, w, d = WORKLOADS[args.workload]
    cfg = config.LiTSConfig(stage) if args.workload == "cfg4" else config.heart_config(stage, h, w, d)
    if args.workload == "cfg4":       # BASELINE configs[4]: "LiTS_2017 config, same pipeline" -- all heads and losses in
        cfg.STAGE_SPLIT = False       # one step (the fork itself trains detector and mask branch in separate phases)
    torch.manual_seed(0)                       # identical replicated weights on every rank
    net = step.CFUNHotPath(cfg).to(dev)
    sharded = args.sharded and world > 1
    sample = step.synthetic_inputs(cfg, dev, seed=0 if sharded else rank)
    assert sample["p_rois"].shape[0] == 4 and sample["n_rois"].shape[0] == 8   # heads must not be skipped
    timer = ops.LaunchTimer(dominant_kernel_match(cfg))
    # N > 1: data-parallel replicas (one volume per GPU).  Gradients are averaged in 64 MB flat buckets whose RCCL
    # all-reduce is issued on a side stream as soon as backward has filled them (cfun_amd.dist.GradientReducer)
    reducer = None
    if world > 1:
        cdist.prepare_zshard_groups(device=dev)     # the z-shard sub-groups (collective new_group calls: at set-up, every rank)
        reducer = cdist.GradientReducer(net.parameters(), average=not sharded)      # sharded: additive shares -> sum

    step_no = [0]

    def one_sharded_step():     # the ranks' loss shares / gradients add up to the single-GPU step (the reducer sums)
        step_no[0] += 1
        reducer.zero_grad()
        with cdist.depth_sharded():
            losses, total, _ = cdist.sharded_training_step(net, sample_s, dropout_seed=step_no[0])
        reducer.finish()
        return losses

    def one_step():
        if sharded:
            return one_sharded_step()
        if reducer is None:
            net.zero_grad(set_to_none=True)
        else:
            reducer.zero_grad()
        out, losses, total = step.training_step(net, sample)
        if reducer is not None:
            reducer.finish()
        return losses

    def fence():
        torch.cuda.synchronize()
        if world > 1:     # RCCL barriers run on a device: name this rank's (the default guess is rank % device_count)
            dist.barrier(device_ids=[local]) if backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup):
        """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize on both sides; MAX over ranks."""
        losses = None
        for _ in range(warmup):
            losses = fn()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            losses = fn()
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed, losses

    sample_s = sample            # (the sharded step's sample: every rank holds the same volume)
    for _ in range(args.warmup):
        losses = one_step()
    fence()
    ops.set_launch_timer(timer)
    elapsed, losses = timed(one_step, args.steps, 0)
    ops.set_launch_timer(None)
    sharded_parity = None
    if sharded:
        lt = torch.stack([l.detach().float() for l in losses])      # the ranks' shares: report their sum
        dist.all_reduce(lt)
        losses = list(lt)
        sharded_parity = sharded_parity_check(cdist, step, dist, net, cfg, sample, reducer, one_sharded_step, rank, world)
        fence()
    want_sharded_leg = world > 1 and not sharded and not args.no_sharded_leg and d % (16 * world) == 0
    lv = [float(l.detach()) for l in losses]
    assert all(v == v and abs(v) != float("inf") for v in lv), "non-finite loss: %s" % lv

    if rank == 0:
        b = cfg.UNET_MASK_BRANCH_CHANNEL
        side = cfg.MASK_POOL_SIZE
        n_roi_launch = len(range(0, 4, world)) if sharded else 4                     # RoIs of rank 0's mask-head launches
        flops = 2.0 * (2 * b) * (2 * b) * 27 * side[0] * side[1] * side[2] * n_roi_launch   # per launch
        durs = timer.durations_ms("mfma")
        durs_h = timer.durations_ms("hbm")
        kern, executed, kname = dominant_kernel_info(cfg, n_roi_launch)
        t_k = sum(durs) / max(len(durs), 1) * 1e-3
        achieved = flops / t_k / 1e12 if t_k > 0 else 0.0
        result = {
            "metric": "volumes/sec fwd+bwd, 256x256x128 8-class CT" if args.workload == "cfg2"
                      else "volumes/sec fwd+bwd (%s)" % args.workload,
            "value": (1 if sharded else world) * args.steps / elapsed, "unit": "volumes/s", "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s: %dx%dx%d CT, stage '%s', 4 positive + 8 negative RoIs, U-Net b=%d, "
                                   "%s crops -> %s masks, 6 losses%s, fwd+bwd"
                                   % (args.workload, h, w, d, stage, b, "x".join(map(str, side)),
                                      "x".join(map(str, cfg.MASK_SHAPE)),
                                      " incl. 3-D Sobel edge loss" if stage == "finetune" else ""),
                       "parallelism": ("ONE volume over %d GPUs: depth-sharded FPN/RPN with xGMI halo exchange overlapped "
                                       "with the interior planes, RPN all-gather, classifier RoIs round-robin, every "
                                       "positive RoI's U-Net z-sharded over world/4 ranks when world > 4 (else one RoI per "
                                       "rank), gradient all-reduce; losses = the sum of the ranks' shares" % world) if sharded else
                                      ("1 volume per GPU x %d, bucketed gradient all-reduce (RCCL) overlapped with backward"
                                       % world) if world > 1 else "single GPU"},
            "losses": lv, "when": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                         "traffic": None, "mfma_util_pmc": None,
                         "kernel": "%s (conv_norm_lrelu_l4.0: 3x3x3 %d->%d @ %dx%s)"
                                   % (kname, 2 * b, 2 * b, n_roi_launch, "x".join(map(str, side))),
                         "flops_per_launch": flops, "flops_basis": "algorithmic: 2 * Ci * Co * 27 * voxels of the direct "
                         "convolution (SURVEY.md section 8d), whatever the kernel executes",
                         "mfma_flops_executed_per_launch": executed,
                         "mfma_pipe_frac": executed / t_k / 1e12 / PEAK_FP32_MFMA_TFLOPS if t_k > 0 else 0.0,
                         "frac_note": "`frac` = algorithmic flops / time / peak, as the bench contract defines it; the kernel "
                                      "executes fewer MACs (Winograd domain) on padded channel tiles, so the share "
                                      "of the matrix pipe it keeps busy is `mfma_pipe_frac` (instruction count) / "
                                      "`mfma_util_pmc` (SQ_VALU_MFMA_BUSY_CYCLES) -- read those as the utilisation figure",
                         "avg_launch_ms": t_k * 1e3, "launches_timed": len(durs)},
        }
        if args.workload == "cfg2" and not args.no_hbm_loop and kern == 2:
            # the same conv back to back outside the step: kernel only (comparable to rocprofv3's per-kernel average under
            # profiles/) and with the per-call weight transform a caller outside a WeightScope pays
            t_ko, t_wt = dominant_back_to_back(cfg, n_roi_launch, dev)
            result["roofline"].update(
                avg_launch_ms_note="`avg_launch_ms` (behind `frac`): one HIP-event pair per call INSIDE the step -- the conv "
                                   "kernel plus the finalize of its epilogue's InstanceNorm statistics, the mask head sharing the "
                                   "GPU with the detector stream; the weight transform is NOT in it (hoisted into the step's one "
                                   "cfun_weight_prepare launch)",
                avg_launch_ms_kernel_only=t_ko * 1e3, frac_kernel_only=flops / t_ko / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                avg_launch_ms_with_weight_transform=t_wt * 1e3,
                frac_with_weight_transform=flops / t_wt / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                back_to_back="8 calls between one HIP-event pair, 4 repetitions, first dropped, median of 3 (as roofline_hbm)")
        for leg in ("dgrad", "wgrad"):      # the same conv's data / weight gradient calls inside the step (HIP events, incl. the
            dl = timer.durations_ms("mfma_" + leg)      # weight-gradient chunk reduction; the mask head shares the GPU with the detector stream)
            if dl:
                result["roofline"]["in_step_%s_ms" % leg] = sum(dl) / len(dl)
        if ops.WGRAD_STREAM:
            result["roofline"]["in_step_note"] = ("since round 5 the weight gradient runs on its own HIP stream BESIDE the data "
                                                  "gradient (cfun_amd.ops.WGRAD_STREAM): the two in-step event pairs above time "
                                                  "kernels that share the GPU -- each is slower than alone, their sum is not "
                                                  "step time; the isolated figures are tools/bench_layers.py's (profiles/)")
        if args.workload == "cfg2":
            r = result["roofline"]
            r["traffic"], r["traffic_source"] = pmc_record(PMC_TRAFFIC_JSON, "traffic_bytes_per_launch")
            r["mfma_util_pmc"], r["mfma_util_pmc_source"] = pmc_record(PMC_MFMA_JSON, "mfma_util")
        if sharded_parity is not None:
            result["sharded_parity"] = sharded_parity
        if durs_h:   # north_star's "HBM roofline on the 3x3x3 conv kernel": the C_in = 1 stem, algorithmic bytes / time
            t_step = sum(durs_h) / len(durs_h) * 1e-3       # one event pair per launch inside the step (~10 us of overhead)
            t_h, nb2b, hname = stem_back_to_back(cfg, net, n_roi_launch, dev, launches=1 if args.no_hbm_loop else 64)
            vox = n_roi_launch * side[0] * side[1] * side[2]
            nbytes = 4.0 * (vox + vox * b + 27 * b)
            result["roofline_hbm"] = {
                "bound": "hbm", "achieved": nbytes / t_h / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": nbytes / t_h / 1e9 / PEAK_HBM_GBS, "traffic": None,
                "kernel": "%s (conv3d_c1_1: 3x3x3 1->%d @ %dx%d^3)" % (hname, b, n_roi_launch, side[0]),
                "bytes_per_launch": nbytes, "avg_launch_ms": t_h * 1e3, "launches_timed": nb2b,
                "repetitions_ms": list(B2B_REPS),
                "timing": "%d back-to-back launches of the step's own C-ABI call between ONE pair of HIP events on the "
                          "launch stream (launch-to-launch rate), four repetitions, the first dropped (clock ramp after the "
                          "idle queue), median of the other three; all four in `repetitions_ms`" % nb2b,
                "avg_launch_ms_in_step": t_step * 1e3, "launches_timed_in_step": len(durs_h),
                "frac_in_step": nbytes / t_step / 1e9 / PEAK_HBM_GBS,
                "methodology": "since round 3 `frac` is the back-to-back launch rate above; rounds 1-2 reported the in-step "
                               "per-launch event timing, which this line keeps as `frac_in_step` / `avg_launch_ms_in_step` "
                               "(one event pair per launch: ~10 us of event overhead on an ~80 us kernel); until round 3 the "
                               "back-to-back figure was ONE repetition (= repetitions_ms[0]) -- compare like with like "
                               "across rounds"}
            if not args.no_hbm_loop:
                t_p, pbytes = pointwise_back_to_back(cfg, net, n_roi_launch, dev)
                result["roofline_hbm"]["second_leg"] = {
                    "kernel": "k_conv_pointwise_t<8, 40> (conv3d_l4: 1x1x1 %d->%d @ %dx%d^3, rows staged through LDS)"
                              % (2 * b, cfg.NUM_CLASSES, n_roi_launch, side[0]),
                    "achieved": pbytes / t_p / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": pbytes / t_p / 1e9 / PEAK_HBM_GBS,
                    "bytes_per_launch": pbytes, "avg_launch_ms": t_p * 1e3, "launches_timed": 64,
                    "repetitions_ms": list(B2B_REPS)}
                if net.fused_mask_losses():       # the step's unhidden HBM-bound pair, alone
                    t_lf, t_lb, lbytes = mask_losses_back_to_back(cfg, n_roi_launch, dev)
                    result["roofline_hbm"]["mask_losses"] = {
                        "kernel": "k_mask_fused_fwd (softmax + CE + 3-D Sobel edge loss, one pass over the logits) / k_mask_fused_bwd "
                                  "(2-D stencil per plane + softmax backward) @ %dx%s" % (n_roi_launch, "x".join(map(str, cfg.MASK_SHAPE))),
                        "bytes_per_pass": lbytes, "fwd_ms": t_lf * 1e3, "bwd_ms": t_lb * 1e3,
                        "fwd_frac": lbytes / t_lf / 1e9 / PEAK_HBM_GBS, "bwd_frac": lbytes / t_lb / 1e9 / PEAK_HBM_GBS,
                        "timing": "each pass between its own HIP-event pair (the backward incl. autograd's two scalar launches), "
                                  "median of 5 after 2 warm-up rounds"}
                # both C_in = 1 convs of the path on the line: the P3D stem is NOT HBM-bound -- 147
                # taps per output at 49 flop/B put it on the vector ALU; both fractions are reported, the binding one is `bound`
                t_s, sbytes, sflops = p3d_stem_back_to_back(cfg, net, dev)
                result["roofline_hbm"]["p3d_stem"] = {
                    "kernel": "k_conv_stem<3,7,7,2,16> + folded BN + ReLU, then k_maxpool2 (fpn.C1: Conv3d 1->16 k(3,7,7) s2 + BN + "
                              "ReLU + MaxPool3d(2,2) on the %dx%dx%d volume, backbone.py:123-128)" % (h, w, d),
                    "bytes_per_call": sbytes, "flops_per_call": sflops, "avg_call_ms": t_s * 1e3,
                    "hbm_frac": sbytes / t_s / 1e9 / PEAK_HBM_GBS, "valu_frac": sflops / t_s / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                    "bound": "valu (fp32 vector peak = %.1f TFLOP/s with packed FMAs)" % PEAK_FP32_MFMA_TFLOPS,
                    "launches_timed": 16, "repetitions_ms": list(B2B_REPS)}
        parity_fail = None
        if world == 1 and not args.no_cpu_baseline:
            # full-size parity check: ONE more (untimed) GPU step with the Dropout3d masks the oracle leg uses, so that the
            # six losses of the two legs are the same computation and must agree
            unet = net.mask.modified_u_net
            prev_masks = unet.dropout_masks
            unet.dropout_masks = parity_dropout_masks(cfg, 4)
            try:
                gl = [float(l.detach()) for l in one_step()]
            finally:
                unet.dropout_masks = prev_masks
            torch.cuda.synchronize()
            named = dict(net.named_parameters())
            pkeys = grad_parity_keys(net)
            gg = {k: named[k].grad.detach().cpu().clone() for k in pkeys if k in named and named[k].grad is not None}
            torch.cuda.empty_cache()
            phys, _ = physical_cores()
            cb = cpu_baseline(cfg, net, sample, threads=max(1, min(phys, 64)), iters=args.cpu_baseline_iters, grad_keys=pkeys)
            cg = cb.pop("_grads") or {}
            result["cpu_baseline"] = cb
            rel = [abs(g - c) / max(abs(c), 1e-12) for g, c in zip(gl, cb["losses"])]
            tol = 1e-4
            result["loss_parity"] = {"what": "the six losses of one extra, untimed GPU step vs the oracle's CPU step at the "
                                             "benchmarked size -- same weights, inputs, RoIs and Dropout3d masks (seed 1)",
                                     "gpu": gl, "cpu_oracle": cb["losses"], "rel_diff": rel, "tolerance_rel": tol,
                                     "ok": bool(max(rel) <= tol)}
            if max(rel) > tol:
                parity_fail = "loss parity FAILED at full size: rel diff %s > %g" % (rel, tol)
            # ... and the parameter gradients of the same two steps (the oracle leg calls backward() anyway)
            ref64 = load_grad_fp64(net, cfg, args.workload)
            gp = {}
            for k, bound in pkeys.items():
                if k in gg and k in cg:
                    a, c = gg[k].double(), cg[k].double()
                    e = {"rel_l2_vs_cpu_fp32": float((a - c).norm() / c.norm().clamp(min=1e-300)), "norm_cpu": float(c.norm())}
                    if ref64 is not None and k in ref64:
                        floor = ref64[k][2]
                        e.update(rel_l2_vs_fp64=grad_fp64_error(a, ref64[k]), cpu_fp32_vs_fp64=grad_fp64_error(c, ref64[k]),
                                 reference_fp32_floor=floor)
                        # the reference arithmetic's deviation from fp64: the larger of the fixture's two evaluations (torch CPU
                        # fp32 with 8 and with 96 threads) and THIS run's CPU oracle leg -- it moves with the thread count where a
                        # LeakyReLU kink flip reaches the tensor (l4.0: 1.8e-4 / 6.8e-4)
                        # (the in-run term is capped at twice the fixture's recorded floor, so a noisy CPU leg
                        # cannot widen the bound the GPU is held to without limit)
                        floor = min(max(floor, e["cpu_fp32_vs_fp64"]), 2.0 * floor)
                        e.update(bound=GRAD_FP64_FACTOR * floor + GRAD_FP64_FLOOR,
                                 rule="relL2(GPU, fp64) <= %g * relL2(reference fp32, fp64) + %g" % (GRAD_FP64_FACTOR, GRAD_FP64_FLOOR))
                        e["rel_l2"] = e["rel_l2_vs_fp64"]
                    else:
                        e.update(rel_l2=e["rel_l2_vs_cpu_fp32"], bound=bound, rule="relL2(GPU, CPU fp32) <= bound")
                    gp[k] = e
            ok = bool(gp) and all(v["rel_l2"] <= v["bound"] for v in gp.values())
            result["grad_parity"] = {"what": "parameter gradients of the same extra GPU step at the benchmarked size: all 27 U-Net "
                                             "conv weights against the mask head's fp64 gradients (%s; bound = 3 x the deviation "
                                             "of the reference's own fp32 arithmetic from them + 2e-5), detector tensors "
                                             "against the oracle's CPU fp32 backward" % GRAD_FP64_FIXTURE,
                                     "fp64_fixture_applies": ref64 is not None, "tensors": gp, "ok": ok}
            if not ok and parity_fail is None:
                parity_fail = "gradient parity FAILED at full size: %s" % {k: (v["rel_l2"], v["bound"]) for k, v in gp.items()}
        if sharded_parity is not None and not sharded_parity["ok"]:
            parity_fail = "sharded step does not reproduce the single-GPU losses: rel diff %s" % sharded_parity["rel_diff"]
        if preflight is not None:
            result["preflight"] = preflight
            result["rccl_ranks_seen"] = preflight["rccl_ranks_seen"]
            result["backend"] = backend + ("" if backend == "nccl" else " (NOT RCCL: %d ranks on %d device(s), collectives on "
                                           "the host -- a smoke test of the N-rank path, not a scaling measurement)"
                                           % (world, preflight["devices_seen"]))
            if not preflight["ok"] and parity_fail is None:
                parity_fail = "communication pre-flight FAILED: %s" % preflight["max_rel_err"]
    else:
        result, parity_fail = None, None

    if want_sharded_leg:
        # the OTHER curve SURVEY.md section 8(e) asks to report, from the same invocation: ONE volume per step over all ranks
        # (strong scaling; model.py:1391-1514 sharded as cfun_amd.dist describes).  `value` above stays the data-parallel
        # figure; this leg has its own reducer (sum, not mean), warm-up, timed region and the parity check against the
        # single-process step.  A watchdog keeps a stuck collective in this EXTRA leg from costing the run its headline:
        # after the limit rank 0 prints the finished data-parallel line without the leg and every rank leaves.
        import threading
        limit = float(os.environ.get("CFUN_BENCH_SHARDED_LEG_TIMEOUT", "420"))

        def bail():
            if rank == 0:
                result["sharded_one_volume"] = {"error": "the one-volume leg did not finish within %.0f s; dropped" % limit}
                print(json.dumps(result), flush=True)
            os._exit(0)
        dog = threading.Timer(limit, bail)
        dog.daemon = True
        dog.start()
        leg = None
        try:
            reducer.remove()
            reducer = cdist.GradientReducer(net.parameters(), average=False)
            sample_s = step.synthetic_inputs(cfg, dev, seed=0)
            sharded = True
            el, ls = timed(one_sharded_step, args.steps, max(1, min(args.warmup, 2)))
            lt = torch.stack([l.detach().float() for l in ls])
            dist.all_reduce(lt)
            par = sharded_parity_check(cdist, step, dist, net, cfg, sample_s, reducer, one_sharded_step, rank, world)
            fence()
            leg = {"what": "ONE volume per step over all %d ranks: depth-sharded FPN/RPN with halo exchange overlapped with "
                           "the interior planes, one candidate all-gather, classifier RoIs round-robin, positive RoIs' U-Nets "
                           "one per rank / z-sharded over world/4 ranks when world > 4, gradient all-reduce (sum)" % world,
                   "value": args.steps / el, "unit": "volumes/s", "ms_per_step": 1e3 * el / args.steps, "scaling": "strong",
                   "steps": args.steps, "losses": lt.tolist(), "sharded_parity": par}
        except Exception as e:       # (an error every rank raises alike; a one-sided failure ends in the watchdog)
            leg = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        dog.cancel()
        if rank == 0:
            result["sharded_one_volume"] = leg
            if leg.get("sharded_parity") is not None:
                result["sharded_parity"] = leg["sharded_parity"]
                if not leg["sharded_parity"]["ok"] and parity_fail is None:
                    parity_fail = "sharded step does not reproduce the single-GPU losses: rel diff %s" % leg["sharded_parity"]["rel_diff"]
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if parity_fail:
        sys.stderr.write(parity_fail + "\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
