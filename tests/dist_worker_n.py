"""Worker for the world-4 / world-8 tests of the depth-sharded layout: BASELINE configs[3] is ONE
volume over the 8 GPUs of a node, so what must be exercised is everything a 2-rank run cannot reach --

  * INTERIOR ranks (a previous AND a next neighbour: four P2P operations per exchange, forward and backward);
  * slabs as thin as the 8-way split of the real volumes makes them (256 / 8 = 32 input planes -> 2 planes at p3; here
    1 and 2 p3 planes per rank), i.e. the padded-slab fall-back of ``dist.halo_conv`` next to its interior / edge split;
  * the z-shard plan of 8 GPUs: 4 positive RoIs x sub-groups of 2 consecutive ranks (``dist.prepare_zshard_groups`` /
    ``zshard_plan``; the collective ``new_group`` order), and a RoI spread over 4 / 8 ranks, where the U-Net's slabs are
    thinner than the folded levels' planes;
  * the ordered ``GradientReducer`` with autograd graphs that differ per rank (ranks without any RoI).

tests/test_dist_gloo.py (CPU tier) runs it on CPU tensors through the HIP emulator build of the kernels;
tests/test_dist_gpu.py (GPU tier) runs the SAME sections with 4 processes on cuda:0 and the real library (gloo
collectives: RCCL refuses several ranks on one device).  argv: rank world port out-pattern [device] [sections]"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def flat_grads(named):
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for _, p in named])


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    backend = os.environ.get("CFUN_DIST_BACKEND", "gloo")        # "nccl": RCCL, one GPU per rank (test_rccl_two_gpus)
    dev = torch.device(sys.argv[5] if len(sys.argv) > 5 else "cpu")
    if backend == "nccl":
        dev = torch.device("cuda", rank % torch.cuda.device_count())
    sections = (sys.argv[6] if len(sys.argv) > 6 else "halo,conv,rpn,step,rr,dp,unet").split(",")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    torch.set_num_threads(1)            # `world` processes share the host's cores
    if dev.type == "cpu":
        os.environ["CFUN_CONV_ALGO"] = os.environ.get("CFUN_DIST_WORKER_ALGO", "direct")
    else:                               # GPU tier: the real library, AUTO algorithm
        os.environ.pop("CFUN_LIB_PATH", None)
        os.environ.pop("CFUN_CONV_ALGO", None)
        torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from cfun_amd import config as ccfg
    from cfun_amd import dist as cdist
    from cfun_amd import step
    from cfun_amd.layers import Conv3dParams
    import module_cases as mc

    # the sub-groups of every z-shard plan, created once at set-up by every rank in the same order (new_group is collective)
    plans = cdist.prepare_zshard_groups()
    res = {"plan_sizes": np.array(sorted(plans.keys()))}
    g = torch.Generator().manual_seed(0)

    if "halo" in sections:
        print("[rank %d] section halo" % rank, flush=True)
        # 1) halo exchange, 2 planes per rank, forward + backward against plain padding of the full tensor
        full = torch.randn(1, 2 * world, 3, 4, 4, generator=g).to(dev)
        gy_full = torch.randn(1, 4 * world, 3, 4, 4, generator=g).to(dev)            # per-rank padded slabs, concatenated
        with cdist.depth_sharded():
            x = cdist.slab(full, dim=1).clone().requires_grad_(True)
            y = cdist.halo_exchange(x, 1, 1)
            (y * gy_full[:, 4 * rank:4 * rank + 4]).sum().backward()
        res["halo_y"], res["halo_gx"] = y.detach().cpu().numpy(), x.grad.cpu().numpy()

    if "conv" in sections:
        print("[rank %d] section conv" % rank, flush=True)
        # 2) depth-coupled convs trained through the exchange at 1, 2 and 4 planes per rank: thinner than the kernel (padded
        #    slab), edges only, and interior + edges (the overlapped split) -- 3x3x3 stride 1 on the direct / MFMA kernel and
        #    on the Winograd kernels (16 -> 32), and the stride-2 down-conv (halo from the previous rank only)
        for tag, ci, co, stride in (("d", 4, 8, 1), ("w", 16, 32, 1), ("s2", 4, 8, 2)):
            for planes in (1, 2, 4):
                if stride == 2 and planes == 1:
                    continue                                                         # (slabs start on multiples of the stride)
                torch.manual_seed(11)
                conv = Conv3dParams(ci, co, 3, stride=stride, padding=1).to(dev)
                D = planes * world
                xs = torch.randn(1, D, 4, 6, ci, generator=g).to(dev)
                gys = torch.randn(1, D // stride, 4 // stride, 6 // stride, co, generator=g).to(dev)
                with cdist.depth_sharded():
                    xl = cdist.slab(xs, dim=1).clone().requires_grad_(True)
                    yl = conv(xl)
                    (yl * cdist.slab(gys, dim=1)).sum().backward()
                    wg = torch.cat([conv.weight.grad.reshape(-1), conv.bias.grad.reshape(-1)])
                    dist.all_reduce(wg)
                key = "conv_%s%d_" % (tag, planes)
                res[key + "y"], res[key + "gx"], res[key + "gw"] = yl.detach().cpu().numpy(), xl.grad.cpu().numpy(), wg.cpu().numpy()
                if rank == 0:
                    conv.weight.grad = conv.bias.grad = None
                    xr = xs.clone().requires_grad_(True)
                    yr = conv(xr)
                    (yr * gys).sum().backward()
                    res["ref_" + key + "y"], res["ref_" + key + "gx"] = yr.detach().cpu().numpy(), xr.grad.cpu().numpy()
                    res["ref_" + key + "gw"] = torch.cat([conv.weight.grad.reshape(-1), conv.bias.grad.reshape(-1)]).cpu().numpy()

    tiny = dict(MASK_POOL_SIZE=[32, 32, 32], POOL_SIZE=[4, 4, 4], UNET_MASK_BRANCH_CHANNEL=4, TOP_DOWN_PYRAMID_SIZE=16,
                RPN_CONV_CHANNELS=16, FPN_CLASSIFY_FC_LAYERS_SIZE=16, RPN_ANCHOR_SCALES=(16, 32), PRE_NMS_LIMIT=64,
                POST_NMS_ROIS_TRAINING=16)

    def tiny_cfg(stage, hw, depth):
        cls = type("TinyHeartN", (ccfg.HeartConfig,), dict(tiny, IMAGE_MAX_DIM=hw, IMAGE_MIN_DIM=depth))
        cfg = cls(stage)
        side = 64 if stage == "finetune" else 32
        cfg.MASK_SHAPE = cfg.MINI_MASK_SHAPE = (side, side, side)
        return cfg

    if "rpn" in sections:
        print("[rank %d] section rpn" % rank, flush=True)
        # 3) depth-sharded FPN -> RPN -> proposals (ONE all-gather of candidates) == the single-process result, with 1 and
        #    with 2 p3 planes per rank (cfg3's 8-way split: 2)
        for planes3 in (1, 2):
            D = 16 * planes3 * world
            cfg = tiny_cfg("beginning", 32 if planes3 == 1 else 16, D)
            h, w = cfg.image_dhw[1:]
            torch.manual_seed(0)
            net = step.CFUNHotPath(cfg).to(dev).eval()
            image = torch.randn(1, 1, D, h, w, generator=g).to(dev)
            with torch.no_grad():
                with cdist.depth_sharded():
                    p2, p3, logits, probs, bbox, rois = cdist.sharded_backbone_rpn(net, cdist.slab(image, dim=2))
                key = "rpn%d_" % planes3
                res.update({key + "p2": p2.cpu().numpy(), key + "p3": p3.cpu().numpy(), key + "logits": logits.cpu().numpy(),
                            key + "bbox": bbox.cpu().numpy(), key + "rois": rois.cpu().numpy()})
                if rank == 0:
                    rp2, rp3, rlogits, rprobs, rbbox = net.backbone_rpn(image)
                    rrois = net.proposals(rprobs, rbbox, "inference")
                    res.update({"ref_" + key + "p2": rp2.cpu().numpy(), "ref_" + key + "p3": rp3.cpu().numpy(),
                                "ref_" + key + "logits": rlogits.cpu().numpy(), "ref_" + key + "bbox": rbbox.cpu().numpy(),
                                "ref_" + key + "rois": rrois.cpu().numpy()})

    def step_cases(cfg, prefix, cases, bucket_bytes):
        """ONE volume over `world` ranks ('finetune': edge loss, folded 5^3 conv) against the single-process step of rank 0:
        depth-sharded FPN / RPN, heads on RoI crops summed over the ranks' slabs, gradients summed through the ordered
        GradientReducer.  cases: (tag, n_pos, n_neg, zshard)."""
        torch.manual_seed(4)
        net = step.CFUNHotPath(cfg).to(dev)
        s0 = step.synthetic_inputs(cfg, dev, 0)
        b = cfg.UNET_MASK_BRANCH_CHANNEL
        gm = torch.Generator().manual_seed(9)
        masks = [torch.empty(4, c).bernoulli_(0.4, generator=gm) / 0.4 for c in (b, 2 * b, 4 * b, 8 * b, 16 * b)]
        named = [(k, p) for k, p in net.named_parameters() if p.requires_grad]
        unet = net.mask.modified_u_net
        red = cdist.GradientReducer(net.parameters(), bucket_bytes=bucket_bytes, average=False)
        assert len(red.buckets) > 2
        for tag, n_pos, n_neg, zshard in cases:
            tag = prefix + tag
            print("[rank %d] step case %s" % (rank, tag), flush=True)
            s = dict(s0)
            s["p_rois"], s["mask_labels"], s["n_rois"] = s0["p_rois"][:n_pos], s0["mask_labels"][:n_pos], s0["n_rois"][:n_neg]
            keep = list(range(n_pos)) + list(range(4, 4 + n_neg))
            s["target_class_ids"], s["target_deltas"] = s0["target_class_ids"][keep], s0["target_deltas"][keep]
            plan = cdist.zshard_plan(cdist.ShardContext(), n_pos) if zshard else None
            if plan is not None:                # this rank's sub-group shares RoI rank // rs
                roi = rank // plan[0]
                unet.dropout_masks = [mk[roi:roi + 1] for mk in masks]
            else:                               # round-robin: RoIs rank, rank + world, ...
                unet.dropout_masks = [mk[:n_pos][rank::world] for mk in masks]
            red.zero_grad()
            with cdist.depth_sharded():
                losses, _, rois = cdist.sharded_training_step(net, s, zshard_unet=zshard)
            red.finish()
            lv = torch.stack([l.detach().float() for l in losses])
            dist.all_reduce(lv)
            res[tag + "_losses"], res[tag + "_grads"] = lv.cpu().numpy(), flat_grads(named).cpu().numpy()
            res[tag + "_rois"] = rois.detach().cpu().numpy()
            res[tag + "_zsharded"] = np.array([0 if plan is None else plan[0]])
            if rank == 0:
                unet.dropout_masks = [mk[:n_pos] for mk in masks]
                red.zero_grad()
                red.arm(sync=False)             # the single-process step accumulates into the buckets, no collective
                out_r, losses_r, _ = step.training_step(net, s)
                res["ref_" + tag + "_losses"] = np.array([float(l.detach()) for l in losses_r], np.float32)
                res["ref_" + tag + "_grads"] = flat_grads(named).cpu().numpy()
                res["ref_" + tag + "_rois"] = out_r["rpn_rois"].detach().cpu().numpy()
        red.remove()
        for _, p in named:
            p.grad = None
        if rank == 0:
            res[prefix + "grad_sizes"] = np.array([int(p.numel()) for _, p in named])
            res[prefix + "grad_names"] = np.array([k for k, _ in named])

    cases = []
    # (a) world / 2 positive RoIs, each U-Net z-sharded over a sub-group of 2 ranks (8 GPUs: the 4 x 2 plan of cfg3);
    # (b) ONE positive RoI over all `world` ranks: 32 / world planes per rank at level 1, one plane at the fold
    if "step" in sections or "stepa" in sections:
        cases += [("za", world // 2, world, True)]
    if "step" in sections or "stepb" in sections:
        cases += [("zb", 1, 3, True)]
    if "rr" in sections:
        # (c) round-robin heads with more ranks than RoIs: 2 positive + 1 negative RoI, ranks >= 3 hold no RoI at all and
        #     ranks >= 2 no mask RoI (their U-Net buckets never complete through the hooks)
        cases += [("rr", 2, 1, False)]
    if cases:                                   # tiny channel counts, 1 p3 plane per rank
        step_cases(tiny_cfg("finetune", 32, 16 * world), "", cases, 4096)
    if "lits" in sections:
        # the LiTS fork's shapes and mask losses (BASELINE configs[4]: "same pipeline", all heads in one step): P3D35 with the
        # 5x7x7 stem (depth halo 2 / 1), 3 classes, non-cubic 32x48x32 crops, class-weighted CE (global weight sum) and the
        # raw-Sobel edge loss on slabs -- z-sharded RoIs and round-robin heads
        lcfg = mc.tiny_lits_config("finetune", 32, 16 * world)
        lcfg.STAGE_SPLIT = False
        step_cases(lcfg, "l_", [("z", max(world // 2, 1), 2, True), ("rr", 2, 1, False)], 4096)
    if "litsplit" in sections:
        # the fork's DEFAULT: two training phases (LiTSConfig.STAGE_SPLIT = True; LiTS_2017/model.py:985-1001, 1518-1548) --
        # 'beginning' = detector only (no mask head, mask losses 0), 'finetune' = mask branch only (FPN / RPN frozen, no
        # classifier head, detector losses 0) -- through the sharded step, z-sharded and round-robin
        for stg, pre in (("beginning", "sb_"), ("finetune", "sf_")):
            scfg = mc.tiny_lits_config(stg, 32, 16 * world)
            assert scfg.STAGE_SPLIT
            step_cases(scfg, pre, [("z", max(world // 2, 1), 2, True), ("rr", 2, 1, False)], 4096)
    if "cfg1" in sections:
        # BASELINE configs[1]'s volume (128x128x64) with the REAL channel counts, 'finetune', 96^3 -> 192^3 masks: (a) the 4 + 8
        # RoIs of the benchmarked step, one positive RoI per rank when world == 4; (b) 2 positive RoIs, each U-Net z-sharded
        # over world / 2 ranks (b = 20: slabs of 96 / rs and 48 / rs planes, all-reduced InstanceNorm statistics)
        step_cases(ccfg.heart_config("finetune", 128, 128, 64), "c1_", [("rr", 4, 8, False), ("z", 2, 4, True)], 64 << 20)

    if "dp" in sections:
        print("[rank %d] section dp" % rank, flush=True)
        # 4b) data-parallel replicas of the WHOLE step (bench.py --gpus N): every rank its own volume (sample seed = rank),
        #     the mask head and its backward on their own HIP stream (step.OVERLAP_MASK_HEAD), 64 MB buckets that mix mask-head
        #     and detector gradients, the bucket all-reduces launched from the hooks under whichever stream produced a
        #     bucket's last gradient (the communication stream must also wait for the MAIN stream) --
        #     against the mean of the ranks' single-process gradients, computed on rank 0
        cfgd = ccfg.heart_config("finetune", 64, 64, 32) if dev.type == "cuda" else tiny_cfg("finetune", 32, 16)
        torch.manual_seed(5)
        netd = step.CFUNHotPath(cfgd).to(dev)
        namedd = [(k, p) for k, p in netd.named_parameters() if p.requires_grad]
        bd = cfgd.UNET_MASK_BRANCH_CHANNEL

        def masks_of(r):
            gd = torch.Generator().manual_seed(30 + r)
            return [torch.empty(4, c).bernoulli_(0.4, generator=gd) / 0.4 for c in (bd, 2 * bd, 4 * bd, 8 * bd, 16 * bd)]
        def sample_of(r):                       # (emulator tier: 1 positive + 2 negative RoIs, the step is slow there)
            s = step.synthetic_inputs(cfgd, dev, seed=r)
            if dev.type == "cuda":
                return s, 4
            s["p_rois"], s["mask_labels"], s["n_rois"] = s["p_rois"][:1], s["mask_labels"][:1], s["n_rois"][:2]
            s["target_class_ids"], s["target_deltas"] = s["target_class_ids"][[0, 4, 5]], s["target_deltas"][[0, 4, 5]]
            return s, 1
        redd = cdist.GradientReducer(netd.parameters(), average=True)
        for it in range(2 if dev.type == "cuda" else 1):        # second step: re-armed buckets, streams already warm
            sd, npd = sample_of(rank)
            netd.mask.modified_u_net.dropout_masks = [m[:npd] for m in masks_of(rank)]
            redd.zero_grad()
            step.training_step(netd, sd)
            redd.finish()
        res["dp_grads"] = flat_grads(namedd).cpu().numpy()
        redd.remove()
        if rank == 0:
            mean = None
            for r_ in range(world):
                for _, p in namedd:
                    p.grad = None
                sd, npd = sample_of(r_)
                netd.mask.modified_u_net.dropout_masks = [m[:npd] for m in masks_of(r_)]
                step.training_step(netd, sd)
                gflat = flat_grads(namedd) / world
                mean = gflat if mean is None else mean + gflat
            res["ref_dp_grads"] = mean.cpu().numpy()
            res["dp_sizes"] = np.array([int(p.numel()) for _, p in namedd])
            res["dp_names"] = np.array([k for k, _ in namedd])

    if "unet" in sections:
        print("[rank %d] section unet" % rank, flush=True)
        # 5) ONE RoI's U-Net ('finetune', Dropout3d active) z-sharded over ALL ranks: logits slabs and summed gradients
        from cfun_amd.mask_branch import Modified3DUNet
        torch.manual_seed(7)
        unet = Modified3DUNet(1, 8, "finetune", 4).to(dev)
        unet.train()
        g7 = torch.Generator().manual_seed(17)
        unet.dropout_masks = [torch.empty(1, c).bernoulli_(0.4, generator=g7) / 0.4 for c in (4, 8, 16, 32, 64)]
        x7 = torch.randn(1, 32, 32, 32, 1, generator=g7).to(dev)
        gy7 = torch.randn(1, 64, 64, 64, 8, generator=g7).to(dev)
        zs = cdist.ShardContext()
        named7 = list(unet.named_parameters())
        y7 = unet.forward_ndhwc(cdist.slab(x7, dim=1, shard=zs).contiguous(), zshard=zs)
        (y7 * cdist.slab(gy7, dim=1, shard=zs)).sum().backward()
        flat7 = flat_grads(named7)
        dist.all_reduce(flat7)
        res["zu_y"], res["zu_g"] = y7.detach().cpu().numpy(), flat7.cpu().numpy()
        if rank == 0:
            for _, p_ in named7:
                p_.grad = None
            yr7 = unet.forward_ndhwc(x7)
            (yr7 * gy7).sum().backward()
            res["ref_zu_y"], res["ref_zu_g"] = yr7.detach().cpu().numpy(), flat_grads(named7).cpu().numpy()
            res["zu_sizes"] = np.array([int(p_.numel()) for _, p_ in named7])
            res["zu_names"] = np.array([k for k, _ in named7])
    np.savez(out % rank, **res)
    dist.barrier(device_ids=[dev.index]) if backend == "nccl" else dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
