"""CPU tier: the multi-GPU path (cfun_amd.dist) over gloo with 2 ranks (tests/dist_worker.py: halo exchange forward and
backward, depth-sharded FPN -> RPN -> proposal all-gather vs the single-rank result, depth-coupled convs trained through the
exchange, the whole sharded step, the bucketed reducer, gradient accumulation) and with 4 and 8 ranks (tests/dist_worker_n.py,
round 5: interior ranks, slabs of 1 - 2 p3 planes, the 4 RoIs x 2 ranks z-shard plan of an 8-GPU node, idle ranks).  The
kernels run through the HIP emulator build (CPU tensors); tests/test_dist_gpu.py runs the same workers on the real library."""
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def run_world(tmp_path, env, world, worker="dist_worker.py", worker_args=(), timeout=900):
    """Spawn tests/<worker> as `world` ranks over gloo and return the ranks' result dicts."""
    import time
    port = _free_port()
    out = str(tmp_path / "rank%d.npz")
    logf = [open(str(tmp_path / ("rank%d.log" % r)), "wb") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", worker), str(r), str(world), port, out]
                              + list(worker_args), env=env, stdout=logf[r], stderr=subprocess.STDOUT)
             for r in range(world)]
    deadline = time.time() + timeout
    try:
        while any(p.poll() is None for p in procs):
            # one rank dying leaves its peers waiting in a collective: stop them instead of running into the timeout
            if any(p.poll() not in (None, 0) for p in procs) or time.time() > deadline:
                break
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
        for f in logf:
            f.close()
    logs = [open(str(tmp_path / ("rank%d.log" % r)), "rb").read().decode(errors="replace") for r in range(world)]
    bad = [r for r, p in enumerate(procs) if p.returncode != 0]
    first = [r for r in bad if procs[r].returncode != -9] or bad        # (the peers this harness stopped report -9)
    assert not bad, "rank %d of %d failed (exit codes %s):\n%s" % (
        first[0], world, [p.returncode for p in procs], logs[first[0]][-4000:])
    return [dict(np.load(out % k)) for k in range(world)]


def run_world2(tmp_path, env, worker_args=(), timeout=900):
    return run_world(tmp_path, env, 2, worker_args=worker_args, timeout=timeout)


def test_depth_sharding_world2(emu_lib, tmp_path):
    env = dict(os.environ, CFUN_LIB_PATH=emu_lib, PYTHONPATH=ROOT)
    check_world2(run_world2(tmp_path, env))


def check_world2(r, conv_tol=1.0):
    """The assertions shared by the CPU tier (emulator kernels) and the GPU tier (tests/test_dist_gpu.py, real kernels)."""

    # halo exchange: rank k's padded slab = planes [8k-1, 8k+5) of the zero-padded full tensor
    import torch
    g = torch.Generator().manual_seed(0)
    full = torch.randn(1, 8, 3, 4, 4, generator=g)
    gy_full = torch.randn(1, 12, 3, 4, 4, generator=g)
    padded = torch.nn.functional.pad(full, (0, 0, 0, 0, 0, 0, 1, 1)).numpy()
    gx_ref = np.zeros((1, 10, 3, 4, 4), np.float32)
    for k in range(2):
        np.testing.assert_array_equal(r[k]["halo_y"], padded[:, 4 * k:4 * k + 6])
        gx_ref[:, 4 * k:4 * k + 6] += gy_full[:, 6 * k:6 * k + 6].numpy()
    for k in range(2):
        np.testing.assert_allclose(r[k]["halo_gx"], gx_ref[:, 1 + 4 * k:5 + 4 * k], rtol=0, atol=1e-6)

    # sharded backbone / RPN / proposals
    ref = r[0]
    np.testing.assert_allclose(np.concatenate([r[0]["p2"], r[1]["p2"]], axis=1), ref["ref_p2"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(np.concatenate([r[0]["p3"], r[1]["p3"]], axis=1), ref["ref_p3"], rtol=1e-5, atol=1e-5)
    for k in range(2):
        np.testing.assert_allclose(r[k]["logits"], ref["ref_logits"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(r[k]["bbox"], ref["ref_bbox"], rtol=1e-5, atol=1e-5)
        assert r[k]["rois"].shape == ref["ref_rois"].shape
        np.testing.assert_allclose(r[k]["rois"], ref["ref_rois"], rtol=0, atol=1e-5)
    np.testing.assert_array_equal(r[0]["rois"], r[1]["rois"])     # every rank holds the identical proposal set

    # conv through the exchange
    np.testing.assert_allclose(np.concatenate([r[0]["conv_y"], r[1]["conv_y"]], axis=1), ref["ref_conv_y"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(np.concatenate([r[0]["conv_gx"], r[1]["conv_gx"]], axis=1), ref["ref_conv_gx"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r[0]["conv_gw"], ref["ref_conv_gw"], rtol=1e-4, atol=1e-5)
    # ... and through the Winograd kernels (16 -> 32 channels: slabs with depth padding 0, data gradient with padding 2)
    np.testing.assert_allclose(np.concatenate([r[0]["wconv_y"], r[1]["wconv_y"]], axis=1), ref["ref_wconv_y"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(np.concatenate([r[0]["wconv_gx"], r[1]["wconv_gx"]], axis=1), ref["ref_wconv_gx"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(r[0]["wconv_gw"], ref["ref_wconv_gw"], rtol=1e-4, atol=2e-5)

    # bucketed data-parallel gradient averaging: every rank ends with the mean of the per-rank gradients
    mean = 0.5 * (r[0]["dp_local"] + r[1]["dp_local"])
    for k in range(2):
        np.testing.assert_allclose(r[k]["dp_grads"], mean, rtol=1e-5, atol=1e-6)
    # gradient accumulation (two backward passes per optimizer step) with data-parallel ranks through FlatSGD: identical
    # parameters on both ranks, equal to the hand-written torch arithmetic; a parameter no rank reached did not move
    np.testing.assert_array_equal(r[0]["acc_params"], r[1]["acc_params"])
    n_ref = r[0]["acc_ref_params"].size
    np.testing.assert_allclose(r[0]["acc_params"][:n_ref], r[0]["acc_ref_params"], rtol=2e-5, atol=2e-6)
    assert float(r[0]["acc_never_moved"][0]) == 0.0 and float(r[1]["acc_never_moved"][0]) == 0.0
    assert np.abs(r[0]["dp_local"] - r[1]["dp_local"]).max() > 1e-3      # the ranks really saw different data

    # one volume over 2 ranks (sharded_training_step): loss shares and gradient sums equal the single-process step
    np.testing.assert_allclose(r[0]["sh_losses"], ref["ref_losses"], rtol=2e-4, atol=1e-6)
    np.testing.assert_array_equal(r[0]["sh_losses"], r[1]["sh_losses"])
    assert r[0]["sh_rois"].shape == ref["ref_rois5"].shape
    np.testing.assert_allclose(r[0]["sh_rois"], ref["ref_rois5"], rtol=0, atol=1e-5)
    off = 0
    for name, n in zip(ref["grad_names"], ref["grad_sizes"]):
        a, b = r[0]["sh_grads"][off:off + n], ref["ref_grads"][off:off + n]
        off += n
        den = max(np.linalg.norm(b), 1e-12)
        assert np.linalg.norm(a - b) / den < 2e-3 or np.abs(a - b).max() < 1e-6, (str(name), np.linalg.norm(a - b) / den)

    # R > RoIs of a kind + differing autograd graphs per rank, through the ordered GradientReducer (sum)
    np.testing.assert_allclose(r[0]["sh6_losses"], ref["ref6_losses"], rtol=2e-4, atol=1e-6)
    np.testing.assert_array_equal(r[0]["sh6_grads"], r[1]["sh6_grads"])      # both ranks hold the same reduced sums
    off = 0
    for name, n in zip(ref["grad_names"], ref["grad_sizes"]):
        a, b = r[0]["sh6_grads"][off:off + n], ref["ref6_grads"][off:off + n]
        off += n
        den = max(np.linalg.norm(b), 1e-12)
        assert np.linalg.norm(a - b) / den < 2e-3 or np.abs(a - b).max() < 1e-6, (str(name), np.linalg.norm(a - b) / den)

    # z-sharded per-RoI U-Net: slabs of the logits and summed gradients equal the single-process U-Net
    for stage in ("beginning", "finetune"):
        y = np.concatenate([r[0]["zu_y_" + stage], r[1]["zu_y_" + stage]], axis=1)
        assert y.shape == ref["zu_ref_y_" + stage].shape
        assert np.abs(y - ref["zu_ref_y_" + stage]).max() < 1e-4 * max(1.0, np.abs(ref["zu_ref_y_" + stage]).max())
        np.testing.assert_array_equal(r[0]["zu_g_" + stage], r[1]["zu_g_" + stage])
        off = 0
        for name, n in zip(ref["zu_names"], ref["zu_sizes"]):
            a, b = r[0]["zu_g_" + stage][off:off + n], ref["zu_ref_g_" + stage][off:off + n]
            off += n
            den = max(np.linalg.norm(b), 1e-12)
            assert np.linalg.norm(a - b) / den < 2e-3 or np.abs(a - b).max() < 1e-6, (stage, str(name), np.linalg.norm(a - b) / den)

    # the whole sharded step with the positive RoI's U-Net z-sharded over both ranks
    np.testing.assert_allclose(r[0]["sh6b_losses"], ref["ref6_losses"], rtol=2e-4, atol=1e-6)
    off = 0
    for name, n in zip(ref["grad_names"], ref["grad_sizes"]):
        a, b = r[0]["sh6b_grads"][off:off + n], ref["ref6_grads"][off:off + n]
        off += n
        den = max(np.linalg.norm(b), 1e-12)
        # U-Net tensors of this 32^3 toy configuration carry a 1e-2 fp32 noise floor of their own (InstanceNorm over
        # 2^3..4^3 voxels; measured against fp64 by check_training_step_vs_oracle): a different summation order of the
        # statistics (slab sums combined across ranks) moves them by a few 1e-3; everything else is held to 2e-3
        tol = 2e-2 if str(name).startswith("mask.") else 2e-3
        assert np.linalg.norm(a - b) / den < tol or np.abs(a - b).max() < 1e-6, (str(name), np.linalg.norm(a - b) / den)


# ---------------------------------------------------------------------------------------------------------------------
# world 4 / world 8 (tests/dist_worker_n.py): interior ranks, slabs of 1 - 2 planes, the 4 RoIs x 2 ranks z-shard plan
# ---------------------------------------------------------------------------------------------------------------------
def _grad_table(names, sizes, got, ref, tol, tol_mask):
    off, worst = 0, ("", 0.0)
    for name, n in zip(names, sizes):
        a, b = got[off:off + n], ref[off:off + n]
        off += n
        den = max(np.linalg.norm(b), 1e-12)
        err = np.linalg.norm(a - b) / den
        t = tol_mask if str(name).startswith("mask.") else tol
        assert err < t or np.abs(a - b).max() < 1e-6, (str(name), err)
        if err > worst[1]:
            worst = (str(name), err)
    return worst


def check_worldn(r, sections="halo,conv,rpn,step,rr,dp,unet", tol_mask=2e-2, unet_tol=2e-3):
    """Assertions over the result dicts of tests/dist_worker_n.py (any world size); shared by the CPU tier (emulator) and
    the GPU tier (4 processes on one GPU)."""
    import torch
    world = len(r)
    ref = r[0]
    sections = sections.split(",")
    # prepare_zshard_groups: one plan per proper divisor of the world
    assert list(ref["plan_sizes"]) == [n for n in range(1, world) if world % n == 0]
    g = torch.Generator().manual_seed(0)
    if "halo" in sections:
        # rank k's padded slab = planes [2k-1, 2k+3) of the zero-padded full tensor; the gradient of a plane is the sum over
        # every padded slab that holds it (interior ranks: three contributions on each of their planes' neighbours)
        full = torch.randn(1, 2 * world, 3, 4, 4, generator=g)
        gy_full = torch.randn(1, 4 * world, 3, 4, 4, generator=g)
        padded = torch.nn.functional.pad(full, (0, 0, 0, 0, 0, 0, 1, 1)).numpy()
        gx_ref = np.zeros((1, 2 * world + 2, 3, 4, 4), np.float32)
        for k in range(world):
            np.testing.assert_array_equal(r[k]["halo_y"], padded[:, 2 * k:2 * k + 4])
            gx_ref[:, 2 * k:2 * k + 4] += gy_full[:, 4 * k:4 * k + 4].numpy()
        for k in range(world):
            np.testing.assert_allclose(r[k]["halo_gx"], gx_ref[:, 1 + 2 * k:3 + 2 * k], rtol=0, atol=1e-6)
    if "conv" in sections:
        for tag, tol in (("d", 1e-5), ("w", 3e-5), ("s2", 1e-5)):
            for planes in (1, 2, 4):
                key = "conv_%s%d_" % (tag, planes)
                if key + "y" not in ref:
                    assert tag == "s2" and planes == 1
                    continue
                for what, t in (("y", tol), ("gx", tol)):
                    got = np.concatenate([r[k][key + what] for k in range(world)], axis=1)
                    np.testing.assert_allclose(got, ref["ref_" + key + what], rtol=t, atol=t, err_msg=key + what)
                np.testing.assert_allclose(ref[key + "gw"], ref["ref_" + key + "gw"], rtol=1e-4, atol=3e-5, err_msg=key + "gw")
    if "rpn" in sections:
        for planes3 in (1, 2):
            key = "rpn%d_" % planes3
            assert r[0][key + "p3"].shape[1] == planes3
            for lv in ("p2", "p3"):
                got = np.concatenate([r[k][key + lv] for k in range(world)], axis=1)
                np.testing.assert_allclose(got, ref["ref_" + key + lv], rtol=2e-5, atol=2e-5, err_msg=key + lv)
            for k in range(world):
                np.testing.assert_allclose(r[k][key + "logits"], ref["ref_" + key + "logits"], rtol=2e-5, atol=2e-5)
                np.testing.assert_allclose(r[k][key + "bbox"], ref["ref_" + key + "bbox"], rtol=2e-5, atol=2e-5)
                np.testing.assert_array_equal(r[k][key + "rois"], r[0][key + "rois"])     # identical proposals everywhere
            assert r[0][key + "rois"].shape == ref["ref_" + key + "rois"].shape
            np.testing.assert_allclose(r[0][key + "rois"], ref["ref_" + key + "rois"], rtol=0, atol=1e-5)
    report = {}
    for tag, sec, rs, pre in (("za", "stepa", 2, ""), ("zb", "stepb", world, ""), ("rr", "rr", 0, ""),
                              ("c1_rr", "cfg1", 0, "c1_"), ("c1_z", "cfg1", world // 2, "c1_"),
                              ("l_z", "lits", 2 if world > 2 else world, "l_"), ("l_rr", "lits", 0, "l_"),
                              ("sb_z", "litsplit", 2 if world > 2 else world, "sb_"), ("sb_rr", "litsplit", 0, "sb_"),
                              ("sf_z", "litsplit", 2 if world > 2 else world, "sf_"), ("sf_rr", "litsplit", 0, "sf_")):
        if sec not in sections and not (sec.startswith("step") and "step" in sections):
            continue
        assert int(ref[tag + "_zsharded"][0]) == rs, (tag, ref[tag + "_zsharded"])
        np.testing.assert_allclose(ref[tag + "_losses"], ref["ref_" + tag + "_losses"], rtol=2e-4, atol=1e-6, err_msg=tag)
        assert ref[tag + "_rois"].shape == ref["ref_" + tag + "_rois"].shape
        np.testing.assert_allclose(ref[tag + "_rois"], ref["ref_" + tag + "_rois"], rtol=0, atol=1e-5)
        for k in range(1, world):       # the reducer left the same sums on every rank
            np.testing.assert_array_equal(r[k][tag + "_grads"], ref[tag + "_grads"])
            np.testing.assert_array_equal(r[k][tag + "_losses"], ref[tag + "_losses"])
        assert np.abs(ref["ref_" + tag + "_grads"]).max() > 0
        if sec == "litsplit":           # the phase's other losses are exactly zero, in the sharded step as in the reference's
            dead = slice(4, 6) if pre == "sb_" else slice(0, 4)
            assert not np.any(ref[tag + "_losses"][dead]) and not np.any(ref["ref_" + tag + "_losses"][dead]), tag
            live = [i for i in range(6) if not (dead.start <= i < dead.stop)]
            assert all(ref[tag + "_losses"][i] > 0 for i in live), tag
        # U-Net tensors of the 32^3 toy configuration carry a 1e-2 fp32 noise floor of their own (InstanceNorm over 2^3..4^3
        # voxels; check_training_step_vs_oracle measures it against fp64): slab sums combined across ranks move them by a
        # few 1e-3 when the RoI is z-sharded; everything else 2e-3
        # (real channel counts, "cfg1": the reference's own fp32-vs-fp64 deviation of the U-Net gradients at b = 20 / 96^3 is
        # 2e-4 ... 7e-3, DESIGN section 6: the ranks' other partial-sum orders are held to 1e-2 round-robin, 2e-2 z-sharded)
        tm = (tol_mask if rs else 1e-2) if pre else (tol_mask if rs else 2e-3)
        report[tag] = _grad_table(ref[pre + "grad_names"], ref[pre + "grad_sizes"], ref[tag + "_grads"],
                                  ref["ref_" + tag + "_grads"], 2e-3, tm)
    if "dp" in sections:
        for k in range(1, world):
            np.testing.assert_array_equal(r[k]["dp_grads"], ref["dp_grads"])
        report["dp"] = _grad_table(ref["dp_names"], ref["dp_sizes"], ref["dp_grads"], ref["ref_dp_grads"], 1e-4, 1e-4)
    if "unet" in sections:
        y = np.concatenate([r[k]["zu_y"] for k in range(world)], axis=1)
        assert y.shape == ref["ref_zu_y"].shape
        assert np.abs(y - ref["ref_zu_y"]).max() < 1e-4 * max(1.0, np.abs(ref["ref_zu_y"]).max())
        for k in range(1, world):
            np.testing.assert_array_equal(r[k]["zu_g"], ref["zu_g"])
        report["unet"] = _grad_table(ref["zu_names"], ref["zu_sizes"], ref["zu_g"], ref["ref_zu_g"], unet_tol, unet_tol)
    return report


def test_lits_stage_split_world2(emu_lib, tmp_path):
    """The LiTS fork's two training phases (STAGE_SPLIT, its default) through the sharded step on 2 ranks: the detector phase
    runs no mask head, the mask phase no classifier / RPN losses -- losses and summed gradients equal step.training_step's."""
    env = dict(os.environ, CFUN_LIB_PATH=emu_lib, PYTHONPATH=ROOT)
    print(check_worldn(run_world(tmp_path, env, 2, "dist_worker_n.py", ("cpu", "litsplit"), timeout=1500), "litsplit"))


def test_depth_sharding_world4(emu_lib, tmp_path):
    """4 ranks: two interior ranks, 2 RoIs x 2 ranks and 1 RoI x 4 ranks z-sharded, round-robin with idle ranks."""
    env = dict(os.environ, CFUN_LIB_PATH=emu_lib, PYTHONPATH=ROOT)
    # (round-robin with idle ranks and the U-Net over all ranks: at world 8; the LiTS fork's losses on 4 ranks: GPU tier,
    #  test_four_ranks_on_real_kernels -- here test_lits_stage_split_world2 holds them, with the fork's phase split)
    sections = "halo,conv,rpn,stepa,stepb,dp"
    print(check_worldn(run_world(tmp_path, env, 4, "dist_worker_n.py", ("cpu", sections), timeout=1500), sections))


def test_depth_sharding_world8(emu_lib, tmp_path):
    """8 ranks -- the layout BASELINE configs[3] names: six interior ranks, slabs of 1 - 2 p3 planes, the 4 RoIs x 2 ranks
    z-shard plan, one RoI over all 8 ranks (4 / 2 / 1 planes per rank at the U-Net's sharded levels)."""
    env = dict(os.environ, CFUN_LIB_PATH=emu_lib, PYTHONPATH=ROOT)
    sections = "halo,conv,rpn,stepa,rr,unet"      # (one RoI over all ranks inside a step and the data-parallel step: at world 4)
    print(check_worldn(run_world(tmp_path, env, 8, "dist_worker_n.py", ("cpu", sections), timeout=2400), sections))
