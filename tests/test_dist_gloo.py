"""CPU tier: the multi-GPU path (cfun_amd.dist) with world_size 2 over gloo -- halo exchange (forward and
backward), depth-sharded FPN -> RPN -> proposal all-gather vs the single-rank result, and a depth-coupled conv
trained through the exchange.  The kernels run through the HIP emulator build (CPU tensors)."""
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


def run_world2(tmp_path, env, worker_args=(), timeout=900):
    """Spawn tests/dist_worker.py as 2 ranks over gloo and return the two result dicts."""
    port = _free_port()
    out = str(tmp_path / "rank%d.npz")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r), "2", port, out]
                              + list(worker_args), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=timeout)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    return [dict(np.load(out % k)) for k in range(2)]


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
