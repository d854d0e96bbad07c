"""GPU tier: the multi-GPU path on the REAL kernels with two ranks.

The test box has one MI355X, so both ranks run on cuda:0 and the collectives go through gloo (RCCL refuses two ranks on
one device); everything else is the product: libcfun_hip.so, HIP streams, the side-stream halo exchange of
``dist.halo_conv`` overlapped with the interior planes, the z-sharded per-RoI U-Net, the ordered ``GradientReducer``.
Same worker and same assertions as tests/test_dist_gloo.py, plus BASELINE configs[0]'s volume with the real channel
counts ('finetune', 4 + 8 RoIs, 96^3 -> 192^3 masks) as one volume over two ranks against the single-process step
(reference dataflow: model.py:1391-1514)."""
import os

import numpy as np
import pytest

from conftest import ROOT
from test_dist_gloo import check_world2, run_world2

pytestmark = pytest.mark.gpu


def _grads_close(names, sizes, got, ref, tol, tol_mask):
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


def test_two_ranks_on_real_kernels(gpu, tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("CFUN_LIB_PATH", "CFUN_CONV_ALGO")}
    env["PYTHONPATH"] = ROOT
    r = run_world2(tmp_path, env, worker_args=("cuda:0", "cfg0"), timeout=1500)
    check_world2(r)
    ref = r[0]
    # (a) cfg0 volume, real channel counts, one volume over 2 ranks: depth-sharded FPN/RPN + round-robin heads, reduced
    # through the GradientReducer (both ranks hold the same sums)
    np.testing.assert_allclose(r[0]["c0_losses"], ref["c0_ref_losses"], rtol=2e-4, atol=1e-6)
    np.testing.assert_array_equal(r[0]["c0_losses"], r[1]["c0_losses"])
    np.testing.assert_array_equal(r[0]["c0_grads"], r[1]["c0_grads"])
    assert r[0]["c0_rois"].shape == ref["c0_ref_rois"].shape
    np.testing.assert_allclose(r[0]["c0_rois"], ref["c0_ref_rois"], rtol=0, atol=1e-5)
    # U-Net tensors: the ranks run 2 RoIs each instead of 4 in one batch (other partial-sum orders in the weight
    # gradients) and at b = 20 / 96^3 the reference's own fp32-vs-fp64 deviation of these gradients is 2e-4 ... 7e-3
    # (DESIGN section 6, table printed by test_training_step_finetune_b20_vs_oracle): held to 1e-2, everything else 2e-3
    w = _grads_close(ref["c0_names"], ref["c0_sizes"], r[0]["c0_grads"], ref["c0_ref_grads"], 2e-3, 1e-2)
    print("cfg0 sharded step, worst gradient rel-L2 vs single process:", w)
    # (b) one positive RoI's U-Net (b = 20, 96^3) z-sharded over both ranks: the slab-wise InstanceNorm statistics are
    # combined in a different order than the single-process sums (deep levels: 6^3 voxels), hence the looser mask bound
    np.testing.assert_allclose(r[0]["c0z_losses"], ref["c0z_ref_losses"], rtol=5e-4, atol=1e-6)
    w = _grads_close(ref["c0_names"], ref["c0_sizes"], r[0]["c0z_grads"], ref["c0z_ref_grads"], 2e-3, 2e-2)
    print("cfg0 z-sharded U-Net step, worst gradient rel-L2 vs single process:", w)


def test_two_ranks_cfg1_volume_sharded_step(gpu, tmp_path):
    """The sharded 'finetune' step (depth-sharded FPN / RPN with overlapped halos, one all-gather for the proposals,
    slab-wise RoIAlign + mask losses, round-robin heads, ordered GradientReducer) at BASELINE configs[1]'s volume
    (128x128x64, real channel counts, 4 + 8 RoIs, 96^3 -> 192^3) over two ranks against the single-process step, and one
    RoI's U-Net z-sharded over both ranks -- until round 4 this step had only run at 64x64x32."""
    env = {k: v for k, v in os.environ.items() if k not in ("CFUN_LIB_PATH", "CFUN_CONV_ALGO")}
    env["PYTHONPATH"] = ROOT
    r = run_world2(tmp_path, env, worker_args=("cuda:0", "cfg1vol_only"), timeout=1500)
    ref = r[0]
    np.testing.assert_allclose(r[0]["c0_losses"], ref["c0_ref_losses"], rtol=2e-4, atol=1e-6)
    np.testing.assert_array_equal(r[0]["c0_losses"], r[1]["c0_losses"])
    np.testing.assert_array_equal(r[0]["c0_grads"], r[1]["c0_grads"])
    assert r[0]["c0_rois"].shape == ref["c0_ref_rois"].shape
    np.testing.assert_allclose(r[0]["c0_rois"], ref["c0_ref_rois"], rtol=0, atol=1e-5)
    w = _grads_close(ref["c0_names"], ref["c0_sizes"], r[0]["c0_grads"], ref["c0_ref_grads"], 2e-3, 1e-2)
    print("128x128x64 sharded step, worst gradient rel-L2 vs single process:", w)
    np.testing.assert_allclose(r[0]["c0z_losses"], ref["c0z_ref_losses"], rtol=5e-4, atol=1e-6)
    w = _grads_close(ref["c0_names"], ref["c0_sizes"], r[0]["c0z_grads"], ref["c0z_ref_grads"], 2e-3, 2e-2)
    print("128x128x64 z-sharded U-Net step, worst gradient rel-L2 vs single process:", w)


def test_four_ranks_on_real_kernels(gpu, tmp_path):
    """FOUR ranks on the real kernels (4 processes on cuda:0, gloo) -- interior ranks with a
    previous and a next neighbour, slabs of 1 - 2 p3 planes, 2 RoIs x 2 ranks and 1 RoI x 4 ranks z-sharded, round-robin
    with idle ranks through the ordered GradientReducer (tiny channel counts), and BASELINE configs[1]'s volume with the
    real channel counts: 4 + 8 RoIs one positive RoI per rank, then 2 positive RoIs z-sharded over 2 ranks each."""
    from test_dist_gloo import check_worldn, run_world
    env = {k: v for k, v in os.environ.items() if k not in ("CFUN_LIB_PATH", "CFUN_CONV_ALGO")}
    env["PYTHONPATH"] = ROOT
    sections = "halo,conv,rpn,step,rr,dp,unet,lits,cfg1"
    r = run_world(tmp_path, env, 4, "dist_worker_n.py", ("cuda:0", sections), timeout=1800)
    print("4 ranks on one GPU, worst gradient rel-L2 vs single process:", check_worldn(r, sections))


def _bench(args, env_extra, timeout):
    """`python bench.py ...` as PLAIN python (no torchrun): the script launches its ranks itself.  Returns the JSON line."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("CFUN_LIB_PATH", "CFUN_CONV_ALGO", "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PYTHONPATH=ROOT, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout)
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout.decode()[-1500:], r.stderr.decode()[-3000:])
    return json.loads(lines[0])


def test_bench_plain_python_gpus2_is_a_two_rank_job(gpu):
    """`python bench.py --gpus 2` (no torchrun) on the real kernels -- two ranks on this box's one GPU,
    collectives over gloo -- prints n_gpus = 2, has run the communication pre-flight, and the same invocation carries BOTH
    curves: the data-parallel `value` and the one-volume `sharded_one_volume` with its parity against the single-process step."""
    d = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "cfg1"], dict(CFUN_BENCH_BACKEND="gloo"), 1500)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["preflight"]["ok"] and d["preflight"]["ranks_seen"] == 2 and d["rccl_ranks_seen"] == 0 and "NOT RCCL" in d["backend"]
    leg = d["sharded_one_volume"]
    assert "error" not in leg and leg["scaling"] == "strong" and leg["value"] > 0
    assert leg["sharded_parity"]["ok"] and max(leg["sharded_parity"]["rel_diff"]) <= 5e-4
    assert d["sharded_parity"] == leg["sharded_parity"]


def test_bench_cfg3_one_volume_over_eight_ranks(gpu):
    """BASELINE configs[3] in its specified form inside the driver-run tier: ONE 512x512x256 volume
    depth-sharded over 8 ranks (32 input planes = 2 p3 planes per rank, 4 RoIs x 2 ranks z-sharded U-Nets), 2 timed steps;
    the summed loss shares reproduce the single-process step.  (8 processes on this box's one GPU, gloo.)"""
    d = _bench(["--gpus", "8", "--sharded", "--workload", "cfg3", "--steps", "2", "--warmup", "1"],
               dict(CFUN_BENCH_BACKEND="gloo"), 2400)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["steps"] == 2
    assert d["preflight"]["ok"] and d["preflight"]["ranks_seen"] == 8
    assert d["sharded_parity"]["ok"], d["sharded_parity"]
    assert all(np.isfinite(v) and v > 0 for v in d["losses"])


def test_bench_cfg4_one_volume_over_eight_ranks(gpu):
    """BASELINE configs[4] ("LiTS_2017 config, same pipeline, 8x MI355X") the same way: the fork's 320x320x256 volume, P3D35 with
    the 5x7x7 stem, 3 classes, 32x80x80 crops, class-weighted CE, as ONE volume over 8 ranks (gloo on this box's one GPU); the
    summed loss shares reproduce the single-process step."""
    d = _bench(["--gpus", "8", "--sharded", "--workload", "cfg4", "--steps", "2", "--warmup", "1"],
               dict(CFUN_BENCH_BACKEND="gloo"), 2400)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["preflight"]["ok"]
    assert d["sharded_parity"]["ok"], d["sharded_parity"]
    assert all(np.isfinite(v) for v in d["losses"]) and d["losses"][0] > 0 and d["losses"][4] > 0


def test_rccl_two_gpus(tmp_path):
    """The `nccl` (= RCCL) branch of ``dist._exchange`` -- device buffers straight into batch_isend_irecv on the side
    stream -- and the reducer's RCCL all-reduces, with a real peer: 2 processes on 2 GPUs.  Skips cleanly on the 1-GPU test
    box; the first multi-GPU lease runs it before bench.py does."""
    import torch
    from test_dist_gloo import check_worldn, run_world
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (RCCL refuses two ranks on one device)")
    env = {k: v for k, v in os.environ.items() if k not in ("CFUN_LIB_PATH", "CFUN_CONV_ALGO")}
    env.update(PYTHONPATH=ROOT, CFUN_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    sections = "halo,conv,rpn,step,rr,dp,unet"
    r = run_world(tmp_path, env, 2, "dist_worker_n.py", ("cuda", sections), timeout=1800)
    print("2 ranks over RCCL, worst gradient rel-L2 vs single process:", check_worldn(r, sections))
