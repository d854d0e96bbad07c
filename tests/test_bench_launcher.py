"""CPU tier: `python bench.py --gpus N` must BE an N-rank job, however it is started.

Until round 6 `--gpus` was parsed and never read -- the world size came from WORLD_SIZE alone, so a plain-python
`bench.py --gpus 8` printed a one-rank number.  Held here, without a GPU:
  * plain python with N > 1 re-executes itself under torch.distributed.run with N ranks (the real launch, end to end:
    `--preflight-only` on CPU tensors through the HIP emulator build, gloo) and the line says n_gpus = N with the
    communication pre-flight (cfun_amd.dist_selftest) green on every section;
  * WORLD_SIZE that disagrees with --gpus is refused, as are more ranks than GPUs over RCCL;
  * without a GPU nothing of the product runs (no CPU fallback)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CFUN_LIB_PATH", "CFUN_BENCH_BACKEND")}
    env.update(PYTHONPATH=ROOT, **env_extra)
    return subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


def test_launch_command_is_the_contract_line():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(["--gpus", "8", "--steps", "3"], 8, port=29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-5:] == [BENCH, "--gpus", "8", "--steps", "3"]
    a = bench.parse_args(["--gpus", "4", "--sharded"])
    assert a.gpus == 4 and a.sharded and not a.preflight_only


@pytest.mark.parametrize("world", [2, 4])
def test_plain_python_gpus_n_becomes_n_ranks(emu_lib, world):
    r = _run(["--gpus", str(world), "--preflight-only"], dict(CFUN_BENCH_BACKEND="gloo", CFUN_LIB_PATH=emu_lib))
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world
    pf = d["preflight"]
    assert pf["ok"] and pf["world"] == world and pf["ranks_seen"] == world and pf["reducer_buckets"] >= 3
    assert pf["rccl_ranks_seen"] == 0          # gloo: nothing here claims to have been RCCL
    assert sorted(pf["max_rel_err"]) == ["conv_gw", "conv_gx", "conv_s2_gw", "conv_s2_gx", "conv_s2_y", "conv_y", "gather",
                                         "halo_bwd", "halo_fwd", "reducer"]
    assert max(pf["max_rel_err"].values()) <= pf["tolerance_rel"]


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "2", "--steps", "1"], dict(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and b"--gpus 2 but WORLD_SIZE = 4" in r.stderr


def test_more_ranks_than_gpus_is_refused_over_rccl():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    r = _run(["--gpus", "2", "--steps", "1"], {})
    assert r.returncode != 0 and b"GPU(s) visible" in r.stderr


def test_no_gpu_no_number():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = _run(["--steps", "1"], {})
    assert r.returncode != 0 and b"needs a GPU" in r.stderr and not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
