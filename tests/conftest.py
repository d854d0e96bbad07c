import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def golden_state_dict(g, gain=1.0):
    """Regenerate the closed-form state dict a golden was produced with (oracle/formula.py)."""
    import torch
    from oracle import formula
    shapes = {str(k): tuple(int(v) for v in str(s).split(",") if v != "")
              for k, s in zip(g["sd_keys"], g["sd_shapes"])}
    arrs = formula.fill_state_dict(shapes, gain)
    return {k: torch.from_numpy(v) for k, v in arrs.items()}


@pytest.fixture(scope="session")
def golden():
    return load_golden


EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libcfun_emu.so")


def _build_emulator():
    """Host build of cfun_amd/csrc/*.hip against tests/emu (a functional HIP emulator).  CPU tier only."""
    import shutil
    import subprocess
    cxx = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(cxx) or shutil.which("make") is None:
        return False
    r = subprocess.run(["make", "-C", EMU_DIR, "-j8"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("emulator build failed:\n" + r.stdout.decode()[-4000:])
    return os.path.exists(EMU_LIB)


@pytest.fixture(scope="session")
def emu_lib():
    if not _build_emulator():
        pytest.skip("no host clang++ to build the HIP emulator")
    return EMU_LIB


@pytest.fixture()
def emu(emu_lib, monkeypatch):
    """Route cfun_amd to the emulator build for one test (CPU tensors, same kernel sources)."""
    monkeypatch.setenv("CFUN_LIB_PATH", emu_lib)
    monkeypatch.setenv("CFUN_CONV_ALGO", os.environ.get("CFUN_EMU_CONV_ALGO", "auto"))
    return "cpu"


@pytest.fixture()
def gpu(monkeypatch):
    """The real library on cuda:0; the GPU tier must never run on the emulator or a CPU fallback."""
    import torch
    monkeypatch.delenv("CFUN_LIB_PATH", raising=False)
    monkeypatch.delenv("CFUN_CONV_ALGO", raising=False)
    assert torch.cuda.is_available(), "GPU tier needs a GPU"
    from cfun_amd import _lib
    _lib.load()
    assert not _lib.is_emulator()
    return "cuda:0"
