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
