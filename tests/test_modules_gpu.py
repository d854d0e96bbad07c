"""GPU tier: module-level parity on the real library -- golden vectors of the reference, oracle comparison
of whole training steps, state-dict contract."""
import numpy as np
import pytest
import torch

import module_cases as mc
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["unet_beginning_eval", "unet_beginning_train", "unet_finetune_train",
                                  "unet_lits_eval"])
def test_unet_golden(gpu, name):
    mc.check_unet_golden(gpu, name)


def test_fpn_rpn_golden(gpu):
    mc.check_fpn_rpn_golden(gpu)


def test_proposal_layer_golden(gpu):
    mc.check_proposal_layer_golden(gpu)


def test_pyramid_roi_align_golden(gpu):
    mc.check_pyramid_roi_align_golden(gpu)


def test_classifier_golden(gpu):
    mc.check_classifier_golden(gpu)


def test_nms_dropin(gpu):
    mc.check_nms_dropin(gpu)


@pytest.mark.parametrize("stage", ["beginning", "finetune"])
def test_training_step_tiny_vs_oracle(gpu, stage):
    mc.check_training_step_vs_oracle(gpu, mc.tiny_config(stage))


def test_training_step_cfg0_vs_oracle(gpu):
    """BASELINE.json configs[0] shape (64x64x32, b = 20, 96^3 crops, 'beginning'), 2 positive RoIs:
    the full hot path forward + losses + backward against the oracle on the host."""
    from cfun_amd import config
    torch.set_num_threads(max(1, torch.get_num_threads()))
    mc.check_training_step_vs_oracle(gpu, config.heart_config("beginning", 64, 64, 32), n_pos=2)


def test_unet_b20_96_forward_properties(gpu):
    """Full-size mask head (b = 20, 96^3 -> 192^3, 'finetune'): size-independent properties --
    softmax rows sum to 1, eval is deterministic, batch entries are independent (InstanceNorm is per sample)."""
    from cfun_amd.mask_branch import Modified3DUNet
    torch.manual_seed(0)
    net = Modified3DUNet(1, 8, "finetune", 20).to(gpu).eval()
    x = torch.randn(2, 1, 96, 96, 96, device=gpu)
    with torch.no_grad():
        y = net(x)
        y1 = net(x[1:2])
        y_again = net(x)
    assert tuple(y.shape) == (2, 8, 192, 192, 192)
    assert torch.equal(y, y_again)
    assert float((y[1:2] - y1).abs().max()) < 1e-4
    from cfun_amd import ops
    p = ops.softmax_channels(y.permute(0, 2, 3, 4, 1))
    assert float((p.sum(-1) - 1).abs().max()) < 1e-5


def test_state_dict_contract(gpu):
    """220 entries with the reference's key names and shapes; strict load round-trip."""
    from cfun_amd import config, step
    g = load_golden("predict_cfg0")
    net = step.CFUNHotPath(config.heart_config("beginning", 64, 64, 32))
    sd = net.state_dict()
    ref = {str(k): tuple(int(v) for v in str(s).split(",") if v != "") for k, s in zip(g["sd_keys"], g["sd_shapes"])}
    assert set(sd) == set(ref)
    assert len(sd) == 220
    for k, v in sd.items():
        assert tuple(v.shape) == ref[k], k
