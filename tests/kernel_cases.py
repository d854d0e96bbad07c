"""Kernel-level parity cases shared by the CPU tier (HIP emulator, tests/test_kernels_emu.py) and the GPU
tier (real libcfun_hip.so on cuda:0, tests/test_kernels_gpu.py).  Every case compares one HIP op
(forward and gradients, called through the C ABI via cfun_amd.ops) with the plain-torch fp32 restatement
used by the oracle, on seeded inputs.  Tolerances are relative to the reference tensor's max-abs."""
import numpy as np
import torch
import torch.nn.functional as F

from cfun_amd import ops
from cfun_amd._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ALGO_AUTO, ALGO_DIRECT, ALGO_MFMA, ALGO_WINO, ALGO_WINO2
from oracle import cfun_oracle as orc

RTOL = 2e-5


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def assert_close(a, b, what, tol=RTOL):
    assert tuple(a.shape) == tuple(b.shape), "%s: shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    e = rel_err(a, b)
    assert e < tol, "%s: rel err %.3e >= %.1e" % (what, e, tol)


def _gen(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def randn(gen, *shape):
    return torch.randn(*shape, generator=gen)


# ------------------------------------------------------------------------------------------ conv
def ref_conv(x, w, spec, scale, shift, res):
    xc = x.permute(0, 4, 1, 2, 3)
    if spec.up2:
        xc = F.interpolate(xc, scale_factor=2, mode="nearest")
    y = F.conv3d(xc, w, None, stride=spec.stride, padding=spec.pad)
    if scale is not None:
        y = y * (scale.view(scale.shape[0], -1, 1, 1, 1) if spec.scale_per_n else scale.view(1, -1, 1, 1, 1))
    if shift is not None:
        y = y + shift.view(1, -1, 1, 1, 1)
    if spec.d2s:   # depth-to-space x2: channel (pz,py,px,o) -> voxel (2z+pz, 2y+py, 2x+px), channel o
        n, c8, d, h, w = y.shape
        y = y.view(n, 2, 2, 2, c8 // 8, d, h, w).permute(0, 4, 5, 1, 6, 2, 7, 3).reshape(n, c8 // 8, 2 * d, 2 * h, 2 * w)
    if res is not None:
        r = res.permute(0, 4, 1, 2, 3)
        if spec.res_up2 or spec.d2s:
            r = F.interpolate(r, scale_factor=2, mode="nearest")
        y = y + r
    if spec.act == ACT_RELU:
        y = F.relu(y)
    elif spec.act == ACT_LRELU:
        y = F.leaky_relu(y, 0.01)
    return y.permute(0, 2, 3, 4, 1)


CONV_CASES = {
    # name: (n, dhw, ci, co, k, kwargs)
    "direct_stem_c1": (1, (5, 6, 7), 1, 20, (3, 3, 3), dict(algo=ALGO_DIRECT)),
    "direct_p3d_stem": (1, (6, 10, 12), 1, 16, (3, 7, 7), dict(stride=2, pad=(1, 3, 3), act=ACT_RELU, scale=True, shift=True, algo=ALGO_DIRECT)),
    "direct_lits_stem": (1, (6, 8, 8), 1, 24, (5, 7, 7), dict(stride=2, pad=(2, 3, 3), act=ACT_RELU, scale=True, shift=True, algo=ALGO_DIRECT)),
    "direct_s2": (2, (6, 8, 8), 4, 8, (3, 3, 3), dict(stride=2, algo=ALGO_DIRECT)),
    "direct_up2_res": (1, (3, 4, 4), 4, 8, (3, 3, 3), dict(up2=True, res=True, res_up2=True, act=ACT_LRELU, algo=ALGO_DIRECT)),
    "direct_1x1_co3": (1, (4, 4, 5), 8, 3, (1, 1, 1), dict(algo=ALGO_DIRECT)),
    "direct_1x1_32_8": (2, (4, 4, 4), 32, 8, (1, 1, 1), dict(algo=ALGO_DIRECT)),
    "direct_dropout_scale": (2, (4, 4, 4), 4, 8, (3, 3, 3), dict(scale=True, per_n=True, algo=ALGO_DIRECT)),
    "mfma_333_8_20": (1, (5, 6, 18), 8, 20, (3, 3, 3), dict(algo=ALGO_MFMA)),
    "mfma_333_epilogue": (2, (4, 5, 7), 8, 40, (3, 3, 3), dict(act=ACT_LRELU, scale=True, per_n=True, res=True, algo=ALGO_MFMA)),
    "mfma_333_s2": (1, (8, 8, 10), 4, 40, (3, 3, 3), dict(stride=2, algo=ALGO_MFMA)),
    "mfma_111_res_up2": (1, (4, 6, 18), 12, 8, (1, 1, 1), dict(shift=True, res=True, res_up2=True, algo=ALGO_MFMA)),
    "mfma_111_32_8": (2, (4, 4, 4), 32, 8, (1, 1, 1), dict(algo=ALGO_MFMA)),
    "mfma_111_s2_bn_relu": (1, (8, 8, 8), 16, 64, (1, 1, 1), dict(stride=2, pad=(0, 0, 0), scale=True, shift=True, act=ACT_RELU, algo=ALGO_MFMA)),
    "mfma_133_bn_relu": (1, (5, 6, 7), 16, 16, (1, 3, 3), dict(scale=True, shift=True, act=ACT_RELU, algo=ALGO_MFMA)),
    "mfma_311_bn_relu": (1, (5, 6, 7), 16, 16, (3, 1, 1), dict(scale=True, shift=True, act=ACT_RELU, algo=ALGO_MFMA)),
    "mfma_333_up2": (1, (3, 3, 5), 8, 20, (3, 3, 3), dict(up2=True, algo=ALGO_MFMA)),
    "mfma_555_up2_res": (1, (3, 3, 4), 8, 8, (5, 5, 5), dict(up2=True, res=True, res_up2=True, algo=ALGO_MFMA)),
    "mfma_333_two_cotiles": (1, (3, 4, 5), 4, 96, (3, 3, 3), dict(algo=ALGO_MFMA)),
    "mfma_333_ci20_co20": (1, (4, 4, 16), 20, 20, (3, 3, 3), dict(algo=ALGO_MFMA)),
    "auto_stem_c1_wgrad_mfma": (2, (5, 6, 18), 1, 20, (3, 3, 3), dict(algo=ALGO_AUTO)),
    "auto_p3d_stem_wgrad_mfma": (1, (6, 10, 36), 1, 16, (3, 7, 7), dict(stride=2, pad=(1, 3, 3), act=ACT_RELU, scale=True, shift=True, algo=ALGO_AUTO)),
    "auto_lits_stem_wgrad_mfma": (1, (6, 8, 8), 1, 24, (5, 7, 7), dict(stride=2, pad=(2, 3, 3), algo=ALGO_AUTO)),
    "mfma_333_splitk_epilogue": (2, (4, 4, 8), 32, 16, (3, 3, 3), dict(algo=ALGO_MFMA, act=ACT_LRELU, scale=True, per_n=True, shift=True, res=True)),
    "mfma_111_splitk_res_up2": (1, (4, 4, 8), 64, 8, (1, 1, 1), dict(algo=ALGO_MFMA, res=True, res_up2=True)),
    "mfma_333_rem_20_20": (1, (5, 6, 18), 20, 20, (3, 3, 3), dict(algo=ALGO_MFMA, act=ACT_LRELU, res=True, scale=True, per_n=True)),
    "mfma_333_rem_40_40": (2, (4, 5, 7), 40, 40, (3, 3, 3), dict(algo=ALGO_MFMA)),
    "mfma_333_rem_16_8": (1, (4, 5, 17), 16, 8, (3, 3, 3), dict(algo=ALGO_MFMA, shift=True)),
    "mfma_111_rem_40_8_res_up2": (1, (4, 6, 18), 40, 8, (1, 1, 1), dict(algo=ALGO_MFMA, res=True, res_up2=True)),
    "mfma_333_s2_rem_20_40": (1, (8, 8, 10), 20, 40, (3, 3, 3), dict(algo=ALGO_MFMA, stride=2)),
    "mfma_333_d2s_res": (2, (3, 4, 5), 8, 64, (3, 3, 3), dict(algo=ALGO_MFMA, d2s=True, res=True)),
    "auto_111_8_8_pointwise": (1, (16, 32, 64), 8, 8, (1, 1, 1), dict(algo=ALGO_AUTO, res=True, res_up2=True, shift=True, act=ACT_RELU, scale=True)),
    # k_conv_pointwise_t (rows staged through LDS): 40 = one segment, 64 = two segments of 32; 33 in x: a ragged last group
    "auto_111_40_8_pointwise_t": (1, (32, 32, 33), 40, 8, (1, 1, 1), dict(algo=ALGO_AUTO, shift=True, act=ACT_LRELU)),
    "auto_111_64_8_pointwise_t": (2, (16, 32, 33), 64, 8, (1, 1, 1), dict(algo=ALGO_AUTO, res=True, scale=True, per_n=True)),
    # the input prologue's (mean, rstd) table next to the staged rows: N * C_in = 3 200 > 2 560 does not fit 64 KB of LDS with
    # them -> the per-row kernel must take the prologue launches (the staged launch failed with invalid value)
    "auto_111_40_8_pointwise_prologue_big_n": (80, (8, 8, 8), 40, 8, (1, 1, 1), dict(algo=ALGO_AUTO, shift=True)),
    "direct_333_d2s_res_cq3": (1, (3, 4, 5), 3, 24, (3, 3, 3), dict(algo=ALGO_DIRECT, d2s=True, res=True, act=ACT_LRELU)),
    # Winograd F(2,3) along x (conv3d_wino.hip): forward and data gradient; ragged tiles, odd widths, epilogue, split-K
    "wino_333_8_48_ragged": (1, (5, 6, 18), 8, 48, (3, 3, 3), dict(algo=ALGO_WINO)),
    "wino_333_epilogue_odd_w": (2, (4, 5, 7), 8, 40, (3, 3, 3), dict(algo=ALGO_WINO, act=ACT_LRELU, scale=True, per_n=True, res=True, shift=True)),
    "wino_333_two_cotiles": (1, (3, 4, 5), 4, 96, (3, 3, 3), dict(algo=ALGO_WINO)),
    "wino_333_splitk_epilogue": (2, (4, 4, 8), 32, 16, (3, 3, 3), dict(algo=ALGO_WINO, act=ACT_LRELU, scale=True, per_n=True, shift=True, res=True)),
    "wino_333_40_40": (1, (4, 5, 19), 40, 40, (3, 3, 3), dict(algo=ALGO_WINO)),
    "wino2_333_8_48_ragged": (1, (5, 6, 18), 8, 48, (3, 3, 3), dict(algo=ALGO_WINO2)),
    "wino2_333_epilogue_odd_w": (2, (4, 5, 7), 8, 40, (3, 3, 3), dict(algo=ALGO_WINO2, act=ACT_LRELU, scale=True, per_n=True, res=True, shift=True)),
    "wino2_333_two_cotiles_splitk": (2, (4, 4, 8), 32, 96, (3, 3, 3), dict(algo=ALGO_WINO2, shift=True)),
    "wino2_333_d2s_slab": (1, (6, 4, 5), 8, 64, (3, 3, 3), dict(algo=ALGO_WINO2, d2s=True, res=True, pad=(0, 1, 1))),
    "wino_333_80_80_nsub1_tiles": (1, (5, 4, 10), 80, 80, (3, 3, 3), dict(algo=ALGO_WINO, act=ACT_LRELU)),      # 5 co tiles of 16, 5 ci subtiles
    "wino_333_12_20_small_ci": (1, (4, 6, 9), 12, 20, (3, 3, 3), dict(algo=ALGO_WINO)),                        # C_in = 12: one half-empty ci subtile
    "wino2_333_12_24_odd_chunks": (2, (5, 4, 9), 12, 24, (3, 3, 3), dict(algo=ALGO_WINO2, act=ACT_LRELU, shift=True)),   # 3 channel chunks: a half-filled register pair
    "wino2_333_40_40": (1, (4, 5, 19), 40, 40, (3, 3, 3), dict(algo=ALGO_WINO2, res=True)),
    "wino_333_slab_pd0": (2, (7, 5, 9), 16, 32, (3, 3, 3), dict(algo=ALGO_WINO, pad=(0, 1, 1), shift=True)),   # a depth slab with its halo
    "wino_333_d2s_res": (2, (3, 4, 5), 8, 64, (3, 3, 3), dict(algo=ALGO_WINO, d2s=True, res=True, act=ACT_LRELU)),
}
# bigger shapes: many workgroups, several chunks per wgrad block, channel counts of the real nets (GPU tier)
CONV_CASES_LARGE = {
    "mfma_333_40_40_32cube": (2, (32, 32, 32), 40, 40, (3, 3, 3), dict(algo=ALGO_MFMA, act=ACT_LRELU, res=True)),
    "wino_333_40_40_32cube": (2, (32, 32, 32), 40, 40, (3, 3, 3), dict(algo=ALGO_WINO, act=ACT_LRELU, res=True)),
    "wino_333_80_80_24cube": (2, (24, 24, 26), 80, 80, (3, 3, 3), dict(algo=ALGO_WINO, shift=True)),
    "wino2_333_40_40_32cube": (2, (32, 32, 32), 40, 40, (3, 3, 3), dict(algo=ALGO_WINO2, act=ACT_LRELU, res=True)),
    "wino2_333_160_160_24cube": (1, (24, 24, 26), 160, 160, (3, 3, 3), dict(algo=ALGO_WINO2, shift=True)),
    "wino_333_128_256_fpn": (1, (8, 16, 16), 128, 256, (3, 3, 3), dict(algo=ALGO_WINO, shift=True, act=ACT_RELU)),
    "mfma_333_128_256_fpn": (1, (8, 16, 16), 128, 256, (3, 3, 3), dict(algo=ALGO_MFMA, shift=True, act=ACT_RELU)),
    "mfma_333_320_160_up2": (2, (6, 6, 6), 320, 160, (3, 3, 3), dict(algo=ALGO_MFMA, up2=True)),
    "mfma_333_s2_80_160": (2, (24, 24, 24), 80, 160, (3, 3, 3), dict(algo=ALGO_MFMA, stride=2)),
    "mfma_111_160_80": (2, (24, 24, 24), 160, 80, (1, 1, 1), dict(algo=ALGO_MFMA)),
    "mfma_555_8_8_up2": (1, (24, 24, 24), 8, 8, (5, 5, 5), dict(algo=ALGO_MFMA, up2=True, res=True, res_up2=True)),
    "direct_stem_96": (2, (48, 48, 48), 1, 20, (3, 3, 3), dict(algo=ALGO_DIRECT)),
    "auto_111_40_8_pointwise_res_up2": (2, (24, 32, 34), 40, 8, (1, 1, 1), dict(algo=ALGO_AUTO, res=True, res_up2=True, shift=True)),
    "auto_111_256_8_pointwise": (1, (16, 32, 64), 256, 8, (1, 1, 1), dict(algo=ALGO_AUTO, shift=True, act=ACT_LRELU)),
    # 80 = 2 row segments of 40 channels through k_conv_pointwise_t (the ds3 conv); 33 x 31 x 33 voxels: a ragged last group
    "auto_111_80_8_pointwise_ragged": (1, (33, 31, 33), 80, 8, (1, 1, 1), dict(algo=ALGO_AUTO, shift=True, scale=True)),
    "auto_stem_c1_48cube": (2, (48, 48, 50), 1, 20, (3, 3, 3), dict(algo=ALGO_AUTO, scale=True, per_n=True)),
    "auto_p3d_stem_64": (1, (32, 64, 66), 1, 16, (3, 7, 7), dict(stride=2, pad=(1, 3, 3), act=ACT_RELU, scale=True, shift=True, algo=ALGO_AUTO)),
    "auto_333_20_20_48cube": (2, (48, 48, 48), 20, 20, (3, 3, 3), dict(algo=ALGO_AUTO, scale=True, per_n=True)),
}


def check_conv(device, n, dhw, ci, co, k, stride=1, pad=None, up2=False, act=ACT_NONE, scale=False, shift=False,
               res=False, res_up2=False, per_n=False, algo=ALGO_AUTO, seed=0, tol=RTOL, d2s=False):
    gen = _gen(seed)
    pad = pad if pad is not None else tuple(kk // 2 for kk in k)
    spec = ops.ConvSpec(k=tuple(k), co=co, stride=stride, pad=tuple(pad), up2=up2, act=act, res_up2=res_up2,
                        scale_per_n=per_n, algo=algo, d2s=d2s)
    x = randn(gen, n, *dhw, ci)
    w = randn(gen, co, ci, *k) / float(ci * k[0] * k[1] * k[2]) ** 0.5
    sc = ((torch.rand(n, co, generator=gen) + 0.5) if per_n else (torch.rand(co, generator=gen) + 0.5)) if scale else None
    sf = randn(gen, co) if shift else None
    out_shape = ref_conv(x, w, spec, sc, sf, None).shape
    rs = None
    if res:
        shp = list(out_shape)
        if res_up2 or d2s:
            shp = [shp[0], shp[1] // 2, shp[2] // 2, shp[3] // 2, shp[4]]
        rs = randn(gen, *shp)
    gy = randn(gen, *out_shape)

    def leafs(dev):
        return [None if t is None else t.detach().clone().to(dev).requires_grad_(True) for t in (x, w, sf, rs)]

    xr, wr, sfr, rsr = leafs("cpu")
    yr = ref_conv(xr, wr, spec, sc, sfr, rsr)
    yr.backward(gy)
    xd, wd, sfd, rsd = leafs(device)
    y = ops.conv3d(xd, ops.pack_weight(wd), spec, None if sc is None else sc.to(device), sfd, rsd)
    y.backward(gy.to(device))
    assert_close(y, yr, "y", tol)
    assert_close(xd.grad, xr.grad, "dx", tol)
    assert_close(wd.grad, wr.grad, "dw", tol)
    if shift:
        assert_close(sfd.grad, sfr.grad, "dshift", tol)
    if res:
        assert_close(rsd.grad, rsr.grad, "dres", tol)
    # the product path: OIDHW weight in, OIDHW gradient out (chunk reduction + un-packing fused) -- same bits
    xw, ww, sfw, rsw = leafs(device)
    yw = ops.conv3d_w(xw, ww, spec, None if sc is None else sc.to(device), sfw, rsw)
    yw.backward(gy.to(device))
    assert torch.equal(yw, y) and torch.equal(xw.grad, xd.grad), "conv3d_w: y / dx differ from the packed path"
    assert torch.equal(ww.grad, wd.grad), "conv3d_w: fused OIDHW weight gradient differs from reduce + unpack"
    # InstanceNorm statistics of y from the conv's epilogue (CfunConvFusion.out_stats): same y, and (mean, rstd) per
    # (sample, channel) equal to the two-pass fp64 statistics of the reference output
    slot = ops.StatsSlot(n)
    with torch.no_grad():
        ys = ops.conv3d_w(xw.detach(), ww.detach(), spec, None if sc is None else sc.to(device),
                          None if sfw is None else sfw.detach(), None if rsw is None else rsw.detach(), stats=slot)
    assert torch.equal(ys, y), "conv3d_w(stats=...): y differs from the plain call"
    st = slot.get(n, y.shape[-1])
    if algo in (ALGO_MFMA, ALGO_WINO, ALGO_WINO2) and ci % 4 == 0 and co % 4 == 0:
        assert st is not None, "the MFMA / Winograd kernels must deliver epilogue statistics"
    if st is not None:
        y64 = yr.detach().double().reshape(n, -1, yr.shape[-1])
        mean = y64.mean(dim=1)
        rstd = 1.0 / torch.sqrt(y64.var(dim=1, unbiased=False) + 1e-5)
        assert_close(st[..., 0], mean.float(), "epilogue mean", 2e-5)
        assert_close(st[..., 1], rstd.float(), "epilogue rstd", 2e-5)
    # the input prologue (CfunConvFusion.in_stats / in_act): the conv reads lrelu((x - mean) * rstd) while staging x,
    # forward and weight gradient; the returned input gradient is the one w.r.t. the NORMALISED input
    if not up2 and ops._fusable_input(ops.NormedInput(xw.detach(), None, ACT_LRELU, 0.01), spec, sc, sfw, rsw, True):
        ist = torch.stack([randn(gen, n, ci) * 0.5, torch.rand(n, ci, generator=gen) + 0.5], dim=-1)
        for stats_t in (ist, None):          # InstanceNorm + LeakyReLU, then the plain LeakyReLU
            xn = x if stats_t is None else (x - stats_t[..., 0].view(n, 1, 1, 1, ci)) * stats_t[..., 1].view(n, 1, 1, 1, ci)
            xn = F.leaky_relu(xn, 0.01).detach().requires_grad_(True)
            _, wr2, sfr2, rsr2 = leafs("cpu")
            yn = ref_conv(xn, wr2, spec, sc, sfr2, rsr2)
            yn.backward(gy)
            xp, wp2, sfp, rsp = leafs(device)
            tok = xp.detach().requires_grad_(True)
            yp = ops.conv3d_w(ops.NormedInput(tok, None if stats_t is None else stats_t.to(device), ACT_LRELU, 0.01), wp2, spec,
                              None if sc is None else sc.to(device), sfp, rsp)
            yp.backward(gy.to(device))
            tag = "prologue(norm)" if stats_t is not None else "prologue(lrelu)"
            assert_close(yp, yn, tag + " y", tol)
            # gradients: against the SAME kernels fed the materialised input (bit for bit -- the prologue repeats the
            # stand-alone passes' arithmetic); the CPU reference would add the output activation's kink flips to the picture
            xm, wm, sfm, rsm = leafs(device)
            tokm = xn.detach().to(device).requires_grad_(True)
            ym = ops.conv3d_w(tokm, wm, spec, None if sc is None else sc.to(device), sfm, rsm)
            ym.backward(gy.to(device))
            assert torch.equal(yp, ym), tag + ": y differs from the materialised-input run"
            assert torch.equal(tok.grad, tokm.grad), tag + ": dx differs from the materialised-input run"
            assert torch.equal(wp2.grad, wm.grad), tag + ": dw differs from the materialised-input run"


# ------------------------------------------------------------------------------------------ norm / act / pool
NORM_CASES = {"c3_v60": (2, (3, 4, 5), 3), "c2_v512": (1, (8, 8, 8), 2), "c20_v343": (2, (7, 7, 7), 20), "c32_v512": (2, (8, 8, 8), 32), "c16_v4096": (2, (16, 16, 16), 16),
              "c320_v8": (1, (2, 2, 2), 320), "c4_v30": (3, (2, 3, 5), 4), "c160_v216": (4, (6, 6, 6), 160)}


def check_instnorm_lrelu(device, n, dhw, c, seed=1):
    gen = _gen(seed)
    x = randn(gen, n, *dhw, c) * 2.0 + randn(gen, 1, 1, 1, 1, c)
    gy = randn(gen, n, *dhw, c)
    xr = x.clone().requires_grad_(True)
    yr = F.leaky_relu(F.instance_norm(xr.permute(0, 4, 1, 2, 3), eps=1e-5), 0.01).permute(0, 2, 3, 4, 1)
    yr.backward(gy)
    xd = x.clone().to(device).requires_grad_(True)
    y = ops.instnorm_lrelu(xd)
    y.backward(gy.to(device))
    assert_close(y, yr, "y")
    assert_close(xd.grad, xr.grad, "dx", 1e-4)


def check_fold_up2(device, seed=7):
    """nearest x2 upsample -> 5x5x5 conv (+ up-sampled skip) == folded 3x3x3 conv with depth-to-space epilogue."""
    gen = _gen(seed)
    x = randn(gen, 2, 4, 5, 6, 8)
    w = randn(gen, 8, 8, 5, 5, 5) / 30.0
    gy = randn(gen, 2, 8, 10, 12, 8)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    up = F.interpolate(xr.permute(0, 4, 1, 2, 3), scale_factor=2, mode="nearest")
    yr = (up + F.conv3d(up, wr, padding=2)).permute(0, 2, 3, 4, 1)
    yr.backward(gy)
    xd, wd = x.clone().to(device).requires_grad_(True), w.clone().to(device).requires_grad_(True)
    spec = ops.ConvSpec(k=(3, 3, 3), co=64, pad=(1, 1, 1), d2s=True, res_up2=True)
    y = ops.conv3d(xd, ops.pack_weight(ops.fold_up2_weight(wd)), spec, res=xd)
    y.backward(gy.to(device))
    assert_close(y, yr, "y")
    assert_close(xd.grad, xr.grad, "dx")
    assert_close(wd.grad, wr.grad, "dw")


def check_fold_up2_conv3(device, ci, co, dhw, algo=ALGO_AUTO, n=2, seed=8):
    """nearest x2 upsample -> 3x3x3 conv == parity-folded conv (2x2x2 live taps per parity, the rest skipped)
    with depth-to-space epilogue; forward, dx and dw (through the differentiable fold)."""
    gen = _gen(seed)
    x = randn(gen, n, *dhw, ci)
    w = randn(gen, co, ci, 3, 3, 3) / float(27 * ci) ** 0.5
    gy = randn(gen, n, 2 * dhw[0], 2 * dhw[1], 2 * dhw[2], co)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv3d(F.interpolate(xr.permute(0, 4, 1, 2, 3), scale_factor=2, mode="nearest"), wr, padding=1).permute(0, 2, 3, 4, 1)
    yr.backward(gy)
    xd, wd = x.clone().to(device).requires_grad_(True), w.clone().to(device).requires_grad_(True)
    cqp = (co + 15) // 16 * 16
    spec = ops.ConvSpec(k=(3, 3, 3), co=8 * cqp, pad=(1, 1, 1), d2s=True, d2s_cq=co, tap_skip=True, algo=algo)
    y = ops.conv3d(xd, ops.pack_weight(ops.fold_up2_weight(wd, cqp)), spec)
    y.backward(gy.to(device))
    assert_close(y, yr, "y")
    assert_close(xd.grad, xr.grad, "dx")
    assert_close(wd.grad, wr.grad, "dw")


def check_elementwise(device, seed=2):
    gen = _gen(seed)
    for shape in ((2, 3, 4, 5, 8), (1, 3, 3, 3, 3), (1, 2, 2, 2, 1)):
        x, gy = randn(gen, *shape), randn(gen, *shape)
        xr = x.clone().requires_grad_(True)
        F.leaky_relu(xr, 0.01).backward(gy)
        xd = x.clone().to(device).requires_grad_(True)
        y = ops.lrelu(xd)
        y.backward(gy.to(device))
        assert_close(y, F.leaky_relu(x, 0.01), "lrelu")
        assert_close(xd.grad, xr.grad, "lrelu dx")
        b = randn(gen, *shape)
        assert_close(ops.add(x.to(device), b.to(device)), x + b, "add")
    g2 = randn(gen, 1000, 16)
    assert_close(ops.channel_sum(g2.to(device)), g2.sum(0), "channel_sum")


def check_norm_passthrough(device, seed=31):
    """The norm / activation nodes' second output (an alias of x for x's other consumer, gradients summed inside the
    backward kernel) and their lazy form (a NormedInput materialized once) against plain torch."""
    gen = _gen(seed)
    for shape in ((2, 3, 4, 5, 8), (1, 2, 3, 3, 20), (1, 2, 2, 2, 3)):
        x = randn(gen, *shape) * 1.5 + randn(gen, 1, 1, 1, 1, shape[-1])
        g1, g2 = randn(gen, *shape), randn(gen, *shape)
        for norm in (True, False):
            def ref_fn(t):
                if norm:
                    return F.leaky_relu(F.instance_norm(t.permute(0, 4, 1, 2, 3), eps=1e-5), 0.01).permute(0, 2, 3, 4, 1)
                return F.leaky_relu(t, 0.01)
            xr = x.clone().requires_grad_(True)
            ((ref_fn(xr) * g1).sum() + (xr * g2).sum()).backward()
            for lazy in (False, True):
                xd = x.clone().to(device).requires_grad_(True)
                xs = ops.add(xd, torch.zeros_like(xd))          # a non-leaf producer, like a conv's output
                fn = ops.instnorm_lrelu if norm else ops.lrelu
                y, x2 = fn(xs, lazy=lazy, passthrough=True)
                if lazy:
                    assert isinstance(y, ops.NormedInput)
                    assert y.materialize() is y.materialize()     # one apply pass however many consumers ask
                    y = y.materialize()
                ((y * g1.to(device)).sum() + (x2 * g2.to(device)).sum()).backward()
                what = "%s lazy=%s %s" % ("norm" if norm else "lrelu", lazy, shape)
                assert_close(y, ref_fn(x), what + " y")
                assert_close(xd.grad, xr.grad, what + " dx", 1e-4)
            # only the passthrough branch used: the gradient passes through untouched
            xd = x.clone().to(device).requires_grad_(True)
            _, x2 = (ops.instnorm_lrelu if norm else ops.lrelu)(ops.add(xd, torch.zeros_like(xd)), passthrough=True)
            (x2 * g2.to(device)).sum().backward()
            assert torch.equal(xd.grad.cpu(), g2)


def check_roi_align_slabs(device, seed=33):
    """RoIAlign on depth slabs: the slabs' additive shares sum to the crop of the whole map, their gradients are the
    whole map's gradient cut into the same slabs (dist.sharded_training_step relies on both)."""
    gen = _gen(seed)
    d, h, w, c = 9, 7, 6, 8
    fm = randn(gen, d, h, w, c)
    boxes = torch.tensor([[0.0, 0.0, 0.0, 1.0, 1.0, 1.0], [0.1, 0.2, 0.3, 0.6, 0.9, 0.8], [0.4, 0.0, 0.5, 0.45, 0.3, 0.55],
                          [-0.2, -0.1, 0.0, 0.5, 1.2, 0.7], [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]])
    pool = (4, 3, 5)
    gy = randn(gen, boxes.shape[0], *pool, c)
    fd = fm.clone().to(device).requires_grad_(True)
    full, b_full = ops.roi_align(fd, boxes.to(device), pool)
    (full * gy.to(device)).sum().backward()
    for cuts in ((0, 3, 4, 9), (0, 1, 9), (0, 9)):
        total, grads = None, []
        for z0, z1 in zip(cuts[:-1], cuts[1:]):
            sd = fm[z0:z1].clone().to(device).requires_grad_(True)
            part, b = ops.roi_align(sd, boxes.to(device), pool, slab=(z0, d))
            assert torch.equal(b.cpu(), b_full.cpu())           # the crop bounds are those of the whole map
            (part * gy.to(device)).sum().backward()
            grads.append(sd.grad)
            total = part if total is None else total + part
        assert float((total - full).detach().abs().max()) < 2e-6 * max(1.0, float(full.detach().abs().max())), cuts
        assert float((torch.cat(grads, 0) - fd.grad).abs().max()) < 1e-5 * max(1.0, float(fd.grad.abs().max())), cuts


def check_fc(device, seed=35):
    """The weight-streaming FC kernels (classifier head) against torch, including the weight gradient in row chunks
    when the rows' g + x slices exceed the kernel's LDS (wide layers)."""
    from cfun_amd import _lib
    gen = _gen(seed)
    lib = _lib.load()
    for r, k, o, act in ((5, 64, 12, ACT_RELU), (33, 200, 24, ACT_NONE), (64, 128, 1024, ACT_NONE), (1, 8, 3, ACT_RELU)):
        x, w = randn(gen, r, k), randn(gen, o, k) * (k ** -0.5)
        shift, gy = randn(gen, o), randn(gen, r, o)
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), shift.clone().requires_grad_(True)
        yr = F.linear(xr, wr, br)
        yr = F.relu(yr) if act == ACT_RELU else yr
        yr.backward(gy)
        xd, wd, bd = [t.clone().to(device).requires_grad_(True) for t in (x, w, shift)]
        y = ops.fc(xd, wd, None, bd, act)
        y.backward(gy.to(device))
        what = "fc r%d k%d o%d" % (r, k, o)
        assert_close(y, yr, what + " y")
        assert_close(xd.grad, xr.grad, what + " dx")
        assert_close(wd.grad, wr.grad, what + " dw")
        assert_close(bd.grad, br.grad, what + " dshift")
    assert 1 <= int(lib.cfun_fc_bwd_weight_max_rows(1024)) <= 64
    assert int(lib.cfun_fc_bwd_weight_max_rows(16)) == 64
    # more rows than the kernel's LDS holds is an argument error at the C ABI, not a launch failure
    rmax = int(lib.cfun_fc_bwd_weight_max_rows(4096))
    if rmax < 64:
        t = torch.zeros(64 * 4096, device=device)
        assert lib.cfun_fc_bwd_weight(ops.ptr(t), ops.ptr(t), ops.ptr(t), rmax + 1, 16, 4096, ops.stream(t)) == -1   # CFUN_EINVAL


def check_maxpool(device, seed=3):
    gen = _gen(seed)
    x = F.relu(randn(gen, 2, 4, 6, 8, 16))   # ReLU output: many exact ties at 0 like the real stem
    gy = randn(gen, 2, 2, 3, 4, 16)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool3d(xr.permute(0, 4, 1, 2, 3), 2, 2).permute(0, 2, 3, 4, 1)
    yr.backward(gy)
    xd = x.clone().to(device).requires_grad_(True)
    y = ops.maxpool2(xd)
    y.backward(gy.to(device))
    assert_close(y, yr, "y", 1e-7)
    # gradients routed to tied zeros may pick a different zero; compare where the input is positive
    m = (x > 0)
    assert_close(xd.grad.cpu() * m, xr.grad * m, "dx", 1e-7)


# ------------------------------------------------------------------------------------------ RoIAlign / NMS
def py_bounds(boxes, dhw):
    """oracle roi_bounds + python slice clamping (what fm[:, lo:hi] actually reads)."""
    ib = orc.roi_bounds(boxes, dhw).numpy()
    out = np.zeros_like(ib)
    for r in range(ib.shape[0]):
        for a in range(3):
            lo, hi, _ = slice(int(ib[r, a]), int(ib[r, 3 + a])).indices(int(dhw[a]))
            out[r, a], out[r, 3 + a] = lo, max(hi, lo)
    return out


def check_roi_align(device, fm, boxes, pool, gy=None, expect=None, expect_grad=None):
    """fm [C,D,H,W] numpy, boxes [R,6]; compares against the oracle (and golden arrays when given)."""
    fm_t = torch.from_numpy(fm)
    bx = torch.from_numpy(boxes)
    fr = fm_t.clone().requires_grad_(True)
    ref = orc.roi_align(fr, pool, bx)
    fd = fm_t.permute(1, 2, 3, 0).contiguous().to(device).requires_grad_(True)
    out, bounds = ops.roi_align(fd, bx.to(device), pool)
    np.testing.assert_array_equal(bounds.cpu().numpy()[:boxes.shape[0]], py_bounds(bx, fm.shape[1:]))  # bit-exact
    got = out.permute(0, 4, 1, 2, 3)
    assert float((got.detach().cpu() - ref.detach()).abs().max()) < 2e-6
    if expect is not None:
        assert float((got.detach().cpu() - torch.from_numpy(expect)).abs().max()) < 2e-6
    if gy is not None:
        g = torch.from_numpy(gy)
        (ref * g).sum().backward()
        (got * g.to(device)).sum().backward()
        gd = fd.grad.permute(3, 0, 1, 2).cpu()
        # (the backward is a gather in RoI order; the oracle sums in torch's order: tolerance relative to the gradient's scale)
        assert float((gd - fr.grad).abs().max()) < 1e-5 * max(1.0, float(fr.grad.abs().max()))
        if expect_grad is not None:
            assert float((gd - torch.from_numpy(expect_grad)).abs().max()) < 1e-5


def check_nms(device, boxes, scores, thr, max_num, expect):
    keep, count = ops.nms3d(torch.from_numpy(boxes).to(device), torch.from_numpy(scores).to(device), thr, max_num)
    k = int(count.item())
    got = keep[:k].cpu().numpy()
    assert got.dtype == np.int32
    np.testing.assert_array_equal(got, expect)   # bit-exact keep list, pick order


# ------------------------------------------------------------------------------------------ losses
def check_mask_losses(device, logits_ncdhw, labels, expect=None):
    """logits [n,C,D,H,W] numpy, labels uint8 [n,D,H,W]."""
    lg = torch.from_numpy(logits_ncdhw)
    lab = torch.from_numpy(labels.astype(np.int64))
    c = lg.shape[1]
    onehot = torch.stack([(lab == k) for k in range(c)], dim=1).double()
    lr = lg.clone().requires_grad_(True)
    ce_r = orc.mask_ce_loss(onehot, lr)
    ce_r.backward()
    g_ce_r = lr.grad.clone()
    lr.grad = None
    el_r = orc.edge_loss(onehot, torch.softmax(lr, dim=1))[0]
    el_r.backward()
    g_el_r = lr.grad.clone()

    ld = lg.permute(0, 2, 3, 4, 1).contiguous().to(device).requires_grad_(True)
    labd = torch.from_numpy(labels).to(device)
    ce = ops.mask_cross_entropy(ld, labd)
    ce.backward()
    g_ce = ld.grad.clone()
    ld.grad = None
    probs = ops.softmax_channels(ld)
    assert_close(probs, torch.softmax(lg, dim=1).permute(0, 2, 3, 4, 1), "softmax", 1e-5)
    el = ops.edge_loss(probs, labd)
    el.backward()
    g_el = ld.grad.clone()
    assert abs(float(ce) - float(ce_r)) < 1e-5 * abs(float(ce_r))
    assert abs(float(el) - float(el_r)) < 1e-4 * abs(float(el_r))
    assert_close(g_ce.permute(0, 4, 1, 2, 3), g_ce_r, "dCE/dlogits", 1e-4)
    assert_close(g_el.permute(0, 4, 1, 2, 3), g_el_r, "dEdge/dlogits", 1e-3)
    # the fused backward (both losses, one pass) = 0.7 * dCE + 1.3 * dEdge of the separate kernels
    ld.grad = None
    ce2, el2 = ops.mask_losses(ld, ops.softmax_channels(ld), labd)
    assert float(ce2) == float(ce) and float(el2) == float(el)
    (0.7 * ce2 + 1.3 * el2).backward()
    assert_close(ld.grad, 0.7 * g_ce + 1.3 * g_el, "fused d(CE+Edge)/dlogits", 1e-5)
    assert_close(ld.grad.permute(0, 4, 1, 2, 3), 0.7 * g_ce_r + 1.3 * g_el_r, "fused vs oracle", 1e-3)
    # round 6: ONE forward pass (softmax + CE + edge loss) and ONE backward pass that recomputes the edge coefficients
    # (cfun_mask_fused_fwd / _bwd) against the separate kernels: identical probabilities, same losses, same gradient
    if ops.mask_losses_fused_supported(ld):
        sep_grad = ld.grad.clone()
        ld.grad = None
        ce3, el3, p3 = ops.mask_losses_fused(ld, labd)
        # (same softmax up to the fused pass's v_exp_f32 and its one division per voxel: a few ulp)
        assert float((p3 - probs.detach()).abs().max()) <= 2e-6, "fused forward: probabilities differ from cfun_softmax_fwd"
        assert not p3.requires_grad
        assert abs(float(ce3) - float(ce)) <= 2e-6 * abs(float(ce)) and abs(float(el3) - float(el)) <= 1e-5 * abs(float(el))
        with torch.no_grad():       # forward-only build of the kernel (no backward operand written): the same sums in the same order
            ce4, el4, p4 = ops.mask_losses_fused(ld.detach(), labd)
        assert float(ce4) == float(ce3) and float(el4) == float(el3) and torch.equal(p4, p3)
        (0.7 * ce3 + 1.3 * el3).backward()
        assert_close(ld.grad, sep_grad, "one-pass d(CE+Edge)/dlogits vs the separate kernels", 2e-5)
        assert_close(ld.grad.permute(0, 4, 1, 2, 3), 0.7 * g_ce_r + 1.3 * g_el_r, "one-pass backward vs oracle", 1e-3)
    if expect is not None:
        assert abs(float(ce) - float(expect["ce"])) < 1e-5 * abs(float(expect["ce"]))
        assert abs(float(el) - float(expect["edge"].reshape(-1)[0])) < 1e-4 * abs(float(expect["edge"].reshape(-1)[0]))
        assert_close(g_ce.permute(0, 4, 1, 2, 3), torch.from_numpy(expect["ce_grad"]), "golden dCE", 1e-4)
        assert_close(g_el.permute(0, 4, 1, 2, 3), torch.from_numpy(expect["edge_grad_logits"]), "golden dEdge", 1e-3)


def check_halo(device, seed=5):
    gen = _gen(seed)
    x = randn(gen, 2, 6, 3, 4, 8)
    xd = x.to(device)
    buf = ops.halo_pack(xd, 4, 2)
    assert_close(buf, x[:, 4:6], "halo_pack", 1e-9)
    dst = torch.zeros(2, 8, 3, 4, 8, device=device)
    ops.halo_unpack(buf, dst, 0)
    assert_close(dst[:, 0:2], x[:, 4:6], "halo_unpack", 1e-9)
    assert float(dst[:, 2:].abs().max()) == 0.0


def check_mask_target_labels(device, seed=11):
    """cfun_mask_target_labels vs the oracle's crop + nearest resize of the one-hot GT (model.py:481-493), bit-exact:
    ragged, thin (1-voxel) and boundary-touching boxes, up- and down-sampling, non-cubic mask shapes."""
    gen = _gen(seed)
    D, H, W, C = 12, 20, 17, 5
    lab = torch.randint(0, C, (D, H, W), generator=gen, dtype=torch.int64)
    onehot = torch.stack([(lab == k) for k in range(C)], dim=0).float()
    lo = torch.rand(9, 3, generator=gen) * 0.6
    hi = lo + 0.08 + torch.rand(9, 3, generator=gen) * 0.35
    rois = torch.cat([lo, hi.clamp(max=1.0)], dim=1)
    rois[0] = torch.tensor([0.0, 0.0, 0.0, 1.0, 1.0, 1.0])            # whole volume
    rois[1] = torch.tensor([0.5, 0.5, 0.5, 0.5 + 1.01 / D, 1.0, 0.9])  # one plane thick
    for shape in ((8, 8, 8), (24, 40, 34), (5, 9, 7)):
        ref = orc.mask_targets(rois, onehot, shape).argmax(1).to(torch.uint8)
        out = ops.mask_target_labels(lab.to(torch.uint8).to(device), rois.to(device), shape)
        assert torch.equal(out.cpu(), ref), shape


def check_weight_layouts(device, seed=13):
    """cfun_weight_pack / _pack_transpose / _unpack vs permute + pad, bit-exact (C_out not a multiple of 16, 1 and
    147 taps, tiles that straddle 32)."""
    gen = _gen(seed)
    for co, ci, k in ((20, 8, (3, 3, 3)), (3, 40, (1, 1, 1)), (16, 1, (3, 7, 7)), (70, 33, (1, 3, 3)), (8, 8, (5, 5, 5))):
        w = randn(gen, co, ci, *k)
        t = k[0] * k[1] * k[2]
        cop, cip = (co + 15) // 16 * 16, (ci + 15) // 16 * 16
        ref = torch.zeros(t, ci, cop)
        ref[:, :, :co] = w.permute(2, 3, 4, 1, 0).reshape(t, ci, co)
        wd = w.to(device).requires_grad_(True)
        wp = ops.pack_weight(wd)
        assert torch.equal(wp.detach().cpu(), ref), (co, ci, k)
        ref_t = torch.zeros(t, co, cip)
        ref_t[:, :, :ci] = ref[:, :, :co].transpose(1, 2)
        assert torch.equal(ops._transpose_pack(wp.detach(), co).cpu(), ref_t), (co, ci, k)
        wp2, wpt2 = ops._pack(wd, both=True)                      # both layouts in one launch
        assert torch.equal(wp2.cpu(), ref) and torch.equal(wpt2.cpu(), ref_t), (co, ci, k)
        g = randn(gen, t, ci, cop)
        wp.backward(g.to(device))
        assert torch.equal(wd.grad.cpu(), g[:, :, :co].reshape(*k, ci, co).permute(4, 3, 0, 1, 2)), (co, ci, k)


def check_mask_losses_lits(device, g):
    """LiTS fork mask losses on the device (ops.mask_cross_entropy(weight=...), ops.edge_loss_raw through the
    differentiable softmax) vs the fork's own values and gradients (golden ``g`` = losses_lits.npz)."""
    lg = torch.from_numpy(g["logits"])
    ld = lg.permute(0, 2, 3, 4, 1).contiguous().to(device).requires_grad_(True)
    labd = torch.from_numpy(g["labels"]).to(device)
    ce = ops.mask_cross_entropy(ld, labd, weight=g["class_weights"])
    ce.backward()
    assert abs(float(ce) - float(g["ce"])) < 1e-5 * abs(float(g["ce"]))
    assert_close(ld.grad.permute(0, 4, 1, 2, 3), torch.from_numpy(g["ce_grad"]), "golden weighted dCE", 1e-4)
    ld.grad = None
    el = ops.edge_loss_raw(ops.softmax_channels(ld), labd)
    el.backward()
    assert abs(float(el) - float(g["edge"].reshape(-1)[0])) < 1e-4 * abs(float(g["edge"].reshape(-1)[0]))
    assert_close(ld.grad.permute(0, 4, 1, 2, 3), torch.from_numpy(g["edge_grad_logits"]), "golden raw-Sobel dEdge", 1e-3)
    # forward-only call keeps no difference field
    with torch.no_grad():
        el2 = ops.edge_loss_raw(ops.softmax_channels(ld.detach()), labd)
    assert float(el2) == float(el)


# ---- batched weight preparation (ops.WeightScope / cfun_weight_prepare): every operand kind against the per-conv path
WEIGHT_SCOPE_CASES = [
    # ci, co, k, stride, algo, (D,H,W), gather (None | 0 | 1: output / input channels gathered from a wider weight)
    (8, 40, (3, 3, 3), 1, ALGO_MFMA, (4, 5, 7), None),       # PACK / PACKT
    (16, 32, (3, 3, 3), 1, ALGO_WINO, (4, 5, 9), None),      # WINO1 / WINO1_T
    (16, 48, (3, 3, 3), 1, ALGO_WINO2, (5, 4, 9), None),     # WINO2 / WINO2_T
    (8, 16, (3, 3, 3), 2, ALGO_AUTO, (8, 8, 8), None),       # PACK / S2FOLD
    (20, 36, (3, 3, 3), 2, ALGO_AUTO, (4, 6, 8), None),      # S2FOLD with padding columns (8*20 = 160 -> 160, 36 rows: ragged tiles)
    (24, 8, (1, 1, 1), 1, ALGO_AUTO, (4, 4, 8), None),       # pointwise
    (16, 16, (1, 3, 3), 1, ALGO_AUTO, (3, 6, 7), None),
    (16, 20, (3, 1, 1), 1, ALGO_AUTO, (5, 4, 6), None),
    (3, 5, (3, 3, 3), 1, ALGO_AUTO, (4, 4, 5), None),        # C % 4 != 0: the direct kernels read the same packs
    (20, 12, (3, 3, 3), 1, ALGO_AUTO, (4, 4, 8), 0),         # Dropout3d slices: 12 of 20 output channels gathered
    (12, 20, (3, 3, 3), 1, ALGO_AUTO, (4, 4, 8), 1),         # ... 12 of 20 input channels
    (32, 40, (3, 3, 3), 1, ALGO_WINO2, (4, 4, 8), 1),        # gathered input channels into a Winograd operand
]


def check_weight_scope(device, seed=41):
    """Each case runs forward + backward three times inside ops.WeightScope: pass 1 records and packs per conv, passes 2
    and 3 take their operands from the ONE cfun_weight_prepare launch of the scope.  y, dx, dw must agree bit for bit and
    the later passes must not miss (a miss = the table's operand kinds differ from what the conv's dispatch wants)."""
    gen = _gen(seed)
    for ci, co, k, stride, algo, dhw, gather in WEIGHT_SCOPE_CASES:
        wide = 20 if gather is not None and max(ci, co) <= 20 else max(ci, co) + 8
        wshape = (wide if gather == 0 else co, wide if gather == 1 else ci) + tuple(k)
        w = torch.nn.Parameter((randn(gen, *wshape) / float(ci * k[0] * k[1] * k[2]) ** 0.5).to(device))
        x = randn(gen, 2, *dhw, ci).to(device)
        pad = tuple(kk // 2 for kk in k)
        spec = ops.ConvSpec(k=tuple(k), co=co, stride=stride, pad=pad, algo=algo)
        idx = None
        if gather is not None:
            n = co if gather == 0 else ci
            idx = torch.sort(torch.randperm(wide, generator=gen)[:n]).values.to(device)
        owner = torch.nn.Module()
        ref = None
        for it in range(3):
            xi = x.clone().requires_grad_(True)
            w.grad = None
            with ops.WeightScope(owner, dyn={"lv": [idx]}) as scope:
                wi = w if idx is None else ops.gather_slices(w, gather, [idx], key="lv")[0]
                y = ops.conv3d_w(xi, wi, spec)
                y.backward(torch.ones_like(y) * 0.5 + y.detach() * 0.25)
                if it:
                    assert scope.hits == 1 and scope.misses == 0, (ci, co, k, stride, algo, gather, scope.hits, scope.misses)
            cur = (y.detach().clone(), xi.grad.clone(), w.grad.clone())
            if ref is None:
                ref = cur
                yr = ref_conv(x.cpu(), (w.detach() if idx is None else w.detach().index_select(gather, idx)).cpu(), spec, None, None, None)
                assert_close(y, yr, "y (%s)" % (spec,), 2e-5)
            else:
                for a, c, nm in zip(ref, cur, ("y", "dx", "dw")):
                    assert torch.equal(a, c), "%s differs on the prepared pass %d of case %s" % (nm, it, (ci, co, k, stride, algo, gather))


def check_fold_up2_kernels(device, seed=41):
    """cfun_fold_up2_fwd / _bwd against the fold written out with torch: F[pqr][tuv][abc] = f[p,a,t] f[q,b,u] f[r,c,v],
    f[p][a][t] = 1 where hi-res tap t of output parity p reads low-res offset a = floor((p + t - k//2) / 2) + 1; the forward is
    w . F (parity groups padded to cqp rows), the backward its transpose (the formulation rounds 3-4 ran as a batched matmul)."""
    gen = _gen(seed)
    for k, o, i, cqp in ((3, 20, 40, 32), (3, 40, 8, 48), (3, 8, 4, 8), (5, 8, 8, 8), (5, 3, 4, 16)):
        f = torch.zeros(2, 3, k)
        for p in range(2):
            for t in range(k):
                f[p, (p + t - k // 2) // 2 + 1, t] = 1.0
        big = torch.einsum("pat,qbu,rcv->pqrtuvabc", f, f, f).reshape(8, k ** 3, 27)
        w = randn(gen, o, i, k, k, k)
        g = randn(gen, 8 * cqp, i, 3, 3, 3)
        a = torch.nn.functional.pad(w.reshape(o, i * k ** 3), (0, 0, 0, cqp - o))
        wf_ref = torch.matmul(a.reshape(1, cqp * i, k ** 3).double(), big.double()).reshape(8 * cqp, i, 3, 3, 3)
        dw_ref = torch.matmul(g.reshape(8, cqp * i, 27).double(), big.double().transpose(1, 2)).sum(0).reshape(cqp, i, k, k, k)[:o]
        wd = w.clone().to(device).requires_grad_(True)
        wf = ops.fold_up2_weight(wd, cqp)
        assert tuple(wf.shape) == (8 * cqp, i, 3, 3, 3)
        wf.backward(g.to(device))
        assert_close(wf.detach(), wf_ref.float(), "folded weight (k=%d)" % k, 1e-6)
        assert_close(wd.grad, dw_ref.float(), "fold backward (k=%d)" % k, 1e-6)
        pad_rows = wf.detach().reshape(8, cqp, -1)[:, o:]
        assert pad_rows.numel() == 0 or float(pad_rows.abs().max()) == 0.0          # the padding rows of a parity group are zeros
