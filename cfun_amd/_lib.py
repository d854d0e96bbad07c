"""ctypes binding of libcfun_hip.so (include/cfun_hip.h).

The library is built in-tree by ``cfun_amd/csrc/Makefile`` (``__graft_entry__.build()``).  There is no CPU
fallback: if the shared object is missing, or a CPU tensor reaches a kernel of the real library, this module
raises.  ``CFUN_LIB_PATH`` may point at another build of the same C ABI; the CPU test tier uses it to load
``tests/emu/libcfun_emu.so`` (the same kernel sources compiled against a functional HIP emulator).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libcfun_hip.so")

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
ALGO_AUTO, ALGO_DIRECT, ALGO_MFMA = 0, 1, 2
ALGO_WINO = 4    # every supported 3x3x3 stride-1 conv on the Winograd F(2,3)-along-x kernel (AUTO picks it per shape)
ALGO_WINO2 = 5   # ... with y in the Winograd domain as well (forward / data gradient)


class ConvParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "Di", "Hi", "Wi", "Ci", "Do", "Ho", "Wo", "Co", "CoP", "CiP",
                                         "kd", "kh", "kw", "stride", "pd", "ph", "pw", "up2", "act")] + \
               [("slope", C.c_float)] + \
               [(n, C.c_int32) for n in ("scale_mode", "has_shift", "res_mode", "res_up2", "d2s", "d2s_cq", "tap_skip", "algo",
                                         "w_prepared")]


class ConvFusion(C.Structure):
    """CfunConvFusion (include/cfun_hip.h): InstanceNorm / LeakyReLU hooks of the conv kernels."""
    _fields_ = [("in_stats", C.c_void_p), ("in_act", C.c_int32), ("in_slope", C.c_float), ("out_stats", C.c_void_p),
                ("out_eps", C.c_float)]


FUSE_OUT_STATS, FUSE_IN_NORM, FUSE_IN_NORM_WGRAD = 1, 2, 4


class WeightJob(C.Structure):
    """CfunWeightJob (include/cfun_hip.h): one conv's weight operands in the batched cfun_weight_prepare launch."""
    _fields_ = [("w", C.c_void_p), ("fwd", C.c_void_p), ("dgrad", C.c_void_p), ("co_idx", C.c_void_p), ("ci_idx", C.c_void_p),
                ("Co", C.c_int32), ("Ci", C.c_int32), ("T", C.c_int32), ("src_ci", C.c_int32), ("fwd_kind", C.c_int32),
                ("dgrad_kind", C.c_int32), ("block_begin", C.c_uint32), ("reserved", C.c_int32)]


WOP_NONE = 0

_P = C.c_void_p
_I = C.c_int32
_L = C.c_int64
_F = C.c_float
_Z = C.c_size_t
_PP = C.POINTER(ConvParams)
_PF = C.POINTER(ConvFusion)

_SIGNATURES = {
    "cfun_version": (C.c_int, []),
    "cfun_error_string": (C.c_char_p, [C.c_int]),
    "cfun_conv3d_fwd_workspace_bytes": (_Z, [_PP]),
    "cfun_conv3d_fwd_kernel": (_I, [_PP]),
    "cfun_conv3d_wino_plan": (C.c_int, [_PP, C.POINTER(C.c_int32)]),
    "cfun_conv3d_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _PP, _P, _Z, _P]),
    "cfun_conv3d_fused_support": (C.c_int, [_PP]),
    "cfun_conv3d_fwd_fused_workspace_bytes": (_Z, [_PP, _PF]),
    "cfun_conv3d_fwd_fused": (C.c_int, [_P, _P, _P, _P, _P, _P, _PP, _PF, _P, _Z, _P]),
    "cfun_conv3d_bwd_weight_fused": (C.c_int, [_P, _P, _P, _I, _PP, _PF, _P, _Z, _P]),
    "cfun_conv3d_bwd_data_workspace_bytes": (_Z, [_PP]),
    "cfun_conv3d_bwd_data": (C.c_int, [_P, _P, _P, _PP, _P, _Z, _P]),
    "cfun_conv3d_bwd_weight_workspace_bytes": (_Z, [_PP]),
    "cfun_conv3d_bwd_weight": (C.c_int, [_P, _P, _P, _PP, _P, _Z, _P]),
    "cfun_conv3d_bwd_weight_oidhw": (C.c_int, [_P, _P, _P, _PP, _P, _Z, _P]),
    "cfun_act_bwd": (C.c_int, [_P, _P, _P, _P, _L, _I, _L, _I, _F, _I, _P]),
    "cfun_channel_sum_workspace_bytes": (_Z, [_L, _I]),
    "cfun_channel_sum": (C.c_int, [_P, _P, _L, _I, _P, _Z, _P]),
    "cfun_fc_workspace_bytes": (_Z, [_I, _I, _I]),
    "cfun_fc_fwd": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _Z, _P]),
    "cfun_fc_bwd_weight": (C.c_int, [_P, _P, _P, _I, _I, _I, _P]),
    "cfun_fc_bwd_weight_max_rows": (_I, [_I]),
    "cfun_fc_bwd_data": (C.c_int, [_P, _P, _P, _I, _I, _I, _P]),
    "cfun_lrelu_fwd": (C.c_int, [_P, _P, _L, _F, _P]),
    "cfun_lrelu_bwd": (C.c_int, [_P, _P, _P, _L, _F, _P]),
    "cfun_add": (C.c_int, [_P, _P, _P, _L, _P]),
    "cfun_lrelu_bwd_add": (C.c_int, [_P, _P, _P, _P, _L, _I, _L, _F, _P]),
    "cfun_instnorm_lrelu_bwd_add": (C.c_int, [_P, _P, _P, _P, _P, _I, _L, _I, _L, _F, _P, _Z, _P]),
    "cfun_upsample2_bwd": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "cfun_instnorm_workspace_bytes": (_Z, [_I, _L, _I]),
    "cfun_instnorm_stats": (C.c_int, [_P, _P, _I, _L, _I, _F, _P, _Z, _P]),
    "cfun_instnorm_lrelu_fwd": (C.c_int, [_P, _P, _P, _I, _L, _I, _F, _P]),
    "cfun_instnorm_lrelu_bwd": (C.c_int, [_P, _P, _P, _P, _I, _L, _I, _F, _P, _Z, _P]),
    "cfun_instnorm_lrelu_fwd_strided": (C.c_int, [_P, _P, _P, _I, _L, _I, _L, _F, _P]),
    "cfun_instnorm_lrelu_bwd_strided": (C.c_int, [_P, _P, _P, _P, _I, _L, _I, _L, _F, _P, _Z, _P]),
    "cfun_instnorm_bwd_means": (C.c_int, [_P, _P, _P, _P, _I, _L, _I, _L, _F, _P, _Z, _P]),
    "cfun_instnorm_lrelu_bwd_apply": (C.c_int, [_P, _P, _P, _P, _P, _I, _L, _I, _L, _F, _P]),
    "cfun_lrelu_fwd_strided": (C.c_int, [_P, _P, _L, _I, _L, _L, _F, _P]),
    "cfun_lrelu_bwd_strided": (C.c_int, [_P, _P, _P, _L, _I, _L, _F, _P]),
    "cfun_maxpool2_fwd": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cfun_maxpool2_bwd": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cfun_roi_align3d_fwd": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "cfun_roi_align3d_bwd": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "cfun_roi_align3d_slab_fwd": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "cfun_roi_align3d_slab_bwd": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "cfun_nms3d_workspace_bytes": (_Z, [_I]),
    "cfun_nms3d": (C.c_int, [_P, _P, _I, _F, _I, _P, _P, _P, _Z, _P]),
    "cfun_softmax_fwd": (C.c_int, [_P, _P, _L, _I, _P]),
    "cfun_softmax_bwd": (C.c_int, [_P, _P, _P, _L, _I, _P]),
    "cfun_loss_workspace_bytes": (_Z, [_L]),
    "cfun_softmax_ce_fwd": (C.c_int, [_P, _P, _P, _L, _I, _P, _Z, _P]),
    "cfun_softmax_ce_bwd": (C.c_int, [_P, _P, _P, _P, _L, _I, _P]),
    "cfun_edge_loss_fwd": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "cfun_edge_loss_bwd_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "cfun_edge_loss_bwd": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "cfun_mask_losses_bwd": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "cfun_edge_loss_fwd_save": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "cfun_mask_losses_bwd_saved": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cfun_mask_fused_workspace_bytes": (_Z, []),
    "cfun_mask_fused_supported": (C.c_int, [_I, _I, _I, _I, _I]),
    "cfun_mask_fused_fwd": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "cfun_mask_fused_u_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "cfun_mask_fused_bwd": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cfun_mask_target_labels": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "cfun_ce_weighted_workspace_bytes": (_Z, []),
    "cfun_softmax_ce_weighted_fwd": (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _P, _Z, _P]),
    "cfun_softmax_ce_weighted_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "cfun_edge_raw_dc_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "cfun_edge_raw_fwd": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "cfun_edge_raw_bwd": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cfun_unmold_argmax": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "cfun_unmold_overlap": (C.c_int, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "cfun_sumsq_partials_count": (C.c_int32, []),
    "cfun_sumsq_partials": (C.c_int, [_P, C.c_int64, _P, _P]),
    "cfun_norm_finalize": (C.c_int, [_P, _I, _P, _P]),
    "cfun_sgd_momentum_step": (C.c_int, [_P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, _P, _I, _P]),
    "cfun_weight_pack": (C.c_int, [_P, _P, _I, _I, _I, _P]),
    "cfun_weight_pack_transpose": (C.c_int, [_P, _P, _I, _I, _I, _P]),
    "cfun_weight_pack_both": (C.c_int, [_P, _P, _P, _I, _I, _I, _P]),
    "cfun_fold_up2_fwd": (C.c_int, [_P, _P, _I, _I, _I, _I, _P]),
    "cfun_fold_up2_bwd": (C.c_int, [_P, _P, _I, _I, _I, _I, _P]),
    "cfun_weight_prepare_kinds": (C.c_int, [_PP, C.POINTER(C.c_int32), C.POINTER(C.c_size_t)]),
    "cfun_weight_prepare_plan": (C.c_int, [C.POINTER(WeightJob), _I, C.POINTER(C.c_int64)]),
    "cfun_weight_prepare": (C.c_int, [_P, _I, _L, _P]),
    "cfun_weight_unpack": (C.c_int, [_P, _P, _I, _I, _I, _P]),
    "cfun_halo_pack": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "cfun_halo_unpack": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "cfun_resize3d": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
}

EXPORTS = tuple(_SIGNATURES)

_lib = None
_lib_path = None
_is_emulator = False


def lib_path():
    return os.environ.get("CFUN_LIB_PATH") or DEFAULT_LIB


def load():
    """Load (once) and return the ctypes library.  Raises if it is not built."""
    global _lib, _lib_path, _is_emulator
    path = lib_path()
    if _lib is not None and _lib_path == path:
        return _lib
    if not os.path.exists(path):
        raise RuntimeError(
            "cfun_amd: %s not found -- the HIP library is required (no CPU fallback). Build it with "
            "`make -C cfun_amd/csrc -j8` or `python -c 'import __graft_entry__ as g; g.build()'`." % path)
    lib = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = ABI mismatch, fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib, _lib_path = lib, path
    _is_emulator = os.path.basename(path).startswith("libcfun_emu")
    return lib


def is_emulator():
    load()
    return _is_emulator


def check(rc, what):
    if rc != 0:
        msg = load().cfun_error_string(rc)
        raise RuntimeError("cfun_amd: %s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def ptr(t):
    """Device pointer of a dense tensor (or NULL for None)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise RuntimeError("cfun_amd: kernel argument must be contiguous, got strides %s" % (t.stride(),))
    if not t.is_cuda and not is_emulator():
        raise RuntimeError("cfun_amd: CPU tensor passed to a HIP kernel (there is no CPU fallback)")
    return t.data_ptr()


def stream(t):
    if t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    return None


def workspace(nbytes, like):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=like.device)
