"""ctypes binding of libvsseg_hip.so (C ABI declared in include/vsseg_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("VSSEG_LIB_PATH") or os.path.join(_HERE, "libvsseg_hip.so")  # VSSEG_LIB_PATH: tuning builds (tools/prof_phases.sh)

F32, BF16 = 0, 1
ACT_NONE, ACT_PRELU, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3
RES_NONE, RES_ADD, RES_RELUMASK, RES_GATE, RES_IN1 = 0, 1, 2, 3, 5
MAX_TAPS = 27
STAT_SHARDS = 256
SEED_INDIRECT = 0x80000000
EINVAL, ELAUNCH = -1, -2
ZERO_PADDED = 1  # vsseg_tensor.reserved of a destination whose channels c .. pitch-1 are zero padding


class Tensor(C.Structure):
    # positional construction Tensor(ptr, dtype, c, pitch, n, x, y, z) leaves ptr2/csplit zero = an ordinary tensor
    _fields_ = [("ptr", C.c_void_p), ("dtype", C.c_int32), ("c", C.c_int32), ("pitch", C.c_int32), ("n", C.c_int32), ("x", C.c_int32), ("y", C.c_int32), ("z", C.c_int32),
                ("ptr2", C.c_void_p), ("csplit", C.c_int32), ("reserved", C.c_int32)]

    @classmethod
    def two_part(cls, a: "Tensor", b: "Tensor") -> "Tensor":
        """The channel concatenation [a | b] of two tensors of the same geometry, dtype and pitch (nothing is copied)."""
        assert (a.dtype, a.pitch, a.n, a.x, a.y, a.z) == (b.dtype, b.pitch, b.n, b.x, b.y, b.z) and not a.ptr2 and not b.ptr2 and a.c % 16 == 0
        return cls(a.ptr, a.dtype, a.c + b.c, a.pitch, a.n, a.x, a.y, a.z, b.ptr, a.c, 0)


class CropJob(C.Structure):  # vsseg_crop_job
    _fields_ = [("src", C.c_void_p), ("sdims", C.c_int32 * 3), ("origin", C.c_int32 * 3), ("flip_x", C.c_int32)]


class IgemmDesc(C.Structure):
    _fields_ = [
        ("inp", Tensor),
        ("out", Tensor),
        ("q", C.c_int32 * 3),
        ("is_", C.c_int32 * 3),
        ("os", C.c_int32 * 3),
        ("oo", C.c_int32 * 3),
        ("ntaps", C.c_int32),
        ("tap_off", (C.c_int32 * 3) * MAX_TAPS),
        ("tile", C.c_int32 * 3),
        ("mtw", C.c_int32),
        ("nt", C.c_int32),
        ("nsplit", C.c_int32),
        ("ck", C.c_int32),
        ("nchunks", C.c_int32),
        ("ksteps", C.c_int32),
        ("depth", C.c_int32),
        ("wpack", C.c_void_p),
        ("bias", C.c_void_p),
        ("bias2", C.c_void_p),
        ("scale", C.c_void_p),
        ("shift", C.c_void_p),
        ("alpha", C.c_void_p),
        ("act", C.c_int32),
        ("res_mode", C.c_int32),
        ("res", Tensor),
        ("accumulate", C.c_int32),
        ("stats", C.c_void_p),
        ("stats_stride", C.c_int32),
        ("cout_mod", C.c_int32),
        ("gate", C.c_void_p),
        ("in_gate", C.c_void_p),
        ("class_split", C.c_int32),
        ("class_oo", (C.c_int32 * 3) * 8),
        ("class_ntaps", C.c_int32 * 8),
        ("class_tap", (C.c_int32 * 8) * 8),
        ("res_tiles", C.c_int32),
        ("wpack_res", C.c_void_p),
        ("bias_res", C.c_void_p),
        ("res_out", Tensor),
        ("in1", C.c_void_p),
        ("in1_w", C.c_void_p),
        ("in1_b", C.c_void_p),
    ]


DICE_MAX_LEVELS = 8


class DiceTailDesc(C.Structure):  # vsseg_dice_tail_desc
    _fields_ = [
        ("n", C.c_int32),
        ("nlevels", C.c_int32),
        ("src", C.c_void_p),
        ("sdims", C.c_int32 * 3),
        ("att", C.c_void_p * DICE_MAX_LEVELS),
        ("label", C.c_void_p * DICE_MAX_LEVELS),
        ("dims", (C.c_int32 * 3) * DICE_MAX_LEVELS),
        ("sums", C.c_void_p),
    ]


class DiceBwdLevelsDesc(C.Structure):  # vsseg_dice_bwd_levels_desc
    _fields_ = [
        ("n", C.c_int32),
        ("nlevels", C.c_int32),
        ("label", C.c_void_p * DICE_MAX_LEVELS),
        ("datt", C.c_void_p * DICE_MAX_LEVELS),
        ("coef", C.c_void_p * DICE_MAX_LEVELS),
        ("nvox", C.c_int64 * DICE_MAX_LEVELS),
        ("gscale", C.c_void_p),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("p", Tensor),
        ("h", Tensor),
        ("q", C.c_int32 * 3),
        ("hs", C.c_int32 * 3),
        ("ntaps", C.c_int32),
        ("tap_off", (C.c_int32 * 3) * MAX_TAPS),
        ("tap_widx", C.c_int32 * MAX_TAPS),
        ("tile", C.c_int32 * 3),
        ("ntp", C.c_int32),
        ("dw", C.c_void_p),
        ("stride_p", C.c_int64),
        ("stride_h", C.c_int64),
        ("stride_tap", C.c_int64),
        ("cp_valid", C.c_int32),
        ("ch_valid", C.c_int32),
        ("persistent_blocks", C.c_int32),
        ("scratch", C.c_void_p),
        ("scratch_elems", C.c_int64),
        ("single_buffer", C.c_int32),
        ("hgroup", C.c_int32),
        ("dbias_p", C.c_void_p),
        ("march", C.c_int32),
        ("h_gate", C.c_void_p),
    ]


class ConvBwdDesc(C.Structure):
    _fields_ = [
        ("y", Tensor),
        ("dout", Tensor),
        ("x", Tensor),
        ("dx", Tensor),
        ("mean", C.c_void_p),
        ("invstd", C.c_void_p),
        ("gamma", C.c_void_p),
        ("scale", C.c_void_p),
        ("shift", C.c_void_p),
        ("alpha", C.c_void_p),
        ("mean_dz", C.c_void_p),
        ("mean_dzx", C.c_void_p),
        ("p_drop", C.c_float),
        ("keep", C.c_void_p),
        ("wpack", C.c_void_p),
        ("dw", C.c_void_p),
        ("tile", C.c_int32 * 3),
        ("scratch", C.c_void_p),
        ("scratch_elems", C.c_int64),
        ("dres", Tensor),
        ("wpack_res", C.c_void_p),
        ("dw_res", C.c_void_p),
        ("x_gate", C.c_void_p),
    ]


class ChainDesc(C.Structure):  # vsseg_chain_desc
    _fields_ = [
        ("inp", Tensor),
        ("out", Tensor),
        ("cmid", C.c_int32),
        ("wpack_a", C.c_void_p),
        ("bias_a", C.c_void_p),
        ("scale_a", C.c_void_p),
        ("shift_a", C.c_void_p),
        ("alpha_a", C.c_void_p),
        ("act_a", C.c_int32),
        ("wpack_b", C.c_void_p),
        ("bias_b", C.c_void_p),
        ("scale_b", C.c_void_p),
        ("shift_b", C.c_void_p),
        ("alpha_b", C.c_void_p),
        ("act_b", C.c_int32),
        ("in1_w", C.c_void_p),
        ("in1_b", C.c_void_p),
        ("res_tiles", C.c_int32),
        ("wpack_res", C.c_void_p),
        ("bias_res", C.c_void_p),
        ("tz", C.c_int32),
        ("mtw", C.c_int32),
        ("lx", C.c_int32),
        ("waves", C.c_int32),
        ("lead", C.c_int32),
    ]


# every exported entry point of include/vsseg_hip.h (the non-GPU tests check the .so exports each of them)
SYMBOLS = [
    "vsseg_last_error", "vsseg_version", "vsseg_fx_status", "vsseg_memset_zero", "vsseg_copy_bytes", "vsseg_store_u64", "vsseg_crop_flip", "vsseg_normalize_intensity", "vsseg_igemm", "vsseg_igemm_lds_bytes", "vsseg_conv_chain", "vsseg_conv_chain_lds_bytes", "vsseg_conv_to1", "vsseg_wgrad", "vsseg_conv_bwd_fused", "vsseg_wgrad_narrow", "vsseg_wgrad_narrow_bn", "vsseg_gather_cast", "vsseg_merge_residual_grads", "vsseg_stage_input",
    "vsseg_bn_finalize", "vsseg_bn_fold_eval", "vsseg_bn_act_fwd", "vsseg_bn_act_fwd_res1", "vsseg_bn_act_bwd_reduce", "vsseg_bn_act_bwd_finalize", "vsseg_bn_act_bwd_apply",
    "vsseg_dropout_mask", "vsseg_att_apply_fwd", "vsseg_att_apply_bwd", "vsseg_channel_sum", "vsseg_add_inplace", "vsseg_copy_cast",
    "vsseg_maxpool_label", "vsseg_dice_pred_sums", "vsseg_dice_att_sums", "vsseg_dice_finalize", "vsseg_dice_pred_bwd", "vsseg_dice_pred_bwd_to", "vsseg_dice_att_bwd", "vsseg_dice_level_sums", "vsseg_dice_tail_sums", "vsseg_dice_att_bwd_levels", "vsseg_fork_event_create", "vsseg_fork_event_destroy", "vsseg_fork_arm", "vsseg_fork_disarm", "vsseg_stream_wait_event",
    "vsseg_adam", "vsseg_swi_accumulate", "vsseg_swi_finalize", "vsseg_hard_dice_counts", "vsseg_argmax2",
]  # fmt: skip

_lib = None


def lib():
    """Load libvsseg_hip.so (built in-tree by `__graft_entry__.build()` / `make -C vs_seg_amd/csrc`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(f"vs_seg_amd: HIP extension {SO_PATH} is missing — build it with `make -C vs_seg_amd/csrc` (there is no CPU fallback)")
        L = C.CDLL(SO_PATH)
        L.vsseg_last_error.restype = C.c_char_p
        for name in SYMBOLS:
            fn = getattr(L, name)
            if name != "vsseg_last_error":
                fn.restype = C.c_int
        vp, i32, i64, f32, f64, u64, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_uint64, C.c_uint32
        I3 = C.POINTER(C.c_int32)
        L.vsseg_store_u64.argtypes = [vp, u64, vp]
        L.vsseg_fx_status.argtypes = [i32, vp]
        L.vsseg_memset_zero.argtypes = [vp, i64, vp]
        L.vsseg_copy_bytes.argtypes = [vp, vp, i64, vp]
        L.vsseg_crop_flip.argtypes = [vp, i32, vp, I3, vp]
        L.vsseg_normalize_intensity.argtypes = [vp, vp, i64, vp, vp]
        L.vsseg_igemm.argtypes = [C.POINTER(IgemmDesc), vp]
        L.vsseg_igemm_lds_bytes.argtypes = [C.POINTER(IgemmDesc)]
        L.vsseg_conv_chain.argtypes = [C.POINTER(ChainDesc), vp]
        L.vsseg_conv_chain_lds_bytes.argtypes = [C.POINTER(ChainDesc)]
        L.vsseg_conv_to1.argtypes = [Tensor, vp, vp, i32, Tensor, i32, vp]
        L.vsseg_wgrad.argtypes = [C.POINTER(WgradDesc), vp]
        L.vsseg_conv_bwd_fused.argtypes = [C.POINTER(ConvBwdDesc), vp]
        L.vsseg_wgrad_narrow.argtypes = [Tensor, vp, i32, i32, vp, C.c_int64, vp, vp, C.c_int64, vp]
        L.vsseg_wgrad_narrow_bn.argtypes = [Tensor, Tensor, vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, C.c_int64, vp, C.c_int64, vp]
        L.vsseg_gather_cast.argtypes = [vp, vp, vp, vp, i64, i32, vp]
        L.vsseg_merge_residual_grads.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
        L.vsseg_stage_input.argtypes = [vp, i32, I3, I3, Tensor, vp]
        L.vsseg_bn_finalize.argtypes = [vp, i32, i32, f64, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp]
        L.vsseg_bn_fold_eval.argtypes = [vp, vp, vp, vp, f32, vp, vp, i32, vp]
        L.vsseg_bn_act_fwd.argtypes = [Tensor, vp, vp, vp, f32, u64, u32, Tensor, i32, Tensor, vp, vp]
        L.vsseg_bn_act_fwd_res1.argtypes = [Tensor, vp, vp, vp, f32, u64, u32, vp, vp, vp, Tensor, vp, vp]
        L.vsseg_bn_act_bwd_reduce.argtypes = [Tensor, Tensor, vp, vp, vp, vp, vp, vp, vp, f32, u64, u32, vp, i32, vp, vp, vp]
        L.vsseg_bn_act_bwd_finalize.argtypes = [vp, i32, vp, i32, f64, vp, vp, vp, vp, vp, vp, vp]
        L.vsseg_bn_act_bwd_apply.argtypes = [Tensor, Tensor, vp, vp, vp, vp, vp, vp, vp, f32, u64, u32, vp, vp, Tensor, vp, vp]
        L.vsseg_dropout_mask.argtypes = [vp, i64, i32, f32, u64, u32, vp]
        L.vsseg_att_apply_fwd.argtypes = [Tensor, vp, Tensor, vp]
        L.vsseg_att_apply_bwd.argtypes = [Tensor, vp, Tensor, vp, Tensor, i32, Tensor, vp, vp, vp]
        L.vsseg_channel_sum.argtypes = [Tensor, vp, vp]
        L.vsseg_add_inplace.argtypes = [Tensor, Tensor, vp]
        L.vsseg_copy_cast.argtypes = [Tensor, Tensor, vp]
        L.vsseg_maxpool_label.argtypes = [vp, i32, I3, I3, vp, vp]
        L.vsseg_dice_pred_sums.argtypes = [vp, i32, vp, i32, i64, i32, vp, vp]
        L.vsseg_dice_att_sums.argtypes = [vp, vp, i32, i64, vp, vp]
        L.vsseg_dice_finalize.argtypes = [vp, vp, i32, i32, vp, vp, vp]
        L.vsseg_dice_pred_bwd.argtypes = [vp, i32, vp, i32, i64, i32, vp, vp, vp, vp]
        L.vsseg_dice_pred_bwd_to.argtypes = [vp, i32, vp, i32, i64, i32, vp, vp, Tensor, vp]
        L.vsseg_dice_att_bwd.argtypes = [vp, i32, i64, vp, f32, vp, vp, vp]
        L.vsseg_dice_level_sums.argtypes = [vp, vp, vp, i32, I3, i32, vp, vp, vp, vp]
        L.vsseg_dice_tail_sums.argtypes = [C.POINTER(DiceTailDesc), vp]
        L.vsseg_dice_att_bwd_levels.argtypes = [C.POINTER(DiceBwdLevelsDesc), vp]
        L.vsseg_fork_event_create.argtypes, L.vsseg_fork_event_create.restype = [], vp
        L.vsseg_fork_event_destroy.argtypes = [vp]
        L.vsseg_fork_arm.argtypes = [vp]
        L.vsseg_fork_disarm.argtypes = []
        L.vsseg_stream_wait_event.argtypes = [vp, vp]
        L.vsseg_adam.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, f32, f32, f32, vp]
        L.vsseg_swi_accumulate.argtypes = [vp, vp, I3, I3, i32, vp, vp, I3, vp]
        L.vsseg_swi_finalize.argtypes = [vp, vp, I3, I3, I3, i32, vp, vp]
        L.vsseg_hard_dice_counts.argtypes = [vp, i32, vp, i64, vp, vp]
        L.vsseg_argmax2.argtypes = [vp, i32, i64, vp, vp]
        _lib = L
    return _lib


class VssegError(RuntimeError):
    """A library call returned a VSSEG_E* code: `rc` is the code (EINVAL = the library rejected the arguments, ELAUNCH = HIP refused the launch),
    `detail` the library's message (for ELAUNCH it ends with HIP's own error string)."""

    def __init__(self, rc: int, what: str, detail: str):
        super().__init__(f"vsseg {what} failed ({rc}): {detail}")
        self.rc, self.what, self.detail = rc, what, detail

    @property
    def during_capture(self) -> bool:
        """HIP refused the call only because the stream is being captured into a graph (an allocation, attribute change or synchronisation inside
        the captured region): the same launch list is fine when launched eagerly."""
        return self.rc == ELAUNCH and "capture" in self.detail.lower()


def check(rc: int, what: str = ""):
    if rc != 0:
        raise VssegError(rc, what, lib().vsseg_last_error().decode())


def fx_status(reset: bool = False) -> bool:
    """True if a fixed-point partial sum (BatchNorm statistics / backward sums, Dice sums) was non-finite or out of range since the last reset
    (include/vsseg_hip.h: the kernels that decode those sums return NaN while the flag is set).  Synchronises the current stream."""
    import torch

    rc = lib().vsseg_fx_status(1 if reset else 0, torch.cuda.current_stream().cuda_stream)
    if rc < 0:
        check(rc, "fx_status")
    return bool(rc)


def i3(v):
    return (C.c_int32 * 3)(*[int(a) for a in v])
