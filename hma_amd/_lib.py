"""ctypes binding of libhma_hip.so (include/hma_hip.h).  There is NO fallback: if the library is
missing or a launch fails, the product path raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhma_hip.so")

A_BF16, A_F32, A_BF16_AFFINE, A_BF16_FRAG32, A_BF16_HEADBLK = 0, 1, 2, 3, 4
EPI_BF16, EPI_F32, EPI_RESID, EPI_GELU2, EPI_SILU2, EPI_DGELU, EPI_DSILU, EPI_ATOMIC_F32 = range(8)

c_i32, c_i64, c_f32, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class GemmNT(C.Structure):
    _fields_ = [
        ("A", c_vp), ("lda", c_i64), ("a_kind", c_i32), ("_pad0", c_i32),
        ("a_group_rows", c_i64), ("a_group_stride", c_i64),
        ("gamma", c_vp), ("beta", c_vp),
        ("W", c_vp), ("ldw", c_i64),
        ("M", c_i64), ("N", c_i64), ("K", c_i64),
        ("epi", c_i32), ("_pad1", c_i32),
        ("bias", c_vp),
        ("C", c_vp), ("ldc", c_i64), ("c_group_rows", c_i64), ("c_group_stride", c_i64),
        ("C2", c_vp), ("ldc2", c_i64),
        ("U", c_vp), ("ldu", c_i64),
        ("batch", c_i32), ("_pad2", c_i32),
        ("sA", c_i64), ("sW", c_i64), ("sBias", c_i64), ("sC", c_i64), ("sC2", c_i64), ("sU", c_i64),
        ("ln_xhat", c_vp), ("ln_rstd", c_vp), ("ln_ss", c_vp), ("ln_xm", c_vp),
        ("ln_eps", C.c_float), ("ln_rows_per_frame", c_i32),
        ("drop_p", C.c_float), ("drop_salt", c_i32), ("drop_seed", c_vp),
    ]


class GemmTN(C.Structure):
    _fields_ = [
        ("dY", c_vp), ("ldy", c_i64), ("y_kind", c_i32), ("_pad0", c_i32),
        ("y_group_rows", c_i64), ("y_group_stride", c_i64),
        ("A", c_vp), ("lda", c_i64), ("a_kind", c_i32), ("_pad1", c_i32),
        ("a_group_rows", c_i64), ("a_group_stride", c_i64),
        ("gamma", c_vp), ("beta", c_vp),
        ("M", c_i64), ("N", c_i64), ("K", c_i64),
        ("dW", c_vp), ("lddw", c_i64), ("dBias", c_vp),
        ("splits", c_i32), ("batch", c_i32),
        ("sY", c_i64), ("sA", c_i64), ("sdW", c_i64), ("sdBias", c_i64),
        ("ws", c_vp), ("ws_elems", c_i64),
        ("w_master", c_vp), ("dgamma", c_vp), ("dbeta", c_vp),
    ]


class MlpFwd(C.Structure):
    _fields_ = [
        ("xhat", c_vp), ("x", c_vp),
        ("w1p", c_vp), ("w2p", c_vp), ("b1", c_vp), ("b2", c_vp),
        ("ln_xhat", c_vp), ("ln_rstd", c_vp), ("ln_eps", C.c_float), ("_pad", c_i32),
        ("M", c_i64),
    ]


class MlpBwd(C.Structure):
    _fields_ = [
        ("xhat", c_vp), ("rstd", c_vp), ("dy", c_vp),
        ("dx", c_vp), ("dx_bf16", c_vp),
        ("w1p", c_vp), ("w2tp", c_vp), ("w1tp", c_vp), ("b1", c_vp),
        ("hg", c_vp), ("du", c_vp),
        ("M", c_i64),
        ("drop_p", C.c_float), ("drop_salt", c_i32), ("drop_seed", c_vp), ("dy_drop", c_vp),
    ]


class ChainWeights(C.Structure):
    _fields_ = [("seg", c_vp * 4), ("bundles", c_i32 * 4)]


class ChainAFwd(C.Structure):
    _fields_ = [
        ("w", ChainWeights),
        ("o", c_vp), ("x", c_vp), ("ss", c_vp),
        ("b_proj", c_vp), ("b_lin", c_vp), ("b_qkv", c_vp),
        ("xhat", c_vp), ("xm", c_vp), ("rstd", c_vp), ("x_bf16", c_vp),
        ("qkv", c_vp), ("ldq", c_i64), ("q_group_rows", c_i64), ("q_group_stride", c_i64),
        ("M", c_i64), ("rows_per_frame", c_i32), ("use_mod", c_i32),
    ]


class ChainABwd(C.Structure):
    _fields_ = [
        ("w", ChainWeights),
        ("dqkv", c_vp), ("ldq", c_i64), ("dx", c_vp),
        ("xhat", c_vp), ("rstd", c_vp), ("ss", c_vp),
        ("dx2_bf16", c_vp), ("dx1_bf16", c_vp), ("d_o", c_vp), ("dss", c_vp),
        ("M", c_i64), ("rows_per_frame", c_i32), ("use_mod", c_i32),
    ]


class ChainBFwd(C.Structure):
    _fields_ = [
        ("w", ChainWeights),
        ("o", c_vp), ("x", c_vp),
        ("b_proj", c_vp), ("b1", c_vp), ("b2", c_vp), ("b_qkv", c_vp),
        ("qkv", c_vp), ("ldq", c_i64),
        ("M", c_i64), ("ln_eps", C.c_float), ("_pad", c_i32),
        ("xhat2", c_vp), ("rstd2", c_vp), ("xhat1n", c_vp), ("rstd1n", c_vp),
        ("drop_p", C.c_float), ("drop_salt", c_i32), ("drop_seed", c_vp),
    ]


class ChainSBwd(C.Structure):
    _fields_ = [
        ("w", ChainWeights),
        ("dqkv", c_vp), ("ldq", c_i64), ("dx", c_vp),
        ("xhat", c_vp), ("rstd", c_vp),
        ("dx_bf16", c_vp),
        ("M", c_i64),
        ("hb_rows", c_i64),
    ]


class ChainTBwd(C.Structure):
    _fields_ = [
        ("w", ChainWeights),
        ("dy_bf16", c_vp), ("qkv", c_vp), ("dqkv", c_vp),
        ("B", c_i64), ("T", c_i32), ("SA", c_i32),
        ("attn_scale", C.c_float), ("_pad", c_i32),
    ]


class PackJob(C.Structure):
    """hma_pack_job_t: the arguments of one hma_chain_pack / hma_mlp_pack / hma_transpose_cast_bf16 call."""
    _fields_ = [
        ("src", c_vp), ("row_stride", c_i64), ("col_stride", c_i64), ("row_scale", c_vp), ("col_scale", c_vp), ("dst", c_vp),
        ("kind", c_i32), ("rows", c_i32), ("cols", c_i32), ("batch", c_i32),
        ("src_batch_stride", c_i64), ("dst_batch_stride", c_i64), ("bundle_stride", c_i32), ("reserved", c_i32),
    ]


def pack_jobs(jobs):
    """A ctypes array of PackJob from keyword dictionaries."""
    arr = (PackJob * len(jobs))()
    for a, j in zip(arr, jobs):
        for k, v in j.items():
            setattr(a, k, v)
    return arr


class ChainABFwd(C.Structure):
    _fields_ = [
        ("seg", c_vp * 6), ("bundles", c_i32 * 6),
        ("o_s", c_vp), ("x", c_vp), ("ss", c_vp),
        ("b_proj_s", c_vp), ("b_lin", c_vp), ("b_qkv_t", c_vp), ("b_proj_t", c_vp), ("b1", c_vp), ("b2", c_vp), ("b_qkv_s", c_vp),
        ("xhat_m", c_vp), ("xm", c_vp), ("rstd_m", c_vp), ("x2b", c_vp),
        ("qkv_t", c_vp), ("o_t", c_vp),
        ("xhat2", c_vp), ("rstd2", c_vp), ("xhat1n", c_vp), ("rstd1n", c_vp), ("qkv_s", c_vp),
        ("B", c_i64), ("T", c_i32), ("SA", c_i32),
        ("attn_scale", C.c_float), ("ln_eps", C.c_float),
        ("drop_p", C.c_float), ("drop_salt", c_i32), ("drop_seed", c_vp),
    ]


class ReadoutCE(C.Structure):
    _fields_ = [
        ("w", ChainWeights),
        ("x", c_vp), ("bias", c_vp),
        ("input_ids", c_vp), ("labels", c_vp),
        ("stats", c_vp), ("dlogits", c_vp), ("grad_scale_dev", c_vp),
        ("rows", c_i64), ("mask_id", c_i64),
        ("S", c_i32), ("SA", c_i32), ("T", c_i32), ("grad_scale", C.c_float), ("label_smoothing", C.c_float), ("_pad", c_i32),
    ]


_PROTOS = {
    "hma_gemm_nt": [c_vp, C.POINTER(GemmNT)],
    "hma_gemm_tn": [c_vp, C.POINTER(GemmTN)],
    "hma_gemm_tn_pair": [c_vp, C.POINTER(GemmTN), C.POINTER(GemmTN)],
    "hma_gemm_tn_multi": [c_vp, C.POINTER(C.POINTER(GemmTN)), c_i32],
    "hma_ln_fwd": [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32],
    "hma_ln_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp],
    "hma_qknorm_fwd": [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_f32, c_i64, c_i64, c_i64],
    "hma_qknorm_bwd": [c_vp, c_vp, c_i64, c_vp, c_vp, c_f32, c_vp, c_vp, c_i64],
    "hma_modln_fwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_f32],
    "hma_modln_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp],
    "hma_attn_spatial_fwd": [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_f32],
    "hma_attn_spatial_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_f32],
    "hma_attn_spatial_bwd_blocked": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_f32],
    "hma_attn_temporal_fwd": [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_f32],
    "hma_attn_temporal_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_f32],
    "hma_embed_fwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64],
    "hma_embed_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64],
    "hma_action_stem_fwd": [c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                            c_vp, c_i64, c_i32, c_i32],
    "hma_action_stem_bwd": [c_vp] * 15 + [c_i64, c_i32],
    "hma_count_masked": [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i64],
    "hma_ce_fwd_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_i64, c_i32, c_i32, c_i64, c_f32],
    "hma_maskgit_step": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_i32],
    "hma_maskgit_step_wide": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_i32],
    "hma_maskgit_step_sampled": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32,
                                 c_i32],
    "hma_attn_temporal_cached": [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_f32],
    "hma_sqnorm": [c_vp, c_vp, c_i64, c_vp],
    "hma_adamw": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_i32, c_vp, c_f32, c_vp],
    "hma_adamw_counted": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_vp, c_i32, c_vp, c_f32,
                          c_vp],
    "hma_cast_bf16": [c_vp, c_vp, c_vp, c_i64],
    "hma_diff_prepare": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32],
    "hma_silu_cast": [c_vp, c_vp, c_vp, c_i64],
    "hma_silu_bwd": [c_vp, c_vp, c_vp, c_vp, c_i64],
    "hma_adaln_fwd": [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_f32, c_vp, c_i64, c_i32],
    "hma_adaln_bwd": [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32],
    "hma_adaln_bwd_acc": [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32],
    "hma_gate_fwd": [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_i64, c_i32],
    "hma_gate_bwd": [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_i64, c_i32],
    "hma_diff_loss": [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_i64, c_i32],
    "hma_diff_p_sample": [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_i32, c_i64, c_i32],
    "hma_diff_p_sample_cfg": [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_i32, c_i64, c_i32, c_f32],
    "hma_mar_patchify": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32],
    "hma_mar_mask_token_bwd": [c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32],
    "hma_mar_embed_fwd": [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32],
    "hma_mar_embed_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32],
    "hma_mar_readout_fwd": [c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32],
    "hma_mar_readout_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32],
    "hma_maskgit_collate": [c_vp, c_vp, c_vp, c_vp, C.c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp],
    "hma_dropout_bf16": [c_vp, c_vp, c_vp, c_i64, c_i32, c_f32, c_vp, c_i32],
    "hma_transpose_cast_bf16": [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i64, c_i64],
    "hma_fold_ln_bf16": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i64, c_i64, c_i64],
    "hma_mlp_pack": [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_i64],
    "hma_mlp_fwd": [c_vp, C.POINTER(MlpFwd)],
    "hma_mlp_bwd": [c_vp, C.POINTER(MlpBwd)],
    "hma_chain_pack": [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_i32],
    "hma_chain_pack_multi": [c_vp, C.POINTER(PackJob), c_i32],
    "hma_mlp_pack_multi": [c_vp, C.POINTER(PackJob), c_i32],
    "hma_transpose_cast_bf16_multi": [c_vp, C.POINTER(PackJob), c_i32],
    "hma_readout_ce": [c_vp, C.POINTER(ReadoutCE)],
    "hma_chain_b_fwd": [c_vp, C.POINTER(ChainBFwd)],
    "hma_chain_a_fwd": [c_vp, C.POINTER(ChainAFwd)],
    "hma_chain_a_bwd": [c_vp, C.POINTER(ChainABwd)],
    "hma_chain_s_bwd": [c_vp, C.POINTER(ChainSBwd)],
    "hma_chain_t_bwd": [c_vp, C.POINTER(ChainTBwd)],
    "hma_chain_ab_fwd": [c_vp, C.POINTER(ChainABFwd)],
    "hma_zero_f32": [c_vp, c_vp, c_i64],
    "hma_abi_version": [],
}

EXPORTS = tuple(_PROTOS)
_lib = None


class HmaKernelError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the in-tree shared object; raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        # torch must be imported first: it brings its own libamdhip64.so.7, and the library has to bind to
        # THAT runtime (a second copy from /opt/rocm would see no device inside this process).
        import torch  # noqa: F401

        if not os.path.exists(LIB_PATH):
            raise HmaKernelError(
                f"{LIB_PATH} is missing: build it with `python -m hma_amd.build` (hipcc --offload-arch=gfx950). "
                "hma_amd has no CPU or eager fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, argtypes in _PROTOS.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = c_i32
        if lib.hma_abi_version() != 0x484D4104:
            raise HmaKernelError("libhma_hip.so ABI mismatch: rebuild with `python -m hma_amd.build --force`")
        _lib = lib
    return _lib


def check(rc: int, name: str) -> None:
    if rc != 0:
        raise HmaKernelError(f"{name} failed with code {rc}")


def call(name: str, *args) -> None:
    check(getattr(load(), name)(*args), name)
