"""Tensor-level wrappers over the C ABI (include/hma_hip.h).

PyTorch here is plumbing only: device memory, streams.  Every function enqueues HIP kernels on
the current stream through libhma_hip.so and raises if the library is missing -- there is no
eager / CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import (A_BF16, A_BF16_AFFINE, A_F32, EPI_ATOMIC_F32, EPI_BF16, EPI_DGELU, EPI_DSILU, EPI_F32, EPI_GELU2,
                   EPI_RESID, EPI_SILU2, ChainABFwd, ChainABwd, ChainAFwd, ChainBFwd, ChainSBwd, ChainTBwd, GemmNT, GemmTN, MlpBwd, MlpFwd, ReadoutCE)

BF16 = torch.bfloat16
F32 = torch.float32


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _kind(t: torch.Tensor, affine: bool = False) -> int:
    if t.dtype == F32:
        assert not affine
        return A_F32
    assert t.dtype == BF16, t.dtype
    return A_BF16_AFFINE if affine else A_BF16


def make_gemm_nt(*, A: int, lda: int, a_kind: int, W: int, ldw: int, M: int, N: int, K: int, epi: int, Cp: int, ldc: int,
                 bias: Optional[int] = None, gamma: Optional[int] = None, beta: Optional[int] = None,
                 a_group=(0, 0), c_group=(0, 0), C2: Optional[int] = None, ldc2: int = 0, U: Optional[int] = None,
                 ldu: int = 0, batch: int = 1, sA: int = 0, sW: int = 0, sBias: int = 0, sC: int = 0, sC2: int = 0,
                 sU: int = 0, ln_xhat: Optional[int] = None, ln_rstd: Optional[int] = None, ln_eps: float = 0.0,
                 ln_ss: Optional[int] = None, ln_xm: Optional[int] = None, ln_rows_per_frame: int = 0, drop_p: float = 0.0,
                 drop_salt: int = 0, drop_seed: Optional[int] = None) -> GemmNT:
    g = GemmNT()
    g.A, g.lda, g.a_kind = A, lda, a_kind
    g.a_group_rows, g.a_group_stride = a_group
    g.gamma, g.beta = gamma, beta
    g.W, g.ldw = W, ldw
    g.M, g.N, g.K = M, N, K
    g.epi, g.bias = epi, bias
    g.C, g.ldc = Cp, ldc
    g.c_group_rows, g.c_group_stride = c_group
    g.C2, g.ldc2, g.U, g.ldu = C2, ldc2, U, ldu
    g.batch, g.sA, g.sW, g.sBias, g.sC, g.sC2, g.sU = batch, sA, sW, sBias, sC, sC2, sU
    g.ln_xhat, g.ln_rstd, g.ln_ss, g.ln_xm = ln_xhat, ln_rstd, ln_ss, ln_xm
    g.ln_eps, g.ln_rows_per_frame = ln_eps, ln_rows_per_frame
    g.drop_p, g.drop_salt, g.drop_seed = drop_p, drop_salt, drop_seed
    return g


def make_gemm_tn(*, dY: int, ldy: int, y_kind: int, A: int, lda: int, a_kind: int, M: int, N: int, K: int, dW: int,
                 lddw: int, dBias: Optional[int] = None, gamma: Optional[int] = None, beta: Optional[int] = None,
                 y_group=(0, 0), a_group=(0, 0), splits: int = 0, batch: int = 1, sY: int = 0, sA: int = 0, sdW: int = 0,
                 sdBias: int = 0, ws: Optional[int] = None, ws_elems: int = 0, w_master: Optional[int] = None,
                 dgamma: Optional[int] = None, dbeta: Optional[int] = None) -> GemmTN:
    g = GemmTN()
    g.dY, g.ldy, g.y_kind = dY, ldy, y_kind
    g.y_group_rows, g.y_group_stride = y_group
    g.A, g.lda, g.a_kind = A, lda, a_kind
    g.a_group_rows, g.a_group_stride = a_group
    g.gamma, g.beta = gamma, beta
    g.M, g.N, g.K = M, N, K
    g.dW, g.lddw, g.dBias = dW, lddw, dBias
    if splits <= 0:
        tiles = (N // 128) * (K // 128) * max(batch, 1)
        splits = max(1, min((M + 63) // 64, 1024 // max(tiles, 1)))
    g.splits, g.batch = splits, batch
    g.sY, g.sA, g.sdW, g.sdBias = sY, sA, sdW, sdBias
    g.ws, g.ws_elems = ws, ws_elems
    g.w_master, g.dgamma, g.dbeta = w_master, dgamma, dbeta
    return g


def make_mlp_fwd(*, M: int, xhat: int, x: int, w1p: int, w2p: int, b1: int, b2: Optional[int] = None,
                 ln_xhat: Optional[int] = None, ln_rstd: Optional[int] = None, ln_eps: float = 1e-5) -> MlpFwd:
    g = MlpFwd()
    g.xhat, g.x, g.w1p, g.w2p, g.b1, g.b2 = xhat, x, w1p, w2p, b1, b2
    g.ln_xhat, g.ln_rstd, g.ln_eps, g.M = ln_xhat, ln_rstd, ln_eps, M
    return g


def make_mlp_bwd(*, M: int, xhat: int, rstd: int, dy: int, dx: int, dx_bf16: int, w1p: int, w2tp: int, w1tp: int, b1: int,
                 hg: int, du: int, drop_p: float = 0.0, drop_salt: int = 0, drop_seed: Optional[int] = None,
                 dy_drop: Optional[int] = None) -> MlpBwd:
    g = MlpBwd()
    g.xhat, g.rstd, g.dy, g.dx, g.dx_bf16 = xhat, rstd, dy, dx, dx_bf16
    g.w1p, g.w2tp, g.w1tp, g.b1, g.hg, g.du, g.M = w1p, w2tp, w1tp, b1, hg, du, M
    g.drop_p, g.drop_salt, g.drop_seed, g.dy_drop = drop_p, drop_salt, drop_seed, dy_drop
    return g


def chain_pack(src: torch.Tensor, *, kind: int, rows: int, cols: int, row_stride: int, col_stride: int,
               row_scale: Optional[torch.Tensor] = None, col_scale: Optional[torch.Tensor] = None,
               out: Optional[torch.Tensor] = None, bundle_stride: int = 1, bundle_offset: int = 0) -> torch.Tensor:
    """Bundles of one logical [rows][cols] matrix (hma_chain_pack) in the chains' streaming order; bf16, 8192 elements per bundle.
    Bundle b lands at bundle slot bundle_offset + b * bundle_stride of `out`."""
    nb = (rows if kind == 0 else cols) // 32
    if out is None:
        out = torch.empty((bundle_offset + (nb - 1) * bundle_stride + 1) * 8192, dtype=BF16, device=src.device)
    _lib.call("hma_chain_pack", stream_ptr(), ptr(src), row_stride, col_stride, ptr(row_scale), ptr(col_scale),
              out.data_ptr() + 2 * 8192 * bundle_offset, kind, rows, cols, 1, 0, 0, bundle_stride)
    return out


def _chain_weights(w, segs) -> None:
    for i in range(4):
        w.seg[i] = segs[i][0] if i < len(segs) else None
        w.bundles[i] = segs[i][1] if i < len(segs) else 0


def make_chain_a_fwd(*, M: int, segs, o: int, x: int, qkv: int, ldq: int = 768, ss: Optional[int] = None,
                     b_proj: Optional[int] = None, b_lin: Optional[int] = None, b_qkv: Optional[int] = None,
                     xhat: Optional[int] = None, xm: Optional[int] = None, rstd: Optional[int] = None,
                     x_bf16: Optional[int] = None, q_group=(0, 0), rows_per_frame: int = 0, use_mod: bool = True) -> ChainAFwd:
    """segs: [(device pointer, bundles)] of the packed proj (8), linear_out (8, use_mod only) and qkv (24) weights."""
    g = ChainAFwd()
    _chain_weights(g.w, segs)
    g.o, g.x, g.ss = o, x, ss
    g.b_proj, g.b_lin, g.b_qkv = b_proj, b_lin, b_qkv
    g.xhat, g.xm, g.rstd, g.x_bf16 = xhat, xm, rstd, x_bf16
    g.qkv, g.ldq = qkv, ldq
    g.q_group_rows, g.q_group_stride = q_group
    g.M, g.rows_per_frame, g.use_mod = M, rows_per_frame, 1 if use_mod else 0
    return g


def make_chain_b_fwd(*, M: int, segs, o: int, x: int, b1: int, b_proj: Optional[int] = None, b2: Optional[int] = None,
                     b_qkv: Optional[int] = None, qkv: Optional[int] = None, ldq: int = 768, ln_eps: float = 1e-5,
                     xhat2: Optional[int] = None, rstd2: Optional[int] = None, xhat1n: Optional[int] = None,
                     rstd1n: Optional[int] = None, drop_p: float = 0.0, drop_salt: int = 0,
                     drop_seed: Optional[int] = None) -> ChainBFwd:
    """segs: packed proj (8), the 64 alternating fc1 / fc2 bundles, and -- with `qkv` -- the next block's folded qkv (24)."""
    g = ChainBFwd()
    _chain_weights(g.w, segs)
    g.o, g.x = o, x
    g.b_proj, g.b1, g.b2, g.b_qkv = b_proj, b1, b2, b_qkv
    g.qkv, g.ldq, g.M, g.ln_eps = qkv, ldq, M, ln_eps
    g.xhat2, g.rstd2, g.xhat1n, g.rstd1n = xhat2, rstd2, xhat1n, rstd1n
    g.drop_p, g.drop_salt, g.drop_seed = drop_p, drop_salt, drop_seed
    return g


def make_readout_ce(*, rows: int, segs, x: int, bias: Optional[int], input_ids: int, labels: int, stats: int, dlogits: Optional[int],
                    grad_scale_dev: Optional[int], grad_scale: float, S: int, SA: int, T: int, mask_id: int,
                    label_smoothing: float) -> ReadoutCE:
    """segs: the 32 N-block bundles of out_x_proj.weight (chain_pack kind 0)."""
    g = ReadoutCE()
    _chain_weights(g.w, segs)
    g.x, g.bias, g.input_ids, g.labels, g.stats, g.dlogits, g.grad_scale_dev = x, bias, input_ids, labels, stats, dlogits, grad_scale_dev
    g.rows, g.mask_id, g.S, g.SA, g.T, g.grad_scale, g.label_smoothing = rows, mask_id, S, SA, T, grad_scale, label_smoothing
    return g


def make_chain_a_bwd(*, M: int, segs, dqkv: int, dx: int, dx1_bf16: int, d_o: int, ldq: int = 768, xhat: Optional[int] = None,
                     rstd: Optional[int] = None, ss: Optional[int] = None, dx2_bf16: Optional[int] = None,
                     dss: Optional[int] = None, rows_per_frame: int = 0, use_mod: bool = True) -> ChainABwd:
    """segs: packed qkv^T (3 x 8), linear_out^T (8, use_mod only), proj^T (8)."""
    g = ChainABwd()
    _chain_weights(g.w, segs)
    g.dqkv, g.ldq, g.dx = dqkv, ldq, dx
    g.xhat, g.rstd, g.ss = xhat, rstd, ss
    g.dx2_bf16, g.dx1_bf16, g.d_o, g.dss = dx2_bf16, dx1_bf16, d_o, dss
    g.M, g.rows_per_frame, g.use_mod = M, rows_per_frame, 1 if use_mod else 0
    return g


def make_chain_ab_fwd(*, B: int, SA: int, segs, o_s: int, x: int, b1: int, x2b: int, qkv_t: int, o_t: int, xhat2: int, rstd2: int,
                      attn_scale: float, ss: Optional[int] = None, xhat_m: Optional[int] = None, xm: Optional[int] = None,
                      rstd_m: Optional[int] = None, b_proj_s: Optional[int] = None, b_lin: Optional[int] = None,
                      b_qkv_t: Optional[int] = None, b_proj_t: Optional[int] = None, b2: Optional[int] = None, b_qkv_s: Optional[int] = None,
                      xhat1n: Optional[int] = None, rstd1n: Optional[int] = None, qkv_s: Optional[int] = None, ln_eps: float = 1e-5,
                      T: int = 16, drop_p: float = 0.0, drop_salt: int = 0, drop_seed: Optional[int] = None) -> ChainABFwd:
    """segs: [(pointer, bundles)] x 6 = proj_s (8), linear_out (8; (None, 0) for blocks without action tokens: then ss / xhat_m / xm /
    rstd_m are not used), temporal qkv (24), proj_t (8), the 64 fc1 / fc2 bundles, the next block's folded spatial qkv (24; (None, 0)
    for the last block).  drop_p > 0: the two nn.Dropout sites of the MLP (salts drop_salt, drop_salt + 1)."""
    g = ChainABFwd()
    for i in range(6):
        g.seg[i] = segs[i][0] if i < len(segs) else None
        g.bundles[i] = segs[i][1] if i < len(segs) else 0
    g.o_s, g.x, g.ss = o_s, x, ss
    g.b_proj_s, g.b_lin, g.b_qkv_t, g.b_proj_t, g.b1, g.b2, g.b_qkv_s = b_proj_s, b_lin, b_qkv_t, b_proj_t, b1, b2, b_qkv_s
    g.xhat_m, g.xm, g.rstd_m, g.x2b, g.qkv_t, g.o_t = xhat_m, xm, rstd_m, x2b, qkv_t, o_t
    g.xhat2, g.rstd2, g.xhat1n, g.rstd1n, g.qkv_s = xhat2, rstd2, xhat1n, rstd1n, qkv_s
    g.B, g.T, g.SA, g.attn_scale, g.ln_eps = B, T, SA, attn_scale, ln_eps
    g.drop_p, g.drop_salt, g.drop_seed = drop_p, drop_salt, drop_seed
    return g


def make_chain_s_bwd(*, M: int, segs, dqkv: int, dx: int, xhat: int, rstd: int, dx_bf16: int, ldq: int = 768, hb_rows: int = 0) -> ChainSBwd:
    """segs: packed qkv^T (3 x 8) of the spatial attention with norm1's gamma folded into the output rows.  hb_rows > 0: dqkv in the
    head-blocked order of hma_attn_spatial_bwd_blocked (rows per frame)."""
    g = ChainSBwd()
    _chain_weights(g.w, segs)
    g.dqkv, g.ldq, g.dx = dqkv, ldq, dx
    g.xhat, g.rstd, g.dx_bf16 = xhat, rstd, dx_bf16
    g.M, g.hb_rows = M, hb_rows
    return g


def make_chain_t_bwd(*, B: int, SA: int, segs, dy_bf16: int, qkv: int, dqkv: int, attn_scale: float, T: int = 16) -> ChainTBwd:
    """segs: the packed transposed temporal projection (8 bundles).  Rows (b, t, s), M = T B SA, 1 <= T <= 16."""
    g = ChainTBwd()
    _chain_weights(g.w, segs)
    g.dy_bf16, g.qkv, g.dqkv = dy_bf16, qkv, dqkv
    g.B, g.T, g.SA, g.attn_scale = B, T, SA, attn_scale
    return g


# ----------------------------------------------------------------------------------------------
# tensor conveniences (used by the module-level API and the parity tests)
# ----------------------------------------------------------------------------------------------
def linear(x: torch.Tensor, w_bf16: torch.Tensor, bias: Optional[torch.Tensor] = None, *, epi: int = EPI_F32,
           gamma=None, beta=None, out: Optional[torch.Tensor] = None, out2: Optional[torch.Tensor] = None,
           aux: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x' @ w^T (+ bias) with x (M, K) bf16/f32 and w (N, K) bf16; epilogue per `epi`."""
    assert x.dim() == 2 and w_bf16.dim() == 2 and w_bf16.dtype == BF16 and x.is_contiguous() and w_bf16.is_contiguous()
    M, K = x.shape
    N = w_bf16.shape[0]
    if out is None:
        dt = F32 if epi in (EPI_F32, EPI_RESID, EPI_ATOMIC_F32) else BF16
        out = torch.empty(M, N, dtype=dt, device=x.device)
    g = make_gemm_nt(A=ptr(x), lda=K, a_kind=_kind(x, gamma is not None), W=ptr(w_bf16), ldw=K, M=M, N=N, K=K, epi=epi,
                     Cp=ptr(out), ldc=N, bias=ptr(bias), gamma=ptr(gamma), beta=ptr(beta), C2=ptr(out2), ldc2=N,
                     U=ptr(aux), ldu=N)
    _lib.call("hma_gemm_nt", stream_ptr(), C.byref(g))
    return out


def linear_wgrad(dy: torch.Tensor, x: torch.Tensor, dW: torch.Tensor, dbias: Optional[torch.Tensor] = None, *,
                 gamma=None, beta=None, splits: int = 0, ws: Optional[torch.Tensor] = None) -> None:
    """dW (N, K) fp32 += dy^T x' ; dbias += column sums of dy."""
    M, N = dy.shape
    K = x.shape[1]
    g = make_gemm_tn(dY=ptr(dy), ldy=N, y_kind=_kind(dy), A=ptr(x), lda=K, a_kind=_kind(x, gamma is not None), M=M, N=N,
                     K=K, dW=ptr(dW), lddw=K, dBias=ptr(dbias), gamma=ptr(gamma), beta=ptr(beta), splits=splits,
                     ws=ptr(ws), ws_elems=0 if ws is None else ws.numel())
    _lib.call("hma_gemm_tn", stream_ptr(), C.byref(g))


def ln_fwd(x: torch.Tensor, eps: float):
    rows = x.numel() // 256
    xhat = torch.empty(x.shape, dtype=BF16, device=x.device)
    rstd = torch.empty(rows, dtype=F32, device=x.device)
    _lib.call("hma_ln_fwd", stream_ptr(), ptr(x), ptr(xhat), ptr(rstd), rows, eps)
    return xhat, rstd


def qknorm_fwd(qkv: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5, save_raw: bool = True):
    """qk_norm=True (hma/model/attention.py:31-35,44-48): per-head LayerNorm of the q | k parts of a packed (rows, 768) bf16 qkv.
    Returns (normalised copy, pre-norm q | k (rows, 512) for the backward or None)."""
    out = qkv.clone()
    raw = torch.empty(qkv.shape[0], 512, dtype=BF16, device=qkv.device) if save_raw else None
    _lib.call("hma_qknorm_fwd", stream_ptr(), ptr(out), 768, ptr(raw), ptr(gamma), ptr(beta), eps, qkv.shape[0], 0, 0)
    return out, raw


def qknorm_bwd(dqkv: torch.Tensor, raw: torch.Tensor, gamma: torch.Tensor, eps: float = 1e-5):
    """(gradient wrt the raw qkv, dgamma [32], dbeta [32]) from the gradient wrt the normalised one."""
    out = dqkv.clone()
    dg = torch.zeros(32, dtype=F32, device=dqkv.device)
    db = torch.zeros(32, dtype=F32, device=dqkv.device)
    _lib.call("hma_qknorm_bwd", stream_ptr(), ptr(out), 768, ptr(raw), ptr(gamma), eps, ptr(dg), ptr(db), dqkv.shape[0])
    return out, dg, db


def attn_spatial_fwd(qkv: torch.Tensor, frames: int, n: int, scale: float):
    o = torch.empty(frames * n, 256, dtype=BF16, device=qkv.device)
    lse = torch.empty(frames * n, 8, dtype=F32, device=qkv.device)
    _lib.call("hma_attn_spatial_fwd", stream_ptr(), ptr(qkv), ptr(o), ptr(lse), frames, n, scale)
    return o, lse


def attn_spatial_bwd(qkv, o, d_o, lse, frames: int, n: int, scale: float, blocked: bool = False):
    """dqkv [frames * n, 768] row-major; `blocked`: in the head-blocked order (`headblk_to_rows` turns it back)."""
    delta = torch.empty_like(lse)
    dqkv = torch.empty_like(qkv)
    _lib.call("hma_attn_spatial_bwd_blocked" if blocked else "hma_attn_spatial_bwd", stream_ptr(), ptr(qkv), ptr(o), ptr(d_o), ptr(lse),
              ptr(delta), ptr(dqkv), frames, n, scale)
    return dqkv


def headblk_to_rows(t: torch.Tensor, n: int) -> torch.Tensor:
    """[M, 256 W] head-blocked (HMA_A_BF16_HEADBLK: [frame][head][part][n][32]) -> row-major [M, 256 W]."""
    M, C = t.shape
    W = C // 256
    return t.reshape(M // n, 8, W, n, 32).permute(0, 3, 2, 1, 4).reshape(M, C).contiguous()


def rows_to_headblk(t: torch.Tensor, n: int) -> torch.Tensor:
    M, C = t.shape
    W = C // 256
    return t.reshape(M // n, n, W, 8, 32).permute(0, 3, 2, 1, 4).reshape(M, C).contiguous()


def attn_temporal_fwd(qkv: torch.Tensor, batch: int, T: int, n_s: int, scale: float):
    o = torch.empty(batch * T * n_s, 256, dtype=BF16, device=qkv.device)
    _lib.call("hma_attn_temporal_fwd", stream_ptr(), ptr(qkv), ptr(o), batch, T, n_s, scale)
    return o


def attn_temporal_bwd(qkv, o, d_o, batch: int, T: int, n_s: int, scale: float):
    dqkv = torch.empty_like(qkv)
    _lib.call("hma_attn_temporal_bwd", stream_ptr(), ptr(qkv), ptr(o), ptr(d_o), ptr(dqkv), batch, T, n_s, scale)
    return dqkv
