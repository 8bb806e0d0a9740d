"""PyTorch custom operators (`torch.ops.hma.*`) over the C ABI of libhma_hip.so.

The operator-level drop-in of INTEGRATION.md section 3 as `torch.library` custom ops: each op allocates its outputs, passes
`tensor.data_ptr()` and the current HIP stream to the `extern "C"` entry point (include/hma_hip.h) and registers a fake
(meta) implementation, so the ops trace under FakeTensor / `torch.compile` graphs and carry autograd formulas where the
reference's modules need them (`linear`, `layer_norm`, the two attentions: hma/model/attention.py:37-61,
st_transformer.py:24-27,104-113).  There is no CPU implementation: a CPU tensor raises, like every other path of this package.

    import hma_amd.torch_ops                      # registers the ops
    y = torch.ops.hma.linear(x_bf16, w_bf16, bias_f32)
    o = torch.ops.hma.attn_spatial(qkv_bf16, frames, n, scale)
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import ops
from ._lib import EPI_BF16, EPI_DGELU, EPI_GELU2

BF16, F32 = torch.bfloat16, torch.float32


def _cuda(*ts):
    for t in ts:
        if t is not None and t.device.type != "cuda":
            raise RuntimeError("hma ops run on the HIP kernels only (got a %s tensor)" % t.device.type)


# ------------------------------------------------------------------------------------------------ linear (nn.Linear under autocast)
@torch.library.custom_op("hma::linear", mutates_args=(), device_types="cuda")
def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y (M, N) bf16 = x (M, K) bf16 @ weight (N, K) bf16 ^T + bias (N) fp32   -- hma_gemm_nt, HMA_EPI_BF16"""
    _cuda(x, weight, bias)
    return ops.linear(x.contiguous(), weight.contiguous(), bias, epi=EPI_BF16)


@linear.register_fake
def _(x, weight, bias=None):
    return x.new_empty(x.shape[0], weight.shape[0], dtype=BF16)


@torch.library.custom_op("hma::linear_wgrad", mutates_args=(), device_types="cuda")
def linear_wgrad(dy: torch.Tensor, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(dW (N, K), dbias (N)) fp32 = (dy^T x, column sums of dy)   -- hma_gemm_tn"""
    _cuda(dy, x)
    dW = torch.zeros(dy.shape[1], x.shape[1], dtype=F32, device=dy.device)
    db = torch.zeros(dy.shape[1], dtype=F32, device=dy.device)
    ops.linear_wgrad(dy.contiguous(), x.contiguous(), dW, db)
    return dW, db


@linear_wgrad.register_fake
def _(dy, x):
    return dy.new_empty(dy.shape[1], x.shape[1], dtype=F32), dy.new_empty(dy.shape[1], dtype=F32)


def _linear_setup(ctx, inputs, output):
    x, weight, bias = inputs
    ctx.save_for_backward(x, weight)
    ctx.has_bias = bias is not None


def _linear_backward(ctx, dy):
    x, weight = ctx.saved_tensors
    dyb = dy.to(BF16).contiguous()
    dx = torch.ops.hma.linear(dyb, weight.t().contiguous(), None) if ctx.needs_input_grad[0] else None
    dW = db = None
    if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
        dW, db = torch.ops.hma.linear_wgrad(dyb, x)
        dW = dW.to(weight.dtype)
    return dx, dW, (db if ctx.has_bias else None)


linear.register_autograd(_linear_backward, setup_context=_linear_setup)


# ------------------------------------------------------------------------------------------------ LayerNorm statistics (d = 256)
@torch.library.custom_op("hma::layer_norm", mutates_args=(), device_types="cuda")
def layer_norm(x: torch.Tensor, eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """(xhat bf16, 1 / sigma fp32) of the fp32 rows x (..., 256); the affine is the caller's (or folded, hma_fold_ln_bf16)"""
    _cuda(x)
    return ops.ln_fwd(x.contiguous(), eps)


@layer_norm.register_fake
def _(x, eps):
    return x.new_empty(x.shape, dtype=BF16), x.new_empty(x.numel() // 256, dtype=F32)


# ------------------------------------------------------------------------------------------------ attention
@torch.library.custom_op("hma::attn_spatial", mutates_args=(), device_types="cuda")
def attn_spatial(qkv: torch.Tensor, frames: int, n: int, scale: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """(o (frames n, 256) bf16, lse (frames n, 8)) of non-causal attention inside each frame, 8 heads of 32, qkv (frames n, 768)"""
    _cuda(qkv)
    return ops.attn_spatial_fwd(qkv.contiguous(), frames, n, scale)


@attn_spatial.register_fake
def _(qkv, frames, n, scale):
    return qkv.new_empty(frames * n, 256, dtype=BF16), qkv.new_empty(frames * n, 8, dtype=F32)


@torch.library.custom_op("hma::attn_spatial_bwd", mutates_args=(), device_types="cuda")
def attn_spatial_bwd(qkv: torch.Tensor, o: torch.Tensor, d_o: torch.Tensor, lse: torch.Tensor, frames: int, n: int,
                     scale: float) -> torch.Tensor:
    _cuda(qkv, o, d_o, lse)
    return ops.attn_spatial_bwd(qkv.contiguous(), o.contiguous(), d_o.contiguous(), lse.contiguous(), frames, n, scale)


@attn_spatial_bwd.register_fake
def _(qkv, o, d_o, lse, frames, n, scale):
    return torch.empty_like(qkv)


def _as_setup(ctx, inputs, output):
    qkv, frames, n, scale = inputs
    o, lse = output
    ctx.save_for_backward(qkv, o, lse)
    ctx.args = (frames, n, scale)


def _as_backward(ctx, d_o, d_lse):
    qkv, o, lse = ctx.saved_tensors
    return torch.ops.hma.attn_spatial_bwd(qkv, o, d_o.to(BF16).contiguous(), lse, *ctx.args), None, None, None


attn_spatial.register_autograd(_as_backward, setup_context=_as_setup)


@torch.library.custom_op("hma::attn_temporal", mutates_args=(), device_types="cuda")
def attn_temporal(qkv: torch.Tensor, batch: int, T: int, n_s: int, scale: float) -> torch.Tensor:
    """o (batch T n_s, 256) bf16 of causal attention over the T frames of each (sample, position) column, rows (b, t, s)"""
    _cuda(qkv)
    return ops.attn_temporal_fwd(qkv.contiguous(), batch, T, n_s, scale)


@attn_temporal.register_fake
def _(qkv, batch, T, n_s, scale):
    return qkv.new_empty(batch * T * n_s, 256, dtype=BF16)


@torch.library.custom_op("hma::attn_temporal_bwd", mutates_args=(), device_types="cuda")
def attn_temporal_bwd(qkv: torch.Tensor, o: torch.Tensor, d_o: torch.Tensor, batch: int, T: int, n_s: int,
                      scale: float) -> torch.Tensor:
    _cuda(qkv, o, d_o)
    return ops.attn_temporal_bwd(qkv.contiguous(), o.contiguous(), d_o.contiguous(), batch, T, n_s, scale)


@attn_temporal_bwd.register_fake
def _(qkv, o, d_o, batch, T, n_s, scale):
    return torch.empty_like(qkv)


def _at_setup(ctx, inputs, output):
    qkv, batch, T, n_s, scale = inputs
    ctx.save_for_backward(qkv, output)
    ctx.args = (batch, T, n_s, scale)


def _at_backward(ctx, d_o):
    qkv, o = ctx.saved_tensors
    return torch.ops.hma.attn_temporal_bwd(qkv, o, d_o.to(BF16).contiguous(), *ctx.args), None, None, None, None


attn_temporal.register_autograd(_at_backward, setup_context=_at_setup)



# ------------------------------------------------------------------------------------------------ qk_norm (attention.py:31-35,44-48)
@torch.library.custom_op("hma::qknorm", mutates_args=(), device_types="cuda")
def qknorm(qkv: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """(qkv with its q | k parts LayerNorm-ed per head (32 values, shared affine), the raw q | k (rows, 512) for the backward)
    -- hma_qknorm_fwd on a copy of the packed (rows, 768) bf16 buffer"""
    _cuda(qkv, gamma, beta)
    return ops.qknorm_fwd(qkv.contiguous(), gamma.float().contiguous(), beta.float().contiguous(), eps)


@qknorm.register_fake
def _(qkv, gamma, beta, eps):
    return torch.empty_like(qkv), qkv.new_empty(qkv.shape[0], 512)


@torch.library.custom_op("hma::qknorm_bwd", mutates_args=(), device_types="cuda")
def qknorm_bwd(dqkv: torch.Tensor, raw: torch.Tensor, gamma: torch.Tensor, eps: float) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    _cuda(dqkv, raw, gamma)
    return ops.qknorm_bwd(dqkv.contiguous(), raw, gamma.float().contiguous(), eps)


@qknorm_bwd.register_fake
def _(dqkv, raw, gamma, eps):
    return torch.empty_like(dqkv), gamma.new_empty(32, dtype=F32), gamma.new_empty(32, dtype=F32)


def _qkn_setup(ctx, inputs, output):
    qkv, gamma, beta, eps = inputs
    ctx.save_for_backward(output[1], gamma)
    ctx.eps = eps


def _qkn_backward(ctx, dqkv, draw_unused):
    raw, gamma = ctx.saved_tensors
    d, dg, db = torch.ops.hma.qknorm_bwd(dqkv.to(BF16).contiguous(), raw, gamma, ctx.eps)
    return d, dg.to(gamma.dtype), db.to(gamma.dtype), None


qknorm.register_autograd(_qkn_backward, setup_context=_qkn_setup)


# ------------------------------------------------------------------------------------------------ Mlp (fc1 -> GELU -> fc2)
@torch.library.custom_op("hma::mlp", mutates_args=(), device_types="cuda")
def mlp(x: torch.Tensor, w1: torch.Tensor, b1: Optional[torch.Tensor], w2: torch.Tensor,
        b2: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(y (M, d) bf16, u (M, hid) bf16, gelu(u) (M, hid) bf16) = fc2(gelu(fc1 x)) -- st_transformer.py:24-27 (dropout p = 0):
    two hma_gemm_nt launches, the first with the HMA_EPI_GELU2 epilogue; u and gelu(u) are what the backward re-reads"""
    _cuda(x, w1, b1, w2, b2)
    u = torch.empty(x.shape[0], w1.shape[0], dtype=BF16, device=x.device)
    h = torch.empty_like(u)
    ops.linear(x.contiguous(), w1.contiguous(), b1, epi=EPI_GELU2, out=u, out2=h)
    return ops.linear(h, w2.contiguous(), b2, epi=EPI_BF16), u, h


@mlp.register_fake
def _(x, w1, b1, w2, b2):
    return (x.new_empty(x.shape[0], w2.shape[0], dtype=BF16), x.new_empty(x.shape[0], w1.shape[0], dtype=BF16),
            x.new_empty(x.shape[0], w1.shape[0], dtype=BF16))


@torch.library.custom_op("hma::mlp_bwd", mutates_args=(), device_types="cuda")
def mlp_bwd(dy: torch.Tensor, x: torch.Tensor, u: torch.Tensor, h: torch.Tensor, w1: torch.Tensor,
            w2: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """(dx bf16, dW1, db1, dW2, db2 fp32): dU = (dy W2) gelu'(u) in the HMA_EPI_DGELU epilogue, then the two weight gradients
    (hma_gemm_tn) and dx = dU W1"""
    _cuda(dy, x, u, h, w1, w2)
    dy = dy.contiguous()
    dW2 = torch.zeros(w2.shape, dtype=F32, device=dy.device)
    db2 = torch.zeros(w2.shape[0], dtype=F32, device=dy.device)
    ops.linear_wgrad(dy, h, dW2, db2)
    du = ops.linear(dy, w2.t().contiguous(), None, epi=EPI_DGELU, aux=u)
    dW1 = torch.zeros(w1.shape, dtype=F32, device=dy.device)
    db1 = torch.zeros(w1.shape[0], dtype=F32, device=dy.device)
    ops.linear_wgrad(du, x.contiguous(), dW1, db1)
    dx = ops.linear(du, w1.t().contiguous(), None, epi=EPI_BF16)
    return dx, dW1, db1, dW2, db2


@mlp_bwd.register_fake
def _(dy, x, u, h, w1, w2):
    return (torch.empty_like(x), w1.new_empty(w1.shape, dtype=F32), w1.new_empty(w1.shape[0], dtype=F32),
            w2.new_empty(w2.shape, dtype=F32), w2.new_empty(w2.shape[0], dtype=F32))


def _mlp_setup(ctx, inputs, output):
    x, w1, b1, w2, b2 = inputs
    _, u, h = output
    ctx.save_for_backward(x, u, h, w1, w2)
    ctx.has_b1, ctx.has_b2 = b1 is not None, b2 is not None


def _mlp_backward(ctx, dy, du_unused, dh_unused):
    x, u, h, w1, w2 = ctx.saved_tensors
    dx, dW1, db1, dW2, db2 = torch.ops.hma.mlp_bwd(dy.to(BF16).contiguous(), x, u, h, w1, w2)
    return dx, dW1.to(w1.dtype), (db1 if ctx.has_b1 else None), dW2.to(w2.dtype), (db2 if ctx.has_b2 else None)


mlp.register_autograd(_mlp_backward, setup_context=_mlp_setup)

# ------------------------------------------------------------------------------------------------ embedding, modulation, loss, decode, update
from . import _lib  # noqa: E402
from .ops import ptr, stream_ptr  # noqa: E402


@torch.library.custom_op("hma::embed", mutates_args=(), device_types="cuda")
def embed(ids: torch.Tensor, e0: torch.Tensor, e1: torch.Tensor, mask_embed: torch.Tensor, pos: torch.Tensor,
          a_emb: Optional[torch.Tensor], action_tokens: int, mask_id: int) -> torch.Tensor:
    """x (B, T, S + A, 256) fp32 = FactorizedEmbedding(ids) (+ the frame's embedded actions on A appended tokens) + pos_embed_TSC
    -- hma_embed_fwd (factorization_utils.py:31-54, st_mask_git.py:640-672).  ids int64 (B, T, S); pos (1, T', S', 256)."""
    _cuda(ids, e0, e1, mask_embed, pos, a_emb)
    B, T, S = ids.shape
    A = action_tokens if a_emb is not None else 0
    x = torch.empty(B, T, S + A, 256, dtype=F32, device=ids.device)
    _lib.call("hma_embed_fwd", stream_ptr(), ptr(ids.contiguous()), ptr(e0.float().contiguous()), ptr(e1.float().contiguous()),
              ptr(mask_embed.float().contiguous()), ptr(pos.float().contiguous()), ptr(None if a_emb is None else a_emb.float().contiguous()),
              ptr(x), B, T, S, A, pos.shape[-2], e0.shape[0], mask_id)
    return x


@embed.register_fake
def _(ids, e0, e1, mask_embed, pos, a_emb, action_tokens, mask_id):
    B, T, S = ids.shape
    return e0.new_empty(B, T, S + (action_tokens if a_emb is not None else 0), 256, dtype=F32)


@torch.library.custom_op("hma::modulate", mutates_args=(), device_types="cuda")
def modulate(x: torch.Tensor, ss: torch.Tensor, rows_per_frame: int, eps: float) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """ModulateLayer's prologue (st_mask_git.py:71-74) on fp32 rows x (frames * rows_per_frame, 256): (xhat = LN(x) without affine,
    xm = xhat (1 + scale) + shift, 1 / sigma) with ss (frames, 512) = [shift | scale] per frame -- hma_modln_fwd"""
    _cuda(x, ss)
    rows = x.numel() // 256
    frames = rows // rows_per_frame
    xhat = torch.empty(rows, 256, dtype=BF16, device=x.device)
    xm = torch.empty_like(xhat)
    rstd = torch.empty(rows, dtype=F32, device=x.device)
    _lib.call("hma_modln_fwd", stream_ptr(), ptr(x.contiguous()), ptr(ss.float().contiguous()), ptr(xhat), ptr(xm), ptr(rstd), frames,
              rows_per_frame, eps)
    return xhat, xm, rstd


@modulate.register_fake
def _(x, ss, rows_per_frame, eps):
    rows = x.numel() // 256
    return x.new_empty(rows, 256, dtype=BF16), x.new_empty(rows, 256, dtype=BF16), x.new_empty(rows, dtype=F32)


@torch.library.custom_op("hma::readout_ce", mutates_args=(), device_types="cuda")
def readout_ce(logits: torch.Tensor, input_ids: torch.Tensor, labels: torch.Tensor, mask_id: int, label_smoothing: float,
               grad_scale: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """(stats [sum loss, sum correct, masked rows, -] fp32, dlogits bf16) of the factorised cross-entropy over the masked tokens of
    frames >= 1 (st_mask_git.py:603-630): hma_count_masked + hma_ce_fwd_bwd on fp32 logits (B * T * S, 1024); ids / labels (B, T, S)"""
    _cuda(logits, input_ids, labels)
    B, T, S = input_ids.shape
    stats = torch.zeros(8, dtype=F32, device=logits.device)  # HMA_CE_STATS_FLOATS (the kernels keep a ticket and an accumulator behind the three sums)
    dlogits = torch.empty(logits.shape, dtype=BF16, device=logits.device)
    ids = input_ids.contiguous()
    _lib.call("hma_count_masked", stream_ptr(), ptr(ids), ptr(stats), B, T, S, mask_id)
    _lib.call("hma_ce_fwd_bwd", stream_ptr(), ptr(logits.contiguous()), ptr(ids), ptr(labels.contiguous()), ptr(stats), ptr(dlogits), None,
              grad_scale, B, T, S, mask_id, label_smoothing)
    return stats[:4].clone(), dlogits


@readout_ce.register_fake
def _(logits, input_ids, labels, mask_id, label_smoothing, grad_scale):
    return logits.new_empty(4, dtype=F32), logits.new_empty(logits.shape, dtype=BF16)


@torch.library.custom_op("hma::maskgit_step", mutates_args=("prompt", "unmasked"), device_types="cuda")
def maskgit_step(logits: torch.Tensor, prompt: torch.Tensor, unmasked: torch.Tensor, out_t: int, n_mask: int, last: bool,
                 mask_id: int) -> torch.Tensor:
    """One greedy MaskGIT step (st_mask_git.py:397-453) on fp32 logits (B, T, S, 1024): writes the sampled ids into prompt[:, out_t]
    (int64 (B, T, S)), updates `unmasked` (uint8 (B, S)); returns the model confidences (B, S) -- hma_maskgit_step"""
    _cuda(logits, prompt, unmasked)
    B, T, S = prompt.shape
    conf = torch.empty(B, S, dtype=F32, device=logits.device)
    _lib.call("hma_maskgit_step", stream_ptr(), ptr(logits), ptr(prompt), ptr(unmasked), None, ptr(conf), B, T, S, out_t, n_mask, int(last),
              mask_id, 0, 0)
    return conf


@maskgit_step.register_fake
def _(logits, prompt, unmasked, out_t, n_mask, last, mask_id):
    return logits.new_empty(prompt.shape[0], prompt.shape[2], dtype=F32)


@torch.library.custom_op("hma::adamw", mutates_args=("p", "m", "v"), device_types="cuda")
def adamw(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, beta1: float, beta2: float, eps: float,
          weight_decay: float, step: int, max_norm: float) -> torch.Tensor:
    """Global-norm clip + AdamW on one flat fp32 range (train_multi.py:593-598, 900-922): hma_sqnorm + hma_adamw; returns the bf16 copy
    of the new weights.  `step` = the number of this update (1, 2, ...); max_norm <= 0: no clip."""
    _cuda(p, g, m, v)
    sq = torch.zeros(1, dtype=F32, device=p.device)
    _lib.call("hma_sqnorm", stream_ptr(), ptr(g), g.numel(), ptr(sq))
    pb = torch.empty(p.shape, dtype=BF16, device=p.device)
    _lib.call("hma_adamw", stream_ptr(), ptr(p), ptr(g), ptr(m), ptr(v), ptr(pb), p.numel(), lr, beta1, beta2, eps, weight_decay, step,
              ptr(sq), max_norm, None)
    return pb


@adamw.register_fake
def _(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, max_norm):
    return p.new_empty(p.shape, dtype=BF16)


OPS = ("linear", "linear_wgrad", "layer_norm", "attn_spatial", "attn_spatial_bwd", "attn_temporal", "attn_temporal_bwd", "mlp",
       "mlp_bwd", "qknorm", "qknorm_bwd", "embed", "modulate", "readout_ce", "maskgit_step", "adamw")
