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

OPS = ("linear", "linear_wgrad", "layer_norm", "attn_spatial", "attn_spatial_bwd", "attn_temporal", "attn_temporal_bwd", "mlp",
       "mlp_bwd")
