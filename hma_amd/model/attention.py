"""SelfAttention with the reference's constructor / forward signature (hma/model/attention.py:10-61).

There is one implementation: the gfx950 kernels.  `XFORMERS_DISABLED` has no meaning here; both
reference class names resolve to it."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from .._lib import EPI_BF16, EPI_F32


class SelfAttention(nn.Module):
    def __init__(self, num_heads: int, d_model: int, qkv_bias: bool = False, proj_bias: bool = True, qk_norm: bool = True,
                 use_mup: bool = True, attn_drop: float = 0.0) -> None:
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = d_model // num_heads
        self.scale = 8 / self.head_dim if use_mup else self.head_dim ** -0.5  # attention.py:27
        self.qkv = nn.Linear(d_model, d_model * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)  # constructed but never applied, as in the reference
        self.proj = nn.Linear(d_model, d_model, bias=proj_bias)
        self.qk_norm = qk_norm
        if self.qk_norm:
            self.norm = nn.LayerNorm(self.head_dim, eps=1e-05)

    def forward(self, x: torch.Tensor, causal: bool = False) -> torch.Tensor:
        """x (Bn, N, 256): spatial (causal=False, N in {64, 256, 320}) or temporal (causal=True, N <= 16).  With autograd on and
        anything that requires a gradient in reach, the same kernels run as `torch.ops.hma.*` custom ops with their backward
        formulas (hma_amd/torch_ops.py), so a caller that trains through the module gets gradients for x, qkv and proj."""
        if self.num_heads != 8 or self.head_dim != 32:
            raise NotImplementedError("kernels are built for 8 heads of 32")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return self._forward_autograd(x, causal)
        with torch.no_grad():
            return self._forward_inference(x, causal)

    def _forward_autograd(self, x: torch.Tensor, causal: bool) -> torch.Tensor:
        from .. import torch_ops  # noqa: F401  (registers torch.ops.hma.*)
        Bn, N, Cd = x.shape
        bf = torch.bfloat16
        xb = x.reshape(Bn * N, Cd).to(bf)
        qkv = torch.ops.hma.linear(xb, self.qkv.weight.to(bf), self.qkv.bias)
        if self.qk_norm:  # q, k = norm(q), norm(k) per head (attention.py:44-48)
            qkv, _ = torch.ops.hma.qknorm(qkv, self.norm.weight, self.norm.bias, float(self.norm.eps))
        if causal:  # rows (t, column): one sample of N frames with Bn positions
            o = torch.ops.hma.attn_temporal(qkv.view(Bn, N, 768).transpose(0, 1).reshape(-1, 768), 1, N, Bn, self.scale)
            o = o.view(N, Bn, Cd).transpose(0, 1).reshape(Bn * N, Cd)
        else:
            o, _ = torch.ops.hma.attn_spatial(qkv, Bn, N, self.scale)
        y = torch.ops.hma.linear(o, self.proj.weight.to(bf), self.proj.bias)
        return y.view(Bn, N, Cd).to(x.dtype)

    def _forward_inference(self, x: torch.Tensor, causal: bool) -> torch.Tensor:
        Bn, N, Cd = x.shape
        xf = x.reshape(Bn * N, Cd).contiguous().float()
        qkv = ops.linear(xf, self.qkv.weight.detach().to(torch.bfloat16), self.qkv.bias, epi=EPI_BF16)
        if self.qk_norm:
            qkv, _ = ops.qknorm_fwd(qkv, self.norm.weight.detach().float(), self.norm.bias.detach().float(), float(self.norm.eps), save_raw=False)
        if causal:
            o = ops.attn_temporal_fwd(qkv.view(Bn, N, 768).transpose(0, 1).contiguous().view(-1, 768), 1, N, Bn, self.scale)
            o = o.view(N, Bn, Cd).transpose(0, 1).contiguous().view(Bn * N, Cd)
        else:
            o, _ = ops.attn_spatial_fwd(qkv, Bn, N, self.scale)
        y = ops.linear(o, self.proj.weight.detach().to(torch.bfloat16), self.proj.bias, epi=EPI_F32)
        return y.view(Bn, N, Cd).to(x.dtype)


BasicSelfAttention = SelfAttention
MemoryEfficientAttention = SelfAttention
