"""ST-transformer trunk modules with the reference's names, constructor arguments and state-dict
(hma/model/st_transformer.py:9-177).  They hold parameters; arithmetic runs in the HIP kernels.
`STTransformerDecoder.forward` / `STBlock.forward` execute through the owning STMaskGIT's engine, with or without autograd."""
from __future__ import annotations

import weakref
from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from .._lib import EPI_F32, EPI_GELU2
from .attention import SelfAttention


class Mlp(nn.Module):
    def __init__(self, d_model: int, mlp_ratio: float = 4.0, mlp_bias: bool = True, mlp_drop: float = 0.0) -> None:
        super().__init__()
        hidden_dim = int(d_model * mlp_ratio)
        self.fc1 = nn.Linear(d_model, hidden_dim, bias=mlp_bias)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_dim, d_model, bias=mlp_bias)
        self.drop = nn.Dropout(mlp_drop)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """fc2(gelu(fc1 x)) (st_transformer.py:24-27; dropout p = 0 on this path).  With autograd on and anything that requires a
        gradient in reach it runs as the `torch.ops.hma.mlp` custom op, whose backward gives x, fc1 and fc2 their gradients."""
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            if self.training and self.drop.p > 0:
                raise NotImplementedError("Mlp.forward with autograd and mlp_drop > 0: train through STMaskGIT / STMAR (hma_amd.train)")
            from .. import torch_ops  # noqa: F401  (registers torch.ops.hma.*)
            bf = torch.bfloat16
            shp = x.shape
            y, _, _ = torch.ops.hma.mlp(x.reshape(-1, shp[-1]).to(bf), self.fc1.weight.to(bf), self.fc1.bias, self.fc2.weight.to(bf),
                                        self.fc2.bias)
            return y.view(shp).to(x.dtype)
        with torch.no_grad():
            return self._forward_inference(x)

    def _forward_inference(self, x: torch.Tensor) -> torch.Tensor:
        shp = x.shape
        xf = x.reshape(-1, shp[-1]).contiguous().float()
        hid = self.fc1.weight.shape[0]
        u = torch.empty(xf.shape[0], hid, dtype=torch.bfloat16, device=x.device)
        h = torch.empty_like(u)
        ops.linear(xf, self.fc1.weight.detach().to(torch.bfloat16), self.fc1.bias, epi=EPI_GELU2, out=u, out2=h)
        y = ops.linear(h, self.fc2.weight.detach().to(torch.bfloat16), self.fc2.bias, epi=EPI_F32)
        return y.view(shp).to(x.dtype)



def _wants_grad(x: torch.Tensor, a: Optional[torch.Tensor], module: nn.Module) -> bool:
    return torch.is_grad_enabled() and (x.requires_grad or (a is not None and a.requires_grad) or
                                        any(p.requires_grad for p in module.parameters()))


class _TrunkFn(torch.autograd.Function):
    """Layers [l0, l1) of the decoder as ONE autograd node over the engine's recorded forward / backward plans
    (st_transformer.py:79-114, 172-177 are ordinary autograd modules).  Inputs: the residual stream, the embedded actions
    (or None) and the owner's anchor (so the node runs when only parameters require a gradient).  The layers' weight gradients
    do not travel through autograd: the backward ADDS them into the engine's flat gradient buffer and points the parameters'
    `.grad` at their views of it, as `loss.backward()` of STMaskGIT.forward does."""

    @staticmethod
    def forward(ctx, x, a_emb, anchor, owner, domain, l0, l1):
        eng = owner._get_engine(x.device)
        owner._check_versions(eng)  # (an optimizer that wrote through the named parameters since the last pass)
        y, stamp = eng.trunk_autograd_forward(x.detach().float(), None if a_emb is None else a_emb.detach().float(), domain, l0, l1)
        ctx.owner, ctx.stamp, ctx.eng = owner, stamp, eng
        ctx.x_dtype, ctx.a_dtype = x.dtype, (None if a_emb is None else a_emb.dtype)
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        owner, eng = ctx.owner, ctx.eng
        if owner._engine is not eng:
            raise RuntimeError("the model's engine was rebuilt between forward and backward")
        owner._trunk_grads_begin()
        dx, da = eng.trunk_autograd_backward(dy.detach().float().contiguous(), ctx.stamp)
        owner._trunk_grads_publish(ctx.stamp)
        return (dx.to(ctx.x_dtype) if ctx.needs_input_grad[0] else None,
                da.to(ctx.a_dtype) if (da is not None and ctx.needs_input_grad[1]) else None, None, None, None, None, None)


def _trunk_autograd(owner, module: nn.Module, x: torch.Tensor, a_emb: Optional[torch.Tensor], domain, l0: int, l1: int):
    cfg = owner.config
    if (not module.training) and float(getattr(cfg, "mlp_drop", 0.0) or 0.0) > 0.0:
        raise NotImplementedError("gradients through the blocks in eval() mode with mlp_drop > 0 (the saved-activation plans apply the "
                                  "Dropout): call .train(), or run under torch.no_grad()")
    if owner._anchor is None or owner._anchor.device != x.device:
        owner._get_engine(x.device)
    return _TrunkFn.apply(x, a_emb, owner._anchor, owner, domain, l0, l1)

class STBlock(nn.Module):
    def __init__(self, num_heads: int, d_model: int, qkv_bias: bool = False, proj_bias: bool = True, qk_norm: bool = True,
                 use_mup: bool = True, attn_drop: float = 0.05, mlp_ratio: float = 4.0, mlp_bias: bool = True,
                 mlp_drop: float = 0.05, action_processing: str = "mlp", jointly_predict_actions: bool = False,
                 mask_token_id: int = 0) -> None:
        super().__init__()
        self.norm1 = nn.Identity() if qk_norm else nn.LayerNorm(d_model, eps=1e-05)
        kw = dict(num_heads=num_heads, d_model=d_model, qkv_bias=qkv_bias, proj_bias=proj_bias, qk_norm=qk_norm,
                  use_mup=use_mup, attn_drop=attn_drop)
        self.spatial_attn = SelfAttention(**kw)
        self.temporal_attn = SelfAttention(**kw)
        self.action_prediction = jointly_predict_actions
        self.action_processing = action_processing
        self.norm2 = nn.Identity() if qk_norm else nn.LayerNorm(d_model, eps=1e-05)
        self.mlp = Mlp(d_model=d_model, mlp_ratio=mlp_ratio, mlp_bias=mlp_bias, mlp_drop=mlp_drop)
        self.action_projectors = None  # attached by STMaskGIT.init_action_projectors
        self._owner = None
        self._index = -1

    def forward(self, x_TSC: torch.Tensor, action_ids: Optional[torch.Tensor] = None, domain=None) -> torch.Tensor:
        """(B, T, S, C) -> same (st_transformer.py:79-114).  `action_ids` is the (B, T, C) action embedding.  With autograd on and
        anything that requires a gradient in reach, the block runs as one autograd node (`_TrunkFn`): x and action_ids get their
        gradients through autograd, the block's parameters through `.grad` views of the engine's gradient buffer."""
        use = action_ids is not None and domain is not None and self.action_projectors is not None
        if _wants_grad(x_TSC, action_ids if use else None, self):
            owner = self._owner() if self._owner is not None else None
            if owner is None:
                raise RuntimeError("STBlock runs through its STMaskGIT's engine; construct it via STMaskGIT")
            return _trunk_autograd(owner, self, x_TSC, action_ids if use else None, domain if use else None, self._index, self._index + 1)
        with torch.no_grad():
            return self._forward_inference(x_TSC, action_ids, domain)

    def _forward_inference(self, x_TSC, action_ids, domain):
        owner = self._owner() if self._owner is not None else None
        if owner is None:
            raise RuntimeError("STBlock runs through its STMaskGIT's engine; construct it via STMaskGIT")
        use = action_ids is not None and domain is not None and self.action_projectors is not None
        eng = owner._get_engine(x_TSC.device)
        owner._check_versions(eng)
        return eng.run_trunk(x_TSC.float(), action_ids.float() if use else None, domain if use else None, self._index,
                             self._index + 1).to(x_TSC.dtype)


class STTransformerDecoder(nn.Module):
    def __init__(self, num_layers: int, num_heads: int, d_model: int, qkv_bias: bool = False, proj_bias: bool = True,
                 qk_norm: bool = True, use_mup: bool = True, attn_drop: float = 0.0, mlp_ratio: float = 4.0,
                 mlp_bias: bool = True, mlp_drop: float = 0.0, action_processing: str = "mlp",
                 jointly_predict_actions: bool = False, random_dummy_action: bool = True, mask_token_id: int = 0):
        super().__init__()
        self.layers = nn.ModuleList([
            STBlock(num_heads=num_heads, d_model=d_model, qkv_bias=qkv_bias, proj_bias=proj_bias, qk_norm=qk_norm,
                    use_mup=use_mup, attn_drop=attn_drop, mlp_ratio=mlp_ratio, mlp_bias=mlp_bias, mlp_drop=mlp_drop,
                    action_processing=action_processing, jointly_predict_actions=jointly_predict_actions,
                    mask_token_id=mask_token_id) for _ in range(num_layers)])
        self.apply(self._init_weights)
        self._owner = None

    def _init_weights(self, m):
        # xavier-uniform gain 0.1, zero bias; LayerNorm (1, 0)   (st_transformer.py:160-170)
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight, gain=0.1)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _bind(self, owner) -> None:
        self._owner = weakref.ref(owner)
        for i, layer in enumerate(self.layers):
            layer._owner = self._owner
            layer._index = i

    def forward(self, tgt: torch.Tensor, action_ids: Optional[torch.Tensor] = None, domain="") -> torch.Tensor:
        """N x STBlock (st_transformer.py:172-177); under autograd the whole stack is one node (see STBlock.forward)."""
        use = action_ids is not None and bool(domain) and self.layers[0].action_projectors is not None
        if _wants_grad(tgt, action_ids if use else None, self):
            owner = self._owner() if self._owner is not None else None
            if owner is None:
                raise RuntimeError("STTransformerDecoder runs through its STMaskGIT's engine; construct it via STMaskGIT")
            return _trunk_autograd(owner, self, tgt, action_ids if use else None, domain if use else None, 0, len(self.layers))
        with torch.no_grad():
            return self._forward_inference(tgt, action_ids, domain)

    def _forward_inference(self, tgt, action_ids, domain):
        owner = self._owner() if self._owner is not None else None
        if owner is None:
            raise RuntimeError("STTransformerDecoder runs through its STMaskGIT's engine; construct it via STMaskGIT")
        use = action_ids is not None and domain and self.layers[0].action_projectors is not None
        eng = owner._get_engine(tgt.device)
        owner._check_versions(eng)
        return eng.run_trunk(tgt.float(), action_ids.float() if use else None, domain if use else None).to(tgt.dtype)
