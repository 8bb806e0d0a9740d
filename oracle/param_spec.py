"""TEST INFRASTRUCTURE ONLY -- names/shapes of the reference STMaskGIT state-dict.

Restates what `STMaskGIT.__init__` + `init_action_projectors` register
(hma/model/st_mask_git.py:152-251, hma/model/st_transformer.py:30-77,
hma/model/attention.py:23-35, hma/model/factorization_utils.py:26-29); checked
key-for-key against the imported reference by tests/golden/make_golden.py.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Sequence, Tuple

import torch


def state_dict_spec(cfg, domains: Sequence[str] = (), d_actions: Sequence[int] = (),
                    action_dims: Sequence[int] = ()) -> "OrderedDict[str, Tuple[int, ...]]":
    """name -> shape, parameters and buffers.  qk_norm=True (hma/model/st_transformer.py:55,62 and attention.py:31-35): norm1 / norm2 are
    nn.Identity (no parameters) and each attention owns a LayerNorm(head_dim) shared by q and k."""
    d, L = cfg.d_model, cfg.num_layers
    hid = int(d * cfg.mlp_ratio)
    C = cfg.factored_vocab_size * cfg.num_factored_vocabs
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    out["pos_embed_TSC"] = (1, cfg.T, cfg.S + cfg.action_token_size, d)
    out["action_mask_tokens"] = (1, cfg.T, 1, d)
    for l in range(L):
        p = f"decoder.layers.{l}"
        qkn = bool(getattr(cfg, "qk_norm", False))
        if not qkn:
            out[f"{p}.norm1.weight"] = (d,)
            out[f"{p}.norm1.bias"] = (d,)
        for a in ("spatial_attn", "temporal_attn"):
            out[f"{p}.{a}.qkv.weight"] = (3 * d, d)
            if cfg.qkv_bias:
                out[f"{p}.{a}.qkv.bias"] = (3 * d,)
            out[f"{p}.{a}.proj.weight"] = (d, d)
            if cfg.proj_bias:
                out[f"{p}.{a}.proj.bias"] = (d,)
            if qkn:
                out[f"{p}.{a}.norm.weight"] = (d // cfg.num_heads,)
                out[f"{p}.{a}.norm.bias"] = (d // cfg.num_heads,)
        if not qkn:
            out[f"{p}.norm2.weight"] = (d,)
            out[f"{p}.norm2.bias"] = (d,)
        out[f"{p}.mlp.fc1.weight"] = (hid, d)
        out[f"{p}.mlp.fc2.weight"] = (d, hid)
        if cfg.mlp_bias:
            out[f"{p}.mlp.fc1.bias"] = (hid,)
            out[f"{p}.mlp.fc2.bias"] = (d,)
        if "modulate" in cfg.action_network:
            for dom in domains:
                q = f"{p}.action_projectors.{dom}"
                out[f"{q}.linear_out.weight"] = (d, d)
                out[f"{q}.linear_out.bias"] = (d,)
                out[f"{q}.adaLN_modulation.0.weight"] = (d, d)
                out[f"{q}.adaLN_modulation.0.bias"] = (d,)
                out[f"{q}.adaLN_modulation.2.weight"] = (2 * d, d)
                out[f"{q}.adaLN_modulation.2.bias"] = (2 * d,)
    out["token_embed.mask_token_embed"] = (1, d)
    for i in range(cfg.num_factored_vocabs):
        out[f"token_embed.factored_embeds.{i}.weight"] = (cfg.factored_vocab_size, d)
    out["out_x_proj.weight"] = (C, d)
    out["out_x_proj.bias"] = (C,)
    for dom, da, ad in zip(domains, d_actions, action_dims):
        out[f"action_preprocessor.{dom}.mean"] = (ad,)
        out[f"action_preprocessor.{dom}.std"] = (ad,)
        out[f"action_mlp.{dom}.model.0.weight"] = (d, da)
        out[f"action_mlp.{dom}.model.0.bias"] = (d,)
        out[f"action_mlp.{dom}.model.1.weight"] = (d,)
        out[f"action_mlp.{dom}.model.1.bias"] = (d,)
        out[f"action_mlp.{dom}.model.3.weight"] = (d, d)
        out[f"action_mlp.{dom}.model.3.bias"] = (d,)
        out[f"action_out_projectors.{dom}.weight"] = (da, d)
        out[f"action_out_projectors.{dom}.bias"] = (da,)
    return out


def seeded_state_dict(spec: "OrderedDict[str, Tuple[int, ...]]", seed: int = 0, std: float = 0.05,
                      embed_std: float = 0.5) -> Dict[str, torch.Tensor]:
    """Deterministic fill (sorted-name order, CPU generator).

    The default init leaves pos-embed / mask tokens at exactly zero which would hide indexing
    bugs, so every tensor is N(0, std) except: LayerNorm-like
    `norm*.weight` / `model.1.weight` = 1 + N(0, std); `.std` buffers = 0.5 + |N(0,1)|.
    """
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name in sorted(spec):
        t = torch.randn(spec[name], generator=g, dtype=torch.float32) * std
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("model.1.weight") or name.endswith("attn.norm.weight"):
            t = t + 1.0
        if name.endswith(".std"):
            t = t.abs() / std + 0.5
        if "factored_embeds" in name:
            t = t / std * embed_std
        sd[name] = t
    return sd
