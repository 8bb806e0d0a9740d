"""GenieConfig -- field-for-field mirror of the reference's checkpoint config (hma/config.py:9-82).

`config.json` written by the reference's `save_pretrained` loads here unchanged and vice versa
(the JSON is part of the checkpoint layout).
"""
from __future__ import annotations

import dataclasses
import json
from dataclasses import dataclass
from typing import List, Optional


def _exact_root(x: int, n: int) -> int:
    r = round(x ** (1.0 / n))
    if r ** n != x:
        raise AssertionError((x, n, r))  # same contract as factorization_utils.nth_root (:99-102)
    return r


@dataclass
class GenieConfig:
    num_layers: int
    num_heads: int
    d_model: int
    T: int = 12
    S: int = 256
    image_vocab_size: Optional[int] = 262144
    use_mup: bool = False
    dataloader_apply_mask: bool = True
    dataloader_apply_corruption: bool = True
    dataloader_mask_ratio_min: float = 0.2
    drop_action_ratio: float = 0.0
    arch: str = "STTransformerDecoder"
    random_dummy_action: bool = True
    num_factored_vocabs: int = 1
    factored_vocab_size: Optional[int] = None
    max_corrupt_rate: float = 0.2
    non_mlm_ratio: float = 0.2
    num_prompt_frames: int = 4
    init_actions: bool = False
    d_action: int = 28
    use_actions: bool = True
    action_domains: Optional[List[str]] = None
    d_actions: Optional[List[int]] = None
    action_stats: Optional[List[List[List[float]]]] = None
    action_network: str = "mlp"
    shared_action_mlps: bool = True
    action_contrastive_loss: bool = False
    jointly_predict_actions: bool = False
    jointly_predict_states: bool = True
    action_token_size: int = 64
    label_drop_prob: float = 0.5
    action_loss_weight: float = 0.5
    qkv_bias: bool = False
    proj_bias: bool = True
    attn_drop: float = 0.0
    qk_norm: bool = True
    mlp_ratio: float = 4.0
    mlp_drop: float = 0.0
    mlp_bias: bool = True

    def __post_init__(self):
        if self.image_vocab_size is None:
            self.factored_vocab_size = 64
        else:
            self.factored_vocab_size = _exact_root(self.image_vocab_size, self.num_factored_vocabs)

    def to_dict(self) -> dict:
        return dict(vars(self))

    def save_pretrained(self, json_path) -> None:
        with open(json_path, "w") as f:
            json.dump(self.to_dict(), f)

    @classmethod
    def from_pretrained(cls, json_path) -> "GenieConfig":
        with open(json_path, "r") as f:
            raw = json.load(f)
        return cls.from_dict(raw)

    @classmethod
    def from_dict(cls, raw: dict) -> "GenieConfig":
        known = {f.name for f in dataclasses.fields(cls)}
        return cls(**{k: v for k, v in raw.items() if k in known})

    def shallow_copy(self) -> "GenieConfig":
        return type(self)(**vars(self))


@dataclasses.dataclass
class DiffusionGenieConfig(GenieConfig):
    """Mirror of hma/config.py:84-117 (the continuous-latent / diffusion-head model, STMAR)."""

    Diffusion: bool = True
    dim: int = 512
    dataloader_apply_mask: bool = True
    dataloader_apply_corruption: bool = False
    dataloader_mask_ratio_min: float = 0.1
    vae_stride: int = 1
    patch_size: int = 1
    vae_embed_dim: int = 4
    mask_ratio_min: float = 0.7
    attn_dropout: float = 0.1
    proj_dropout: float = 0.1
    buffer_size: int = 64
    diffloss_d: int = 4
    diffloss_w: int = 1024
    num_sampling_steps: str = "100"
    diffusion_batch_mul: int = 1
    grad_checkpointing: bool = False
    use_actions: bool = True
    jointly_predict_actions: bool = False
    jointly_predict_states: bool = True
    action_token_size: int = 64
    label_drop_prob: float = 0.5
    action_loss_weight: float = 1.0
    predict_unmask: bool = False
    maskgit_steps: int = 16
