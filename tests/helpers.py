"""Shared test helpers: golden loading, the tiny seeded state-dict, tolerant comparisons."""
import os

import torch
from safetensors.torch import load_file

from oracle.param_spec import seeded_state_dict, state_dict_spec
from oracle.st_maskgit_ref import RefConfig
from tests.golden.golden_cfg import TINY, tiny_inputs  # noqa: F401

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return load_file(os.path.join(GOLDEN, name + ".safetensors"))


def tiny_ref_config(**over):
    kw = {k: v for k, v in TINY["config"].items() if k in RefConfig.__dataclass_fields__}
    kw.update(over)
    return RefConfig(**kw)


def tiny_state_dict(cfg=None, initlike: bool = False):
    """The seeded fixture weights: N(0, 0.05) with unit-scale embeddings (adversarially large activations),
    or -- `initlike` -- N(0, 0.02) everywhere like the reference's init_weights (st_mask_git.py:737-753)."""
    cfg = cfg or tiny_ref_config()
    spec = state_dict_spec(cfg, TINY["domains"], TINY["d_actions"], [len(s[0]) for s in TINY["action_stats"]])
    if initlike:
        sd = seeded_state_dict(spec, seed=TINY["seed"] + 1, std=0.02, embed_std=0.02)
    else:
        sd = seeded_state_dict(spec, seed=TINY["seed"])
    for dom, st in zip(TINY["domains"], TINY["action_stats"]):
        sd[f"action_preprocessor.{dom}.mean"] = torch.tensor(st[0], dtype=torch.float32)
        sd[f"action_preprocessor.{dom}.std"] = torch.tensor(st[1], dtype=torch.float32)
    return sd


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / (max |b| + tiny): scale-aware max error."""
    a = a.double().cpu()
    b = b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def rms_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.double().cpu()
    b = b.double().cpu()
    return ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30)).item()
