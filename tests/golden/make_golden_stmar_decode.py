#!/usr/bin/env python3
"""Golden vectors for STMAR.maskgit_generate (MAR decode of one frame) from the REAL reference -> g12_stmar_decode.safetensors.
The random generation order and every Gaussian draw (DiffLoss.sample's initial noise and each p_sample's randn_like) are
recorded and stored with the reference's output.  Build container only."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: F401,E402
from stmar_cfg import CFG, DOMAINS, D_ACTIONS, STATS, seeded_state, inputs  # noqa: E402

from hma.config import DiffusionGenieConfig  # noqa: E402
from hma.model.st_mar import STMAR  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

cfg = DiffusionGenieConfig(**CFG)
model = STMAR(cfg)
model.init_action_projectors(DOMAINS, D_ACTIONS, STATS, cfg.action_network)
model.load_state_dict(seeded_state(model.state_dict()))
model.eval()
inp = inputs()
g = torch.Generator().manual_seed(99)
B, T, H = 2, 3, 32
prompt = inp["latents"].reshape(B, T, H, H, 4).clone()
prompt[:, 2] = model.mask_token.detach().reshape(-1)      # the frame to generate holds the mask latent
orders = torch.stack([torch.randperm(256, generator=g) for _ in range(B)])
model.sample_orders = lambda bsz: orders.clone()
calls = []      # per DiffLoss.sample call: [noise0, randn_like draws...]
o_randn, o_randn_like = torch.randn, torch.randn_like
def randn(*shape, **k):
    v = o_randn(*shape, generator=g)
    calls.append([v])
    return v
def randn_like(x, **k):
    v = o_randn(x.shape, generator=g)
    calls[-1].append(v)
    return v
torch.randn, torch.randn_like = randn, randn_like
try:
    frame, orig, _ = model.maskgit_generate(prompt.clone(), 2, action_ids=inp["actions_domA"], domain=["domA"] * B, maskgit_steps=2,
                                            temperature=0.9, h=[H] * B, w=[H] * B)
finally:
    torch.randn, torch.randn_like = o_randn, o_randn_like
out = {"prompt": prompt, "orders": orders, "frame": frame.contiguous(), "orig_latents": orig.contiguous()}
for k, c in enumerate(calls):
    out[f"noise0.{k}"] = c[0]
    out[f"steps.{k}"] = torch.stack(c[1:])
save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "g12_stmar_decode.safetensors"))
print("wrote g12:", {k: tuple(v.shape) for k, v in out.items()}, os.path.getsize(os.path.join(HERE, "g12_stmar_decode.safetensors")) // 1024, "KB")
