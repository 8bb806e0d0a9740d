#!/usr/bin/env python3
"""G17: STMAR with jointly_predict_actions=True from the REAL reference (st_mar.py:231-273).  Build container only.

    python tests/golden/make_golden_stmar_jpa.py   -> tests/golden/g17_stmar_jpa.safetensors
The random draws of the forward are replaced by recorded ones, told apart by shape: the action-mask start frames
(`torch.randint(0, T, (B, 1))`, :236), the video head's timesteps / noise and the action head's timesteps / noise."""
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

cfg = DiffusionGenieConfig(**dict(CFG, jointly_predict_actions=True))
model = STMAR(cfg)
model.init_action_projectors(DOMAINS, D_ACTIONS, STATS, cfg.action_network)
model.load_state_dict(seeded_state(model.state_dict()))
model.train()
inp = inputs()
B, T = 2, CFG["T"]
g = torch.Generator().manual_seed(77)
start = torch.tensor([[1], [2]])                       # frames >= start are masked: neither all nor none
t_act = torch.randint(0, 1000, (B * T,), generator=g)
noise_act = torch.randn(B * T, 7, generator=g)
o_randint, o_randn_like = torch.randint, torch.randn_like


def fake_randint(*a, **k):
    shape = a[-1] if isinstance(a[-1], (tuple, list, torch.Size)) else a[2]
    shape = tuple(shape)
    if shape == (B, 1):
        return start.clone()
    if shape == (B * T,):
        return t_act
    assert shape == tuple(inp["t"].shape), shape
    return inp["t"]


def fake_randn_like(x, *a, **k):
    if tuple(x.shape) == tuple(noise_act.shape):
        return noise_act
    assert tuple(x.shape) == tuple(inp["noise"].shape), x.shape
    return inp["noise"]


torch.randint, torch.randn_like = fake_randint, fake_randn_like
try:
    out = model(input_ids=inp["latents"].clone(), labels=inp["latents"].clone(), action_ids=inp["actions_domA"], domain=["domA"] * 2,
                masked_tokens_indicator=inp["masked"], h=[32, 32], w=[32, 32])
finally:
    torch.randint, torch.randn_like = o_randint, o_randn_like
(out.loss + 1.0 * out.action_loss).backward()   # DiffusionGenieConfig.action_loss_weight = 1.0 (config.py:113)
mask = torch.zeros(B, T)
for i in range(B):
    mask[i, int(start[i]):] = 1
fix = {"loss": out.loss.detach().reshape(1), "action_loss": out.action_loss.detach().reshape(1), "actions": out.actions.detach().contiguous(),
       "action_mask": mask, "t_act": t_act, "noise_act": noise_act,
       "z": out.logits.detach().permute(0, 2, 3, 4, 1).reshape(2, 3, 256, 256).contiguous()}
names = ["token_embed.weight", "out_x_proj.weight", "pos_embed_TSC", "decoder.layers.0.spatial_attn.qkv.weight", "decoder.layers.1.mlp.fc1.weight",
         "decoder.layers.0.action_projectors.domA.linear_out.weight", "action_mlp.domA.model.0.weight", "diffloss.net.cond_embed.weight",
         "action_diff_losses.domA.net.cond_embed.weight", "action_diff_losses.domA.net.final_layer.linear.weight",
         "action_diff_losses.domA.net.res_blocks.0.mlp.0.weight", "action_diff_losses.domA.net.input_proj.weight"]
params = dict(model.named_parameters())
for n in names:
    fix["grad." + n] = params[n].grad.detach().clone()
fix["grad_is_none.domB_head"] = torch.tensor([float(params["action_diff_losses.domB.net.cond_embed.weight"].grad is None)])
save_file({k: v.contiguous() for k, v in fix.items()}, os.path.join(HERE, "g17_stmar_jpa.safetensors"))
print("wrote g17_stmar_jpa:", len(fix), "tensors; loss", float(out.loss), "action_loss", float(out.action_loss))
