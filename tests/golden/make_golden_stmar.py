#!/usr/bin/env python3
"""Golden vectors for STMAR's training forward / backward from the REAL reference.

    python tests/golden/make_golden_stmar.py   -> tests/golden/g11_stmar.safetensors
The state dict is regenerated from a seed on both sides (tests/golden/stmar_cfg.py); the fixture holds the reference's
loss, latents z and a selection of gradients for the seeded inputs and diffusion draws.  Build container only.
"""
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
model.train()
inp = inputs()
draws = [inp["t"], inp["noise"]]
o_randint, o_randn_like = torch.randint, torch.randn_like
torch.randint = lambda *a, **k: inp["t"]
torch.randn_like = lambda x, *a, **k: inp["noise"]
try:
    out = model(input_ids=inp["latents"].clone(), labels=inp["latents"].clone(), action_ids=inp["actions_domA"], domain=["domA"] * 2,
                masked_tokens_indicator=inp["masked"], h=[32, 32], w=[32, 32])
finally:
    torch.randint, torch.randn_like = o_randint, o_randn_like
out.loss.backward()
fix = {"loss": out.loss.detach().reshape(1), "z": out.logits.detach().permute(0, 2, 3, 4, 1).reshape(2, 3, 256, 256).contiguous()}
names = ["token_embed.weight", "mask_token", "z_proj_ln.weight", "z_proj_ln.bias", "decoder_norm.weight", "decoder_norm.bias",
         "out_x_proj.weight", "out_x_proj.bias", "diffusion_pos_embed_learned", "pos_embed_TSC",
         "decoder.layers.0.spatial_attn.qkv.weight", "decoder.layers.0.spatial_attn.qkv.bias", "decoder.layers.1.mlp.fc1.weight",
         "decoder.layers.0.action_projectors.domA.linear_out.weight", "action_mlp.domA.model.0.weight",
         "diffloss.net.cond_embed.weight", "diffloss.net.final_layer.linear.weight", "diffloss.net.res_blocks.0.mlp.0.weight"]
params = dict(model.named_parameters())
for n in names:
    fix["grad." + n] = params[n].grad.detach().clone()
fix["grad_is_none.domB"] = torch.tensor([float(params["action_mlp.domB.model.0.weight"].grad is None)])
with open(os.path.join(HERE, "g11_stmar_keys.txt"), "w") as f:
    f.write("\n".join(f"{k} {tuple(v.shape)}" for k, v in model.state_dict().items()))
save_file({k: v.contiguous() for k, v in fix.items()}, os.path.join(HERE, "g11_stmar.safetensors"))
print("wrote g11_stmar:", len(fix), "tensors,", os.path.getsize(os.path.join(HERE, "g11_stmar.safetensors")) // 1024, "KB; loss", float(out.loss))
