#!/usr/bin/env python3
"""G11b: STMAR's training forward / backward WITHOUT action conditioning (`action_ids=None`, hma/model/st_mar.py:146-197: no action
tokens, the decoder unconditioned) from the REAL reference.  Build container only.

    python tests/golden/make_golden_stmar_noact.py   -> tests/golden/g11b_stmar_noact.safetensors"""
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
o_randint, o_randn_like = torch.randint, torch.randn_like
torch.randint = lambda *a, **k: inp["t"]
torch.randn_like = lambda x, *a, **k: inp["noise"]
try:
    out = model(input_ids=inp["latents"].clone(), labels=inp["latents"].clone(), action_ids=None, domain=None,
                masked_tokens_indicator=inp["masked"], h=[32, 32], w=[32, 32])
finally:
    torch.randint, torch.randn_like = o_randint, o_randn_like
out.loss.backward()
fix = {"loss": out.loss.detach().reshape(1), "z": out.logits.detach().permute(0, 2, 3, 4, 1).reshape(2, 3, 256, 256)[:, :, ::4].contiguous()}
names = ["token_embed.weight", "mask_token", "z_proj_ln.weight", "decoder_norm.bias", "out_x_proj.weight", "diffusion_pos_embed_learned",
         "decoder.layers.0.spatial_attn.qkv.weight", "decoder.layers.0.temporal_attn.proj.bias", "decoder.layers.1.mlp.fc1.weight",
         "diffloss.net.cond_embed.weight", "diffloss.net.final_layer.linear.weight"]
params = dict(model.named_parameters())
for n in names:
    g = params[n].grad.detach()
    fix["grad." + n] = (g[::4] if g.dim() == 2 and g.shape[0] >= 256 else g[:, ::4] if g.dim() == 3 else g).clone()  # (row subsets: a small fixture)
fix["grad.pos_embed_TSC.image_rows"] = params["pos_embed_TSC"].grad.detach()[:, :, :256:4].clone()
assert float(params["pos_embed_TSC"].grad[:, :, 256:].abs().max()) == 0.0          # no action rows in this mode
assert params["action_mlp.domA.model.0.weight"].grad is None and params["decoder.layers.0.action_projectors.domA.linear_out.weight"].grad is None
save_file({k: v.contiguous() for k, v in fix.items()}, os.path.join(HERE, "g11b_stmar_noact.safetensors"))
print("wrote g11b_stmar_noact:", len(fix), "tensors,", os.path.getsize(os.path.join(HERE, "g11b_stmar_noact.safetensors")) // 1024, "KB; loss",
      float(out.loss))
