#!/usr/bin/env python3
"""G14: two clip(1.0) + AdamW steps of the tiny STMAR from the REAL reference (build container only).

    python tests/golden/make_golden_stmar_adamw.py   -> tests/golden/g14_stmar_adamw.safetensors

The optimizer is built exactly as hma/train_multi.py:907-922 builds it (names containing "bias" or "layer_norm.weight" are
un-decayed -- for this model that is the biases only: decoder_norm.weight, z_proj_ln.weight, the head's in_ln.weight,
mask_token and the positional embeddings ARE decayed), AdamW(0.9 / 0.95, 1e-8, wd 0.05), clip_grad_norm_(1.0) (:593-598).
domB and the action-diffusion heads receive no gradient (grad None): torch.optim.AdamW never touches them.
The diffusion draws (t, noise) of both steps are the seeded ones of tests/golden/stmar_cfg.py.
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

LR = 1e-3
cfg = DiffusionGenieConfig(**CFG)
model = STMAR(cfg)
model.init_action_projectors(DOMAINS, D_ACTIONS, STATS, cfg.action_network)
model.load_state_dict(seeded_state(model.state_dict()))
model.train()
before = {n: p.detach().clone() for n, p in model.named_parameters()}
no_decay = ["bias", "layer_norm.weight"]
groups = [
    {"params": [p for n, p in model.named_parameters() if not any(nd in n for nd in no_decay)], "weight_decay": 0.05},
    {"params": [p for n, p in model.named_parameters() if any(nd in n for nd in no_decay)], "weight_decay": 0.0},
]
opt = torch.optim.AdamW(groups, lr=LR, betas=(0.9, 0.95), eps=1e-8)
inp = inputs()
fix = {}
o_randint, o_randn_like = torch.randint, torch.randn_like
for it in range(2):
    torch.randint = lambda *a, **k: inp["t"]
    torch.randn_like = lambda x, *a, **k: inp["noise"]
    try:
        out = model(input_ids=inp["latents"].clone(), labels=inp["latents"].clone(), action_ids=inp["actions_domA"], domain=["domA"] * 2,
                    masked_tokens_indicator=inp["masked"], h=[32, 32], w=[32, 32])
    finally:
        torch.randint, torch.randn_like = o_randint, o_randn_like
    out.loss.backward()
    norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    opt.zero_grad()
    fix[f"step{it}.loss"] = out.loss.detach().reshape(1)
    fix[f"step{it}.grad_norm"] = norm.detach().reshape(1)
untouched = []
for n, p in model.named_parameters():
    flat = p.detach().reshape(-1)
    idx = torch.linspace(0, flat.numel() - 1, 64).long()
    fix[f"param_samp.{n}"] = flat[idx].clone()
    fix[f"delta_abs.{n}"] = (p.detach() - before[n]).abs().sum().reshape(1)
    if torch.equal(p.detach(), before[n]):
        untouched.append(n)
with open(os.path.join(HERE, "g14_stmar_untouched.txt"), "w") as f:
    f.write("\n".join(untouched))
save_file({k: v.contiguous() for k, v in fix.items()}, os.path.join(HERE, "g14_stmar_adamw.safetensors"))
print("wrote g14_stmar_adamw:", len(fix), "tensors,", os.path.getsize(os.path.join(HERE, "g14_stmar_adamw.safetensors")) // 1024, "KB; losses",
      float(fix["step0.loss"]), float(fix["step1.loss"]), "norms", float(fix["step0.grad_norm"]), float(fix["step1.grad_norm"]),
      "untouched", len(untouched))
