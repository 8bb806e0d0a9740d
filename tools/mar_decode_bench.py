#!/usr/bin/env python3
"""Timed MAR decode (hma/model/st_mar.py:358-557 with the 100-step sampler of gaussian_diffusion.py:394-441): STMAR.generate at the
C4 model (mar_n32_h8_d256_action.json, 32 layers, diffusion head 1024 x 4), T0 prompt frames + NEW generated frames of 32 x 32 x 4
latents, `maskgit_steps` MAR iterations per frame, each drawing every masked token through the 100-step diffusion sampler.
   python tools/mar_decode_bench.py            (B=4, 2 prompt + 2 generated frames, 8 MAR iterations)
Prints frames/s and ms per generated frame."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hma_amd.config import DiffusionGenieConfig
from hma_amd.model.st_mar import STMAR

dev = torch.device("cuda", 0)
B = int(os.environ.get("B", 4))
T0, NEW, STEPS, L = int(os.environ.get("T0", 2)), int(os.environ.get("NEW", 2)), int(os.environ.get("MG_STEPS", 8)), int(os.environ.get("LAYERS", 32))
T = T0 + NEW
cfgd = dict(num_layers=L, num_heads=8, d_model=256, T=T, S=1024, use_mup=True, action_network="concat+modulate", num_factored_vocabs=2,
            qkv_bias=True, proj_bias=True, qk_norm=False, mlp_drop=0.05, mlp_bias=False, patch_size=2, vae_embed_dim=4, diffloss_w=1024,
            diffloss_d=4, num_sampling_steps="100", attn_drop=0.0)
m = STMAR(DiffusionGenieConfig(**cfgd))
m.init_action_projectors(["dom0"], [14], [[[0.0] * 7, [1.0] * 7]], cfgd["action_network"])
with torch.no_grad():
    for p_ in m.parameters():
        if p_.dim() >= 2 and float(p_.abs().max()) == 0.0:
            p_.normal_(0, 0.02)
m = m.to(dev).eval()
g = torch.Generator(device=dev).manual_seed(0)
prompt = torch.randn(B, T0 * 1024, 4, device=dev, generator=g) * 0.7
act = torch.randn(B, T, 14, device=dev, generator=g)
kw = dict(max_new_tokens=NEW * 1024, action_ids=act, domain=["dom0"] * B, maskgit_steps=STEPS, temperature=1.0)
with torch.no_grad():
    out = m.generate(prompt, None, **kw)  # warm-up (plans, casts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = int(os.environ.get("REPS", 2))
    for _ in range(reps):
        out = m.generate(prompt, None, **kw)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
assert torch.isfinite(out).all() and out.shape[1] == T * 1024
print(f"MAR decode: B={B}, {T0} prompt + {NEW} generated frames of 32x32x4 latents, {STEPS} MAR iterations x 100 diffusion steps, L={L}: "
      f"{dt * 1e3:.1f} ms per rollout = {B * NEW / dt:.2f} frames/s, {dt * 1e3 / NEW:.1f} ms per generated frame (batch {B})")
