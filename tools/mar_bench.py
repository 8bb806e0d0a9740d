#!/usr/bin/env python3
"""STMAR train-step timing at the C4 shape (BASELINE configs[3]): 32 layers, T = 16, 32x32x4 latents (256 patch tokens +
64 action tokens per frame), batch 16 per GPU, diffusion head width 1024 / depth 4; forward + backward + clip + AdamW."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hma_amd import _lib
if os.environ.get("HMA_LIB"):  # a variant build (tools/chain_variants.sh)
    _lib.LIB_PATH = os.environ["HMA_LIB"]
from hma_amd.config import DiffusionGenieConfig
from hma_amd.model.st_mar import STMAR

B = int(os.environ.get("BATCH", 16)); T = 16; L = int(os.environ.get("LAYERS", 32))
cfg = DiffusionGenieConfig(num_layers=L, num_heads=8, d_model=256, T=T, S=1024, use_mup=True, action_network="concat+modulate",
                           num_factored_vocabs=2, qkv_bias=True, proj_bias=True, qk_norm=False, mlp_drop=float(os.environ.get("MLP_DROP", 0.0)), mlp_bias=False, patch_size=2,
                           vae_embed_dim=4, diffloss_w=1024, diffloss_d=4, num_sampling_steps="100", attn_drop=0.0)
m = STMAR(cfg)
doms = ["d0", "d1"]
m.init_action_projectors(doms, [14, 7], [[[0.0] * 7, [1.0] * 7]] * 2, cfg.action_network)
with torch.no_grad():
    for n, p in m.named_parameters():
        if p.dim() >= 2 and p.abs().max() == 0: p.normal_(0, 0.02)
m = m.to("cuda").train()
g = torch.Generator(device="cuda").manual_seed(0)
lat = torch.randn(B, T * 1024, 4, device="cuda", generator=g) * 0.7
masked = torch.rand(B, T, 32, 32, device="cuda", generator=g) < 0.6
act = torch.randn(B, T, 14, device="cuda", generator=g)
def step():
    m.zero_grad()
    out = m(input_ids=lat, labels=lat, action_ids=act, domain=["d0"] * B, masked_tokens_indicator=masked, h=[32] * B, w=[32] * B)
    out.loss.backward()
    m.optimizer_step(1e-4, "d0")
    return out.loss
for _ in range(2): l = step()
torch.cuda.synchronize(); t0 = time.time(); n = 3
for _ in range(n): l = step()
torch.cuda.synchronize(); dt = (time.time() - t0) / n
tok = B * T * 256
print(f"STMAR C4 step: B={B} T={T} L={L}: {dt * 1e3:.1f} ms/step, {tok / dt / 1e3:.1f} k patch-tokens/s ({B * T * 1024 / dt / 1e6:.2f} M latent positions/s), loss {l.item():.4f}, "
      f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
