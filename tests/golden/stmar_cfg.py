"""Tiny STMAR configuration, seeded state dict and inputs shared by make_golden_stmar.py and the tests."""
import torch

CFG = dict(num_layers=2, num_heads=8, d_model=256, T=3, S=1024, image_vocab_size=262144, use_mup=True,
           action_network="concat+modulate", num_factored_vocabs=2, qkv_bias=True, proj_bias=True, attn_drop=0.0, qk_norm=False,
           mlp_ratio=4.0, mlp_drop=0.0, mlp_bias=False, patch_size=2, vae_embed_dim=4, diffloss_w=256, diffloss_d=2,
           num_sampling_steps="10", diffusion_batch_mul=1, use_actions=True)
DOMAINS, D_ACTIONS = ["domA", "domB"], [7, 14]
STATS = [[[0.05 * i for i in range(7)], [0.6 + 0.1 * i for i in range(7)]],
         [[-0.1 * i for i in range(7)], [1.0 + 0.05 * i for i in range(7)]]]


def seeded_state(template: dict, seed: int = 21) -> dict:
    """Deterministic values for every tensor of a state dict (same on both sides: ordered by name)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(template):
        v = template[k]
        if k.endswith(".mean") or k.endswith(".std"):
            out[k] = v.clone()
        elif v.dim() >= 2 and "pos_embed" not in k and "mask_token" not in k:
            out[k] = torch.randn(v.shape, generator=g) * (0.5 / v.shape[-1] ** 0.5)
        elif k.endswith("norm1.weight") or k.endswith("norm2.weight") or k.endswith("ln.weight") or k.endswith("norm.weight") \
                or k.endswith("model.1.weight"):
            out[k] = 1 + 0.1 * torch.randn(v.shape, generator=g)
        else:
            out[k] = 0.05 * torch.randn(v.shape, generator=g)
    return out


def inputs(seed: int = 5, B: int = 2):
    g = torch.Generator().manual_seed(seed)
    T, H = CFG["T"], 32
    lat = torch.randn(B, T * H * H, 4, generator=g) * 0.18215 * 4
    masked = torch.rand(B, T, H, H, generator=g) < 0.55
    masked[:, 0] = False
    n = B * T * 256
    return {"latents": lat, "masked": masked, "actions_domA": torch.randn(B, T, 7, generator=g),
            "t": torch.randint(0, 1000, (n,), generator=g), "noise": torch.randn(n, 16, generator=g)}
