"""Tiny model configuration + seeded inputs shared by make_golden.py and the tests."""
import math

import torch

TINY = {
    "seed": 7,
    "config": dict(num_layers=2, num_heads=8, d_model=256, T=3, S=256, image_vocab_size=262144,
                   use_mup=True, action_network="concat+modulate", num_factored_vocabs=2,
                   qkv_bias=False, proj_bias=True, attn_drop=0.0, qk_norm=False, mlp_ratio=4.0,
                   mlp_drop=0.0, mlp_bias=True),
    "domains": ["domA", "domB"],
    "d_actions": [7, 14],
    "action_stats": [
        [[0.05 * i for i in range(7)], [0.6 + 0.1 * i for i in range(7)]],
        [[-0.1 * i for i in range(7)], [1.0 + 0.05 * i for i in range(7)]],
    ],
}


def tiny_inputs(B: int = 2, seed: int = 11):
    """labels ~ U{0..8191}; inputs masked per (b,t) at rate cos(u*pi/2) on frames >= 1 (data.py:74-83)."""
    T = TINY["config"]["T"]
    g = torch.Generator().manual_seed(seed)
    labels = torch.randint(0, 8192, (B, T, 16, 16), generator=g)
    # include ids beyond 8191 so both factors see the whole range
    labels[:, :, 0, :4] = torch.randint(0, 262144, (B, T, 4), generator=g)
    u = torch.rand(B, T - 1, 1, 1, generator=g)
    prob = torch.cos(u * math.pi / 2)
    m = torch.rand(B, T - 1, 16, 16, generator=g) < prob
    inputs = labels.clone()
    inputs[:, 1:][m] = TINY["config"]["image_vocab_size"]
    return {
        "labels": labels.reshape(B, -1),
        "input_ids": inputs.reshape(B, -1),
        "actions_domA": torch.randn(B, T, 7, generator=g),
        "actions_domB": torch.randn(B, T, 14, generator=g),
    }
