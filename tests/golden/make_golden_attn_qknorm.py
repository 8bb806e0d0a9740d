#!/usr/bin/env python3
"""G3 with qk_norm=True: the stand-alone SelfAttention of the REAL reference (hma/model/attention.py:10-61) with its per-head LayerNorm
of q and k, forward and backward, both attention scales.  Build container only.

    python tests/golden/make_golden_attn_qknorm.py     # writes tests/golden/g3_attention_qknorm.safetensors"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
import torch  # noqa: E402


def g3_attention_qknorm():
    """G3 with qk_norm=True (attention.py:31-35,44-48: a LayerNorm(head_dim) with one shared affine on q and k), both attention scales,
    forward outputs and the gradients of a fixed upstream gradient (the stand-alone module is trainable)."""
    g = torch.Generator().manual_seed(33)
    out = {}
    w = dict(qkv=MG.rnd(g, 768, 256, std=0.08), proj_w=MG.rnd(g, 256, 256, std=0.08), proj_b=MG.rnd(g, 256, std=0.1),
             norm_w=1 + MG.rnd(g, 32, std=0.2), norm_b=MG.rnd(g, 32, std=0.2))
    out.update(w)  # (one set of weights for both scales: the fixture stays small)
    for use_mup in (True, False):
        att = MG.SelfAttention(num_heads=8, d_model=256, qkv_bias=False, proj_bias=True, qk_norm=True, use_mup=use_mup)
        with torch.no_grad():
            att.qkv.weight.copy_(w["qkv"])
            att.proj.weight.copy_(w["proj_w"])
            att.proj.bias.copy_(w["proj_b"])
            att.norm.weight.copy_(w["norm_w"])
            att.norm.bias.copy_(w["norm_b"])
        tag = "mup" if use_mup else "std"
        # (the production frame of 320 tokens under the training scale, 64 under the plain one)
        for kind, shape, causal in (("spatial", (1, 320 if use_mup else 64, 256), False), ("temporal", (3, 16, 256), True)):
            x = MG.rnd(g, *shape).requires_grad_(True)
            dy = MG.rnd(g, *shape)
            att.zero_grad(set_to_none=True)
            y = att(x, causal=causal)
            y.backward(dy)
            out.update({f"{tag}.x_{kind}": x.detach().clone(), f"{tag}.y_{kind}": y.detach().clone(), f"{tag}.dy_{kind}": dy,
                        f"{tag}.dx_{kind}": x.grad.clone(), f"{tag}.dqkv_w_{kind}": att.qkv.weight.grad[::8].clone(),
                        f"{tag}.dnorm_w_{kind}": att.norm.weight.grad.clone(), f"{tag}.dnorm_b_{kind}": att.norm.bias.grad.clone(),
                        f"{tag}.dproj_w_{kind}": att.proj.weight.grad[::4].clone(), f"{tag}.dproj_b_{kind}": att.proj.bias.grad.clone()})
    MG.save("g3_attention_qknorm", out)




if __name__ == "__main__":
    g3_attention_qknorm()
