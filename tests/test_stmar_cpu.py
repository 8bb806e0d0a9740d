"""STMAR training forward on CPU: oracle/st_mar_ref.py against the reference's loss, latents and
gradients (tests/golden/make_golden_stmar.py; the state dict is regenerated from a seed on both sides)."""
import os

import torch
from safetensors.torch import load_file

from oracle import st_mar_ref as M
from oracle import st_maskgit_ref as R
from tests.golden.stmar_cfg import CFG, inputs, seeded_state

HERE = os.path.dirname(os.path.abspath(__file__))
G = load_file(os.path.join(HERE, "golden", "g11_stmar.safetensors"))


def template():
    out = {}
    for line in open(os.path.join(HERE, "golden", "g11_stmar_keys.txt")):
        name, shape = line.split(" ", 1)
        out[name] = torch.zeros(eval(shape))
    # buffers keep their constructor values (action statistics)
    from tests.golden.stmar_cfg import STATS, DOMAINS
    for dom, st in zip(DOMAINS, STATS):
        out[f"action_preprocessor.{dom}.mean"] = torch.tensor(st[0])
        out[f"action_preprocessor.{dom}.std"] = torch.tensor(st[1])
    return out


def ref_cfg():
    return R.RefConfig(num_layers=2, num_heads=8, d_model=256, T=3, S=1024, use_mup=True, qkv_bias=True, mlp_bias=False)


def test_stmar_forward_and_gradients_match_reference():
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not (k.endswith(".mean") or k.endswith(".std")) else v)
          for k, v in seeded_state(template()).items()}
    inp = inputs()
    loss, z = M.forward(sd, ref_cfg(), inp["latents"], inp["latents"], inp["actions_domA"], ["domA"] * 2, inp["masked"], inp["t"],
                        inp["noise"], 2, 32, 32, CFG["diffloss_d"])
    assert torch.allclose(z, G["z"], rtol=1e-3, atol=2e-4)
    assert abs(loss.item() - G["loss"].item()) <= 1e-4 * abs(G["loss"].item())
    loss.backward()
    for k in G:
        if k.startswith("grad.") :
            want, got = G[k], sd[k[5:]].grad
            assert torch.allclose(got, want, rtol=5e-3, atol=2e-5 * want.abs().max().item() + 1e-9), k
    assert G["grad_is_none.domB"].item() == 1.0 and sd["action_mlp.domB.model.0.weight"].grad is None


def test_mar_decode_matches_reference():
    """maskgit_generate of one frame (2 MaskGIT steps x 10 diffusion steps) with the reference's order and Gaussian draws."""
    D = load_file(os.path.join(HERE, "golden", "g12_stmar_decode.safetensors"))
    sd = seeded_state(template())
    inp = inputs()
    draws = [(D[f"noise0.{k}"], D[f"steps.{k}"]) for k in range(2)]
    frame, orig = M.maskgit_generate(sd, ref_cfg(), D["prompt"], 2, 2, 0.9, inp["actions_domA"], ["domA"] * 2, D["orders"], draws, 2,
                                     CFG["diffloss_d"], CFG["num_sampling_steps"])
    assert torch.allclose(orig.permute(0, 2, 1).reshape(2, 256, 16, 16), D["orig_latents"], rtol=1e-3, atol=2e-4)
    assert torch.allclose(frame, D["frame"], rtol=1e-3, atol=1e-3)


def test_stmar_jointly_predict_actions_matches_reference():
    """G17 (make_golden_stmar_jpa.py): the per-domain action diffusion head on the pooled action tokens (st_mar.py:119-129, 187-189,
    231-273), gradients of loss + action_loss."""
    G17 = load_file(os.path.join(HERE, "golden", "g17_stmar_jpa.safetensors"))
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not (k.endswith(".mean") or k.endswith(".std")) else v)
          for k, v in seeded_state(template()).items()}
    inp = inputs()
    loss, z, aloss, pooled = M.forward_with_actions(sd, ref_cfg(), inp["latents"], inp["latents"], inp["actions_domA"], ["domA"] * 2,
                                                    inp["masked"], inp["t"], inp["noise"], 2, 32, 32, CFG["diffloss_d"], G17["action_mask"],
                                                    G17["t_act"], G17["noise_act"])
    assert torch.allclose(z, G17["z"], rtol=1e-3, atol=2e-4)
    assert torch.allclose(pooled, G17["actions"], rtol=1e-3, atol=2e-4)
    assert abs(loss.item() - G17["loss"].item()) <= 1e-4 * abs(G17["loss"].item())
    assert abs(aloss.item() - G17["action_loss"].item()) <= 1e-4 * abs(G17["action_loss"].item())
    (loss + aloss).backward()
    for k in G17:
        if k.startswith("grad."):
            want, got = G17[k], sd[k[5:]].grad
            assert torch.allclose(got, want, rtol=5e-3, atol=2e-5 * want.abs().max().item() + 1e-9), k
    assert G17["grad_is_none.domB_head"].item() == 1.0 and sd["action_diff_losses.domB.net.cond_embed.weight"].grad is None


def _sub(name, g):
    """the row subsets tests/golden/make_golden_stmar_noact.py keeps"""
    if name == "pos_embed_TSC.image_rows":
        return None
    return g[::4] if g.dim() == 2 and g.shape[0] >= 256 else g[:, ::4] if g.dim() == 3 else g


def test_stmar_without_actions_matches_reference():
    """`action_ids=None` (hma/model/st_mar.py:146-197: no action tokens, no modulation): the oracle against G11b from the real reference."""
    from safetensors.torch import load_file
    Gn = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g11b_stmar_noact.safetensors"))
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not (k.endswith(".mean") or k.endswith(".std")) else v)
          for k, v in seeded_state(template()).items()}
    inp = inputs()
    loss, z = M.forward(sd, ref_cfg(), inp["latents"], inp["latents"], None, None, inp["masked"], inp["t"], inp["noise"], 2, 32, 32,
                        CFG["diffloss_d"])
    assert torch.allclose(z[:, :, ::4], Gn["z"], rtol=1e-3, atol=2e-4)
    assert abs(loss.item() - Gn["loss"].item()) <= 1e-4 * abs(Gn["loss"].item())
    loss.backward()
    for k, want in Gn.items():
        if not k.startswith("grad."):
            continue
        name = k[5:]
        got = sd["pos_embed_TSC"].grad[:, :, :256:4] if name == "pos_embed_TSC.image_rows" else _sub(name, sd[name].grad)
        assert torch.allclose(got, want, rtol=5e-3, atol=2e-5 * want.abs().max().item() + 1e-9), k
    assert sd["action_mlp.domA.model.0.weight"].grad is None
