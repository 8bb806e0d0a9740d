"""Diffusion head on CPU: the restatement in oracle/diffloss_ref.py against vectors captured from
the real reference (tests/golden/make_golden_diffloss.py)."""
import os

import numpy as np
import torch
from safetensors.torch import load_file

from oracle import diffloss_ref as R

G = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_diffloss.safetensors"))
DEPTH = 2


def params(requires_grad=False):
    return {k[3:]: v.clone().requires_grad_(requires_grad) for k, v in G.items() if k.startswith("sd.")}


def test_schedule_tables():
    tb = R.Tables.train()
    assert tb.n == 1000 and abs(tb.betas[0] - 4.128e-05) < 1e-6 and tb.betas.max() <= 0.999
    gen = R.Tables.sampling("10")
    assert gen.n == 10 and gen.timestep_map == [0, 111, 222, 333, 444, 555, 666, 777, 888, 999]
    assert np.isclose(np.prod(1 - gen.betas), np.prod(1 - R.cosine_betas()))  # same terminal alpha-bar


def test_training_loss_and_gradients_match_reference():
    P = params(True)
    z = G["z"].clone().requires_grad_(True)
    loss, out = R.diffloss_forward(P, G["target"], z, G["mask"], G["t"], G["noise"], DEPTH)
    assert torch.allclose(out, G["net_out"], rtol=1e-4, atol=1e-4)
    rows, _ = R.training_losses(R.Tables.train(), params(), G["target"], G["t"], G["noise"], G["z"], DEPTH)
    assert torch.allclose(rows, G["loss_rows"], rtol=1e-4, atol=1e-4)
    assert abs(loss.item() - G["loss"].item()) <= 1e-5 * abs(G["loss"].item())
    loss.backward()
    assert torch.allclose(z.grad, G["dz"], rtol=1e-3, atol=1e-6)
    for k, p in P.items():
        want = G[f"grad.{k}"]
        assert torch.allclose(p.grad, want, rtol=2e-3, atol=1e-5 * want.abs().max().item() + 1e-8), k


def test_sampling_loop_matches_reference():
    x = R.diffloss_sample(params(), G["s.z"], G["s.noise0"], list(G["s.draws"]), DEPTH, temperature=0.9, num_sampling_steps="10")
    assert torch.allclose(x, G["s.sample"], rtol=1e-4, atol=1e-4)


def test_sampling_with_classifier_free_guidance_matches_reference():
    x = R.diffloss_sample(params(), G["g.z"], G["g.half0"], list(G["g.draws"]), DEPTH, temperature=0.9, num_sampling_steps="10",
                          cfg=float(G["g.cfg"]))
    assert x.shape == G["g.sample"].shape and not torch.equal(x[:32], x[32:])  # the halves draw their own step noise
    assert torch.allclose(x, G["g.sample"], rtol=1e-4, atol=1e-4)
