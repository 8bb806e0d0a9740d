"""STMAR training forward / backward on the GPU against the reference's golden vectors (G11)."""
import os

import pytest
import torch
from safetensors.torch import load_file

from hma_amd.config import DiffusionGenieConfig
from hma_amd.model.st_mar import STMAR
from tests.golden.stmar_cfg import CFG, DOMAINS, D_ACTIONS, STATS, inputs, seeded_state

pytestmark = pytest.mark.gpu
DEV = "cuda"
HERE = os.path.dirname(os.path.abspath(__file__))
G = load_file(os.path.join(HERE, "golden", "g11_stmar.safetensors"))


def build():
    m = STMAR(DiffusionGenieConfig(**CFG))
    m.init_action_projectors(DOMAINS, D_ACTIONS, STATS, CFG["action_network"])
    return m


def rel(a, b):
    return ((a.float().cpu() - b).norm() / (b.norm() + 1e-12)).item()


def test_state_dict_matches_reference_names_and_shapes():
    want = {}
    for line in open(os.path.join(HERE, "golden", "g11_stmar_keys.txt")):
        name, shape = line.split(" ", 1)
        want[name] = tuple(eval(shape))
    got = {k: tuple(v.shape) for k, v in build().state_dict().items()}
    assert got == want, sorted(set(got) ^ set(want))[:10]


def test_forward_backward_match_reference():
    m = build()
    m.load_state_dict(seeded_state(m.state_dict()))
    m = m.to(DEV).train()
    inp = {k: v.to(DEV) for k, v in inputs().items()}
    out = m(input_ids=inp["latents"].clone(), labels=inp["latents"].clone(), action_ids=inp["actions_domA"], domain=["domA"] * 2,
            masked_tokens_indicator=inp["masked"], h=[32, 32], w=[32, 32], diffusion_t=inp["t"], diffusion_noise=inp["noise"])
    z = out.logits.permute(0, 2, 3, 4, 1).reshape(2, 3, 256, 256)
    assert rel(z, G["z"]) < 1e-2                                         # measured 3.7e-3
    assert abs(out.loss.item() - G["loss"].item()) <= 2e-3 * abs(G["loss"].item()), (out.loss.item(), G["loss"].item())  # measured 1e-4
    out.loss.backward()
    params = dict(m.named_parameters())
    worst = {}
    for k in G:
        if k.startswith("grad."):
            g = params[k[5:]].grad
            assert g is not None, k
            worst[k] = rel(g, G[k])
    bad = {k: v for k, v in worst.items() if v > 2e-2}              # worst measured 6.8e-3 (rms, relative)
    assert not bad, bad
    assert params["action_mlp.domB.model.0.weight"].grad is None  # the other domain's head is untouched


def test_train_steps_reduce_the_loss():
    """forward + backward + clip + AdamW (engine ranges for the trunk / active domain, per-tensor for the head) on a fixed batch."""
    m = build()
    m.load_state_dict(seeded_state(m.state_dict()))
    m = m.to(DEV).train()
    inp = {k: v.to(DEV) for k, v in inputs().items()}
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    losses = []
    for _ in range(5):
        m.zero_grad()
        out = m(input_ids=inp["latents"].clone(), labels=inp["latents"].clone(), action_ids=inp["actions_domA"], domain=["domA"] * 2,
                masked_tokens_indicator=inp["masked"], h=[32, 32], w=[32, 32], diffusion_t=inp["t"], diffusion_noise=inp["noise"])
        out.loss.backward()
        m.optimizer_step(2e-3, "domA")
        losses.append(out.loss.item())
    assert losses[-1] < 0.9 * losses[0], losses
    moved = {n for n, p in m.named_parameters() if not torch.equal(p.detach(), before[n])}
    assert {"token_embed.weight", "mask_token", "out_x_proj.weight", "diffloss.net.cond_embed.weight", "decoder.layers.0.mlp.fc1.weight",
            "action_mlp.domA.model.0.weight", "pos_embed_TSC"} <= moved
    assert not any("domB" in n for n in moved) and "action_mask_tokens" not in moved


def test_mar_decode_matches_reference():
    """maskgit_generate of one frame (2 MaskGIT steps x 10 reverse-diffusion steps) with the reference's order and draws."""
    D = load_file(os.path.join(HERE, "golden", "g12_stmar_decode.safetensors"))
    m = build()
    m.load_state_dict(seeded_state(m.state_dict()))
    m = m.to(DEV).eval()
    inp = {k: v.to(DEV) for k, v in inputs().items()}
    draws = [(D[f"noise0.{k}"].to(DEV), D[f"steps.{k}"].to(DEV)) for k in range(2)]
    frame, orig, _ = m.maskgit_generate(D["prompt"].to(DEV), 2, action_ids=inp["actions_domA"], domain=["domA"] * 2, maskgit_steps=2,
                                        temperature=0.9, orders=D["orders"].to(DEV), draws=draws)
    assert rel(orig, D["orig_latents"]) < 1e-2
    # 20 bf16 network evaluations through a sampler clamped to +-10 with seeded random weights: a few elements that sit
    # on the clamp flip side (measured: 22 of 8192 off by > 1), everything else follows the reference closely (measured:
    # median |err| 0.002, 99th percentile 0.083 at a mean magnitude of 7.5)
    assert frame.shape == D["frame"].shape
    err = (frame.cpu() - D["frame"]).abs().flatten()
    assert err.median().item() < 0.01 and err.quantile(0.99).item() < 0.25 and (err > 1).float().mean().item() < 0.01
    # rollout API: two new frames appended to a one-frame prompt, shapes and finiteness
    out = m.generate(D["prompt"][:, :1].reshape(2, -1, 4).to(DEV), None, max_new_tokens=2 * 1024, action_ids=inp["actions_domA"],
                     domain=["domA"] * 2, temperature=0.9)
    assert out.shape == (2, 3 * 1024, 4) and torch.isfinite(out).all()
    assert torch.equal(out[:, :1024].cpu(), D["prompt"][:, 0].reshape(2, 1024, 4))
    # the interactive caller's batch of ONE (round 6: the frame being written is then a contiguous view of the window)
    out1 = m.generate(D["prompt"][:1, :1].reshape(1, -1, 4).to(DEV), None, max_new_tokens=1024, action_ids=inp["actions_domA"][:1],
                      domain=["domA"], temperature=0.9)
    assert out1.shape == (1, 2 * 1024, 4) and torch.isfinite(out1).all()
    assert torch.equal(out1[:, :1024].cpu(), D["prompt"][:1, 0].reshape(1, 1024, 4))


def test_mlp_dropout_trains_and_eval_is_deterministic():
    """The shipped MAR config trains with mlp_drop = 0.05 (hma/configs/mar_n32_h8_d256_action.json): masks change per step in
    training, nothing is dropped in eval, and the fixed-batch loss still falls."""
    cfg = dict(CFG, mlp_drop=0.05)
    m = STMAR(DiffusionGenieConfig(**cfg))
    m.init_action_projectors(DOMAINS, D_ACTIONS, STATS, CFG["action_network"])
    m.load_state_dict(seeded_state(m.state_dict()))
    m = m.to(DEV)
    inp = {k: v.to(DEV) for k, v in inputs().items()}
    kw = dict(input_ids=inp["latents"], labels=inp["latents"], action_ids=inp["actions_domA"], domain=["domA"] * 2,
              masked_tokens_indicator=inp["masked"], h=[32, 32], w=[32, 32], diffusion_t=inp["t"], diffusion_noise=inp["noise"])
    m.eval()
    with torch.no_grad():
        e1, e2 = m(**kw).loss.item(), m(**kw).loss.item()
    # eval: no dropout -> the reference's (no-drop) value, repeatable up to the fp32 atomics order of the loss reduction
    assert abs(e1 - e2) <= 2e-6 * abs(e1) and abs(e1 - G["loss"].item()) <= 2e-3 * abs(G["loss"].item())
    m.train()
    with torch.no_grad():
        pass
    l1 = m(**kw).loss.item()
    l2 = m(**kw).loss.item()
    assert abs(l1 - l2) > 1e-4 * abs(l1) and abs(l1 - e1) < 0.2 * abs(e1)       # different masks per forward, same ballpark
    losses = []
    for _ in range(6):
        m.zero_grad()
        out = m(**kw)
        out.loss.backward()
        m.optimizer_step(2e-3, "domA")
        losses.append(out.loss.item())
    assert min(losses[-2:]) < 0.9 * losses[0], losses


def test_two_adamw_steps_match_reference():
    """G14 (tests/golden/make_golden_stmar_adamw.py): two clip + AdamW steps of the reference, optimizer grouped as
    train_multi.py:907-918 -- only names containing "bias" are un-decayed; domB, the action-diffusion heads and
    action_mask_tokens get no gradient and are never touched."""
    from hma_amd.train import MarTrainer
    g = load_file(os.path.join(HERE, "golden", "g14_stmar_adamw.safetensors"))
    untouched = set(open(os.path.join(HERE, "golden", "g14_stmar_untouched.txt")).read().split())
    m = build()
    state = seeded_state(m.state_dict())
    m.load_state_dict(state)
    m = m.to(DEV).train()
    tr = MarTrainer(m, lr=1e-3, warmup_steps=0)
    # decay grouping of the own flat range, name by name
    own = tr.own
    off = 0
    for name, pv in zip(own["names"], own["pviews"]):
        f = int(own["flags"][off // 64])
        assert f == (1 if "bias" in name else 2), name
        off += (pv.numel() + 63) // 64 * 64
    for n in ("decoder_norm.weight", "z_proj_ln.weight", "mask_token", "pos_embed_TSC", "diffloss.net.res_blocks.0.in_ln.weight"):
        assert n in own["names"]
    inp = {k: v.to(DEV) for k, v in inputs().items()}
    kw = dict(input_ids=inp["latents"], labels=inp["latents"], action_ids=inp["actions_domA"], domain=["domA"] * 2,
              masked_tokens_indicator=inp["masked"], h=[32, 32], w=[32, 32], diffusion_t=inp["t"], diffusion_noise=inp["noise"])
    for it in range(2):
        out = tr.step(**kw)
        ref = g[f"step{it}.loss"].item()
        assert abs(out.loss.item() - ref) <= (2e-3 if it == 0 else 2e-2) * abs(ref), (it, out.loss.item(), ref)  # step 1 sees the update
        norm = tr.engine.grad_norm().item()
        assert abs(norm - g[f"step{it}.grad_norm"].item()) <= 2e-2 * g[f"step{it}.grad_norm"].item(), (it, norm)
    params = dict(m.named_parameters())
    worst = 0.0
    for name, p in params.items():
        flat = p.detach().float().cpu().reshape(-1)
        idx = torch.linspace(0, flat.numel() - 1, 64).long()
        init = state[name].reshape(-1)[idx]
        if name in untouched:
            assert torch.equal(flat[idx], init), f"{name} must not be touched (no gradient: no decay, no moments)"
            continue
        ref = g[f"param_samp.{name}"]
        moved = (ref - init).double().pow(2).mean().sqrt().item()
        err = (flat[idx] - ref).double().pow(2).mean().sqrt().item()
        assert moved > 0, name
        worst = max(worst, err / moved)
        # Adam turns a gradient into +-lr steps: an element whose gradient is bf16 rounding noise can land one step away
        # (measured worst 0.35: temporal_attn.qkv.bias, whose k-bias third has a mathematically zero gradient -- softmax is shift invariant --
        # so its Adam update is +-lr on rounding noise)
        assert err <= 0.40 * moved, (name, err, moved)
    print("stmar adamw worst update err / moved", worst)


def test_mar_trainer_resume_roundtrip(tmp_path):
    """`MarTrainer.save_state` / `load_state`: Accelerate-layout optimizer.bin / scheduler.bin (train_multi.py:310-321) carry the Adam
    moments of both flat ranges and the update counts; a resumed trainer takes the same next step."""
    from hma_amd.train import MarTrainer, reference_param_groups
    m = build()
    m.load_state_dict(seeded_state(m.state_dict()))
    m = m.to(DEV).train()
    tr = MarTrainer(m, lr=1e-3, warmup_steps=0)
    inp = {k: v.to(DEV) for k, v in inputs().items()}
    kw = dict(input_ids=inp["latents"], labels=inp["latents"], action_ids=inp["actions_domA"], domain=["domA"] * 2,
              masked_tokens_indicator=inp["masked"], h=[32, 32], w=[32, 32], diffusion_t=inp["t"], diffusion_noise=inp["noise"])
    for _ in range(2):
        tr.step(**kw)
    tr.save_state(tmp_path)
    assert {"optimizer.bin", "scheduler.bin", "model.safetensors", "config.json"} <= set(os.listdir(tmp_path))
    osd = torch.load(tmp_path / "optimizer.bin", weights_only=False)
    order = sum(reference_param_groups([n for n, _ in m.named_parameters()]), [])
    st = {order[i]: v for i, v in osd["state"].items()}
    assert float(st["diffloss.net.cond_embed.weight"]["step"]) == 2.0 and float(st["decoder.layers.0.mlp.fc1.weight"]["step"]) == 2.0
    assert not any(n.startswith("action_diff_losses.") or ".domB." in n or n == "action_mask_tokens" for n in st)
    m2 = STMAR.from_pretrained(tmp_path).to(DEV).train()
    tr2 = MarTrainer(m2, lr=1e-3, warmup_steps=0)
    tr2.load_state(tmp_path)
    assert tr2.completed == 2 and tr2.engine.opt_step == 2 and tr2.engine.dom_steps == {"domA": 2}
    assert torch.equal(tr2.own["M"], tr.own["M"]) and torch.equal(tr2.own["V"], tr.own["V"])
    tr.step(**kw)
    tr2.step(**kw)
    worst = max((p1.detach() - p2.detach()).abs().max().item() for (_, p1), (_, p2) in zip(m.named_parameters(), m2.named_parameters()))
    assert worst <= 2.1e-3  # at most one Adam step apart where a gradient element is rounding noise


def test_diffusion_batch_mul_scores_every_token_at_several_draws():
    """diffusion_batch_mul = 2 (st_mar.py:133-140: target / z / mask rows repeated before the diffusion loss): with the draws of the two
    copies given, the masked-mean loss is the mean of the two single-draw losses and every gradient the mean of theirs."""
    inp = {k: v.to(DEV) for k, v in inputs().items()}
    gen = torch.Generator().manual_seed(5)
    t2 = torch.randint(0, 1000, inp["t"].shape, generator=gen).to(DEV)
    n2 = torch.randn(inp["noise"].shape, generator=gen).to(DEV)
    names = ["out_x_proj.weight", "diffloss.net.cond_embed.weight", "decoder.layers.1.mlp.fc2.weight", "token_embed.weight"]

    def run(mul, t, noise):
        m = STMAR(DiffusionGenieConfig(**dict(CFG, diffusion_batch_mul=mul)))
        m.init_action_projectors(DOMAINS, D_ACTIONS, STATS, CFG["action_network"])
        m.load_state_dict(seeded_state(m.state_dict()))
        m = m.to(DEV).train()
        out = m(input_ids=inp["latents"].clone(), labels=inp["latents"].clone(), action_ids=inp["actions_domA"], domain=["domA"] * 2,
                masked_tokens_indicator=inp["masked"], h=[32, 32], w=[32, 32], diffusion_t=t, diffusion_noise=noise)
        out.loss.backward()
        p = dict(m.named_parameters())
        return out.loss.item(), {n: p[n].grad.detach().float().cpu().clone() for n in names}

    la, ga = run(1, inp["t"], inp["noise"])
    lb, gb = run(1, t2, n2)
    l2, g2 = run(2, torch.cat([inp["t"], t2]), torch.cat([inp["noise"], n2]))
    assert abs(la - lb) > 1e-4 * abs(la)  # the draws matter
    assert abs(l2 - 0.5 * (la + lb)) <= 2e-4 * abs(l2), (l2, la, lb)
    for n in names:
        ref = 0.5 * (ga[n] + gb[n])
        assert rel(g2[n], ref) < 2e-2, (n, rel(g2[n], ref))


def test_mar_trainer_segmented_backward_equals_plain():
    """MarTrainer's data-parallel backward (hooks: the diffusion head's range reduced when the head backward is done, the trunk bucket
    by bucket from `trunk_train_backward(on_segment=...)`, the input / output stages last) computes the same step as the plain one."""
    from hma_amd.train import MarTrainer
    inp = {k: v.to(DEV) for k, v in inputs().items()}
    kw = dict(input_ids=inp["latents"], labels=inp["latents"], action_ids=inp["actions_domA"], domain=["domA"] * 2,
              masked_tokens_indicator=inp["masked"], h=[32, 32], w=[32, 32], diffusion_t=inp["t"], diffusion_noise=inp["noise"])
    res = []
    for overlap in (False, True):
        m = build()
        m.load_state_dict(seeded_state(m.state_dict()))
        m = m.to(DEV).train()
        tr = MarTrainer(m, lr=1e-3, warmup_steps=0, layers_per_bucket=1, grad_accum=2)
        tr.force_overlap = overlap
        for _ in range(2):
            tr.micro_step(**kw)
            tr.micro_step(**kw)
            tr.optimizer_step()
        assert tr.early_launches == (2 * 3 if overlap else 0)  # per step: head + two one-layer buckets, last micro-batch only
        res.append({n: p.detach().clone() for n, p in m.named_parameters()})
    worst = max((res[0][n] - res[1][n]).abs().max().item() for n in res[0])
    assert worst <= 2.1e-3, worst  # (atomics order: an Adam step of +-lr on a noise-level gradient element can flip)
    m0 = build()
    init = {k: v.to(DEV) for k, v in seeded_state(m0.state_dict()).items()}
    for n in res[0]:
        moved = (res[0][n] - init[n]).double().pow(2).mean().sqrt().item()
        err = (res[0][n] - res[1][n]).double().pow(2).mean().sqrt().item()
        # run-to-run noise of the fp32 atomics: far below the update itself, except for vectors whose gradient IS noise (the k third
        # of temporal_attn.qkv.bias: softmax is shift invariant), where +-lr Adam steps flip
        assert err <= (0.3 if res[0][n].numel() <= 1024 else 0.05) * moved + 1e-12, (n, err, moved)


def test_jointly_predict_actions_matches_reference():
    """G17 (make_golden_stmar_jpa.py, the real reference): the per-domain action diffusion head on the mean-pooled action tokens
    (st_mar.py:119-129, 187-189, 231-273); gradients of loss + action_loss; then MarTrainer steps it (only the active domain's head)."""
    from hma_amd.train import MarTrainer
    G17 = load_file(os.path.join(HERE, "golden", "g17_stmar_jpa.safetensors"))
    m = STMAR(DiffusionGenieConfig(**dict(CFG, jointly_predict_actions=True)))
    m.init_action_projectors(DOMAINS, D_ACTIONS, STATS, CFG["action_network"])
    m.load_state_dict(seeded_state(m.state_dict()))
    m = m.to(DEV).train()
    inp = {k: v.to(DEV) for k, v in inputs().items()}
    kw = dict(input_ids=inp["latents"].clone(), labels=inp["latents"].clone(), action_ids=inp["actions_domA"], domain=["domA"] * 2,
              masked_tokens_indicator=inp["masked"], h=[32, 32], w=[32, 32], diffusion_t=inp["t"], diffusion_noise=inp["noise"],
              action_mask=G17["action_mask"].to(DEV), action_diffusion_t=G17["t_act"].to(DEV), action_diffusion_noise=G17["noise_act"].to(DEV))
    out = m(**kw)
    assert abs(out.loss.item() - G17["loss"].item()) <= 2e-3 * abs(G17["loss"].item())
    assert abs(out.action_loss.item() - G17["action_loss"].item()) <= 5e-3 * abs(G17["action_loss"].item()), (out.action_loss.item(), G17["action_loss"].item())
    assert rel(out.actions, G17["actions"]) < 1e-2
    (out.loss + out.action_loss).backward()
    params = dict(m.named_parameters())
    bad = {}
    for k in G17:
        if k.startswith("grad."):
            g = params[k[5:]].grad
            assert g is not None, k
            e = rel(g, G17[k])
            if e > 3e-2:
                bad[k] = e
    assert not bad, bad
    assert params["action_diff_losses.domB.net.cond_embed.weight"].grad is None
    # trainer: loss + action_loss_weight * action_loss; the active domain's head moves, the other one does not
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    tr = MarTrainer(m, lr=1e-3, warmup_steps=0)
    l0 = None
    for _ in range(3):
        o = tr.step(**kw)
        l0 = l0 if l0 is not None else (o.loss.item(), o.action_loss.item())
    assert o.action_loss.item() < l0[1] and o.loss.item() < l0[0], (l0, o.loss.item(), o.action_loss.item())
    moved = {n for n, p in m.named_parameters() if not torch.equal(p.detach(), before[n])}
    assert "action_diff_losses.domA.net.cond_embed.weight" in moved and "action_diff_losses.domA.net.final_layer.linear.bias" in moved
    assert not any(n.startswith("action_diff_losses.domB") for n in moved)
    # without the action loss in the objective nothing of the action head gets a gradient
    m.zero_grad()
    o2 = m(**kw)
    o2.loss.backward()
    g = dict(m.named_parameters())["action_diff_losses.domA.net.cond_embed.weight"].grad
    assert g is None or float(g.abs().sum()) == 0.0


def test_stmar_with_qk_norm_against_oracle():
    """STMAR on a qk_norm=True trunk (attention.py:31-35,44-48: per-head LayerNorm of q and k, identity norm1 / norm2), the tiny fixture's
    shapes: loss, latents and gradients against oracle/st_mar_ref.py (whose trunk is pinned for qk_norm by G18) on the host."""
    from oracle import st_mar_ref as MR
    from oracle import st_maskgit_ref as R
    m = STMAR(DiffusionGenieConfig(**dict(CFG, qk_norm=True)))
    m.init_action_projectors(DOMAINS, D_ACTIONS, STATS, CFG["action_network"])
    sd = seeded_state(m.state_dict())
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    inp = inputs()
    rc = R.RefConfig(num_layers=2, num_heads=8, d_model=256, T=3, S=1024, use_mup=True, qkv_bias=True, mlp_bias=False, qk_norm=True)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()
            if v.is_floating_point() and not (k.endswith(".mean") or k.endswith(".std")) and not k.startswith("action_diff_losses")}
    full = dict(sd)
    full.update(leaf)
    loss_ref, z_ref = MR.forward(full, rc, inp["latents"], inp["latents"], inp["actions_domA"], ["domA"] * 2, inp["masked"], inp["t"],
                                 inp["noise"], 2, 32, 32, CFG["diffloss_d"])
    loss_ref.backward()
    d = {k: v.to(DEV) for k, v in inp.items()}
    out = m(input_ids=d["latents"].clone(), labels=d["latents"].clone(), action_ids=d["actions_domA"], domain=["domA"] * 2,
            masked_tokens_indicator=d["masked"], h=[32, 32], w=[32, 32], diffusion_t=d["t"], diffusion_noise=d["noise"])
    assert abs(out.loss.item() - loss_ref.item()) <= 2e-3 * abs(loss_ref.item())
    z = out.logits.permute(0, 2, 3, 4, 1).reshape(2, 3, 256, 256)
    assert rel(z, z_ref.detach()) < 1e-2
    out.loss.backward()
    named = dict(m.named_parameters())
    assert "decoder.layers.0.temporal_attn.norm.weight" in named and "decoder.layers.0.norm2.weight" not in named
    bad = {}
    for k, p in leaf.items():
        if p.grad is None or float(p.grad.abs().sum()) == 0.0 or named[k].grad is None:
            continue
        e = rel(named[k].grad, p.grad)
        if e > 3e-2:
            bad[k] = e
    assert not bad, bad


def test_mar_decode_samples_actions_with_jointly_predict_actions():
    """st_mar.py:441-446: with jointly_predict_actions the MAR decode also samples actions from the domain's action diffusion head,
    conditioned on the mean-pooled action tokens of the window; third return value (B * T, d_action).  The two pieces are pinned
    separately (pooled tokens: G17's `actions`; DiffLoss.sample: G9 / G12) -- here: the wiring, with injected draws."""
    D = load_file(os.path.join(HERE, "golden", "g12_stmar_decode.safetensors"))
    m = STMAR(DiffusionGenieConfig(**dict(CFG, jointly_predict_actions=True)))
    m.init_action_projectors(DOMAINS, D_ACTIONS, STATS, CFG["action_network"])
    m.load_state_dict(seeded_state(m.state_dict()))
    m = m.to(DEV).eval()
    inp = {k: v.to(DEV) for k, v in inputs().items()}
    draws = [(D[f"noise0.{k}"].to(DEV), D[f"steps.{k}"].to(DEV)) for k in range(2)]
    g = torch.Generator().manual_seed(1)
    n_steps = D["steps.0"].shape[0]
    adraws = [(torch.randn(6, 7, generator=g).to(DEV), torch.randn(n_steps, 6, 7, generator=g).to(DEV)) for _ in range(2)]
    kw = dict(action_ids=inp["actions_domA"], domain=["domA"] * 2, maskgit_steps=2, temperature=0.9, orders=D["orders"].to(DEV), draws=draws)
    frame, orig, acts = m.maskgit_generate(D["prompt"].to(DEV), 2, action_draws=adraws, **kw)
    assert acts.shape == (6, 7) and torch.isfinite(acts).all()
    assert rel(orig, D["orig_latents"]) < 1e-2          # the video path is what it was without action prediction
    frame2, _, acts2 = m.maskgit_generate(D["prompt"].to(DEV), 2, action_draws=adraws, **kw)
    assert torch.equal(acts, acts2) and torch.equal(frame, frame2)
    # equals the head's own sampler on the pooled action tokens of the final window
    # (a step conditions on the window BEFORE its own write: a one-step decode's actions depend on the prompt and the action draws only)
    _, pooled0 = m.compute_latents(m.patchify(D["prompt"].to(DEV)), action_ids=inp["actions_domA"], domain=["domA"] * 2)
    direct0 = m.action_diff_losses["domA"].sample(pooled0.reshape(-1, 256), 0.9, 1.0, clip_denoised=True, noise0=adraws[0][0], step_noises=adraws[0][1])
    _, _, acts_1step = m.maskgit_generate(D["prompt"].to(DEV), 2, action_draws=adraws[:1], **dict(kw, maskgit_steps=1, draws=None))
    assert torch.equal(acts_1step, direct0)


def test_stmar_without_action_conditioning_matches_reference():
    """VERDICT round 3, missing 5: `STMAR.forward(action_ids=None)` (hma/model/st_mar.py:146-197: no action tokens, the decoder
    unconditioned) forward + backward against G11b from the real reference; `compute_latents` without actions gives the same latents."""
    Gn = load_file(os.path.join(HERE, "golden", "g11b_stmar_noact.safetensors"))
    m = build()
    m.load_state_dict(seeded_state(m.state_dict()))
    m = m.to(DEV).train()
    inp = {k: v.to(DEV) for k, v in inputs().items()}
    out = m(input_ids=inp["latents"].clone(), labels=inp["latents"].clone(), action_ids=None, domain=None,
            masked_tokens_indicator=inp["masked"], h=[32, 32], w=[32, 32], diffusion_t=inp["t"], diffusion_noise=inp["noise"])
    z = out.logits.permute(0, 2, 3, 4, 1).reshape(2, 3, 256, 256)
    assert rel(z[:, :, ::4], Gn["z"]) < 1e-2
    assert abs(out.loss.item() - Gn["loss"].item()) <= 2e-3 * abs(Gn["loss"].item()), (out.loss.item(), Gn["loss"].item())
    out.loss.backward()
    params = dict(m.named_parameters())
    sub = lambda g: g[::4] if g.dim() == 2 and g.shape[0] >= 256 else g[:, ::4] if g.dim() == 3 else g
    bad = {}
    for k, want in Gn.items():
        if not k.startswith("grad."):
            continue
        name = k[5:]
        got = params["pos_embed_TSC"].grad[:, :, :256:4] if name == "pos_embed_TSC.image_rows" else sub(params[name].grad)
        e = rel(got, want)
        if e > 2e-2:
            bad[k] = e
    assert not bad, bad
    assert float(params["pos_embed_TSC"].grad[:, :, 256:].abs().max()) == 0.0            # no action rows took part
    g_a = params["action_mlp.domA.model.0.weight"].grad
    assert g_a is None or float(g_a.abs().max()) == 0.0
    # decode-side entry point on the same (masked) inputs
    m.eval()
    with torch.no_grad():
        x = inp["latents"].reshape(2, 3, 32, 32, 4).clone()
        x[inp["masked"]] = m.mask_token.reshape(-1)
        zl, pooled = m.compute_latents(m.patchify(x), action_ids=None, domain=None)
    assert pooled is None and rel(zl.permute(0, 2, 3, 4, 1).reshape(2, 3, 256, 256)[:, :, ::4], Gn["z"]) < 1e-2
