"""Pin the CPU oracle against every golden vector captured from the real reference (CPU only)."""
import torch
import pytest

from oracle import st_maskgit_ref as R
from tests.golden.golden_cfg import TINY  # noqa: E402
from tests.helpers import golden, tiny_ref_config, tiny_state_dict, tiny_inputs, rel_err

TOL = 2e-5  # fp32 restatement vs fp32 reference: summation-order noise only


def test_g1_factorize_bit_exact():
    g = golden("g1_factorize")
    fac = R.factorize_token_ids(g["ids"], 2, 512)
    assert torch.equal(fac, g["fac"])
    assert torch.equal(R.unfactorize_token_ids(fac, 2, 512), g["unfac"])
    assert torch.equal(R.unfactorize_token_ids(fac, 2, 512), g["ids"])
    assert torch.equal(R.factorize_labels(g["ids"]), g["labels"])


def test_g2_embedding():
    g = golden("g2_embedding")
    cfg = tiny_ref_config()
    sd = {"token_embed.factored_embeds.0.weight": g["E0"], "token_embed.factored_embeds.1.weight": g["E1"],
          "token_embed.mask_token_embed": g["mask_embed"]}
    assert torch.equal(R.token_embed(sd, cfg, g["ids"]), g["out"])


@pytest.mark.parametrize("tag,scale", [("mup", 8 / 32), ("std", 32 ** -0.5)])
def test_g3_attention(tag, scale):
    g = golden("g3_attention")
    for kind, causal in (("spatial", False), ("temporal", True)):
        y = R.self_attention(g[f"{tag}.x_{kind}"], g[f"{tag}.qkv"], None, g[f"{tag}.proj_w"], g[f"{tag}.proj_b"],
                             8, scale, causal)
        assert rel_err(y, g[f"{tag}.y_{kind}"]) < TOL


@pytest.mark.parametrize("tag,scale", [("mup", 8 / 32), ("std", 32 ** -0.5)])
def test_g3_attention_qknorm(tag, scale):
    """The oracle's qk_norm branch against the stand-alone reference module with qk_norm=True (forward and, through autograd, the
    gradients of x, the qkv / proj weights and the per-head LayerNorm's affine)."""
    g = golden("g3_attention_qknorm")
    for kind, causal in (("spatial", False), ("temporal", True)):
        leaf = {k: g[k].clone().requires_grad_(True) for k in ("qkv", "proj_w", "proj_b", "norm_w", "norm_b")}
        x = g[f"{tag}.x_{kind}"].clone().requires_grad_(True)
        y = R.self_attention(x, leaf["qkv"], None, leaf["proj_w"], leaf["proj_b"], 8, scale, causal, qk_norm=(leaf["norm_w"], leaf["norm_b"]))
        assert rel_err(y, g[f"{tag}.y_{kind}"]) < TOL
        y.backward(g[f"{tag}.dy_{kind}"])
        assert rel_err(x.grad, g[f"{tag}.dx_{kind}"]) < TOL
        assert rel_err(leaf["qkv"].grad[::8], g[f"{tag}.dqkv_w_{kind}"]) < TOL
        assert rel_err(leaf["proj_w"].grad[::4], g[f"{tag}.dproj_w_{kind}"]) < TOL
        assert rel_err(leaf["norm_w"].grad, g[f"{tag}.dnorm_w_{kind}"]) < 5 * TOL
        assert rel_err(leaf["norm_b"].grad, g[f"{tag}.dnorm_b_{kind}"]) < 5 * TOL


def test_g4_blocks():
    g = golden("g4_blocks")
    y = R.mlp(g["mlp.x"], g["mlp.fc1_w"], g["mlp.fc1_b"], g["mlp.fc2_w"], g["mlp.fc2_b"])
    assert rel_err(y, g["mlp.y"]) < TOL
    sd = {f"m.{k[4:]}": v for k, v in g.items() if k.startswith("mod.") and k[4:] not in ("x", "c", "y")}
    B, S, T = 2, 5, 4
    y = R.modulate_layer(sd, "m", g["mod.x"].reshape(B, S, T, 256), g["mod.c"]).reshape(B * S, T, 256)
    assert rel_err(y, g["mod.y"]) < TOL
    cfg = tiny_ref_config()
    sd = {"action_preprocessor.d.mean": g["stem.mean"], "action_preprocessor.d.std": g["stem.std"]}
    for k, v in g.items():
        if k.startswith("stem.model."):
            sd["action_mlp.d." + k[5:]] = v
    assert rel_err(R.action_stem(sd, cfg, g["stem.a"], "d"), g["stem.y"]) < TOL


def test_g5_stblock_and_decoder():
    g = golden("g5_stblock")
    cfg = tiny_ref_config()
    sd = tiny_state_dict(cfg)
    y = R.st_block(sd, cfg, 0, g["x"], g["a_emb"], "domA")
    assert rel_err(y[:, :, ::8], g["y_layer0_domA"]) < TOL
    y = R.st_block(sd, cfg, 0, g["x"][:, :, :256], None, None)
    assert rel_err(y[:, :, ::8], g["y_layer0_noact"]) < TOL
    x = g["x"]
    for l in range(cfg.num_layers):
        x = R.st_block(sd, cfg, l, x, g["a_emb"], "domB")
    assert rel_err(x[:, :, ::8], g["y_decoder_domB"]) < TOL


@pytest.mark.parametrize("tag", ["domA", "domB", "noact"])
def test_g6_forward_backward(tag):
    g = golden("g6_forward_backward")
    cfg = tiny_ref_config()
    sd = tiny_state_dict(cfg)
    inp = tiny_inputs()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if not (k.endswith(".mean") or k.endswith(".std"))}
    full = dict(sd)
    full.update(params)
    act = None if tag == "noact" else inp[f"actions_{tag}"]
    dom = None if tag == "noact" else [tag] * 2
    loss, acc, logits = R.forward(full, cfg, inp["input_ids"], inp["labels"], act, dom)
    loss.backward()
    assert abs(loss.item() - g[f"{tag}.loss"].item()) < 1e-5
    assert acc.item() == g[f"{tag}.acc"].item()
    assert rel_err(logits.detach()[:, :, :, ::4, ::4], g[f"{tag}.logits_sub"]) < TOL
    n_checked = 0
    for name, p in params.items():
        key = f"{tag}.grad_head.{name}"
        if key not in g:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, name
            continue
        n_checked += 1
        gf = p.grad.reshape(-1)
        idx = torch.linspace(0, gf.numel() - 1, 64).long()
        scale = g[f"{tag}.grad_abs.{name}"].item() / gf.numel() + 1e-12
        assert (gf[:64] - g[key]).abs().max().item() < 5e-4 * scale + 1e-9, name
        assert (gf[idx] - g[f"{tag}.grad_samp.{name}"]).abs().max().item() < 5e-4 * scale + 1e-9, name
        assert abs(gf.double().abs().sum().item() - g[f"{tag}.grad_abs.{name}"].item()) < 1e-4 * gf.numel() * scale
    assert n_checked > 20


@pytest.mark.parametrize("tag", ["domA", "domB"])
def test_g16_jointly_predict_actions(tag):
    """jointly_predict_actions=True (st_mask_git.py:656-660, 676-678, 724-733): masked action tokens, the pooled action read-out and the
    reference's action loss (legacy `reduce="none"` = mean, times the masked fraction), gradients of loss + 0.5 action_loss."""
    g = golden("g16_jpa")
    cfg = tiny_ref_config()
    sd = tiny_state_dict(cfg)
    inp = tiny_inputs()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if not (k.endswith(".mean") or k.endswith(".std"))}
    full = dict(sd)
    full.update(params)
    loss, acc, logits, aloss, actions = R.forward_with_actions(full, cfg, inp["input_ids"], inp["labels"], inp[f"actions_{tag}"], [tag] * 2,
                                                               g[f"{tag}.action_mask"])
    (loss + 0.5 * aloss).backward()
    assert abs(loss.item() - g[f"{tag}.loss"].item()) < 1e-5
    assert abs(aloss.item() - g[f"{tag}.action_loss"].item()) < 1e-5 * max(1.0, abs(g[f"{tag}.action_loss"].item()))
    assert acc.item() == g[f"{tag}.acc"].item()
    assert rel_err(actions.detach(), g[f"{tag}.actions"]) < TOL
    assert rel_err(logits.detach()[:, :, :, ::4, ::4], g[f"{tag}.logits_sub"]) < TOL
    seen = set()
    for name, p in params.items():
        key = f"{tag}.grad_samp.{name}"
        if key not in g:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, name
            continue
        seen.add(name)
        gf = p.grad.reshape(-1)
        idx = torch.linspace(0, gf.numel() - 1, 64).long()
        scale = g[f"{tag}.grad_abs.{name}"].item() / gf.numel() + 1e-12
        assert (gf[idx] - g[key]).abs().max().item() < 5e-4 * scale + 1e-9, name
    assert {"action_mask_tokens", f"action_out_projectors.{tag}.weight", f"action_out_projectors.{tag}.bias"} <= seen


def test_g16_policy_mode():
    """jointly_predict_actions without action ids (st_mask_git.py:663-666): mask tokens in, predicted actions out."""
    g = golden("g16_jpa")
    cfg = tiny_ref_config()
    sd = tiny_state_dict(cfg)
    inp = tiny_inputs()
    with torch.no_grad():
        logits, actions = R.compute_logits_and_actions(sd, cfg, inp["input_ids"].reshape(2, cfg.T, 16, 16)[:1], None, ["domB"], None)
    assert rel_err(actions, g["policy.actions"]) < TOL
    assert rel_err(logits[:, :, :, ::4, ::4], g["policy.logits_sub"]) < TOL


@pytest.mark.parametrize("tag", ["domA", "noact"])
def test_g18_qk_norm(tag):
    """qk_norm=True (attention.py:31-35,44-48; st_transformer.py:55,62): per-head LayerNorm of q and k, identity norm1 / norm2."""
    from oracle.param_spec import seeded_state_dict, state_dict_spec
    g = golden("g18_qknorm")
    cfg = tiny_ref_config(qk_norm=True)
    sd = seeded_state_dict(state_dict_spec(cfg, TINY["domains"], TINY["d_actions"], [len(s[0]) for s in TINY["action_stats"]]),
                           seed=TINY["seed"] + 2, std=0.02, embed_std=0.02)
    for dom, st in zip(TINY["domains"], TINY["action_stats"]):
        sd[f"action_preprocessor.{dom}.mean"] = torch.tensor(st[0], dtype=torch.float32)
        sd[f"action_preprocessor.{dom}.std"] = torch.tensor(st[1], dtype=torch.float32)
    inp = tiny_inputs()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if not (k.endswith(".mean") or k.endswith(".std"))}
    full = dict(sd)
    full.update(params)
    act = None if tag == "noact" else inp[f"actions_{tag}"]
    dom = None if tag == "noact" else [tag] * 2
    loss, acc, logits = R.forward(full, cfg, inp["input_ids"], inp["labels"], act, dom)
    loss.backward()
    assert abs(loss.item() - g[f"{tag}.loss"].item()) < 1e-5
    assert acc.item() == g[f"{tag}.acc"].item()
    assert rel_err(logits.detach()[:, :, :, ::4, ::4], g[f"{tag}.logits_sub"]) < TOL
    seen = 0
    for name, p in params.items():
        key = f"{tag}.grad_samp.{name}"
        if key not in g:
            continue
        seen += 1
        gf = p.grad.reshape(-1)
        idx = torch.linspace(0, gf.numel() - 1, 64).long()
        scale = g[f"{tag}.grad_abs.{name}"].item() / gf.numel() + 1e-12
        assert (gf[idx] - g[key]).abs().max().item() < 5e-4 * scale + 1e-9, name
    assert seen > 20 and f"{tag}.grad_samp.decoder.layers.0.spatial_attn.norm.weight" in g


def test_g6b_initlike_forward_backward():
    g = golden("g6b_initlike")
    cfg = tiny_ref_config()
    sd = tiny_state_dict(cfg, initlike=True)
    inp = tiny_inputs()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if not (k.endswith(".mean") or k.endswith(".std"))}
    full = dict(sd)
    full.update(params)
    loss, acc, logits = R.forward(full, cfg, inp["input_ids"], inp["labels"], inp["actions_domA"], ["domA"] * 2)
    loss.backward()
    assert abs(loss.item() - g["domA.loss"].item()) < 1e-5
    assert rel_err(logits.detach()[:, :, :, ::4, ::4], g["domA.logits_sub"]) < TOL
    for name, p in params.items():
        key = f"domA.grad_samp.{name}"
        if key in g:
            gf = p.grad.reshape(-1)
            idx = torch.linspace(0, gf.numel() - 1, 64).long()
            scale = g[f"domA.grad_abs.{name}"].item() / gf.numel() + 1e-12
            assert (gf[idx] - g[key]).abs().max().item() < 5e-4 * scale + 1e-9, name


def test_g7_generate_ids_bit_exact():
    g = golden("g7_generate")
    cfg = tiny_ref_config()
    sd = tiny_state_dict(cfg)
    inp = tiny_inputs()
    out_t = cfg.T - 1
    for steps in (1, 2, 8):
        p = g["prompt0"].clone()
        s, fl = R.maskgit_generate(sd, cfg, p, out_t, steps, 0.0, "greedy", inp["actions_domA"], ["domA"] * 2)
        assert torch.equal(s, g[f"greedy{steps}.samples"]), steps
        assert torch.equal(p, g[f"greedy{steps}.prompt_after"])
        assert rel_err(fl[:, ::8], g[f"greedy{steps}.logits_sub"]) < TOL
    p = g["prompt0"].clone()
    s, _ = R.maskgit_generate(sd, cfg, p, out_t, 4, 0.0, "random", inp["actions_domA"], ["domA"] * 2,
                              rand_draws=list(g["random4.draws"]))
    assert torch.equal(s, g["random4.samples"])


def test_g8_clip_adamw():
    g = golden("g8_adamw")
    cfg = tiny_ref_config()
    sd = tiny_state_dict(cfg)
    inp = tiny_inputs()
    names = [k for k in sd if not (k.endswith(".mean") or k.endswith(".std"))]
    params = {k: sd[k].clone() for k in names}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v = {k: torch.zeros_like(v) for k, v in params.items()}
    for it in range(2):
        leaf = {k: p.clone().requires_grad_(True) for k, p in params.items()}
        full = dict(sd)
        full.update(leaf)
        loss, _, _ = R.forward(full, cfg, inp["input_ids"], inp["labels"], inp["actions_domA"], ["domA"] * 2)
        loss.backward()
        grads = {k: leaf[k].grad for k in names}
        norm = R.clip_and_adamw(params, grads, m, v, it + 1, 1e-3)
        assert abs(loss.item() - g[f"step{it}.loss"].item()) < 2e-5
        assert abs(norm - g[f"step{it}.grad_norm"].item()) < 1e-4 * max(1.0, norm)
    for k in names:
        flat = params[k].reshape(-1)
        idx = torch.linspace(0, flat.numel() - 1, 64).long()
        assert (flat[idx] - g[f"param_samp.{k}"]).abs().max().item() < 2e-5, k
    # a globally-unused domain (domB) must be untouched: no decay, no moments
    for k in names:
        if ".domB." in k or "action_out_projectors" in k or k == "action_mask_tokens":
            assert torch.equal(params[k], sd[k]), k


def _redraw_q(g, steps, B=2):
    """The Exp(1) draws of the reference's `sampled3` run, re-drawn in torch.multinomial's call order (step, factor 1, factor 0)
    from the recorded seed; `qsum` pins the stream.  Returns one (B, 256, 2, 512) tensor per step (factor v at [..., v, :])."""
    torch.manual_seed(int(g["sampled3.seed"]))
    out, sums = [], []
    for _ in range(steps):
        q = torch.empty(B, 256, 2, 512)
        for v in (1, 0):
            qv = torch.empty(B * 256, 512).exponential_(1)
            sums.append(qv.double().sum())
            q[:, :, v] = qv.reshape(B, 256, 512)
        out.append(q)
    assert torch.equal(torch.stack(sums), g["sampled3.qsum"]), "the CPU RNG stream differs from the one the fixture was drawn with"
    return out


def test_g13_decode_steps_oracle():
    """Per-step MaskGIT records of the reference (tests/golden/make_golden_decode.py): the oracle reproduces every step's ids
    bit for bit -- greedy, random-order and the Categorical branch (temperature > 0, st_mask_git.py:411-416) -- and the
    confidences the reference ranks by."""
    g = golden("g13_decode_steps")
    cfg = tiny_ref_config()
    sd = tiny_state_dict(cfg)
    inp = tiny_inputs()
    out_t = cfg.T - 1
    runs = [("greedy1", 1, 0.0, "greedy", {}), ("greedy2", 2, 0.0, "greedy", {}), ("greedy8", 8, 0.0, "greedy", {}),
            ("random4", 4, 0.0, "random", dict(rand_draws=list(g["random4.draws"]))),
            ("sampled3", 3, 1.0, "greedy", dict(sample_draws=_redraw_q(g, 3)))]
    for tag, steps, temp, mode, kw in runs:
        p = g["prompt0"].clone()
        trace = []
        s, _ = R.maskgit_generate(sd, cfg, p, out_t, steps, temp, mode, inp["actions_domA"], ["domA"] * 2, trace=trace, **kw)
        assert torch.equal(s.reshape(2, 256).int(), g[f"{tag}.frame_out"][-1]), tag
        assert len(trace) == steps
        for k, (lg, conf) in enumerate(trace):
            lg = lg.reshape(2, 2, 512, 256)
            assert torch.equal(lg.argmax(2).to(torch.int16), g[f"{tag}.top1"][k]), (tag, k)
            assert rel_err(lg.reshape(2, 1024, 256)[:, ::32], g[f"{tag}.logits_sub"][k]) < TOL
            if conf is not None:
                ref = g[f"{tag}.conf"][k]
                assert torch.equal(torch.isinf(conf), torch.isinf(ref)), (tag, k)
                fin = ~torch.isinf(ref)
                assert torch.allclose(conf[fin], ref[fin], rtol=2e-5, atol=1e-12), (tag, k)
