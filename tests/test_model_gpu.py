"""End-to-end parity of the HIP path against the pinned CPU oracle and the reference's golden vectors.

Tolerances (bf16 MFMA compute, fp32 accumulation / residual stream / loss):
  * loss: |hip - reference| <= 1e-3 (the north-star bound) on the init-scale fixture (weights N(0, 0.02), what a
    training run starts from); on the adversarial fixture (weights N(0, 0.05), unit-scale embeddings, logits of
    magnitude ~4, loss ~15.7) the bound is 3e-4 RELATIVE (5e-3 absolute); accuracy identical on the fixtures;
  * logits: max error <= 1.2 % of the logits' range (measured 0.66 %); parameter gradients: rms error <= 2 % per tensor on the
    adversarial fixture (measured worst 1.0 %, the factored embedding tables; tightened in round 4 once the headline-shape gradient
    decomposition of tests/test_headline_gpu.py existed), <= 3 % on 64 sampled elements per tensor of the init-scale fixture;
  * token ids from MaskGIT decode: bit-exact given the same logits (kernel-level test); end to end in bf16 an id can flip
    only where the reference's top-2 logit margin is below the logits tolerance -- asserted step by step in
    tests/test_decode_rule_gpu.py (the agreement percentages here are logged, not asserted).
"""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from hma_amd.config import GenieConfig  # noqa: E402
from hma_amd.model import STMaskGIT  # noqa: E402
from hma_amd.train import FusedAdamW, Trainer  # noqa: E402
from oracle import st_maskgit_ref as R  # noqa: E402
from tests.helpers import TINY, golden, rel_err, rms_err, tiny_inputs, tiny_ref_config, tiny_state_dict  # noqa: E402

DEV = "cuda"
REPORT = {}


def _note(key, val):
    REPORT[key] = val
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/parity_report.json", "w") as f:
            json.dump(REPORT, f, indent=1, sort_keys=True)
    except OSError:
        pass


def build_model(train=True, initlike=False, **cfg_over):
    cfg = GenieConfig(**{**TINY["config"], **cfg_over})
    m = STMaskGIT(cfg)
    m.init_action_projectors(TINY["domains"], TINY["d_actions"], TINY["action_stats"], cfg.action_network)
    m.load_state_dict(tiny_state_dict(initlike=initlike), strict=True)
    m = m.to(DEV)
    m.train(train)
    return m


def oracle_grads(tag):
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
    return loss.detach(), acc.detach(), logits.detach(), {k: p.grad for k, p in params.items()}


@pytest.mark.parametrize("tag", ["domA", "domB", "noact", "domA+fused_mlp", "domA+unfused_mlp"])
def test_forward_backward_matches_reference(tag):
    """`+fused_mlp` / `+unfused_mlp`: the same check with the MLP block forced onto the fused kernels (hma_mlp_fwd /
    hma_mlp_bwd: hidden activation on chip, pre-activation recomputed in backward, LayerNorm gradients from the wgrad
    reduction -- the default from one 128-row tile per CU upwards) or onto the GEMM pair (smaller passes)."""
    fused = tag.endswith("+fused_mlp")
    unfused = tag.endswith("+unfused_mlp")
    tag = tag.split("+")[0]
    g = golden("g6_forward_backward")
    m = build_model()
    if fused or unfused:
        eng = m._get_engine(torch.device(DEV, torch.cuda.current_device()))
        eng.fused_mlp_train = fused
        eng.fused_mlp_min_rows = 0 if fused else 1 << 60
    inp = {k: v.to(DEV) for k, v in tiny_inputs().items()}
    kw = dict(input_ids=inp["input_ids"], labels=inp["labels"], h=[16, 16], w=[16, 16])
    if tag != "noact":
        kw.update(action_ids=inp[f"actions_{tag}"], domain=[tag] * 2)
    else:
        kw.update(domain=None)
    out = m(**kw)
    loss_ref, acc_ref, logits_ref, grads_ref = oracle_grads(tag)
    # the oracle itself is pinned to the golden file; check the golden numbers directly as well
    assert abs(out.loss.item() - g[f"{tag}.loss"].item()) <= 5e-3
    assert abs(out.loss.item() - loss_ref.item()) <= 3e-4 * loss_ref.item()
    assert out.acc.item() == g[f"{tag}.acc"].item()
    e = rel_err(out.logits, logits_ref)
    nt = tag + ("+fused_mlp" if fused else "+unfused_mlp" if unfused else "")
    _note(f"{nt}.loss_abs_err", abs(out.loss.item() - loss_ref.item()))
    _note(f"{nt}.logits_rel_err", e)
    assert e <= 1.2e-2  # (measured 6.6e-3)
    assert rel_err(out.logits[:, :, :, ::4, ::4], g[f"{tag}.logits_sub"]) <= 1.2e-2
    out.loss.backward()
    worst = 0.0
    for name, p in m.named_parameters():
        gr = grads_ref[name]
        if gr is None or float(gr.abs().sum()) == 0.0:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, f"{name} should have no gradient"
            continue
        assert p.grad is not None, name
        err = rms_err(p.grad, gr)
        worst = max(worst, err)
        _note(f"{nt}.grad_rms.{name}", err)
        assert err <= 2e-2, f"{name}: rms rel err {err:.3e}"  # (measured worst 1.0e-2: the factored embedding tables)
    _note(f"{nt}.worst_grad_rms", worst)


def test_loss_within_1e3_on_init_scale_weights():
    """The north-star tolerance, on the regime it is quoted for: loss within 1e-3 of the reference."""
    g = golden("g6b_initlike")
    m = build_model(initlike=True)
    inp = {k: v.to(DEV) for k, v in tiny_inputs().items()}
    out = m(input_ids=inp["input_ids"], labels=inp["labels"], action_ids=inp["actions_domA"], domain=["domA"] * 2)
    err = abs(out.loss.item() - g["domA.loss"].item())
    _note("initlike.loss_abs_err", err)
    assert err <= 1e-3
    assert out.acc.item() == g["domA.acc"].item()
    e = rel_err(out.logits[:, :, :, ::4, ::4], g["domA.logits_sub"])
    _note("initlike.logits_rel_err", e)
    assert e <= 2e-2
    out.loss.backward()
    worst = 0.0
    for name, p in m.named_parameters():
        key = f"domA.grad_samp.{name}"
        if key not in g:
            continue
        gf = p.grad.reshape(-1).cpu()
        idx = torch.linspace(0, gf.numel() - 1, 64).long()
        ref = g[key]
        err = (gf[idx] - ref).pow(2).mean().sqrt().item() / (ref.pow(2).mean().sqrt().item() + 1e-20)
        worst = max(worst, err)
        assert err <= 3e-2, f"{name}: {err:.3e}"  # (64 sampled elements per tensor; measured worst 8.6e-3)
    _note("initlike.worst_grad_sample_rms", worst)


def test_stblock_and_decoder_forward():
    g = golden("g5_stblock")
    m = build_model(train=False)
    x, a = g["x"].to(DEV), g["a_emb"].to(DEV)
    with torch.no_grad():  # (the same modules under autograd: tests/test_torch_ops.py::test_blocks_train_under_autograd)
        y = m.decoder.layers[0](x, action_ids=a, domain="domA")
        assert rel_err(y[:, :, ::8], g["y_layer0_domA"]) <= 1e-2
        y = m.decoder.layers[0](x[:, :, :256].contiguous(), action_ids=None, domain=None)
        assert rel_err(y[:, :, ::8], g["y_layer0_noact"]) <= 1e-2
        y = m.decoder(x, action_ids=a, domain="domB")
    e = rel_err(y[:, :, ::8], g["y_decoder_domB"])
    _note("decoder_rel_err", e)
    assert e <= 1e-2


def test_submodule_forwards_match_golden():
    g3, g4, g2 = golden("g3_attention"), golden("g4_blocks"), golden("g2_embedding")
    from hma_amd.model import FactorizedEmbedding, Mlp, SelfAttention
    for tag, mup in (("mup", True), ("std", False)):
        att = SelfAttention(num_heads=8, d_model=256, qkv_bias=False, proj_bias=True, qk_norm=False, use_mup=mup)
        att.load_state_dict({"qkv.weight": g3[f"{tag}.qkv"], "proj.weight": g3[f"{tag}.proj_w"], "proj.bias": g3[f"{tag}.proj_b"]})
        att = att.to(DEV)
        assert rel_err(att(g3[f"{tag}.x_spatial"].to(DEV), causal=False), g3[f"{tag}.y_spatial"]) <= 1.5e-2
        assert rel_err(att(g3[f"{tag}.x_temporal"].to(DEV), causal=True), g3[f"{tag}.y_temporal"]) <= 1.5e-2
    mlp = Mlp(256)
    mlp.load_state_dict({"fc1.weight": g4["mlp.fc1_w"], "fc1.bias": g4["mlp.fc1_b"], "fc2.weight": g4["mlp.fc2_w"],
                         "fc2.bias": g4["mlp.fc2_b"]})
    mlp = mlp.to(DEV)
    assert rel_err(mlp(g4["mlp.x"].to(DEV)), g4["mlp.y"]) <= 1.5e-2
    emb = FactorizedEmbedding(512, 2, 256, 262144)
    emb.load_state_dict({"factored_embeds.0.weight": g2["E0"], "factored_embeds.1.weight": g2["E1"],
                         "mask_token_embed": g2["mask_embed"]})
    emb = emb.to(DEV)
    assert torch.equal(emb(g2["ids"].to(DEV)).cpu(), g2["out"])  # gather + fp32 add: bit-exact


def test_modulate_layer_and_basic_mlp_forward_match_golden():
    """The module-level `forward`s of the two small blocks the reference also exposes (st_mask_git.py:66-76, 101-102), G4."""
    g4 = golden("g4_blocks")
    from hma_amd.model.st_mask_git import BasicMLP, ModulateLayer
    mod = ModulateLayer(256, 256)
    mod.load_state_dict({k[4:]: v for k, v in g4.items() if k.startswith("mod.") and k[4:] not in ("x", "c", "y")})
    mod = mod.to(DEV)
    assert rel_err(mod(g4["mod.x"].to(DEV), g4["mod.c"].to(DEV)), g4["mod.y"]) <= 1.5e-2
    bm = BasicMLP(14, 256)
    bm.load_state_dict({k[5:]: v for k, v in g4.items() if k.startswith("stem.model.")})
    bm = bm.to(DEV)
    assert rel_err(bm(g4["stem.norm"].to(DEV)), g4["stem.y"]) <= 1e-4   # fp32 VALU kernel


def test_maskgit_generate_against_golden():
    g = golden("g7_generate")
    m = build_model(train=False)
    inp = {k: v.to(DEV) for k, v in tiny_inputs().items()}
    cfg = m.config
    out_t = cfg.T - 1
    agree = []
    for steps in (1, 2, 8):
        p = g["prompt0"].to(DEV).clone()
        s, fl, _ = m.maskgit_generate(p, out_t=out_t, maskgit_steps=steps, temperature=0.0, unmask_mode="greedy",
                                      action_ids=inp["actions_domA"], domain=["domA"] * 2)
        assert torch.equal(p[:, out_t], s) and torch.equal(p[:, :out_t].cpu(), g["prompt0"][:, :out_t])
        assert (s != cfg.image_vocab_size).all()
        assert rel_err(fl[:, ::8], g[f"greedy{steps}.logits_sub"]) <= 2e-2
        a = (s.cpu() == g[f"greedy{steps}.samples"]).float().mean().item()
        _note(f"generate.greedy{steps}.id_agreement", a)
        agree.append(a)
    # (end-to-end agreement is logged only: the id rule -- a flip only below the reference's margin -- is asserted step by step
    # in tests/test_decode_rule_gpu.py)
    # replayed "random" unmasking order (the torch.rand_like draws are an input)
    p = g["prompt0"].to(DEV).clone()
    s, _, _ = m.maskgit_generate(p, out_t=out_t, maskgit_steps=4, temperature=0.0, unmask_mode="random",
                                 action_ids=inp["actions_domA"], domain=["domA"] * 2, rand_draws=list(g["random4.draws"].to(DEV)))
    a = (s.cpu() == g["random4.samples"]).float().mean().item()
    _note("generate.random4.id_agreement", a)
    # generate(): two autoregressive frames
    ids = inp["labels"].reshape(2, cfg.T, 256)[:, : cfg.T - 2].reshape(2, -1)
    toks = m.generate(ids, None, max_new_tokens=2 * 256, maskgit_steps=2, temperature=0.0, action_ids=inp["actions_domA"],
                      domain=["domA"] * 2, h=[16, 16], w=[16, 16], unmask_mode="greedy")
    assert toks.shape == g["generate2.tokens"].shape
    assert torch.equal(toks[:, : (cfg.T - 2) * 256].cpu(), g["generate2.tokens"][:, : (cfg.T - 2) * 256])
    _note("generate.generate2.id_agreement", (toks.cpu() == g["generate2.tokens"]).float().mean().item())


def test_cached_decode_equals_full_window_decode():
    """Incremental decode (per-layer temporal K/V cache) reproduces the full-window recompute: same ids, and
    the per-frame logits agree to fp32 rounding of the attention sums."""
    g = golden("g7_generate")
    m = build_model(train=False)
    inp = {k: v.to(DEV) for k, v in tiny_inputs().items()}
    cfg = m.config
    ids = inp["labels"].reshape(2, cfg.T, 256)[:, : cfg.T - 2].reshape(2, -1)
    kw = dict(max_new_tokens=2 * 256, maskgit_steps=2, temperature=0.0, action_ids=inp["actions_domA"], domain=["domA"] * 2,
              h=[16, 16], w=[16, 16], unmask_mode="greedy")
    full = m.generate(ids, None, use_cache=False, **kw)
    cached = m.generate(ids, None, use_cache=True, **kw)
    agree = (full == cached).float().mean().item()
    _note("generate.cached_vs_full_agreement", agree)
    assert torch.equal(full, cached)  # (at this depth the two paths round identically; L = 32: tests/test_fulldepth_gpu.py)
    assert (cached != cfg.image_vocab_size).all()
    _note("generate.cached.golden_agreement", (cached.cpu() == g["generate2.tokens"]).float().mean().item())
    # logits of one cached frame pass vs the full-window forward on the same tokens
    eng = m._engine
    toks = full.reshape(2, cfg.T, 256)
    logits_full, _ = m.compute_logits(toks.reshape(2, cfg.T, 16, 16), action_ids=inp["actions_domA"], domain=["domA"] * 2)
    ref_last = logits_full[:, :, cfg.T - 1].permute(0, 2, 3, 1).reshape(2 * 256, 1024).clone()
    eng.decode_prefill(toks[:, : cfg.T - 1].contiguous(), inp["actions_domA"].float(), "domA", cfg.T)
    lg = eng.decode_frame(toks[:, cfg.T - 1].contiguous(), inp["actions_domA"][:, cfg.T - 1].float(), "domA", cfg.T - 1, cfg.T)
    e = rel_err(lg, ref_last)
    _note("generate.cached_logits_rel_err", e)
    assert e <= 1e-3
    # without actions
    kw2 = dict(max_new_tokens=256, maskgit_steps=3, temperature=0.0, h=[16, 16], w=[16, 16], unmask_mode="greedy")
    ids2 = inp["labels"].reshape(2, cfg.T, 256)[:, : cfg.T - 1].reshape(2, -1)
    a = m.generate(ids2, None, use_cache=False, **kw2)
    b = m.generate(ids2, None, use_cache=True, **kw2)
    assert (a == b).float().mean().item() >= 0.995


def test_decode_ids_bit_exact_vs_oracle_on_same_logits():
    """Index path: given the engine's own fp32 logits, the sampled ids equal the oracle's argmax/rank rule."""
    m = build_model(train=False)
    inp = {k: v.to(DEV) for k, v in tiny_inputs().items()}
    cfg = m.config
    out_t = cfg.T - 1
    prompt = inp["labels"].reshape(2, cfg.T, 16, 16).clone()
    prompt[:, out_t:] = cfg.image_vocab_size
    p = prompt.clone()
    s, fl, _ = m.maskgit_generate(p, out_t=out_t, maskgit_steps=1, action_ids=inp["actions_domA"], domain=["domA"] * 2)
    fl = fl.cpu()  # (B, V, NV, H, W)
    a = fl.argmax(dim=1)  # (B, NV, H, W)
    assert torch.equal(s.cpu(), a[:, 1] * 512 + a[:, 0])


def test_clip_adamw_two_steps_vs_golden():
    g = golden("g8_adamw")
    m = build_model()
    sd0 = tiny_state_dict()
    inp = {k: v.to(DEV) for k, v in tiny_inputs().items()}
    opt = FusedAdamW(m, lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05, max_grad_norm=1.0)
    for it in range(2):
        out = m(input_ids=inp["input_ids"], labels=inp["labels"], action_ids=inp["actions_domA"], domain=["domA"] * 2)
        out.loss.backward()
        opt.step()
        opt.zero_grad()
        assert abs(out.loss.item() - g[f"step{it}.loss"].item()) <= (5e-3 if it == 0 else 1.5e-2), it  # step 1 runs on weights that already took a sign-like Adam step
        norm = m._engine.grad_norm().item()
        _note(f"adamw.step{it}.grad_norm_rel_err", abs(norm - g[f"step{it}.grad_norm"].item()) / g[f"step{it}.grad_norm"].item())
        assert abs(norm - g[f"step{it}.grad_norm"].item()) <= 5e-3 * g[f"step{it}.grad_norm"].item()  # measured 1.1e-3 / 5e-4
    worst, errs = 0.0, []
    for name, p in m.named_parameters():
        flat = p.detach().reshape(-1).cpu()
        idx = torch.linspace(0, flat.numel() - 1, 64).long()
        ref = g[f"param_samp.{name}"]
        old = sd0[name].reshape(-1)[idx]
        if ".domB." in name or "action_out_projectors" in name or name == "action_mask_tokens":
            assert torch.equal(flat[idx], old), f"{name} must be untouched (globally unused)"
            continue
        # Adam's first steps move every weight by ~lr: compare the UPDATE, not the weight
        # (first Adam steps are ~lr * sign(g): an element whose gradient is near zero can flip, so use rms)
        du, dr = flat[idx] - old, ref - old
        err = (du - dr).pow(2).mean().sqrt().item() / (dr.pow(2).mean().sqrt().item() + 1e-12)
        worst = max(worst, err)
        errs.append(err)
        # measured worst 0.21 (one parameter whose sampled elements include near-zero gradients: a bf16-noise sign flip moves an
        # element by 2 lr, i.e. one flipped element of 64 is an rms error of 2 / 8 = 0.25 of the update)
        assert err <= 0.26, f"{name}: update err {err:.3f}"
    errs.sort()
    _note("adamw.worst_update_rel_err", worst)
    _note("adamw.median_update_rel_err", errs[len(errs) // 2])
    assert errs[len(errs) // 2] <= 0.08, errs[len(errs) // 2]  # the typical parameter: no flipped element among the 64 sampled


def test_trainer_step_equals_autograd_path_and_checkpoint_roundtrip(tmp_path):
    inp = {k: v.to(DEV) for k, v in tiny_inputs().items()}
    m1, m2 = build_model(), build_model()
    opt = FusedAdamW(m1, lr=1e-3)
    out = m1(input_ids=inp["input_ids"], labels=inp["labels"], action_ids=inp["actions_domB"], domain=["domB"] * 2)
    (out.loss / 1.0).backward()
    opt.step()
    tr = Trainer(m2, lr=1e-3, device=DEV)
    tr.engine.fused_ce = False  # the same launches as the autograd path (which hands the logits to the caller)
    ws = tr.step(inp["input_ids"], inp["labels"], inp["actions_domB"], ["domB"] * 2)
    loss2, _ = tr.loss_and_acc(ws)
    assert abs(loss2.item() - out.loss.item()) < 1e-4  # fp32 atomics order in the loss reduction
    bad = tot = 0
    for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        d = (p1 - p2).abs()
        assert d.max().item() <= 2.1e-3, n1  # an Adam step moves a weight by at most ~lr
        bad += int((d > 2e-5).sum())
        tot += d.numel()
    assert bad <= 1e-3 * tot  # only near-zero gradients (|g| ~ eps) are sensitive to atomics order
    # the default Trainer step: readout + cross-entropy in one launch (hma_readout_ce, logits not written).  Same loss; the gradient of
    # the logits is rounded to bf16 from differently ordered fp32 sums, so the first Adam step (~lr sign(g)) differs where |g| is noise
    m4 = build_model()
    tr4 = Trainer(m4, lr=1e-3, device=DEV)
    assert tr4.engine.fused_ce
    ws4 = tr4.step(inp["input_ids"], inp["labels"], inp["actions_domB"], ["domB"] * 2)
    loss4, acc4 = tr4.loss_and_acc(ws4)
    assert abs(loss4.item() - out.loss.item()) < 1e-4 and acc4.item() == out.acc.item()
    sd0 = tiny_state_dict()
    worst = 0.0
    for (n1, p1), (n4, p4) in zip(m1.named_parameters(), m4.named_parameters()):
        u1, u4 = p1.detach().cpu() - sd0[n1], p4.detach().cpu() - sd0[n1]
        den = u1.pow(2).mean().sqrt().item()
        if den == 0:
            assert torch.equal(u4, u1), n1
            continue
        worst = max(worst, (u1 - u4).pow(2).mean().sqrt().item() / den)
    _note("fused_ce.worst_update_rel_err_vs_unfused", worst)
    assert worst <= 0.12, worst  # measured 0.056
    m1.save_pretrained(tmp_path)
    assert sorted(os.listdir(tmp_path)) == ["README.md", "config.json", "model.safetensors"]
    m3 = STMaskGIT.from_pretrained(tmp_path).to(DEV).eval()
    m1.eval()
    with torch.no_grad():
        a = m1(input_ids=inp["input_ids"], labels=inp["labels"], action_ids=inp["actions_domB"], domain=["domB"] * 2)
        la, lg = a.loss.item(), a.logits.clone()
        b = m3(input_ids=inp["input_ids"], labels=inp["labels"], action_ids=inp["actions_domB"], domain=["domB"] * 2)
    assert torch.equal(lg, b.logits)  # the forward is deterministic (no atomics)
    assert abs(la - b.loss.item()) <= 1e-5 * abs(la)  # the masked-mean reduction uses fp32 atomics


def test_graph_replay_matches_eager_steps():
    """hipGraph replay (one graph, and one graph per gradient bucket as on the N > 1 path) == eager launches."""
    inp = {k: v.to(DEV) for k, v in tiny_inputs().items()}
    models = [build_model() for _ in range(3)]
    trainers = [Trainer(m, lr=1e-3, device=DEV, layers_per_bucket=1) for m in models]
    trainers[0].use_graphs = False
    trainers[2].force_segments = True
    losses = [[], [], []]
    for it in range(4):  # steps 0-1 eager + capture, steps 2-3 replayed
        dom = "domA" if it % 2 == 0 else "domB"
        for i, tr in enumerate(trainers):
            ws = tr.step(inp["input_ids"], inp["labels"], inp[f"actions_{dom}"], [dom] * 2)
            losses[i].append(tr.loss_and_acc(ws)[0].item())
    for it in range(4, 8):
        for i, tr in enumerate(trainers):
            ws = tr.step(inp["input_ids"], inp["labels"], inp["actions_domA"], ["domA"] * 2)
            losses[i].append(tr.loss_and_acc(ws)[0].item())
    assert len(trainers[1]._graphs) >= 1 and len(trainers[2]._graphs) >= 1
    assert len(next(iter(trainers[2]._graphs.values()))) == 3  # L = 2 layers -> 2 buckets + tail
    for a, b, c in zip(*losses):
        # fp32 atomics make two runs differ in the last bits; eight Adam steps at lr 1e-3 amplify that slightly
        assert abs(a - b) <= 2e-2 and abs(a - c) <= 2e-2, (losses)
    assert losses[0][-1] < losses[0][0]  # it trains


def test_grad_accumulation_and_external_torch_optimizer():
    """The reference loop's shape: loss / accum, backward twice, torch.optim.AdamW on the named parameters."""
    inp = {k: v.to(DEV) for k, v in tiny_inputs().items()}
    m = build_model()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    kw = dict(input_ids=inp["input_ids"], labels=inp["labels"], action_ids=inp["actions_domA"], domain=["domA"] * 2)
    (m(**kw).loss / 2).backward()
    g1 = m.out_x_proj.weight.grad.clone()
    (m(**kw).loss / 2).backward()
    assert torch.allclose(m.out_x_proj.weight.grad, 2 * g1, rtol=1e-3, atol=1e-7)
    before = m.out_x_proj.weight.detach().clone()
    opt.step()
    opt.zero_grad()
    assert not torch.equal(before, m.out_x_proj.weight)
    out = m(**kw)  # the engine must notice the external in-place update and refresh its bf16 copies
    assert torch.isfinite(out.loss)


def test_adam_step_counts_are_per_domain_and_resume_roundtrip(tmp_path):
    """A domain head that first receives a gradient at global step 3 gets Adam's step-1 bias correction (torch AdamW
    keeps `step` per parameter and skips grad-None parameters): |delta| ~ lr, not ~0.52 lr.  Then: save_state /
    load_state reproduces the next step of an uninterrupted run."""
    inp = {k: v.to(DEV) for k, v in tiny_inputs().items()}
    lr = 1e-3
    m = build_model()
    tr = Trainer(m, lr=lr, weight_decay=0.0, max_grad_norm=None, device=DEV)
    tr.use_graphs = False
    for _ in range(2):
        tr.step(inp["input_ids"], inp["labels"], inp["actions_domA"], ["domA"] * 2)
    name = "decoder.layers.0.action_projectors.domB.linear_out.weight"
    before = dict(m.named_parameters())[name].detach().clone()
    tr.step(inp["input_ids"], inp["labels"], inp["actions_domB"], ["domB"] * 2)
    delta = (dict(m.named_parameters())[name].detach() - before).abs()
    med = delta[delta > 0].median().item() / lr
    assert 0.9 < med < 1.05, med
    assert tr.engine.dom_steps == {"domA": 2, "domB": 1} and tr.engine.opt_step == 3
    # resume
    tr.save_state(tmp_path)
    assert {"trainer_state.safetensors", "optimizer.bin", "scheduler.bin", "model.safetensors", "config.json"} <= set(os.listdir(tmp_path))
    # the Accelerate-layout files are what the reference's torch AdamW / LambdaLR would have written (train_multi.py:310-321)
    osd = torch.load(tmp_path / "optimizer.bin", weights_only=False)
    from hma_amd.train import reference_param_groups
    order = sum(reference_param_groups([n for n, _ in m.named_parameters()]), [])
    st = {order[i]: v for i, v in osd["state"].items()}
    assert float(st["decoder.layers.0.mlp.fc1.weight"]["step"]) == 3.0 and float(st[name]["step"]) == 1.0
    assert not any("action_out_projectors" in n or n == "action_mask_tokens" for n in st)  # never stepped: no state
    assert torch.load(tmp_path / "scheduler.bin", weights_only=False)["_step_count"] == 4
    os.remove(tmp_path / "trainer_state.safetensors")  # resume from the Accelerate files alone
    m2 = STMaskGIT.from_pretrained(tmp_path).to(DEV).train()
    tr2 = Trainer(m2, lr=lr, weight_decay=0.0, max_grad_norm=None, device=DEV)
    tr2.use_graphs = False
    tr2.load_state(tmp_path)
    assert tr2.completed == 3 and tr2.engine.dom_steps == {"domA": 2, "domB": 1}
    tr.step(inp["input_ids"], inp["labels"], inp["actions_domA"], ["domA"] * 2)
    tr2.step(inp["input_ids"], inp["labels"], inp["actions_domA"], ["domA"] * 2)
    worst = 0.0
    for (n1, p1), (n2, p2) in zip(m.named_parameters(), m2.named_parameters()):
        worst = max(worst, (p1 - p2).abs().max().item())
    assert worst <= 2.1 * lr  # fp32 atomics order in the reductions moves near-zero-gradient weights by at most a step
    bad = sum(int(((p1 - p2).abs() > 2e-5).sum()) for (_, p1), (_, p2) in zip(m.named_parameters(), m2.named_parameters()))
    assert bad <= 1e-3 * sum(p.numel() for p in m.parameters())


def test_evaluator_loss_matches_plain_cross_entropy():
    from hma_amd.eval_utils import compute_loss
    g = torch.Generator().manual_seed(5)
    B, T, H, W = 2, 3, 16, 16
    labels = torch.randint(0, 262144, (B, T * H * W), generator=g)
    logits = torch.randn(B, 512, 2, T - 1, H, W, generator=g) * 2
    lab = labels.reshape(B, T, H, W)[:, 1:]
    fl = torch.stack((lab % 512, lab // 512), dim=1)  # (B, 2, T-1, H, W): factorize_labels
    want = torch.nn.functional.cross_entropy(logits, fl, reduction="none").sum(dim=1).mean().item()
    got = compute_loss(labels, logits.to(DEV))
    assert abs(got - want) <= 2e-5 * abs(want), (got, want)


def test_mlp_dropout_fused_path_equals_epilogue_path():
    """mlp_drop > 0 in training: chain B forward + hma_mlp_bwd re-create the SAME counter-based masks as the GELU2 / RESID epilogues +
    hma_dropout_bf16 of the unfused path (salts 2 l, 2 l + 1 of one device seed), so the two paths give the same loss and gradients up
    to bf16 rounding -- and both differ from the no-dropout loss."""
    inp = {k: v.to(DEV) for k, v in tiny_inputs().items()}
    res = {}
    for tag in ("fused", "unfused"):
        m = build_model(mlp_drop=0.1)
        eng = m._get_engine(torch.device(DEV, torch.cuda.current_device()))
        eng.fused_mlp_min_rows = 0
        if tag == "unfused":
            eng.fused_mlp = False
        out = m(input_ids=inp["input_ids"], labels=inp["labels"], action_ids=inp["actions_domA"], domain=["domA"] * 2)
        out.loss.backward()
        M = 2 * m.config.T * (256 + m.config.action_token_size)
        assert eng._use_fused(M, True, 256 + m.config.action_token_size) == (tag == "fused")
        res[tag] = (out.loss.item(), {n: p.grad.detach().float().cpu().clone() for n, p in m.named_parameters() if p.grad is not None})
    (la, ga), (lb, gb) = res["fused"], res["unfused"]
    _note("dropout.fused_vs_unfused.loss", [la, lb])
    assert abs(la - lb) <= 2e-3 * abs(lb), (la, lb)
    worst = 0.0
    for n in gb:
        den = gb[n].pow(2).mean().sqrt().item()
        if den == 0:
            continue
        worst = max(worst, (ga[n] - gb[n]).pow(2).mean().sqrt().item() / den)
    _note("dropout.fused_vs_unfused.worst_grad_rms", worst)
    assert worst <= 3e-2, worst
    m0 = build_model()
    l0 = m0(input_ids=inp["input_ids"], labels=inp["labels"], action_ids=inp["actions_domA"], domain=["domA"] * 2).loss.item()
    assert abs(la - l0) > 1e-4 * abs(l0)


def test_shipped_config_window_against_oracle():
    """The shipped discrete config (hma/configs/magvit_n32_h8_d256_action.json: T = 12, use_mup false -- the softmax scale is
    head_dim ** -0.5, attention.py:27 -- qkv_bias false, mlp_bias true) at 3 layers instead of 32, init-scale weights: loss, logits and
    every gradient against the oracle on the host cores.  (The golden fixtures use T = 3 and use_mup true; full depth uses T = 16.)"""
    import math
    from oracle.param_spec import seeded_state_dict, state_dict_spec
    shipped = dict(num_layers=3, num_heads=8, use_actions=True, d_model=256, T=12, S=256, image_vocab_size=262144, use_mup=False,
                   action_network="concat+modulate", num_factored_vocabs=2, qkv_bias=False, proj_bias=True, attn_drop=0.0, qk_norm=False,
                   mlp_ratio=4.0, mlp_drop=0.0, mlp_bias=True)
    doms, das, stats = ["domA"], [7], [TINY["action_stats"][0]]
    rc = R.RefConfig(**{k: v for k, v in shipped.items() if k in R.RefConfig.__dataclass_fields__})
    sd = seeded_state_dict(state_dict_spec(rc, doms, das, [7]), seed=31, std=0.02, embed_std=0.02)
    sd["action_preprocessor.domA.mean"] = torch.tensor(stats[0][0])
    sd["action_preprocessor.domA.std"] = torch.tensor(stats[0][1])
    m = STMaskGIT(GenieConfig(**shipped))
    m.init_action_projectors(doms, das, stats, shipped["action_network"])
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).train()
    gq = torch.Generator().manual_seed(12)
    B, T = 2, 12
    labels = torch.randint(0, 262144, (B, T, 16, 16), generator=gq)
    prob = torch.cos(torch.rand(B, T - 1, 1, 1, generator=gq) * math.pi / 2)
    ids = labels.clone()
    ids[:, 1:][torch.rand(B, T - 1, 16, 16, generator=gq) < prob] = 262144
    act = torch.randn(B, T, 7, generator=gq)
    out = m(input_ids=ids.reshape(B, -1).to(DEV), labels=labels.reshape(B, -1).to(DEV), action_ids=act.to(DEV), domain=["domA"] * B)
    out.loss.backward()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if not (k.endswith(".mean") or k.endswith(".std"))}
    full = dict(sd)
    full.update(params)
    loss, acc, logits = R.forward(full, rc, ids.reshape(B, -1), labels.reshape(B, -1), act, ["domA"] * B)
    loss.backward()
    assert abs(out.loss.item() - loss.item()) <= 1e-3, (out.loss.item(), loss.item())
    assert rel_err(out.logits, logits.detach()) <= 2e-2
    worst = 0.0
    for name, p in m.named_parameters():
        gr = params[name].grad
        if gr is None or float(gr.abs().sum()) == 0.0:
            continue
        worst = max(worst, rms_err(p.grad, gr))
    _note("shipped_T12.loss_abs_err", abs(out.loss.item() - loss.item()))
    _note("shipped_T12.worst_grad_rms", worst)
    assert worst <= 3e-2, worst


@pytest.mark.parametrize("B,T,HW", [(1, 2, 8), (1, 3, 8)])
def test_small_pass_trains_against_oracle(B, T, HW):
    """The smallest passes that can train (the spatial attention takes frames of 64, 256 or 320 tokens and a loss needs T >= 2: M = 128
    and 192 rows, no action tokens).  Chain S backward leaves norm1's dgamma / dbeta to the qkv weight gradient's LDS-DMA path, which
    needs whole 32-row stages and two 64-row slabs: M = 128 is exactly that boundary (below it the engine keeps hma_ln_bwd for norm1
    -- `chain_s_ok` in STEngine._backward_plan, the round-5 advisor finding).  Loss and every gradient against the oracle."""
    import math
    from oracle.param_spec import seeded_state_dict, state_dict_spec
    S = HW * HW
    cfgd = dict(num_layers=2, num_heads=8, d_model=256, T=T, S=S, image_vocab_size=262144, use_mup=True, action_network="concat+modulate",
                num_factored_vocabs=2, qkv_bias=False, proj_bias=True, attn_drop=0.0, qk_norm=False, mlp_ratio=4.0, mlp_drop=0.0, mlp_bias=True)
    rc = R.RefConfig(**{k: v for k, v in cfgd.items() if k in R.RefConfig.__dataclass_fields__})
    sd = seeded_state_dict(state_dict_spec(rc, [], [], []), seed=77, std=0.02, embed_std=0.02)
    m = STMaskGIT(GenieConfig(**cfgd))
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).train()
    gq = torch.Generator().manual_seed(5)
    labels = torch.randint(0, 262144, (B, T, HW, HW), generator=gq)
    ids = labels.clone()
    ids[:, 1:][torch.rand(B, T - 1, HW, HW, generator=gq) < 0.6] = 262144
    ids[:, 1, 0, 0] = 262144  # (at least one masked token per sample)
    out = m(input_ids=ids.reshape(B, -1).to(DEV), labels=labels.reshape(B, -1).to(DEV), domain=None, h=[HW] * B, w=[HW] * B)
    out.loss.backward()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss, acc, logits = R.forward(params, rc, ids.reshape(B, -1), labels.reshape(B, -1), None, None, H=HW, W=HW)
    loss.backward()
    assert abs(out.loss.item() - loss.item()) <= 1e-3, (out.loss.item(), loss.item())
    worst = 0.0
    for name, p in m.named_parameters():
        gr = params[name].grad
        if gr is None or float(gr.abs().sum()) == 0.0:
            continue
        assert p.grad is not None, name
        worst = max(worst, rms_err(p.grad, gr))
    assert math.isfinite(worst) and worst <= 3e-2, worst


def _t16_model(with_actions: bool, mlp_drop: float = 0.0, layers: int = 2):
    """A 16-frame model at `layers` layers with init-scale seeded weights, its oracle config and state dict."""
    from oracle.param_spec import seeded_state_dict, state_dict_spec
    cfgd = dict(num_layers=layers, num_heads=8, d_model=256, T=16, S=256, image_vocab_size=262144, use_mup=True, action_network="concat+modulate",
                num_factored_vocabs=2, qkv_bias=False, proj_bias=True, attn_drop=0.0, qk_norm=False, mlp_ratio=4.0, mlp_drop=mlp_drop, mlp_bias=True)
    rc = R.RefConfig(**{k: v for k, v in cfgd.items() if k in R.RefConfig.__dataclass_fields__})
    doms, das, stats = (["domA"], [7], [TINY["action_stats"][0]]) if with_actions else ([], [], [])
    sd = seeded_state_dict(state_dict_spec(rc, doms, das, [7] if with_actions else []), seed=41, std=0.02, embed_std=0.02)
    if with_actions:
        sd["action_preprocessor.domA.mean"] = torch.tensor(stats[0][0])
        sd["action_preprocessor.domA.std"] = torch.tensor(stats[0][1])
    m = STMaskGIT(GenieConfig(**cfgd))
    if with_actions:
        m.init_action_projectors(doms, das, stats, cfgd["action_network"])
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).train(), rc, sd


def _t16_batch(B=2, seed=21):
    import math
    gq = torch.Generator().manual_seed(seed)
    labels = torch.randint(0, 262144, (B, 16, 16, 16), generator=gq)
    prob = torch.cos(torch.rand(B, 15, 1, 1, generator=gq) * math.pi / 2)
    ids = labels.clone()
    ids[:, 1:][torch.rand(B, 15, 16, 16, generator=gq) < prob] = 262144
    return ids.reshape(B, -1), labels.reshape(B, -1), torch.randn(B, 16, 7, generator=gq)


def _plan_names(m):
    eng = m._get_engine(torch.device(DEV, torch.cuda.current_device()))
    return {name for pl in eng._plans.values() for _, name, _ in pl.calls}


def test_deferred_weight_gradients_are_complete_at_every_bucket_mark():
    """The weight gradients of TWO blocks share a launch (engine.wgrad_layers), but never across a gradient bucket: a data-parallel
    driver all-reduces a bucket at its layer mark, so every weight gradient of the bucket's layers must have been LAUNCHED by then.
    (One rank cannot see the difference in the numbers -- its all-reduce is the identity -- so the plan itself is checked.)"""
    L = 5
    m, rc, sd = _t16_model(with_actions=True, layers=L)
    eng = m._get_engine(torch.device(DEV, torch.cuda.current_device()))
    eng.fused_mlp_min_rows = 0
    ids, labels, act = _t16_batch()
    eng.forward(ids.view(2, 16, 256).to(DEV), labels.to(DEV), act.to(DEV), "domA", train=True, loss_grad=True, need_logits=False)
    for group in (1, 2, 3, L):
        eng.ada_group = group
        pl = eng._backward_plan(2, 16, 256, 64, "domA")
        launched_before = lambda pos: sum(a[1] for _, name, a in pl.calls[:pos] if name == "hma_gemm_tn_multi")
        n_multi = sum(1 for _, name, _ in pl.calls if name == "hma_gemm_tn_multi")
        assert n_multi > 0, "the deferred weight-gradient launch is not in the plan"
        for l in reversed(range(L)):
            if (L - l) % group == 0 or l == 0:  # a bucket ends behind layer l
                assert launched_before(pl.marks[f"layer{l}"]) == 7 * (L - l), (group, l, launched_before(pl.marks[f"layer{l}"]))
        if group == L:  # nothing to respect but the end: pairs from the top, the odd layer alone
            assert n_multi == (L + 1) // 2


def test_fused_forward_chain_without_action_tokens_vs_oracle():
    """Round 6: blocks WITHOUT action tokens (no ModulateLayer, 256 rows per frame) take hma_chain_ab_fwd too at T = 16 (its MOD = false
    form: chain A is the spatial projection + residual, bf16(x1) the temporal qkv's operand).  Loss and every gradient against the oracle;
    the fused launch is asserted to be IN the plan (the T = 3 fixtures run the three-launch forward)."""
    m, rc, sd = _t16_model(with_actions=False)
    eng = m._get_engine(torch.device(DEV, torch.cuda.current_device()))
    eng.fused_mlp_min_rows = 0
    ids, labels, _ = _t16_batch()
    out = m(input_ids=ids.to(DEV), labels=labels.to(DEV), domain=None)
    out.loss.backward()
    assert "hma_chain_ab_fwd" in _plan_names(m) and "hma_chain_t_bwd" in _plan_names(m), _plan_names(m)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss, acc, logits = R.forward(params, rc, ids, labels, None, None)
    loss.backward()
    assert abs(out.loss.item() - loss.item()) <= 1e-3, (out.loss.item(), loss.item())
    assert rel_err(out.logits, logits.detach()) <= 2e-2
    worst = 0.0
    for name, p in m.named_parameters():
        gr = params[name].grad
        if gr is None or float(gr.abs().sum()) == 0.0:
            continue
        assert p.grad is not None, name
        worst = max(worst, rms_err(p.grad, gr))
    _note("t16_noact.loss_abs_err", abs(out.loss.item() - loss.item()))
    _note("t16_noact.worst_grad_rms", worst)
    assert worst <= 3e-2, worst


def test_fused_forward_chain_with_dropout_equals_epilogue_path_at_t16():
    """Round 6: mlp_drop > 0 on the fused forward chain (hma_chain_ab_fwd's DROP form: the MAR configs train with 0.05) re-creates the SAME
    counter-based masks as the GELU2 / RESID epilogues of the unfused path, so both give the same loss and gradients to bf16 rounding.
    The T = 3 version of this check (test_mlp_dropout_fused_path_equals_epilogue_path) runs chain B; 16-frame windows run the fused chain."""
    ids, labels, act = _t16_batch()
    res = {}
    for tag in ("fused", "unfused"):
        m, _, _ = _t16_model(with_actions=True, mlp_drop=0.1)
        eng = m._get_engine(torch.device(DEV, torch.cuda.current_device()))
        eng.fused_mlp_min_rows = 0
        if tag == "unfused":
            eng.fused_mlp = False
        out = m(input_ids=ids.to(DEV), labels=labels.to(DEV), action_ids=act.to(DEV), domain=["domA"] * 2)
        out.loss.backward()
        assert ("hma_chain_ab_fwd" in _plan_names(m)) == (tag == "fused"), (tag, _plan_names(m))
        res[tag] = (out.loss.item(), {n: p.grad.detach().float().cpu().clone() for n, p in m.named_parameters() if p.grad is not None})
    (la, ga), (lb, gb) = res["fused"], res["unfused"]
    assert abs(la - lb) <= 2e-3 * abs(lb), (la, lb)
    worst = 0.0
    for n in gb:
        den = gb[n].pow(2).mean().sqrt().item()
        if den == 0:
            continue
        worst = max(worst, (ga[n] - gb[n]).pow(2).mean().sqrt().item() / den)
    _note("t16_dropout.fused_vs_unfused.worst_grad_rms", worst)
    assert worst <= 3e-2, worst
