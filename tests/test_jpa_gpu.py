"""jointly_predict_actions=True on the GPU (st_mask_git.py:656-660, 676-678, 724-733) against G16 (the real reference, tests/golden/
make_golden_jpa.py): masked action tokens, the pooled action read-out, the reference's action loss, and the gradients of
loss + 0.5 * action_loss -- through `STMaskGIT.forward` + autograd and through the fused `Trainer` step."""
import pytest
import torch

from hma_amd.config import GenieConfig
from hma_amd.model import STMaskGIT
from hma_amd.train import FusedAdamW, Trainer
from tests.helpers import TINY, golden, rel_err, rms_err, tiny_inputs, tiny_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build():
    cfg = GenieConfig(**dict(TINY["config"], jointly_predict_actions=True))
    m = STMaskGIT(cfg)
    m.init_action_projectors(TINY["domains"], TINY["d_actions"], TINY["action_stats"], cfg.action_network)
    m.load_state_dict(tiny_state_dict(), strict=True)
    return m.to(DEV).train()


@pytest.mark.parametrize("tag", ["domA", "domB"])
def test_forward_backward_with_action_prediction_matches_reference(tag):
    g = golden("g16_jpa")
    m = build()
    inp = {k: v.to(DEV) for k, v in tiny_inputs().items()}
    mask = g[f"{tag}.action_mask"].to(DEV)
    out = m(input_ids=inp["input_ids"], labels=inp["labels"], action_ids=inp[f"actions_{tag}"], domain=[tag] * 2, action_mask=mask)
    assert abs(out.loss.item() - g[f"{tag}.loss"].item()) <= 3e-4 * abs(g[f"{tag}.loss"].item())
    assert out.acc.item() == g[f"{tag}.acc"].item()
    assert abs(out.action_loss.item() - g[f"{tag}.action_loss"].item()) <= 2e-3 * abs(g[f"{tag}.action_loss"].item())
    assert rel_err(out.actions, g[f"{tag}.actions"]) <= 2e-2
    assert rel_err(out.logits[:, :, :, ::4, ::4], g[f"{tag}.logits_sub"]) <= 2e-2
    (out.loss + 0.5 * out.action_loss).backward()
    checked = set()
    for name, p in m.named_parameters():
        key = f"{tag}.grad_samp.{name}"
        if key not in g:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, f"{name} should have no gradient"
            continue
        assert p.grad is not None, name
        gf = p.grad.reshape(-1).float().cpu()
        idx = torch.linspace(0, gf.numel() - 1, 64).long()
        ref = g[key]
        err = (gf[idx] - ref).pow(2).mean().sqrt().item() / (ref.pow(2).mean().sqrt().item() + 1e-20)
        assert err <= 6e-2, f"{name}: {err:.3e}"
        checked.add(name)
    assert {"action_mask_tokens", f"action_out_projectors.{tag}.weight", f"action_out_projectors.{tag}.bias",
            f"action_mlp.{tag}.model.0.weight", "pos_embed_TSC"} <= checked
    # leaving the action loss out of the objective: the read-out gets no gradient, the mask tokens still do (through the video loss)
    m2 = build()
    o2 = m2(input_ids=inp["input_ids"], labels=inp["labels"], action_ids=inp[f"actions_{tag}"], domain=[tag] * 2, action_mask=mask)
    o2.loss.backward()
    w = dict(m2.named_parameters())
    assert float(w[f"action_out_projectors.{tag}.weight"].grad.abs().sum()) == 0.0
    assert float(w["action_mask_tokens"].grad.abs().sum()) > 0.0


def test_trainer_step_with_action_prediction_equals_autograd_path():
    inp = {k: v.to(DEV) for k, v in tiny_inputs().items()}
    mask = golden("g16_jpa")["domA.action_mask"].to(DEV)
    m1, m2 = build(), build()
    opt = FusedAdamW(m1, lr=1e-3)
    out = m1(input_ids=inp["input_ids"], labels=inp["labels"], action_ids=inp["actions_domA"], domain=["domA"] * 2, action_mask=mask)
    (out.loss + m1.config.action_loss_weight * out.action_loss).backward()
    opt.step()
    tr = Trainer(m2, lr=1e-3, device=DEV)
    tr.engine.fused_ce = False
    tr.step(inp["input_ids"], inp["labels"], inp["actions_domA"], ["domA"] * 2, action_mask=mask)
    sd0 = tiny_state_dict()
    for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        u1, u2 = p1.detach().cpu() - sd0[n1], p2.detach().cpu() - sd0[n1]
        den = u1.pow(2).mean().sqrt().item()
        if den == 0:
            assert torch.equal(u1, u2), n1
            continue
        assert (u1 - u2).pow(2).mean().sqrt().item() <= 0.1 * den, n1
    moved = {n for n, p in m2.named_parameters() if not torch.equal(p.detach().cpu(), sd0[n])}
    assert {"action_mask_tokens", "action_out_projectors.domA.weight"} <= moved and "action_out_projectors.domB.weight" not in moved


def test_compute_logits_returns_actions_and_random_mask_is_drawn():
    m = build().eval()
    inp = {k: v.to(DEV) for k, v in tiny_inputs().items()}
    T = m.config.T
    with torch.no_grad():
        logits, actions = m.compute_logits(inp["input_ids"].reshape(2, T, 16, 16), action_ids=inp["actions_domA"], domain=["domA"] * 2)
        assert actions.shape == (2, T, 7) and torch.isfinite(actions).all() and logits.shape == (2, 1024, T, 16, 16)
        torch.manual_seed(3)
        a = m(input_ids=inp["input_ids"], labels=inp["labels"], action_ids=inp["actions_domA"], domain=["domA"] * 2)
        torch.manual_seed(4)
        b = m(input_ids=inp["input_ids"], labels=inp["labels"], action_ids=inp["actions_domA"], domain=["domA"] * 2)
    assert a.action_loss.item() != b.action_loss.item() or a.loss.item() != b.loss.item()  # different masks were drawn


def test_policy_mode_predicts_actions_without_action_ids():
    """No action ids (st_mask_git.py:663-666, "as in policies"): every action token is a mask token, the decoder runs unconditioned,
    the domain selects the action read-out.  G16's `policy.*` entries are the reference at B = 1 (its branch only runs for one
    sample); here any batch works and the rows are independent."""
    g = golden("g16_jpa")
    m = build().eval()
    inp = {k: v.to(DEV) for k, v in tiny_inputs().items()}
    T = m.config.T
    x = inp["input_ids"].reshape(2, T, 16, 16)
    with torch.no_grad():
        logits1, actions1 = m.compute_logits(x[:1], action_ids=None, domain=["domB"])
        assert rel_err(actions1, g["policy.actions"]) <= 2e-2
        assert rel_err(logits1[:, :, :, ::4, ::4], g["policy.logits_sub"]) <= 2e-2
        logits2, actions2 = m.compute_logits(x, action_ids=None, domain=["domB"] * 2)
        assert actions2.shape == (2, T, 14)
        assert rel_err(actions2[:1], actions1) <= 2e-3 and rel_err(logits2[:1], logits1) <= 2e-3
        # and with action ids again afterwards (the plans of the two modes do not interfere)
        _, actions3 = m.compute_logits(x, action_ids=inp["actions_domB"], domain=["domB"] * 2)
        assert actions3.shape == (2, T, 14) and rel_err(actions3, actions2) > 1e-3


def test_generate_returns_unnormalised_actions():
    """generate(return_with_actions=True) (st_mask_git.py:304-322): the actions predicted during the LAST frame's MaskGIT steps, mapped
    back through the domain's statistics; the window (non-cached) path, whose every step also yields the action read-out."""
    m = build().eval()
    inp = {k: v.to(DEV) for k, v in tiny_inputs().items()}
    T = m.config.T
    ids = inp["labels"].reshape(2, T, 256)[:, : T - 1].reshape(2, -1)
    kw = dict(max_new_tokens=256, maskgit_steps=2, temperature=0.0, action_ids=inp["actions_domA"], domain=["domA"] * 2, unmask_mode="greedy")
    tokens, actions = m.generate(ids, None, return_with_actions=True, **kw)
    plain = m.generate(ids, None, use_cache=False, **kw)
    assert torch.equal(tokens, plain) and actions.shape == (2, T, 7) and torch.isfinite(actions).all()
    prompt = tokens.reshape(2, T, 16, 16).clone()
    prompt[:, T - 1] = m.mask_token_id
    _, _, raw = m.maskgit_generate(prompt, T - 1, maskgit_steps=2, temperature=0.0, action_ids=inp["actions_domA"], domain=["domA"] * 2,
                                   unmask_mode="greedy")
    st = m.action_preprocessor["domA"]
    assert torch.allclose(actions, raw * (st.std + 1e-10) + st.mean, rtol=1e-6, atol=1e-6)


def test_cached_generate_equals_window_generate_on_a_jpa_model():
    """`generate()` defaults to the K/V-cached loop; on a jointly_predict_actions model its prefill must feed the
    embedded actions as the prompt frames' action tokens (what the window path, st_mask_git.py:656-660 without an action mask, feeds),
    and the policy mode (no action ids: mask tokens on every frame, :663-666) must take the window path."""
    m = build().eval()
    inp = {k: v.to(DEV) for k, v in tiny_inputs().items()}
    T = m.config.T
    ids = inp["labels"].reshape(2, T, 256)[:, : T - 2].reshape(2, -1)
    kw = dict(max_new_tokens=512, maskgit_steps=2, temperature=0.0, action_ids=inp["actions_domA"], domain=["domA"] * 2, unmask_mode="greedy")
    with torch.no_grad():
        eng = m._get_engine(torch.device(DEV, torch.cuda.current_device()))
        window = m.generate(ids, None, use_cache=False, **kw)
        eng._workspace(2, T - 2, 256, m.config.action_token_size, False)["a_tok"].fill_(float("nan"))  # stale content must not matter
        cached = m.generate(ids, None, **kw)
        assert (cached != m.mask_token_id).all()
        agree = (cached == window).float().mean().item()
        assert agree >= 0.97, agree   # (the two paths differ by bf16 noise: near-tie ids may flip, tests/test_fulldepth_gpu.py)
        # policy mode: cached request == window path exactly (it is routed there)
        kwp = dict(kw, action_ids=None, domain=["domB"] * 2)
        assert torch.equal(m.generate(ids, None, **kwp), m.generate(ids, None, use_cache=False, **kwp))


def test_action_loss_alone_trains_the_action_head():
    """`out.action_loss.backward()` with no video loss in the objective used to record a scale and compute nothing.
    It now runs the engine backward at the end of the autograd pass: the gradients equal those of (0 * loss + action_loss)."""
    inp = {k: v.to(DEV) for k, v in tiny_inputs().items()}
    T = TINY["config"]["T"]
    amask = (torch.arange(T)[None, :] >= torch.tensor([[1], [2]])).to(DEV)
    kw = dict(input_ids=inp["input_ids"], labels=inp["labels"], action_ids=inp["actions_domA"], domain=["domA"] * 2, action_mask=amask)
    grads = []
    for alone in (True, False):
        m = build().train()
        out = m(**kw)
        if alone:
            out.action_loss.backward()
        else:
            (0.0 * out.loss + out.action_loss).backward()
        g = {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None}
        assert "action_out_projectors.domA.weight" in g and float(g["action_out_projectors.domA.weight"].abs().max()) > 0
        grads.append(g)
    for n in ("action_out_projectors.domA.weight", "decoder.layers.0.mlp.fc1.weight", "decoder.layers.1.spatial_attn.qkv.weight",
              "action_mask_tokens"):
        a, b = grads[0][n], grads[1][n]
        assert (a - b).norm() <= 2e-2 * (b.norm() + 1e-12), n
