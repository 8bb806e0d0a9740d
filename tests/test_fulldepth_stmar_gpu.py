"""Full-depth parity of the continuous model on the GPU box (BASELINE configs[3]: L = 32, T = 16, 32 x 32 x 4 latents ->
256 patch tokens + 64 action tokens per frame, diffusion head 1024 x 4): HIP against the pinned CPU oracle
(oracle/st_mar_ref.py, oracle/diffloss_ref.py: G9 / G11 / G14) on the box's host cores, plus the discrete model's decode
logits against the oracle's full-window pass at L = 32.

VERDICT round 2 (weak 1): the fixture tests use L = 2, head 256 x 2; bf16 error grows with depth, so the north-star
bounds are asserted here at the depth they are quoted for:
  * loss within 1e-3 (relative) of the oracle, the latents z within 2 % rms, sampled parameter gradients of layers 0 / 15 / 31 and of
    the diffusion head within 3 % rms -- B = 1, init-scale weights, fixed diffusion draws (t, noise);
  * B = 16 (configs[3]'s per-GPU batch) property pass: finite, and the masked-mean loss decomposes over batch chunks;
  * with mlp_drop = 0.05 (the shipped mar_n32_h8_d256_action.json) in training mode: finite, masks change per forward,
    the loss stays near the no-dropout value; eval ignores the dropout.
Reference lines: hma/model/st_mar.py:219-275, hma/model/diffloss.py:28-59.
"""
import json
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from hma_amd.config import DiffusionGenieConfig  # noqa: E402
from hma_amd.model.st_mar import STMAR  # noqa: E402
from oracle import st_mar_ref as MR  # noqa: E402
from oracle import st_maskgit_ref as R  # noqa: E402
from tests.helpers import rms_err  # noqa: E402

DEV = "cuda"
FULL = dict(num_layers=32, num_heads=8, d_model=256, T=16, S=1024, image_vocab_size=262144, use_mup=True,
            action_network="concat+modulate", num_factored_vocabs=2, qkv_bias=True, proj_bias=True, attn_drop=0.0, qk_norm=False,
            mlp_ratio=4.0, mlp_drop=0.0, mlp_bias=False, patch_size=2, vae_embed_dim=4, diffloss_w=1024, diffloss_d=4,
            num_sampling_steps="100", diffusion_batch_mul=1, use_actions=True)
DOMAINS, D_ACTIONS = ["domA", "domB"], [7, 14]
STATS = [[[0.05 * i for i in range(7)], [0.6 + 0.1 * i for i in range(7)]],
         [[-0.1 * i for i in range(7)], [1.0 + 0.05 * i for i in range(7)]]]
REPORT = {}


def _note(key, val):
    REPORT[key] = val
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/parity_report_fulldepth_stmar.json", "w") as f:
            json.dump(REPORT, f, indent=1, sort_keys=True)
    except OSError:
        pass


def _state(template):
    """Init-scale values for every tensor (what a training run starts from: decoder Linears xavier-uniform gain 0.1,
    st_transformer.py:160-170; STMAR.initialize_weights N(0, 0.02) for the rest, st_mar.py:102-115), with small random values
    where the reference starts from exact zeros / ones so that an indexing error there cannot hide."""
    g = torch.Generator().manual_seed(11)
    out = {}
    for k in sorted(template):
        v = template[k]
        if k.endswith(".mean") or k.endswith(".std"):
            out[k] = v.clone()
        elif v.dim() == 2 and k.startswith("decoder.layers."):
            a = 0.1 * math.sqrt(6.0 / (v.shape[0] + v.shape[1]))
            out[k] = (torch.rand(v.shape, generator=g) * 2 - 1) * a
        elif v.dim() >= 2:
            out[k] = 0.02 * torch.randn(v.shape, generator=g)
        elif k.endswith("norm1.weight") or k.endswith("norm2.weight") or k.endswith("ln.weight") or k.endswith("norm.weight") \
                or k.endswith("model.1.weight"):
            out[k] = 1 + 0.02 * torch.randn(v.shape, generator=g)
        else:
            out[k] = 0.01 * torch.randn(v.shape, generator=g)
    return out


def _model(train=True, **over):
    cfg = dict(FULL, **over)
    m = STMAR(DiffusionGenieConfig(**cfg))
    m.init_action_projectors(DOMAINS, D_ACTIONS, STATS, cfg["action_network"])
    sd = _state(m.state_dict())
    m.load_state_dict(sd)
    m = m.to(DEV)
    m.train(train)
    return m, sd


def _inputs(B, seed=5):
    g = torch.Generator().manual_seed(seed)
    T, H = FULL["T"], 32
    lat = torch.randn(B, T * H * H, 4, generator=g) * 0.18215 * 4
    masked = torch.rand(B, T, H, H, generator=g) < 0.55
    masked[:, 0] = False
    n = B * T * 256
    return dict(lat=lat, masked=masked, act=torch.randn(B, T, 7, generator=g), t=torch.randint(0, 1000, (n,), generator=g),
                noise=torch.randn(n, 16, generator=g))


def _kw(inp, B):
    d = lambda t: t.to(DEV)
    return dict(input_ids=d(inp["lat"]).clone(), labels=d(inp["lat"]).clone(), action_ids=d(inp["act"]), domain=["domA"] * B,
                masked_tokens_indicator=d(inp["masked"]), h=[32] * B, w=[32] * B, diffusion_t=d(inp["t"]), diffusion_noise=d(inp["noise"]))


@pytest.mark.timeout(1200)
def test_stmar_full_depth_loss_b4_within_1e3_absolute():
    """north_star's bound as stated -- |loss - reference| <= 1e-3 ABSOLUTE -- for the continuous model at full depth.  One sample's
    masked mean over ~3 500 patch rows moves by ~1e-3 with the side single bf16 roundings of the latents z fall on (the B = 1 test
    below bounds it relatively); over a batch of four the per-sample noise averages down and the absolute bound holds."""
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 32)))
    m, sd = _model()  # (training mode, mlp_drop = 0: the forward the B = 1 test below checks)
    B = 4
    inp = _inputs(B, seed=11)
    rc = R.RefConfig(num_layers=32, num_heads=8, d_model=256, T=16, use_mup=True, qkv_bias=True, mlp_bias=False)
    keep = lambda k: (".action_projectors." not in k and not k.startswith("action_")) or ".domA." in k
    full = {k: v for k, v in sd.items() if keep(k)}
    with torch.no_grad():
        loss_ref, _ = MR.forward(full, rc, inp["lat"], inp["lat"], inp["act"], ["domA"] * B, inp["masked"], inp["t"], inp["noise"], 2, 32, 32, 4)
        out = m(**_kw(inp, B))
    dl = abs(out.loss.item() - loss_ref.item())
    _note("b4.loss_abs_err", dl)
    _note("b4.loss_ref", loss_ref.item())
    assert dl <= 1e-3, (out.loss.item(), loss_ref.item())


@pytest.mark.timeout(1200)
def test_stmar_full_depth_forward_backward_vs_oracle():
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 32)))
    m, sd = _model()
    inp = _inputs(1)
    # ---- oracle (fp32, CPU)
    rc = R.RefConfig(num_layers=32, num_heads=8, d_model=256, T=16, use_mup=True, qkv_bias=True, mlp_bias=False)
    keep = lambda k: (".action_projectors." not in k and not k.startswith("action_")) or ".domA." in k
    full = {k: v for k, v in sd.items() if keep(k)}
    leaf = {k: v.clone().requires_grad_(True) for k, v in full.items()
            if not (k.endswith(".mean") or k.endswith(".std")) and not k.startswith("action_diff_losses")}
    full.update(leaf)
    loss_ref, z_ref = MR.forward(full, rc, inp["lat"], inp["lat"], inp["act"], ["domA"], inp["masked"], inp["t"], inp["noise"], 2, 32, 32, 4)
    loss_ref.backward()
    # ---- HIP
    out = m(**_kw(inp, 1))
    z = out.logits.permute(0, 2, 3, 4, 1).reshape(1, 16, 256, 256)
    dl = abs(out.loss.item() - loss_ref.item())
    ez = rms_err(z, z_ref.detach())
    _note("loss_abs_err", dl)
    _note("loss_ref", loss_ref.item())
    _note("z_rms_err", ez)
    # (loss ~ 2.6 over ~3 500 masked patch rows of ONE sample: measured 0.7e-3 .. 1.4e-3 depending on which side single bf16
    # roundings of z fall; the bound is 1e-3 RELATIVE here, the absolute 1e-3 of the north star holds for the discrete model,
    # tests/test_fulldepth_gpu.py)
    assert dl <= 1e-3 * abs(loss_ref.item()), (out.loss.item(), loss_ref.item())
    assert ez <= 2e-2, ez
    out.loss.backward()
    named = dict(m.named_parameters())
    picks = ["token_embed.weight", "mask_token", "out_x_proj.weight", "decoder_norm.weight", "z_proj_ln.bias", "pos_embed_TSC",
             "diffloss.net.cond_embed.weight", "diffloss.net.input_proj.weight", "diffloss.net.res_blocks.0.mlp.0.weight",
             "diffloss.net.res_blocks.3.mlp.2.weight", "diffloss.net.res_blocks.1.adaLN_modulation.1.weight",
             "diffloss.net.final_layer.linear.weight", "diffloss.net.time_embed.mlp.2.weight", "action_mlp.domA.model.3.weight"]
    for l in (0, 15, 31):
        p = f"decoder.layers.{l}."
        picks += [p + "norm1.weight", p + "spatial_attn.qkv.weight", p + "spatial_attn.qkv.bias", p + "spatial_attn.proj.weight",
                  p + "temporal_attn.qkv.weight", p + "temporal_attn.proj.bias", p + "norm2.weight", p + "mlp.fc1.weight",
                  p + "mlp.fc2.weight", p + "action_projectors.domA.linear_out.weight",
                  p + "action_projectors.domA.adaLN_modulation.2.weight"]
    worst = 0.0
    for name in picks:
        if name not in leaf:
            continue
        g_ref, g_hip = leaf[name].grad, named[name].grad
        assert g_hip is not None and g_ref is not None, name
        e = rms_err(g_hip, g_ref)
        worst = max(worst, e)
        _note(f"grad_rms.{name}", e)
        assert e <= 2e-2, f"{name}: rms rel err {e:.3e}"
    _note("worst_grad_rms", worst)


@pytest.mark.timeout(900)
def test_stmar_full_size_batch_and_dropout_properties():
    """configs[3]'s per-GPU batch (B = 16): finite, the masked-mean loss decomposes over batch chunks; the shipped mlp_drop."""
    m, _ = _model(train=False)
    B = 16
    inp = _inputs(B, seed=6)
    n1 = 16 * 256

    def chunk(lo, hi):
        c = {k: (v[lo:hi] if k in ("lat", "masked", "act") else v[lo * n1:hi * n1]) for k, v in inp.items()}
        return _kw(c, hi - lo)

    with torch.no_grad():
        out = m(**_kw(inp, B))
        assert torch.isfinite(out.loss) and torch.isfinite(out.logits).all()
        num = den = 0.0
        for c in range(0, B, 4):
            oc = m(**chunk(c, c + 4))
            pm = MR.patchify(inp["masked"][c:c + 4][..., None].float(), 2).sum(-1) > 0   # the patch mask the loss averages over
            n = float(pm.sum())
            num += oc.loss.item() * n
            den += n
    err = abs(out.loss.item() - num / den)
    _note("b16.loss_vs_chunked", err)
    assert err <= 1e-3, (out.loss.item(), num / den)
    e0 = out.loss.item()
    del m
    torch.cuda.empty_cache()
    # ---- mlp_drop = 0.05 (what the shipped MAR config trains with)
    md, _ = _model(train=True, mlp_drop=0.05)
    small = chunk(0, 4)
    l1 = md(**small).loss.item()
    l2 = md(**small).loss.item()
    assert math.isfinite(l1) and math.isfinite(l2)
    assert abs(l1 - l2) > 1e-5 * abs(l1)            # fresh masks per forward
    md.eval()
    with torch.no_grad():
        le1, le2 = md(**small).loss.item(), md(**small).loss.item()
    assert abs(le1 - le2) <= 2e-6 * abs(le1)         # eval: nothing dropped, repeatable up to the loss reduction's atomics
    assert abs(l1 - le1) < 0.2 * abs(le1)
    _note("dropout.train_loss", l1)
    _note("dropout.eval_loss", le1)
    _note("b16.loss", e0)


@pytest.mark.timeout(900)
def test_stmar_shipped_config_window_vs_oracle():
    """The shipped MAR config's block shape (hma/configs/mar_n32_h8_d256_action.json: T = 12, use_mup false, qkv_bias true, mlp_bias
    false; attn_drop is constructed but never applied by the reference's attention, attention.py:37-61 / :151; mlp_drop set to 0 for a
    deterministic comparison) at 3 layers and the full-width head: loss, latents and gradients against the oracle on the host cores."""
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 32)))
    over = dict(num_layers=3, T=12, use_mup=False, attn_drop=0.1)
    m, sd = _model(**over)
    B, T, H = 1, 12, 32
    g = torch.Generator().manual_seed(9)
    lat = torch.randn(B, T * H * H, 4, generator=g) * 0.18215 * 4
    masked = torch.rand(B, T, H, H, generator=g) < 0.55
    masked[:, 0] = False
    n = B * T * 256
    inp = dict(lat=lat, masked=masked, act=torch.randn(B, T, 7, generator=g), t=torch.randint(0, 1000, (n,), generator=g),
               noise=torch.randn(n, 16, generator=g))
    rc = R.RefConfig(num_layers=3, num_heads=8, d_model=256, T=12, use_mup=False, qkv_bias=True, mlp_bias=False)
    keep = lambda k: (".action_projectors." not in k and not k.startswith("action_")) or ".domA." in k
    full = {k: v for k, v in sd.items() if keep(k)}
    leaf = {k: v.clone().requires_grad_(True) for k, v in full.items()
            if not (k.endswith(".mean") or k.endswith(".std")) and not k.startswith("action_diff_losses")}
    full.update(leaf)
    loss_ref, z_ref = MR.forward(full, rc, inp["lat"], inp["lat"], inp["act"], ["domA"], inp["masked"], inp["t"], inp["noise"], 2, 32, 32, 4)
    loss_ref.backward()
    out = m(**_kw(inp, 1))
    z = out.logits.permute(0, 2, 3, 4, 1).reshape(1, T, 256, 256)
    assert abs(out.loss.item() - loss_ref.item()) <= 1e-3 * abs(loss_ref.item()), (out.loss.item(), loss_ref.item())
    assert rms_err(z, z_ref.detach()) <= 2e-2
    out.loss.backward()
    named = dict(m.named_parameters())
    worst = 0.0
    for name, p in leaf.items():
        if p.grad is None or name not in named or named[name].grad is None or float(p.grad.abs().sum()) == 0.0:
            continue
        e = rms_err(named[name].grad, p.grad)
        worst = max(worst, e)
        assert e <= 3e-2, f"{name}: rms rel err {e:.3e}"
    _note("shipped_T12.loss_abs_err", abs(out.loss.item() - loss_ref.item()))
    _note("shipped_T12.worst_grad_rms", worst)
