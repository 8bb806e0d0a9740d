#!/usr/bin/env python3
"""Generate golden vectors by importing the REAL reference (runs only in the build container).

    python tests/golden/make_golden.py          # writes tests/golden/*.safetensors

/root/reference never travels to the GPU box, so its outputs on seeded inputs are
frozen here as small fixtures (data only: inputs + expected outputs).  Two modules
the reference imports unconditionally are absent from the image and stubbed exactly
as SURVEY.md Appendix A describes: `xformers` (the in-repo BasicSelfAttention is
selected via XFORMERS_DISABLED) and `mup` (identity at d_model == 256).
"""
import math
import os
import sys
import types

os.environ["XFORMERS_DISABLED"] = "true"
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def _install_stubs():
    xf = types.ModuleType("xformers")
    xops = types.ModuleType("xformers.ops")

    class LowerTriangularMask:  # noqa: D401
        pass

    def memory_efficient_attention(*a, **k):
        raise RuntimeError("xformers stub")

    xops.LowerTriangularMask = LowerTriangularMask
    xops.memory_efficient_attention = memory_efficient_attention
    xops.unbind = lambda x, dim: x.unbind(dim)
    xf.ops = xops
    sys.modules["xformers"] = xf
    sys.modules["xformers.ops"] = xops

    mup = types.ModuleType("mup")

    class MuReadout(nn.Linear):
        def __init__(self, in_features, out_features, bias=True, readout_zero_init=False, output_mult=1.0):
            super().__init__(in_features, out_features, bias=bias)
            self.output_mult = output_mult

        def width_mult(self):
            return self.in_features / 256

    mup.MuReadout = MuReadout
    mup.set_base_shapes = lambda model, base, rescale_params=False: model
    mup.normal_ = torch.nn.init.normal_
    mup.MuAdamW = torch.optim.AdamW
    sys.modules["mup"] = mup
    torch.Tensor.cuda = lambda self, *a, **k: self


_install_stubs()
sys.path.insert(0, REF)
from hma.config import GenieConfig  # noqa: E402
from hma.model import factorization_utils as rfu  # noqa: E402
from hma.model.attention import SelfAttention  # noqa: E402
from hma.model.st_mask_git import STMaskGIT, ModulateLayer, BasicMLP, ActionStat  # noqa: E402
from hma.model.st_transformer import Mlp, STBlock  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

from oracle.param_spec import state_dict_spec, seeded_state_dict  # noqa: E402
from tests.golden.golden_cfg import TINY, tiny_inputs  # noqa: E402


def rnd(g, *shape, std=1.0):
    return torch.randn(*shape, generator=g) * std


def save(name, tensors):
    out = {k: (v.detach().contiguous().clone() if torch.is_tensor(v) else torch.tensor(v)) for k, v in tensors.items()}
    save_file(out, os.path.join(HERE, name + ".safetensors"))
    print(f"{name}: {sum(t.numel() * t.element_size() for t in out.values()) / 1e6:.2f} MB, {len(out)} tensors")


def g1_factorize():
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 262144, (2, 3, 4, 5), generator=g)
    ids.view(-1)[:4] = torch.tensor([0, 511, 512, 262143])
    fac = rfu.factorize_token_ids(ids, 2, 512)
    save("g1_factorize", {"ids": ids, "fac": fac, "unfac": rfu.unfactorize_token_ids(fac, 2, 512),
                          "labels": rfu.factorize_labels(ids)})


def g2_embedding():
    g = torch.Generator().manual_seed(2)
    emb = rfu.FactorizedEmbedding(512, 2, 256, 262144)
    with torch.no_grad():
        emb.mask_token_embed.copy_(rnd(g, 1, 256))
    ids = torch.randint(0, 262144, (2, 3, 64), generator=g)
    ids[torch.rand(ids.shape, generator=g) < 0.4] = 262144
    save("g2_embedding", {"ids": ids, "E0": emb.factored_embeds[0].weight, "E1": emb.factored_embeds[1].weight,
                          "mask_embed": emb.mask_token_embed, "out": emb(ids)})


def g3_attention():
    g = torch.Generator().manual_seed(3)
    out = {}
    for use_mup in (True, False):
        att = SelfAttention(num_heads=8, d_model=256, qkv_bias=False, proj_bias=True, qk_norm=False, use_mup=use_mup)
        with torch.no_grad():
            att.qkv.weight.copy_(rnd(g, 768, 256, std=0.08))
            att.proj.weight.copy_(rnd(g, 256, 256, std=0.08))
            att.proj.bias.copy_(rnd(g, 256, std=0.1))
        xs = rnd(g, 1, 320, 256)
        xt = rnd(g, 3, 16, 256)
        tag = "mup" if use_mup else "std"
        out.update({f"{tag}.qkv": att.qkv.weight, f"{tag}.proj_w": att.proj.weight, f"{tag}.proj_b": att.proj.bias,
                    f"{tag}.x_spatial": xs, f"{tag}.y_spatial": att(xs, causal=False),
                    f"{tag}.x_temporal": xt, f"{tag}.y_temporal": att(xt, causal=True)})
    save("g3_attention", out)


def g4_blocks():
    g = torch.Generator().manual_seed(4)
    out = {}
    m = Mlp(256)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(rnd(g, *p.shape, std=0.06))
    x = rnd(g, 5, 16, 256)
    out.update({"mlp.fc1_w": m.fc1.weight, "mlp.fc1_b": m.fc1.bias, "mlp.fc2_w": m.fc2.weight,
                "mlp.fc2_b": m.fc2.bias, "mlp.x": x, "mlp.y": m(x)})
    mod = ModulateLayer(256, 256)
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_(rnd(g, *p.shape, std=0.06))
    B, S, T = 2, 5, 4
    xm = rnd(g, B * S, T, 256)
    c = rnd(g, B, T, 256)
    for n, p in mod.named_parameters():
        out[f"mod.{n}"] = p
    out.update({"mod.x": xm, "mod.c": c, "mod.y": mod(xm, c)})
    stat = ActionStat([[0.1 * i for i in range(7)], [0.5 + 0.1 * i for i in range(7)]])
    bm = BasicMLP(14, 256)
    with torch.no_grad():
        for p in bm.parameters():
            p.copy_(rnd(g, *p.shape, std=0.1))
        bm.model[1].weight.add_(1.0)
    a = rnd(g, 2, 4, 14)
    for n, p in bm.named_parameters():
        out[f"stem.{n}"] = p
    out.update({"stem.mean": stat.mean, "stem.std": stat.std, "stem.a": a, "stem.norm": stat(a),
                "stem.y": bm(stat(a)), "stem.unnorm": stat.unnormalize(stat(a))})
    save("g4_blocks", out)


def build_tiny():
    cfg = GenieConfig(**TINY["config"])
    model = STMaskGIT(cfg)
    model.init_action_projectors(TINY["domains"], TINY["d_actions"], TINY["action_stats"], cfg.action_network)
    from oracle.st_maskgit_ref import RefConfig
    rc = RefConfig(**{k: v for k, v in TINY["config"].items() if k in RefConfig.__dataclass_fields__})
    spec = state_dict_spec(rc, TINY["domains"], TINY["d_actions"], [len(s[0]) for s in TINY["action_stats"]])
    ref_sd = model.state_dict()
    assert list(sorted(spec)) == list(sorted(ref_sd)), (set(spec) ^ set(ref_sd))
    for k in spec:
        assert tuple(ref_sd[k].shape) == tuple(spec[k]), (k, ref_sd[k].shape, spec[k])
    sd = seeded_state_dict(spec, seed=TINY["seed"])
    # mean/std buffers come from action_stats (they are part of the config), keep those
    for dom in TINY["domains"]:
        sd[f"action_preprocessor.{dom}.mean"] = ref_sd[f"action_preprocessor.{dom}.mean"].clone()
        sd[f"action_preprocessor.{dom}.std"] = ref_sd[f"action_preprocessor.{dom}.std"].clone()
    model.load_state_dict(sd, strict=True)
    return cfg, model, sd


def grad_digest(named):
    out = {}
    for n, p in named:
        if p.grad is None:
            continue
        gflat = p.grad.reshape(-1)
        out[f"grad_sum.{n}"] = gflat.double().sum().float()
        out[f"grad_abs.{n}"] = gflat.double().abs().sum().float()
        out[f"grad_head.{n}"] = gflat[:64].clone()
        idx = torch.linspace(0, gflat.numel() - 1, 64).long()
        out[f"grad_samp.{n}"] = gflat[idx].clone()
    return out


def g5_g6_model():
    cfg, model, sd = build_tiny()
    inp = tiny_inputs()
    # G5: one STBlock + decoder on a raw activations tensor
    g = torch.Generator().manual_seed(5)
    x = rnd(g, 2, cfg.T, 320, 256)
    a_emb = rnd(g, 2, cfg.T, 256, std=0.5)
    with torch.no_grad():
        y1 = model.decoder.layers[0](x, action_ids=a_emb, domain="domA")
        y1n = model.decoder.layers[0](x[:, :, :256], action_ids=None, domain=None)
        y2 = model.decoder(x, action_ids=a_emb, domain="domB")
    save("g5_stblock", {"x": x, "a_emb": a_emb, "y_layer0_domA": y1[:, :, ::8], "y_layer0_noact": y1n[:, :, ::8],
                        "y_decoder_domB": y2[:, :, ::8]})
    # G6: forward + backward for each domain, and without actions
    out = {}
    for tag, dom, act in (("domA", "domA", inp["actions_domA"]), ("domB", "domB", inp["actions_domB"]),
                          ("noact", None, None)):
        model.zero_grad(set_to_none=True)
        kw = dict(input_ids=inp["input_ids"], labels=inp["labels"])
        if act is not None:
            kw.update(action_ids=act, domain=[dom] * 2)
        else:
            kw.update(domain=None)
        o = model(**kw, h=[16, 16], w=[16, 16])
        o.loss.backward()
        out[f"{tag}.loss"] = o.loss.detach()
        out[f"{tag}.acc"] = o.acc.detach()
        out[f"{tag}.logits_sub"] = o.logits.detach()[:, :, :, ::4, ::4]
        out[f"{tag}.logits_sum"] = o.logits.detach().double().sum().float()
        for k, v in grad_digest(model.named_parameters()).items():
            out[f"{tag}.{k}"] = v
    save("g6_forward_backward", out)
    return cfg, model, sd


def g6b_initlike():
    """Same forward/backward on an init-scale state-dict (N(0, 0.02) like STMaskGIT.init_weights, :737-753):
    the regime the north-star loss tolerance (1e-3) is quoted for."""
    cfg, model, _ = build_tiny()
    from oracle.st_maskgit_ref import RefConfig
    rc = RefConfig(**{k: v for k, v in TINY["config"].items() if k in RefConfig.__dataclass_fields__})
    spec = state_dict_spec(rc, TINY["domains"], TINY["d_actions"], [len(s[0]) for s in TINY["action_stats"]])
    sd = seeded_state_dict(spec, seed=TINY["seed"] + 1, std=0.02, embed_std=0.02)
    ref_sd = model.state_dict()
    for dom in TINY["domains"]:
        sd[f"action_preprocessor.{dom}.mean"] = ref_sd[f"action_preprocessor.{dom}.mean"].clone()
        sd[f"action_preprocessor.{dom}.std"] = ref_sd[f"action_preprocessor.{dom}.std"].clone()
    model.load_state_dict(sd, strict=True)
    inp = tiny_inputs()
    out = {}
    model.zero_grad(set_to_none=True)
    o = model(input_ids=inp["input_ids"], labels=inp["labels"], action_ids=inp["actions_domA"], domain=["domA"] * 2,
              h=[16, 16], w=[16, 16])
    o.loss.backward()
    out["domA.loss"] = o.loss.detach()
    out["domA.acc"] = o.acc.detach()
    out["domA.logits_sub"] = o.logits.detach()[:, :, :, ::4, ::4]
    for k, v in grad_digest(model.named_parameters()).items():
        out[f"domA.{k}"] = v
    save("g6b_initlike", out)


def g7_generate(cfg, model):
    inp = tiny_inputs()
    out = {}
    B = 2
    prompt0 = inp["labels"].reshape(B, cfg.T, 16, 16).clone()
    out_t = cfg.T - 1
    prompt0[:, out_t:] = cfg.image_vocab_size
    for steps in (1, 2, 8):
        p = prompt0.clone()
        s, fl, _ = model.maskgit_generate(p, out_t=out_t, maskgit_steps=steps, temperature=0.0,
                                          unmask_mode="greedy", action_ids=inp["actions_domA"], domain=["domA"] * B)
        out[f"greedy{steps}.samples"] = s
        out[f"greedy{steps}.logits_sub"] = fl[:, ::8]
        out[f"greedy{steps}.prompt_after"] = p
    # random mode: record the rand_like draws
    draws = []
    orig = torch.rand_like

    def rec(t, *a, **k):
        r = orig(t, *a, **k)
        draws.append(r.clone())
        return r

    torch.rand_like = rec
    try:
        torch.manual_seed(123)
        p = prompt0.clone()
        s, fl, _ = model.maskgit_generate(p, out_t=out_t, maskgit_steps=4, temperature=0.0, unmask_mode="random",
                                          action_ids=inp["actions_domA"], domain=["domA"] * B)
    finally:
        torch.rand_like = orig
    out["random4.samples"] = s
    out["random4.draws"] = torch.stack(draws)
    out["prompt0"] = prompt0
    # generate(): two new frames, autoregressive (st_mask_git.py:253-329)
    ids = inp["labels"].reshape(B, cfg.T, 256)[:, : cfg.T - 2].reshape(B, -1)
    toks = model.generate(ids, None, max_new_tokens=2 * 256, maskgit_steps=2, temperature=0.0,
                          action_ids=inp["actions_domA"], domain=["domA"] * B, h=[16, 16], w=[16, 16],
                          unmask_mode="greedy")
    out["generate2.tokens"] = toks
    save("g7_generate", out)


def g8_adamw(cfg, model):
    """One clip(1.0) + AdamW(0.9/0.95, 1e-8, wd .05) step exactly as train_multi.py:593-598, 907-922."""
    inp = tiny_inputs()
    model.zero_grad(set_to_none=True)
    no_decay = ["bias", "layer_norm.weight"]
    groups = [
        {"params": [p for n, p in model.named_parameters() if not any(nd in n for nd in no_decay)], "weight_decay": 0.05},
        {"params": [p for n, p in model.named_parameters() if any(nd in n for nd in no_decay)], "weight_decay": 0.0},
    ]
    opt = torch.optim.AdamW(groups, lr=1e-3, betas=(0.9, 0.95), eps=1e-8)
    out = {}
    for it in range(2):
        o = model(input_ids=inp["input_ids"], labels=inp["labels"], action_ids=inp["actions_domA"],
                  domain=["domA"] * 2, h=[16, 16], w=[16, 16])
        o.loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
        out[f"step{it}.loss"] = o.loss.detach()
        out[f"step{it}.grad_norm"] = norm.detach()
    for n, p in model.named_parameters():
        flat = p.detach().reshape(-1)
        out[f"param_sum.{n}"] = flat.double().sum().float()
        idx = torch.linspace(0, flat.numel() - 1, 64).long()
        out[f"param_samp.{n}"] = flat[idx].clone()
    save("g8_adamw", out)


if __name__ == "__main__":
    torch.manual_seed(0)
    g1_factorize()
    g2_embedding()
    g3_attention()
    g4_blocks()
    cfg, model, sd = g5_g6_model()
    g6b_initlike()
    g7_generate(cfg, model)
    cfg, model, sd = build_tiny()
    g8_adamw(cfg, model)
