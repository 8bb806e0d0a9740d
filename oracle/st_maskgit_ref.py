"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the HMA hot path (plain PyTorch, fp32).

A *functional* restatement of the reference algorithm: every function takes the
reference's state-dict (tensor names exactly as `STMaskGIT.state_dict()` emits
them) plus plain tensors, and returns plain tensors.  No
nn.Module tree, no einops: the point is an independent second implementation
that the golden vectors (made by importing the real reference, see
tests/golden/make_golden.py) can pin.

Each function cites the reference lines (relative to /root/reference) it follows.
Parity status: PINNED by tests/test_oracle_golden.py against tests/golden/*.
Third-party arithmetic absent from the reference tree: `mup` (janEbert fork,
`fsdp-fix`) -- identity at d_model == 256 which is the only width handled here;
`xformers 0.0.26.post1` -- replaced by the in-repo equivalent
hma/model/attention.py:37-61 which this file restates.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


@dataclass
class RefConfig:
    """The subset of GenieConfig (hma/config.py:9-60) the hot path reads."""

    num_layers: int
    num_heads: int
    d_model: int
    T: int = 12
    S: int = 256
    image_vocab_size: int = 262144
    num_factored_vocabs: int = 2
    factored_vocab_size: int = 512
    use_mup: bool = True
    action_network: str = "concat+modulate"
    action_token_size: int = 64
    qkv_bias: bool = False
    proj_bias: bool = True
    qk_norm: bool = False
    mlp_ratio: float = 4.0
    mlp_bias: bool = True
    jointly_predict_actions: bool = False
    jointly_predict_states: bool = True

    @property
    def mask_token_id(self) -> int:  # hma/model/st_mask_git.py:181
        return self.image_vocab_size

    @property
    def attn_scale(self) -> float:  # hma/model/attention.py:27
        hd = self.d_model // self.num_heads
        return 8.0 / hd if self.use_mup else hd ** -0.5


# --------------------------------------------------------------------------------------
# integer paths (bit-exact)
# --------------------------------------------------------------------------------------
def factorize_token_ids(ids: torch.Tensor, num_vocabs: int = 2, vocab: int = 512) -> torch.Tensor:
    """hma/model/factorization_utils.py:57-68 -- id -> (id // vocab**i) % vocab, last dim = factor."""
    outs = [(ids // (vocab ** i)) % vocab for i in range(num_vocabs)]
    return torch.stack(outs, dim=-1)


def unfactorize_token_ids(fac: torch.Tensor, num_vocabs: int = 2, vocab: int = 512) -> torch.Tensor:
    """hma/model/factorization_utils.py:71-82."""
    out = torch.zeros(fac.shape[:-1], dtype=fac.dtype)
    for i in range(num_vocabs):
        out = out + fac[..., i] * (vocab ** i)
    return out


def factorize_labels(labels_THW: torch.Tensor, num_vocabs: int = 2, vocab: int = 512) -> torch.Tensor:
    """hma/model/factorization_utils.py:85-96 -- (B,T,H,W) -> (B,num_vocabs,T,H,W)."""
    return factorize_token_ids(labels_THW, num_vocabs, vocab).permute(0, 4, 1, 2, 3).contiguous()


def cosine_schedule(u: float) -> float:
    """hma/model/st_mask_git.py:116-125."""
    return math.cos(u * math.pi / 2)


# --------------------------------------------------------------------------------------
# floating-point blocks
# --------------------------------------------------------------------------------------
def token_embed(sd: SD, cfg: RefConfig, ids_TS: torch.Tensor) -> torch.Tensor:
    """hma/model/factorization_utils.py:31-54 -- mask rows get mask_token_embed, others E0[id%V]+E1[id//V]."""
    V = cfg.factored_vocab_size
    is_mask = ids_TS == cfg.mask_token_id
    safe = torch.where(is_mask, torch.zeros_like(ids_TS), ids_TS)
    emb = torch.zeros(ids_TS.shape + (cfg.d_model,), dtype=torch.float32)
    for i in range(cfg.num_factored_vocabs):
        emb = emb + sd[f"token_embed.factored_embeds.{i}.weight"][(safe // (V ** i)) % V]
    mask_vec = sd["token_embed.mask_token_embed"][0]
    return torch.where(is_mask[..., None], mask_vec.expand_as(emb), emb)


def action_stem(sd: SD, cfg: RefConfig, action_ids: torch.Tensor, domain: str,
                skip_normalization: bool = False) -> torch.Tensor:
    """ActionStat + BasicMLP: hma/model/st_mask_git.py:134-138, 90-102, 646-649."""
    a = action_ids
    if not skip_normalization:
        mean = sd[f"action_preprocessor.{domain}.mean"]
        std = sd[f"action_preprocessor.{domain}.std"]
        B, T, SD_ = a.shape
        d = mean.numel()
        a = ((a.reshape(B, T, SD_ // d, d) - mean) / (std + 1e-10)).reshape(B, T, SD_)
    p = f"action_mlp.{domain}.model"
    h = F.linear(a, sd[f"{p}.0.weight"], sd[f"{p}.0.bias"])
    h = F.layer_norm(h, (cfg.d_model,), sd[f"{p}.1.weight"], sd[f"{p}.1.bias"], 1e-5)
    h = torch.relu(h)
    return F.linear(h, sd[f"{p}.3.weight"], sd[f"{p}.3.bias"])


def self_attention(x: torch.Tensor, w_qkv: torch.Tensor, b_qkv: Optional[torch.Tensor],
                   w_proj: torch.Tensor, b_proj: Optional[torch.Tensor],
                   num_heads: int, scale: float, causal: bool,
                   qk_norm: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> torch.Tensor:
    """BasicSelfAttention.forward, hma/model/attention.py:37-61.  `qk_norm` = (weight, bias) of the attention's LayerNorm(head_dim):
    q and k are normalised per head with the SAME affine before the scale (:44-48).

    q is scaled before the product (`:49`); the causal fill is -finfo.max (`:52-56`);
    attn_drop is constructed but never applied.
    """
    Bn, N, C = x.shape
    hd = C // num_heads
    qkv = F.linear(x, w_qkv, b_qkv).reshape(Bn, N, 3, num_heads, hd)
    q = qkv[:, :, 0].transpose(1, 2)  # (Bn, h, N, hd)
    k = qkv[:, :, 1].transpose(1, 2)
    v = qkv[:, :, 2].transpose(1, 2)
    if qk_norm is not None:
        q = F.layer_norm(q, (hd,), qk_norm[0], qk_norm[1], 1e-5)
        k = F.layer_norm(k, (hd,), qk_norm[0], qk_norm[1], 1e-5)
    q = q * scale
    att = q @ k.transpose(-1, -2)
    if causal:
        keep = torch.ones(N, N, dtype=torch.bool).tril()
        att = att.masked_fill(~keep, -torch.finfo(att.dtype).max)
    att = att.softmax(dim=-1)
    o = (att @ v).transpose(1, 2).reshape(Bn, N, C)
    return F.linear(o, w_proj, b_proj)


def mlp(x: torch.Tensor, w1, b1, w2, b2) -> torch.Tensor:
    """Mlp.forward, hma/model/st_transformer.py:24-27 (exact-erf GELU, dropout p=0)."""
    return F.linear(F.gelu(F.linear(x, w1, b1)), w2, b2)


def modulate_layer(sd: SD, prefix: str, x_bstd: torch.Tensor, c_btd: torch.Tensor) -> torch.Tensor:
    """ModulateLayer.forward, hma/model/st_mask_git.py:66-76.

    x_bstd: (B, S, T, d); c_btd: (B, T, d) (the action embedding).  LN has no affine, eps 1e-6.
    """
    T = x_bstd.shape[2]
    c = c_btd[:, None, :T]
    h = F.silu(F.linear(c, sd[f"{prefix}.adaLN_modulation.0.weight"], sd[f"{prefix}.adaLN_modulation.0.bias"]))
    ss = F.linear(h, sd[f"{prefix}.adaLN_modulation.2.weight"], sd[f"{prefix}.adaLN_modulation.2.bias"])
    shift, scale = ss.chunk(2, dim=-1)
    xn = F.layer_norm(x_bstd, (x_bstd.shape[-1],), None, None, 1e-6)
    xm = xn * (1 + scale) + shift
    return F.linear(xm, sd[f"{prefix}.linear_out.weight"], sd[f"{prefix}.linear_out.bias"])


def st_block(sd: SD, cfg: RefConfig, l: int, x_btsd: torch.Tensor,
             a_emb: Optional[torch.Tensor], domain: Optional[str]) -> torch.Tensor:
    """STBlock.forward, hma/model/st_transformer.py:79-114 ("modulate" action processing)."""
    B, T, S, D = x_btsd.shape
    p = f"decoder.layers.{l}"
    g = lambda n: sd.get(f"{p}.{n}")
    # spatial, bidirectional, pre-LN (:85-86)
    xs = x_btsd.reshape(B * T, S, D)
    xn = xs if cfg.qk_norm else F.layer_norm(xs, (D,), g("norm1.weight"), g("norm1.bias"), 1e-5)
    xs = xs + self_attention(xn, g("spatial_attn.qkv.weight"), g("spatial_attn.qkv.bias"),
                             g("spatial_attn.proj.weight"), g("spatial_attn.proj.bias"),
                             cfg.num_heads, cfg.attn_scale, causal=False,
                             qk_norm=(g("spatial_attn.norm.weight"), g("spatial_attn.norm.bias")) if cfg.qk_norm else None)
    # (B S) T C view (:89)
    xt = xs.reshape(B, T, S, D).permute(0, 2, 1, 3)  # B S T D
    if a_emb is not None and domain is not None and "modulate" in cfg.action_network:
        xt = xt + modulate_layer(sd, f"{p}.action_projectors.{domain}", xt, a_emb)  # :102-104
    elif a_emb is not None and domain is not None and "mlp" in cfg.action_network:
        xt = xt + a_emb[:, None, :T]  # :96-99 (Identity projector)
    xt = xt.reshape(B * S, T, D)
    # causal temporal attention on the UN-normed stream (:111)
    xt = xt + self_attention(xt, g("temporal_attn.qkv.weight"), g("temporal_attn.qkv.bias"),
                             g("temporal_attn.proj.weight"), g("temporal_attn.proj.bias"),
                             cfg.num_heads, cfg.attn_scale, causal=True,
                             qk_norm=(g("temporal_attn.norm.weight"), g("temporal_attn.norm.bias")) if cfg.qk_norm else None)
    xn2 = xt if cfg.qk_norm else F.layer_norm(xt, (D,), g("norm2.weight"), g("norm2.bias"), 1e-5)
    xt = xt + mlp(xn2, g("mlp.fc1.weight"), g("mlp.fc1.bias"), g("mlp.fc2.weight"), g("mlp.fc2.bias"))  # :112
    return xt.reshape(B, S, T, D).permute(0, 2, 1, 3).contiguous()


def trunk_input(sd: SD, cfg: RefConfig, x_THW: torch.Tensor, action_ids: Optional[torch.Tensor],
                domain: Optional[Sequence[str]], skip_normalization: bool = False,
                action_mask: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """compute_logits up to the decoder call, hma/model/st_mask_git.py:640-672.  `action_mask` (B, T), 1 = masked: with
    jointly_predict_actions the concatenated action tokens of a masked frame are `action_mask_tokens[t]` (:656-660); the
    modulation further down still sees the embedded actions."""
    B, T = x_THW.shape[:2]
    x = token_embed(sd, cfg, x_THW.reshape(B, T, -1))
    a_emb = None
    if action_ids is not None:
        a_emb = action_stem(sd, cfg, action_ids, domain[0], skip_normalization)
        if "concat" in cfg.action_network:
            cond = a_emb[:, :T, None].expand(B, T, cfg.action_token_size, cfg.d_model)
            if action_mask is not None:
                m = action_mask[:, :T, None, None].to(cond.dtype)
                cond = m * sd["action_mask_tokens"][:, :T] + (1 - m) * cond
            x = torch.cat([x, cond], dim=2)
    x = x + sd["pos_embed_TSC"][:, :T, : x.shape[2]]
    return x, a_emb


def compute_logits(sd: SD, cfg: RefConfig, x_THW: torch.Tensor, action_ids: Optional[torch.Tensor] = None,
                   domain: Optional[Sequence[str]] = None, skip_normalization: bool = False) -> torch.Tensor:
    """STMaskGIT.compute_logits, hma/model/st_mask_git.py:632-686 -> logits (B, C, T, H, W)."""
    B, T, H, W = x_THW.shape
    x, a_emb = trunk_input(sd, cfg, x_THW, action_ids, domain, skip_normalization)
    dom = domain[0] if domain is not None else None
    for l in range(cfg.num_layers):
        x = st_block(sd, cfg, l, x, a_emb, dom)
    x = x[:, :, : H * W]
    # FixedMuReadout at width_mult == 1, output_mult == 1 is a plain Linear (:784-789)
    logits = F.linear(x, sd["out_x_proj.weight"], sd["out_x_proj.bias"])  # (B,T,S,C)
    return logits.reshape(B, T, H, W, -1).permute(0, 4, 1, 2, 3)


def compute_logits_and_actions(sd: SD, cfg: RefConfig, x_THW: torch.Tensor, action_ids: torch.Tensor, domain: Sequence[str],
                               action_mask: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """compute_logits with jointly_predict_actions (hma/model/st_mask_git.py:656-660, 676-683): also the action read-out -- the
    mean of a frame's action tokens through `action_out_projectors[domain]` (a plain Linear at width_mult == 1) -> (B, T, d_action)."""
    B, T, H, W = x_THW.shape
    if action_ids is None:
        # policy mode (:663-666): every action token is action_mask_tokens[t]; the decoder gets no actions, hence no modulation
        x = token_embed(sd, cfg, x_THW.reshape(B, T, -1))
        cond = sd["action_mask_tokens"][:, :T].expand(B, T, cfg.action_token_size, cfg.d_model)
        x = torch.cat([x, cond], dim=2) + sd["pos_embed_TSC"][:, :T, : H * W + cfg.action_token_size]
        a_emb = None
    else:
        x, a_emb = trunk_input(sd, cfg, x_THW, action_ids, domain, action_mask=action_mask)
    dom = domain[0]
    for l in range(cfg.num_layers):
        x = st_block(sd, cfg, l, x, a_emb, dom if a_emb is not None else None)
    pooled = x[:, :, -cfg.action_token_size:].mean(dim=2)
    actions = F.linear(pooled, sd[f"action_out_projectors.{dom}.weight"], sd[f"action_out_projectors.{dom}.bias"])
    logits = F.linear(x[:, :, : H * W], sd["out_x_proj.weight"], sd["out_x_proj.bias"])
    return logits.reshape(B, T, H, W, -1).permute(0, 4, 1, 2, 3), actions


def forward_with_actions(sd: SD, cfg: RefConfig, input_ids: torch.Tensor, labels: torch.Tensor, action_ids: torch.Tensor,
                         domain: Sequence[str], action_mask: torch.Tensor, H: int = 16, W: int = 16):
    """STMaskGIT.forward with jointly_predict_actions, hma/model/st_mask_git.py:688-733, the (B, T) action mask given instead of
    drawn (:704-710).  action_loss as the reference computes it: `mse_loss(action_labels, action_outputs, reduce="none")` goes
    through the legacy-argument path, where any truthy `reduce` means the MEAN, so the "elementwise" loss is a scalar and
    (scalar * mask).mean() = mean squared error (against the RAW action ids) x the fraction of masked frames (:725-726)."""
    B = input_ids.shape[0]
    x_THW = input_ids.reshape(B, cfg.T, H, W)
    logits, actions = compute_logits_and_actions(sd, cfg, x_THW, action_ids, domain, action_mask)
    loss, acc = video_loss_and_acc(cfg, logits, labels, x_THW)
    action_loss = ((action_ids[:, : cfg.T] - actions) ** 2).mean() * action_mask.to(actions.dtype).mean()
    return loss, acc, logits, action_loss, actions


def video_loss_and_acc(cfg: RefConfig, logits_CTHW: torch.Tensor, labels_flat: torch.Tensor,
                       x_THW: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """compute_video_loss_and_acc + the mask rule of forward: hma/model/st_mask_git.py:603-630, 714-716."""
    B, C, T, H, W = logits_CTHW.shape
    V, NV = cfg.factored_vocab_size, cfg.num_factored_vocabs
    targets = labels_flat.reshape(B, T, H, W)[:, 1:]
    lg = logits_CTHW[:, :, 1:].reshape(B, NV, V, T - 1, H, W).permute(0, 2, 1, 3, 4, 5)  # b V NV t h w
    ft = factorize_labels(targets, 2, 512)  # hard-coded defaults, st_mask_git.py:617
    loss = F.cross_entropy(lg, ft, reduction="none", label_smoothing=0.01).sum(dim=1)
    acc = (lg.argmax(dim=1) == ft).all(dim=1)
    m = (x_THW[:, 1:] == cfg.mask_token_id)
    n = m.sum()
    return (loss * m).sum() / n, (acc * m).sum().float() / n


def forward(sd: SD, cfg: RefConfig, input_ids: torch.Tensor, labels: torch.Tensor,
            action_ids: Optional[torch.Tensor] = None, domain: Optional[Sequence[str]] = None,
            H: int = 16, W: int = 16):
    """STMaskGIT.forward, hma/model/st_mask_git.py:688-735 (jointly_predict_actions=False)."""
    B = input_ids.shape[0]
    x_THW = input_ids.reshape(B, cfg.T, H, W)
    logits = compute_logits(sd, cfg, x_THW, action_ids, domain)
    loss, acc = video_loss_and_acc(cfg, logits, labels, x_THW)
    return loss, acc, logits


@torch.no_grad()
def maskgit_generate(sd: SD, cfg: RefConfig, prompt_THW: torch.Tensor, out_t: int, maskgit_steps: int = 1,
                     temperature: float = 0.0, unmask_mode: str = "random", action_ids=None, domain=None,
                     rand_draws: Optional[List[torch.Tensor]] = None, sample_draws: Optional[List[torch.Tensor]] = None,
                     trace: Optional[list] = None):
    """STMaskGIT.maskgit_generate, hma/model/st_mask_git.py:338-467.

    `rand_draws[step]` (B, H, W) replaces `torch.rand_like` (:435) so the "random" mode is replayable.
    temperature > 1e-8 takes the Categorical branch (:411-416): `Categorical(probs / temperature).sample()` normalises
    the temperature away and draws through torch.multinomial's single-sample path, argmax_k p_k / q_k with q ~ Exp(1);
    `sample_draws[step]` (B, H*W, NV, V) are those q (factor v at [..., v, :]).
    Mutates `prompt_THW` in place like the reference (:453).  Returns (samples_HW, first-pass
    factored logits (B, V, NV, H, W)).  `trace` (a list) receives per step (logits B C H W, confidences handed to argsort).
    """
    assert out_t, "maskgit_generate requires out_t > 0"
    assert torch.all(prompt_THW[:, out_t:] == cfg.mask_token_id)
    B, T, H, W = prompt_THW.shape
    S = H * W
    V, NV = cfg.factored_vocab_size, cfg.num_factored_vocabs
    unmasked = torch.zeros(B, S, dtype=torch.bool)
    first_logits = None
    samples_HW = None
    for step in range(maskgit_steps):
        logits = compute_logits(sd, cfg, prompt_THW, action_ids, domain)[:, :, out_t]  # B C H W
        if first_logits is None:
            first_logits = logits.clone()
        fl = logits.reshape(B, NV, V, H, W)
        probs = fl.softmax(dim=2)
        samples = torch.zeros(B, H, W, dtype=torch.long)
        conf = torch.ones(B, H, W)
        for v in reversed(range(NV)):  # flip(2): highest factor first (:408)
            if temperature <= 1e-8:
                s = probs[:, v].argmax(dim=1)
            else:
                pr = probs[:, v].permute(0, 2, 3, 1) / temperature           # b h w vocab (:413-414)
                pr = pr / pr.sum(-1, keepdim=True)                            # Categorical.__init__ normalises
                q = sample_draws[step][:, :, v].reshape(B, H, W, V)
                s = (pr / q).argmax(dim=-1)                                   # torch.multinomial(p, 1): argmax p / q, q ~ Exp(1)
            samples = samples * V + s
            conf = conf * torch.gather(probs[:, v], 1, s[:, None])[:, 0]
        prev_unmasked = unmasked.clone()
        prev_flat = prompt_THW[:, out_t].reshape(B, S).clone()
        flat = samples.reshape(B, S)
        if step != maskgit_steps - 1:
            n = math.ceil(cosine_schedule((step + 1) / maskgit_steps) * S)
            if unmask_mode == "greedy":
                c = conf.reshape(B, S).clone()
            elif unmask_mode == "random":
                c = rand_draws[step].reshape(B, S).clone()
            else:
                raise NotImplementedError(unmask_mode)
            c[unmasked] = torch.inf
            if trace is not None:
                trace.append((logits.clone(), c.clone()))
            order = torch.argsort(c, dim=1)
            unmasked.scatter_(1, order[:, n:], True)
            flat.scatter_(1, order[:, :n], cfg.mask_token_id)
        elif trace is not None:
            trace.append((logits.clone(), None))
        flat[prev_unmasked] = prev_flat[prev_unmasked]
        samples_HW = flat.reshape(B, H, W)
        prompt_THW[:, out_t] = samples_HW
    return samples_HW, first_logits.reshape(B, NV, V, H, W).permute(0, 2, 1, 3, 4)


def maskgit_select(conf: torch.Tensor, samples: torch.Tensor, unmasked: torch.Tensor, prev: torch.Tensor,
                   n: int, mask_id: int, last: bool):
    """The index half of one MaskGIT step, hma/model/st_mask_git.py:422-453, given confidences.

    conf (B,S) float32, samples (B,S) int64, unmasked (B,S) bool, prev (B,S) int64.
    Returns (new_samples, new_unmasked).  A *stable* rank is used so ties keep index order.
    """
    B, S = conf.shape
    unm = unmasked.clone()
    out = samples.clone()
    if not last:
        c = conf.clone()
        c[unmasked] = torch.inf
        order = torch.argsort(c, dim=1, stable=True)
        unm.scatter_(1, order[:, n:], True)
        out.scatter_(1, order[:, :n], mask_id)
    out[unmasked] = prev[unmasked]
    return out, unm


# --------------------------------------------------------------------------------------
# optimizer step (hma/train_multi.py:593-598, 907-922; torch.optim.AdamW semantics)
# --------------------------------------------------------------------------------------
def clip_and_adamw(params: Dict[str, torch.Tensor], grads: Dict[str, Optional[torch.Tensor]],
                   m: Dict[str, torch.Tensor], v: Dict[str, torch.Tensor], step: int, lr: float,
                   betas=(0.9, 0.95), eps: float = 1e-8, weight_decay: float = 0.05,
                   max_norm: Optional[float] = 1.0) -> float:
    """Global-norm clip over non-None grads then AdamW; names containing "bias" are not decayed.

    Params whose grad is None are skipped entirely (no moment update, no decay) -- what DDP's
    globally-unused rule + zero_grad(set_to_none) gives.  `step` is 1-based.
    Returns the pre-clip total norm.
    """
    live = [n for n in params if grads.get(n) is not None]
    total = torch.sqrt(sum((grads[n].double() ** 2).sum() for n in live)).item()
    coef = 1.0
    if max_norm is not None:
        coef = min(1.0, max_norm / (total + 1e-6))
    b1, b2 = betas
    for n in live:
        g = grads[n] * coef
        wd = 0.0 if "bias" in n else weight_decay
        params[n].mul_(1 - lr * wd)
        m[n].mul_(b1).add_(g, alpha=1 - b1)
        v[n].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** step
        bc2 = 1 - b2 ** step
        denom = (v[n].sqrt() / math.sqrt(bc2)).add_(eps)
        params[n].addcdiv_(m[n], denom, value=-lr / bc1)
    return total
