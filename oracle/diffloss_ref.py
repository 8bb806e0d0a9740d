"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain PyTorch fp32, numpy fp64 tables) of the diffusion head.

Follows /root/reference/hma/model/diffloss.py (DiffLoss :10-59, SimpleMLPAdaLN :152-233, ResBlock :99-124,
FinalLayer :127-149, TimestepEmbedder :66-96) and /root/reference/hma/diffusion/gaussian_diffusion.py (cosine betas
:112-137, tables :139-189, q_sample :203-219, p_mean_variance :250-325, _vb_terms_bpd :650-673, training_losses
:675-745, p_sample :358-394, p_sample_loop :396-493), respace.py:8-119 (timestep map of the sampling process),
diffusion_utils.py (normal_kl, discretized_gaussian_log_likelihood).  Pinned by tests/golden/g9_diffloss.safetensors
(captured from the real reference by tests/golden/make_golden_diffloss.py).  Only tests may import this.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------- schedule tables
def cosine_betas(n: int = 1000, max_beta: float = 0.999) -> np.ndarray:
    ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    return np.array([min(1 - ab((i + 1) / n) / ab(i / n), max_beta) for i in range(n)], dtype=np.float64)


def space_timesteps(num_timesteps: int, section_counts) -> list:
    if isinstance(section_counts, str):
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per, extra = num_timesteps // len(section_counts), num_timesteps % len(section_counts)
    start, steps = 0, []
    for i, cnt in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        stride = 1 if cnt <= 1 else (size - 1) / (cnt - 1)
        cur = 0.0
        for _ in range(cnt):
            steps.append(start + round(cur))
            cur += stride
        start += size
    return sorted(set(steps))


class Tables:
    """fp64 schedule tables of GaussianDiffusion.__init__; `timestep_map[i]` is the original step fed to the network."""

    def __init__(self, betas: np.ndarray, timestep_map=None):
        betas = np.asarray(betas, dtype=np.float64)
        self.betas = betas
        self.n = len(betas)
        ac = np.cumprod(1.0 - betas)
        ac_prev = np.append(1.0, ac[:-1])
        self.sqrt_ac, self.sqrt_1mac = np.sqrt(ac), np.sqrt(1.0 - ac)
        self.sqrt_recip_ac, self.sqrt_recipm1_ac = np.sqrt(1.0 / ac), np.sqrt(1.0 / ac - 1)
        pv = betas * (1.0 - ac_prev) / (1.0 - ac)
        self.post_logvar = np.log(np.append(pv[1], pv[1:])) if self.n > 1 else np.array([])
        self.coef1 = betas * np.sqrt(ac_prev) / (1.0 - ac)
        self.coef2 = (1.0 - ac_prev) * np.sqrt(1.0 - betas) / (1.0 - ac)
        self.log_betas = np.log(betas)
        self.timestep_map = list(range(self.n)) if timestep_map is None else list(timestep_map)

    @staticmethod
    def train():
        return Tables(cosine_betas())

    @staticmethod
    def sampling(num_sampling_steps="100"):
        base = np.cumprod(1.0 - cosine_betas())
        use = space_timesteps(1000, num_sampling_steps)
        last, nb = 1.0, []
        for i in use:
            nb.append(1 - base[i] / last)
            last = base[i]
        return Tables(np.array(nb), use)

    def at(self, arr, t, like):
        return torch.from_numpy(arr)[t].float()[:, None] + torch.zeros_like(like)


# ----------------------------------------------------------------------------------------------- network
def timestep_embedding(t, dim=256, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def net_forward(P, x, t, c, depth, pre="net."):
    """SimpleMLPAdaLN.forward: x (N, C), t (N,), c (N, z) -> (N, 2C)."""
    g = lambda k: P[pre + k]
    lin = lambda v, k: F.linear(v, g(k + ".weight"), g(k + ".bias"))
    x = lin(x, "input_proj")
    te = lin(F.silu(lin(timestep_embedding(t), "time_embed.mlp.0")), "time_embed.mlp.2")
    y = te + lin(c, "cond_embed")
    sy = F.silu(y)
    w = x.shape[-1]
    for i in range(depth):
        b = f"res_blocks.{i}."
        shift, scale, gate = lin(sy, b + "adaLN_modulation.1").chunk(3, dim=-1)
        h = F.layer_norm(x, (w,), g(b + "in_ln.weight"), g(b + "in_ln.bias"), 1e-6) * (1 + scale) + shift
        h = lin(F.silu(lin(h, b + "mlp.0")), b + "mlp.2")
        x = x + gate * h
    shift, scale = lin(sy, "final_layer.adaLN_modulation.1").chunk(2, dim=-1)
    return lin(F.layer_norm(x, (w,), None, None, 1e-6) * (1 + scale) + shift, "final_layer.linear")


# ----------------------------------------------------------------------------------------------- losses
def _cdf(x):
    return 0.5 * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def _disc_loglik(x, means, log_scales):
    cx = x - means
    inv = torch.exp(-log_scales)
    cp, cm = _cdf(inv * (cx + 1.0 / 255.0)), _cdf(inv * (cx - 1.0 / 255.0))
    lcp = torch.log(cp.clamp(min=1e-12))
    l1m = torch.log((1.0 - cm).clamp(min=1e-12))
    return torch.where(x < -0.999, lcp, torch.where(x > 0.999, l1m, torch.log((cp - cm).clamp(min=1e-12))))


def p_mean_logvar(tb, x, t, out, clip_denoised=False):
    C = x.shape[1]
    eps, v = out[:, :C], out[:, C:]
    min_log, max_log = tb.at(tb.post_logvar, t, x), tb.at(tb.log_betas, t, x)
    frac = (v + 1) / 2
    logvar = frac * max_log + (1 - frac) * min_log
    x0 = tb.at(tb.sqrt_recip_ac, t, x) * x - tb.at(tb.sqrt_recipm1_ac, t, x) * eps
    if clip_denoised:
        x0 = x0.clamp(-10, 10)
    mean = tb.at(tb.coef1, t, x) * x0 + tb.at(tb.coef2, t, x) * x
    return mean, logvar, x0


def training_losses(tb, P, x0, t, noise, z, depth):
    """Per-row loss = mse(eps) + vb(learned range, mean frozen)  (LossType.MSE, ModelVarType.LEARNED_RANGE)."""
    C = x0.shape[1]
    x_t = tb.at(tb.sqrt_ac, t, x0) * x0 + tb.at(tb.sqrt_1mac, t, x0) * noise
    out = net_forward(P, x_t, torch.tensor(tb.timestep_map)[t], z, depth)
    eps, v = out[:, :C], out[:, C:]
    frozen = torch.cat([eps.detach(), v], dim=1)
    true_mean = tb.at(tb.coef1, t, x0) * x0 + tb.at(tb.coef2, t, x0) * x_t
    true_lv = tb.at(tb.post_logvar, t, x0)
    mean, logvar, _ = p_mean_logvar(tb, x_t, t, frozen)
    kl = 0.5 * (-1.0 + logvar - true_lv + torch.exp(true_lv - logvar) + (true_mean - mean) ** 2 * torch.exp(-logvar))
    kl = kl.mean(dim=1) / math.log(2.0)
    nll = -_disc_loglik(x0, mean, 0.5 * logvar).mean(dim=1) / math.log(2.0)
    vb = torch.where(t == 0, nll, kl)
    mse = ((noise - eps) ** 2).mean(dim=1)
    return mse + vb, out


def diffloss_forward(P, target, z, mask, t, noise, depth):
    """DiffLoss.forward with the draws passed in: (sum loss * mask) / (sum mask + 1e-8)."""
    loss, out = training_losses(Tables.train(), P, target, t, noise, z, depth)
    if mask is not None:
        loss = (loss * mask).sum() / (mask.sum() + 1e-8)
    return loss.mean(), out


@torch.no_grad()
def diffloss_sample(P, z, noise0, step_noises, depth, temperature=1.0, num_sampling_steps="100", clip_denoised=False, cfg=1.0):
    """DiffLoss.sample (diffloss.py:37-59): p_sample_loop over the respaced process; step_noises[k] is the k-th randn_like.
    cfg != 1: noise0 holds the half-batch start noise (duplicated, :40-41) and every step runs forward_with_cfg (:235-243) --
    the network sees the FIRST half of x twice (conditions z = [cond | uncond]), eps of both halves becomes
    uncond + cfg * (cond - uncond), the variance channels stay per row."""
    tb = Tables.sampling(num_sampling_steps)
    x = noise0 if cfg == 1.0 else torch.cat([noise0, noise0], dim=0)
    C = x.shape[1]
    for k, i in enumerate(reversed(range(tb.n))):
        t = torch.full((x.shape[0],), i, dtype=torch.long)
        tm = torch.tensor(tb.timestep_map)[t]
        if cfg == 1.0:
            out = net_forward(P, x, tm, z, depth)
        else:
            h = x.shape[0] // 2
            out = net_forward(P, torch.cat([x[:h], x[:h]], dim=0), tm, z, depth)
            eps = out[h:, :C] + cfg * (out[:h, :C] - out[h:, :C])
            out = torch.cat([torch.cat([eps, eps], dim=0), out[:, C:]], dim=1)
        mean, logvar, _ = p_mean_logvar(tb, x, t, out, clip_denoised)
        nz = 0.0 if i == 0 else 1.0
        x = mean + nz * torch.exp(0.5 * logvar) * step_noises[k] * temperature
    return x
