"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain PyTorch fp32) of STMAR's training forward.

Follows /root/reference/hma/model/st_mar.py: patchify :199-207, forward :219-275 (mask-latent fill :245, patch mask
:260), compute_latents :146-197 (token_embed Linear, action tokens, pos_embed, z_proj_ln, trunk, out_x_proj +
decoder_norm + diffusion_pos_embed), compute_video_loss_and_acc :131-144 (DiffLoss on every patch row with the patch
mask).  The trunk is oracle/st_maskgit_ref.py's, the head oracle/diffloss_ref.py's.  Pinned by
tests/golden/g11_stmar.safetensors (tests/golden/make_golden_stmar.py).  Only tests may import this.
"""
import torch
import torch.nn.functional as F

from . import diffloss_ref as DR
from . import st_maskgit_ref as R


def patchify(x, p):
    """(B, T, H, W, C) -> (B, T, H/p, W/p, p*p*C), channel order (p, q, c)  (st_mar.py:199-207)."""
    B, T, H, W, C = x.shape
    x = x.reshape(B, T, H // p, p, W // p, p, C)
    return torch.einsum("nthpwqc->nthwpqc", x).reshape(B, T, H // p, W // p, C * p * p)


def compute_latents(sd, cfg, x_patches, action_ids, domain, with_actions=False):
    """x_patches (B, T, h, w, 16) -> z (B, T, h*w, d); `with_actions` (jointly_predict_actions, st_mar.py:187-189): also the mean of
    every frame's action tokens (B, T, d) -- the action mask is NOT applied to the inputs here (:146-172 ignore it)."""
    B, T, h, w, _ = x_patches.shape
    x = F.linear(x_patches.reshape(B, T, h * w, -1).float(), sd["token_embed.weight"])
    a_emb = None
    if action_ids is not None:  # (st_mar.py:154-172; without actions: no action tokens, the decoder runs unconditioned)
        a_emb = R.action_stem(sd, cfg, action_ids, domain[0])
        x = torch.cat([x, a_emb[:, :T, None].expand(B, T, cfg.action_token_size, cfg.d_model)], dim=2)
    x = x + sd["pos_embed_TSC"][:, :T, : x.shape[2]]
    x = F.layer_norm(x, (cfg.d_model,), sd["z_proj_ln.weight"], sd["z_proj_ln.bias"], 1e-6)
    for l in range(cfg.num_layers):
        x = R.st_block(sd, cfg, l, x, a_emb, domain[0] if action_ids is not None else None)
    pooled = x[:, :, -cfg.action_token_size:].mean(dim=2)
    x = x[:, :, : h * w]
    y = F.linear(x, sd["out_x_proj.weight"], sd["out_x_proj.bias"])
    y = F.layer_norm(y, (cfg.d_model,), sd["decoder_norm.weight"], sd["decoder_norm.bias"], 1e-6)
    z = y + sd["diffusion_pos_embed_learned"].view(1, -1, h * w, cfg.d_model)[:, :T]
    return (z, pooled) if with_actions else z


def forward(sd, cfg, input_ids, labels, action_ids, domain, masked, t, noise, patch_size, H, W, diff_depth):
    """STMAR.forward loss with the diffusion draws (t, noise) passed in.  input_ids / labels (B, T*H*W, C) float."""
    B = input_ids.shape[0]
    T = cfg.T
    x = input_ids.reshape(B, T, H, W, -1).clone()
    x[masked] = sd["mask_token"].reshape(-1)
    z = compute_latents(sd, cfg, patchify(x, patch_size), action_ids, domain)
    target = patchify(labels.reshape(B, T, H, W, -1), patch_size)
    pmask = patchify(masked[..., None].float(), patch_size).sum(-1) > 0
    n = B * T * z.shape[2]
    P = {k[len("diffloss."):]: v for k, v in sd.items() if k.startswith("diffloss.")}
    loss, _ = DR.diffloss_forward(P, target.reshape(n, -1).float(), z.reshape(n, -1), pmask.reshape(n).float(), t, noise, diff_depth)
    return loss, z


def forward_with_actions(sd, cfg, input_ids, labels, action_ids, domain, masked, t, noise, patch_size, H, W, diff_depth, action_mask,
                         t_act, noise_act):
    """STMAR.forward with jointly_predict_actions (st_mar.py:231-273): the video loss as `forward`, plus the per-domain action
    diffusion head (`action_diff_losses[domain]`, :119-129) on the pooled action tokens against the RAW action ids, masked-mean over
    the (B, T) action mask (frames from a drawn timestep on, :234-240, given here)."""
    B = input_ids.shape[0]
    T = cfg.T
    x = input_ids.reshape(B, T, H, W, -1).clone()
    x[masked] = sd["mask_token"].reshape(-1)
    z, pooled = compute_latents(sd, cfg, patchify(x, patch_size), action_ids, domain, with_actions=True)
    target = patchify(labels.reshape(B, T, H, W, -1), patch_size)
    pmask = patchify(masked[..., None].float(), patch_size).sum(-1) > 0
    n = B * T * z.shape[2]
    P = {k[len("diffloss."):]: v for k, v in sd.items() if k.startswith("diffloss.")}
    loss, _ = DR.diffloss_forward(P, target.reshape(n, -1).float(), z.reshape(n, -1), pmask.reshape(n).float(), t, noise, diff_depth)
    pre = f"action_diff_losses.{domain[0]}."
    PA = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    aloss, _ = DR.diffloss_forward(PA, action_ids[:, :T].reshape(B * T, -1).float(), pooled.reshape(B * T, -1),
                                   action_mask.reshape(B * T).float(), t_act, noise_act, diff_depth)
    return loss, z, aloss, pooled


def unpatchify(x, p, c):
    """(B, T, h, w, p*p*c) -> (B, T, h*p, w*p, c)  (st_mar.py:209-217)."""
    B, T, h, w, _ = x.shape
    return torch.einsum("nthwpqc->nthpwqc", x.reshape(B, T, h, w, p, p, c)).reshape(B, T, h * p, w * p, c)


@torch.no_grad()
def maskgit_generate(sd, cfg, prompt_THWC, out_t, maskgit_steps, temperature, action_ids, domain, orders, draws, patch_size, diff_depth,
                     num_sampling_steps):
    """STMAR.maskgit_generate (st_mar.py:362-452, cfg = 1) with the random order and every Gaussian draw passed in:
    draws[k] = (noise0, step_noises) of the k-th DiffLoss.sample call.  NB the reference never updates `unmasked`, so
    every step re-predicts all tokens outside the first mask_len entries of the order, and the last step all of them."""
    import math
    import numpy as np
    x = patchify(prompt_THWC, patch_size).clone()
    B, T, h, w, pc = x.shape
    S = h * w
    P = {k[len("diffloss."):]: v for k, v in sd.items() if k.startswith("diffloss.")}
    z = compute_latents(sd, cfg, x, action_ids, domain)[:, out_t]
    orig = z.clone()
    for step in range(maskgit_steps):
        if step > 0:
            z = compute_latents(sd, cfg, x, action_ids, domain)[:, out_t]
        mask_len = max(1, min(S - 1, int(np.floor(S * np.cos(math.pi / 2.0 * (step + 1) / maskgit_steps)))))
        mask_next = torch.zeros(B, S).scatter(-1, orders[:, :mask_len], torch.ones(B, S)).bool()
        to_pred = torch.ones(B, S, dtype=torch.bool) if step >= maskgit_steps - 1 else ~mask_next
        rows = z[to_pred]
        noise0, step_noises = draws[step]
        smp = DR.diffloss_sample(P, rows, noise0, list(step_noises), diff_depth, temperature, num_sampling_steps, clip_denoised=True)
        xt = x[:, out_t].reshape(B, S, pc)
        xt[to_pred] = smp
        x[:, out_t] = xt.reshape(B, h, w, pc)
    c = pc // (patch_size ** 2)
    return unpatchify(x, patch_size, c)[:, out_t], orig
