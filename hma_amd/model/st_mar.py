"""Spatial-temporal MAR: the continuous-latent model with a diffusion head, `STMAR`.

Mirror of hma/model/st_mar.py for training: same constructor (`DiffusionGenieConfig`), parameter names / shapes
(`mask_token`, `token_embed.weight (d, p*p*c)`, `z_proj_ln`, `decoder.*`, `out_x_proj (d, d)`, `decoder_norm`,
`diffusion_pos_embed_learned`, `pos_embed_TSC`, `diffloss.net.*`, per-domain `action_*`), `init_action_projectors`,
and `forward(input_ids, labels, action_ids, domain, masked_tokens_indicator=..., h=, w=) -> ModelOutput(loss, acc, logits)`.

Execution: the ST-transformer trunk, action stems and per-layer modulation are the discrete model's engine (the decoder
modules and their parameters are shared with an internal `STMaskGIT` whose flat buffers they live in); around it
`hma_mar_patchify` -> token_embed GEMM -> `hma_mar_embed_fwd` -> trunk -> out_x_proj GEMM -> `hma_mar_readout_fwd` ->
`DiffLoss` (hma_amd/model/diffloss.py), and the mirror image backward.  No CPU / eager-PyTorch path.
`maskgit_generate` / `generate` are the MAR decode (st_mar.py:277-452) with `DiffLoss.sample`.
`jointly_predict_actions`: training (the per-domain action diffusion heads, st_mar.py:119-129, 231-273) and the MAR decode's action
sampling (:441-446: `maskgit_generate` returns the sampled actions as its third value).  Not built: cfg != 1 in the MAR decode (the reference's own branch,
st_mar.py:417-418, indexes bs rows of latents with a 2 bs mask and cannot run; `DiffLoss.sample(cfg=...)` itself is built).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional

import torch
import torch.nn as nn
from huggingface_hub import PyTorchModelHubMixin
from transformers.utils import ModelOutput

from .. import _lib
from .._lib import A_BF16, A_F32, EPI_F32
from ..config import DiffusionGenieConfig, GenieConfig
from ..ops import make_gemm_nt, make_gemm_tn, ptr
from .diffloss import DiffLoss
from .st_mask_git import FixedMuReadout, STMaskGIT

BF16, F32 = torch.bfloat16, torch.float32
_PAD = 128


class _MarLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, owner, value):
        ctx.owner = owner
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        ctx.owner.__dict__["_bwd_ran"] = True
        ctx.owner._backward(g)
        return None, None, None


class _MarActionLoss(torch.autograd.Function):
    """jointly_predict_actions: the action loss as a second autograd leaf whose backward only records d total / d action_loss; it is
    created after `_MarLoss`' node, so autograd runs it first and the one `_backward` of the step finds the scale."""

    @staticmethod
    def forward(ctx, anchor, owner, value):
        ctx.owner = owner
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        owner = ctx.owner
        owner._act_scale = g.detach().clone()
        owner.__dict__["_bwd_ran"] = False

        def after_pass():  # (an objective without the video loss: the one backward of the step still has to run, with a zero video scale)
            if not owner.__dict__.get("_bwd_ran", False):
                owner.__dict__["_bwd_ran"] = True
                owner._backward(torch.zeros_like(g))

        torch.autograd.Variable._execution_engine.queue_callback(after_pass)
        return None, None, None


class STMAR(nn.Module, PyTorchModelHubMixin):
    """Spatial-Time MAR with VisionTransformer backbone (st_mar.py:36-78)."""

    def __init__(self, config: DiffusionGenieConfig):
        super().__init__()
        if isinstance(config, dict):
            config = DiffusionGenieConfig.from_dict(config)
        if config.diffusion_batch_mul < 1:
            raise ValueError("diffusion_batch_mul must be >= 1")
        self.config = config
        self.patch_size, self.vae_embed_dim = config.patch_size, config.vae_embed_dim
        self.h = self.w = math.isqrt(config.S)
        assert self.h ** 2 == config.S, "Expected S to be square"
        self.seq_len = config.S // (config.patch_size ** 2)          # patch tokens per frame
        d, T, pc = config.d_model, config.T, config.vae_embed_dim * config.patch_size ** 2
        if pc > 64:
            raise NotImplementedError("patch channels > 64")
        # the trunk (+ action stems / modulation) of the discrete model, on the patch-token grid
        core_cfg = GenieConfig.from_dict({**config.to_dict(), "S": self.seq_len, "jointly_predict_actions": False, "init_actions": False,
                                          "action_domains": None, "d_actions": None, "action_stats": None})
        object.__setattr__(self, "_core_box", [STMaskGIT(core_cfg)])  # not a registered sub-module: only shared pieces are exposed
        core = self._core
        self.decoder = core.decoder
        self.action_mask_tokens = core.action_mask_tokens
        self.pos_embed_TSC = nn.Parameter(torch.zeros(1, T, config.S + config.action_token_size, d))
        self.mask_token = nn.Parameter(torch.zeros(1, 1, config.vae_embed_dim))
        self.token_embed = nn.Linear(pc, d, bias=False)
        cls = FixedMuReadout if config.use_mup else nn.Linear
        self.out_x_proj = cls(d, d)
        self.decoder_norm = nn.LayerNorm(d, eps=1e-6)
        self.z_proj_ln = nn.LayerNorm(d, eps=1e-6)
        self.diffusion_pos_embed_learned = nn.Parameter(torch.zeros(1, self.seq_len * T, d))
        self.diffloss = DiffLoss(target_channels=pc, z_channels=d, width=config.diffloss_w, depth=config.diffloss_d,
                                 num_sampling_steps=config.num_sampling_steps, grad_checkpointing=config.grad_checkpointing)
        self.diffusion_batch_mul = config.diffusion_batch_mul
        torch.nn.init.normal_(self.diffusion_pos_embed_learned, std=0.02)
        self._anchor: Optional[torch.Tensor] = None
        self._saved = None
        self._grads_live = False
        if (config.init_actions or config.use_actions) and config.action_domains is not None:
            self.init_action_projectors(config.action_domains, config.d_actions, config.action_stats, config.action_network)

    @property
    def _core(self) -> STMaskGIT:
        return self._core_box[0]

    def init_action_projectors(self, domains: List[str], d_actions: List[int], action_stats, action_network: str = "mlp"):
        """st_mar.py:80-99: the discrete model's per-domain stems / projectors plus one DiffLoss head per domain."""
        core = self._core
        core.init_action_projectors(domains, d_actions, action_stats, action_network)  # (its unused action_out_projectors stay internal)
        self._ver_params = None  # (the parameter set the engine's version check sums over has changed)
        self._drop_param_lists()
        self.config.init_actions = True
        self.config.action_domains, self.config.d_actions, self.config.action_stats = list(domains), list(d_actions), action_stats
        self.action_preprocessor = core.action_preprocessor
        self.action_mlp = core.action_mlp
        self.action_diff_losses = nn.ModuleDict()
        for dom, da in zip(domains, d_actions):
            self.action_diff_losses[dom] = DiffLoss(target_channels=da, z_channels=self.config.d_model, width=self.config.diffloss_w,
                                                    depth=self.config.diffloss_d, num_sampling_steps=self.config.num_sampling_steps)

    # ------------------------------------------------------------------------------------------ helpers
    @staticmethod
    def _cast(stream, w: torch.Tensor, rows: int, cols: int, transpose: bool):
        """bf16 copy of a weight zero-padded to (rows, cols); optionally its transpose."""
        wp = torch.zeros(rows, cols, dtype=F32, device=w.device)
        wp[: w.shape[0], : w.shape[1]] = w.detach()
        wb = torch.empty(rows, cols, dtype=BF16, device=w.device)
        _lib.call("hma_cast_bf16", stream, ptr(wp), ptr(wb), wp.numel())
        wt = None
        if transpose:
            wt = torch.empty(cols, rows, dtype=BF16, device=w.device)
            _lib.call("hma_transpose_cast_bf16", stream, ptr(wp), ptr(wt), rows, cols, 1, 0, 0)
        return wb, wt

    @staticmethod
    def _nt(stream, **kw):
        g = make_gemm_nt(**kw)
        _lib.call("hma_gemm_nt", stream, C.byref(g))

    @staticmethod
    def _tn(stream, **kw):
        g = make_gemm_tn(**kw)
        _lib.call("hma_gemm_tn", stream, C.byref(g))

    def _engine(self, dev):
        core = self._core
        eng = core._get_engine(dev)
        ps = getattr(self, "_ver_params", None)
        if ps is None or getattr(self, "_ver_engine", None) is not eng:  # (the trunk's parameters: what the engine keeps derived copies of)
            self._ver_engine = eng
            am = getattr(self, "action_mlp", None)
            ps = self._ver_params = list(self.decoder.parameters()) + (list(am.parameters()) if am is not None else [])
        ver = STMaskGIT._version_sum(ps)
        if ver != getattr(self, "_seen_versions", None):  # load_state_dict / an optimizer wrote through the named parameters
            self._seen_versions = ver
            eng.weights_changed()
        return eng

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, input_ids, labels, action_ids=None, domain="default", **kwargs):
        assert "masked_tokens_indicator" in kwargs
        cfg = self.config
        if action_ids is None and cfg.jointly_predict_actions:
            raise NotImplementedError("jointly_predict_actions without input actions (mask tokens on every frame) is built for the discrete model only")
        masked = kwargs["masked_tokens_indicator"]
        T, H, W = cfg.T, self.h, self.w
        if "h" in kwargs:
            H, W = int(kwargs["h"][0]), int(kwargs["w"][0])
        dev = input_ids.device
        B, Cc, p = input_ids.shape[0], cfg.vae_embed_dim, cfg.patch_size
        h_, w_ = H // p, W // p
        S, A, d = h_ * w_, (cfg.action_token_size if action_ids is not None else 0), cfg.d_model  # (no actions: st_mar.py:154, no action tokens)
        if S != self.seq_len:
            raise ValueError(f"{H}x{W} latents give {S} patch tokens per frame, the model was built for {self.seq_len}")
        SA, Fr, Mi, M, pc = S + A, B * T, B * T * S, B * T * (S + A), Cc * p * p
        dom = None if action_ids is None else (domain if isinstance(domain, str) else domain[0])
        train = torch.is_grad_enabled() and self.training
        stream = torch.cuda.current_stream(dev).cuda_stream
        e = lambda *s, dt=F32: torch.empty(*s, dtype=dt, device=dev)
        lat = input_ids.detach().reshape(Fr, H, W, Cc).to(F32).contiguous()
        lab = labels.detach().reshape(Fr, H, W, Cc).to(F32).contiguous()
        mk = masked.reshape(Fr, H, W).to(torch.uint8).contiguous()
        patches, target, pmask = e(Mi, _PAD, dt=BF16), e(Mi, pc), e(Mi)
        _lib.call("hma_mar_patchify", stream, ptr(lat), ptr(mk), ptr(self.mask_token), ptr(patches), _PAD, None, ptr(pmask), Fr, H, W, Cc, p)
        _lib.call("hma_mar_patchify", stream, ptr(lab), None, None, None, 0, ptr(target), None, Fr, H, W, Cc, p)
        wtok, wtok_t = self._cast(stream, self.token_embed.weight, d, _PAD, train)
        wout, wout_t = self._cast(stream, self.out_x_proj.weight, d, d, train)
        eng = self._engine(dev)
        xtok, xhat_e, rstd_e = e(Mi, d), e(M, d, dt=BF16), e(M)
        pos_stride = self.pos_embed_TSC.shape[2] * d

        def build_x(ws):
            self._nt(stream, A=ptr(patches), lda=_PAD, a_kind=A_BF16, W=ptr(wtok), ldw=_PAD, M=Mi, N=d, K=_PAD, epi=EPI_F32, Cp=ptr(xtok), ldc=d)
            _lib.call("hma_mar_embed_fwd", stream, ptr(xtok), ws["a_emb"].data_ptr() if A > 0 else None, ptr(self.pos_embed_TSC), pos_stride,
                      ptr(self.z_proj_ln.weight), ptr(self.z_proj_ln.bias), 1e-6, ws["x"].data_ptr(), ptr(xhat_e), ptr(rstd_e), Fr, T, S, A)

        ws = eng.trunk_train_forward(B, T, S, None if action_ids is None else action_ids.to(dev, F32), dom, build_x, train=train)
        y, z, yhat, rstd_r = e(Mi, d), e(Mi, d), e(Mi, d, dt=BF16), e(Mi)
        self._nt(stream, A=ws["x"].data_ptr(), lda=d, a_kind=A_F32, a_group=(S, SA), W=ptr(wout), ldw=d, M=Mi, N=d, K=d, epi=EPI_F32,
                 Cp=ptr(y), ldc=d, bias=ptr(self.out_x_proj.bias))
        _lib.call("hma_mar_readout_fwd", stream, ptr(y), ptr(self.decoder_norm.weight), ptr(self.decoder_norm.bias), 1e-6,
                  ptr(self.diffusion_pos_embed_learned), ptr(z), ptr(yhat), ptr(rstd_r), Mi, T, S)
        act = None
        self._act_scale = None
        if cfg.jointly_predict_actions:
            # st_mar.py:231-240, 187-189, 266-273: the frames from a drawn timestep on are "masked" (the mask only weights the loss:
            # compute_latents does not apply it), every frame's action tokens are mean-pooled after the trunk and the domain's action
            # diffusion head is trained on them against the RAW action ids
            amask = kwargs.get("action_mask")
            if amask is None:
                start = torch.randint(0, T, (B, 1), device=dev)
                amask = (torch.arange(T, device=dev)[None, :] >= start).to(F32)
            pooled = ws["x"].view(Fr, SA, d)[:, S:].mean(dim=1)
            za = pooled.detach().requires_grad_(train)
            mulA = self.diffusion_batch_mul
            tgt_a = action_ids[:, :T].reshape(Fr, -1).to(dev, F32)
            inner_a = self.action_diff_losses[dom](tgt_a.repeat(mulA, 1), za.repeat(mulA, 1), amask.reshape(Fr).to(dev, F32).repeat(mulA),
                                                   t=kwargs.get("action_diffusion_t"), noise=kwargs.get("action_diffusion_noise"))
            act = dict(inner=inner_a, za=za, actions=pooled.view(B, T, d).clone())
        zl = z.detach().requires_grad_(train)
        mul = self.diffusion_batch_mul
        if mul > 1:  # st_mar.py:133-140: every token is scored at `mul` independent (t, noise) draws; the rows are repeated whole
            inner = self.diffloss(target.repeat(mul, 1), zl.repeat(mul, 1), pmask.repeat(mul), t=kwargs.get("diffusion_t"),
                                  noise=kwargs.get("diffusion_noise"))
        else:
            inner = self.diffloss(target, zl, pmask, t=kwargs.get("diffusion_t"), noise=kwargs.get("diffusion_noise"))
        logits = z.view(B, T, h_, w_, d).permute(0, 4, 1, 2, 3)
        acc = torch.zeros((), device=dev)
        if not train:
            if act is not None:
                return ModelOutput(loss=inner.detach(), acc=acc, logits=logits, action_loss=act["inner"].detach(), actions=act["actions"])
            return ModelOutput(loss=inner.detach(), acc=acc, logits=logits)
        self._saved = dict(act=act, inner=inner, zl=zl, yhat=yhat, rstd_r=rstd_r, xhat_e=xhat_e, rstd_e=rstd_e, patches=patches, mk=mk, wtok_t=wtok_t,
                           wout_t=wout_t, dims=(B, T, H, W, Cc, p, S, A, dom), pos_stride=pos_stride)
        if self._anchor is None or self._anchor.device != dev:
            self._anchor = torch.zeros((), device=dev, requires_grad=True)
        loss = _MarLoss.apply(self._anchor, self, inner.detach())
        if act is not None:
            return ModelOutput(loss=loss, acc=acc, logits=logits, action_loss=_MarActionLoss.apply(self._anchor, self, act["inner"].detach()),
                               actions=act["actions"])
        return ModelOutput(loss=loss, acc=acc, logits=logits)

    # ------------------------------------------------------------------------------------------ generation (MAR decode)
    @torch.no_grad()
    def compute_latents(self, x_patches: torch.Tensor, action_ids: torch.Tensor = None, domain=None, **kwargs):
        """st_mar.py:146-197 on already patchified latents (B, T, h, w, p*p*c) -> (decoded_states (B, d, T, h, w), None)."""
        cfg = self.config
        if action_ids is None and cfg.jointly_predict_actions:
            raise NotImplementedError("jointly_predict_actions without input actions (mask tokens on every frame) is built for the discrete model only")
        B, T, h_, w_, pc = x_patches.shape
        dev = x_patches.device
        S, A, d = h_ * w_, (cfg.action_token_size if action_ids is not None else 0), cfg.d_model
        Fr, Mi, M = B * T, B * T * S, B * T * (S + A)
        dom = None if action_ids is None else (domain if isinstance(domain, str) else domain[0])
        stream = torch.cuda.current_stream(dev).cuda_stream
        e = lambda *s, dt=F32: torch.empty(*s, dtype=dt, device=dev)
        patches = torch.zeros(Mi, _PAD, dtype=BF16, device=dev)
        patches[:, :pc] = x_patches.reshape(Mi, pc)          # layout + dtype only
        wtok, _ = self._cast(stream, self.token_embed.weight, d, _PAD, False)
        wout, _ = self._cast(stream, self.out_x_proj.weight, d, d, False)
        eng = self._engine(dev)
        xtok, xhat_e, rstd_e = e(Mi, d), e(M, d, dt=BF16), e(M)
        pos_stride = self.pos_embed_TSC.shape[2] * d

        def build_x(ws):
            self._nt(stream, A=ptr(patches), lda=_PAD, a_kind=A_BF16, W=ptr(wtok), ldw=_PAD, M=Mi, N=d, K=_PAD, epi=EPI_F32, Cp=ptr(xtok), ldc=d)
            _lib.call("hma_mar_embed_fwd", stream, ptr(xtok), ws["a_emb"].data_ptr() if A > 0 else None, ptr(self.pos_embed_TSC), pos_stride,
                      ptr(self.z_proj_ln.weight), ptr(self.z_proj_ln.bias), 1e-6, ws["x"].data_ptr(), ptr(xhat_e), ptr(rstd_e), Fr, T, S, A)

        ws = eng.trunk_train_forward(B, T, S, None if action_ids is None else action_ids.to(dev, F32), dom, build_x, train=False)
        y, z, yhat, rstd_r = e(Mi, d), e(Mi, d), e(Mi, d, dt=BF16), e(Mi)
        self._nt(stream, A=ws["x"].data_ptr(), lda=d, a_kind=A_F32, a_group=(S, S + A), W=ptr(wout), ldw=d, M=Mi, N=d, K=d, epi=EPI_F32,
                 Cp=ptr(y), ldc=d, bias=ptr(self.out_x_proj.bias))
        _lib.call("hma_mar_readout_fwd", stream, ptr(y), ptr(self.decoder_norm.weight), ptr(self.decoder_norm.bias), 1e-6,
                  ptr(self.diffusion_pos_embed_learned), ptr(z), ptr(yhat), ptr(rstd_r), Mi, T, S)
        pooled = None
        if cfg.jointly_predict_actions:  # :187-189: the mean of every frame's action tokens, what the action diffusion head conditions on
            pooled = ws["x"].view(Fr, S + A, d)[:, S:].mean(dim=1).view(B, T, d)
        return z.view(B, T, h_, w_, d).permute(0, 4, 1, 2, 3), pooled

    def patchify(self, x):
        """(B, T, H, W, C) -> (B, T, H/p, W/p, p*p*C) through the patchify kernel (st_mar.py:199-207)."""
        B, T, H, W, Cc = x.shape
        p = self.patch_size
        out = torch.empty(B * T * (H // p) * (W // p), Cc * p * p, dtype=F32, device=x.device)
        _lib.call("hma_mar_patchify", torch.cuda.current_stream(x.device).cuda_stream, ptr(x.to(F32).contiguous()), None, None, None, 0, ptr(out),
                  None, B * T, H, W, Cc, p)
        return out.view(B, T, H // p, W // p, Cc * p * p)

    def unpatchify(self, x):
        """(B, T, h, w, p*p*C) -> (B, T, h*p, w*p, C): a pure re-layout (st_mar.py:209-217)."""
        p, c = self.patch_size, self.vae_embed_dim
        B, T, h_, w_, _ = x.shape
        return x.reshape(B, T, h_, w_, p, p, c).permute(0, 1, 2, 4, 3, 5, 6).reshape(B, T, h_ * p, w_ * p, c)

    def sample_orders(self, bsz):
        """A random generation order per sample (st_mar.py:351-360; numpy's global RNG like the reference)."""
        import numpy as np
        return torch.stack([torch.from_numpy(np.random.permutation(self.seq_len)) for _ in range(bsz)]).long()

    @torch.no_grad()
    def maskgit_generate(self, prompt_THW, out_t: int, unmask_mode: str = "random", action_ids=None, domain="default", maskgit_steps=8,
                         cfg=1.0, temperature=1.0, cfg_schedule="linear", action_only: bool = False, state_only: bool = False,
                         orders: Optional[torch.Tensor] = None, draws=None, **kwargs):
        """MAR decode of frame `out_t` (st_mar.py:362-452; cfg must be 1).  Returns (frame (B, H, W, C), the first pass's latents
        (B, d, h, w), None).  `orders` / `draws` replay the random order and the Gaussian draws of each DiffLoss.sample call
        ([(noise0, step_noises), ...]).  Like the reference, `unmasked` is never updated: every step re-predicts all tokens outside
        the first `mask_len` entries of the order, the last step all of them."""
        import numpy as np
        assert out_t, "maskgit_generate requires out_t > 0"
        if cfg != 1.0:
            raise NotImplementedError("cfg != 1 in the MAR decode: the reference branch (st_mar.py:417-418) cannot run; DiffLoss.sample(cfg=) is built")
        dev = prompt_THW.device
        x = self.patchify(prompt_THW).clone()
        B, T, h_, w_, pc = x.shape
        S = h_ * w_
        orders = (self.sample_orders(B) if orders is None else orders).to(dev)
        z, pooled = self.compute_latents(x, action_ids=action_ids, domain=domain)
        z = z[:, :, out_t]
        orig = z.clone()
        sampled_actions = None
        dom = domain if isinstance(domain, str) else domain[0]
        action_draws = kwargs.get("action_draws")
        for step in range(maskgit_steps):
            if step > 0:
                z, pooled = self.compute_latents(x, action_ids=action_ids, domain=domain)
                z = z[:, :, out_t]
            mask_len = max(1, min(S - 1, int(np.floor(self.seq_len * np.cos(math.pi / 2.0 * (step + 1) / maskgit_steps)))))
            mask_next = torch.zeros(B, S, device=dev).scatter(-1, orders[:, :mask_len], torch.ones(B, S, device=dev)).bool()
            to_pred = torch.ones(B, S, dtype=torch.bool, device=dev) if step >= maskgit_steps - 1 else ~mask_next
            rows = z.reshape(B, -1, S).permute(0, 2, 1)[to_pred].contiguous()          # (n, d): gather of the rows to sample
            kw = {} if draws is None else dict(noise0=draws[step][0], step_noises=draws[step][1])
            smp = self.diffloss.sample(rows, temperature, 1.0, clip_denoised=True, **kw)
            if pooled is not None:  # jointly_predict_actions (:441-446): actions sampled by the domain's action diffusion head, every step
                akw = {} if action_draws is None else dict(noise0=action_draws[step][0], step_noises=action_draws[step][1])
                sampled_actions = self.action_diff_losses[dom].sample(pooled.reshape(-1, pooled.shape[-1]), temperature, 1.0, clip_denoised=True, **akw)
            xt = x[:, out_t].reshape(B, S, pc).clone()  # (at B = 1 the frame is contiguous and reshape is a VIEW: writing it back onto itself raises)
            xt[to_pred] = smp
            x[:, out_t] = xt.reshape(B, h_, w_, pc)
        return self.unpatchify(x)[:, out_t], orig, sampled_actions

    @torch.no_grad()
    def generate(self, input_ids, attention_mask=None, max_new_tokens: int = 0, min_new_tokens: int = None, return_logits: bool = False,
                 return_with_actions: bool = False, temperature: float = 1.0, action_ids=None, domain="default", action_only: bool = False,
                 state_only: bool = False, **kwargs):
        """Frame-by-frame rollout (st_mar.py:277-349): input_ids (B, T0*H*W, C) latents -> (B, (T0 + new)*H*W, C)."""
        assert min_new_tokens in (None, max_new_tokens), "Expecting `min_new_tokens`, if specified, to match `max_new_tokens`."
        if return_with_actions:
            raise NotImplementedError("action prediction (jointly_predict_actions) is not built")
        h, w, c = self.h, self.w, self.vae_embed_dim
        if "h" in kwargs:
            h, w = int(kwargs["h"][0]), int(kwargs["w"][0])
        S = h * w
        new = max_new_tokens // S
        B = input_ids.shape[0]
        x = input_ids.reshape(B, -1, h, w, c).clone()
        x = torch.cat([x, self.mask_token.detach()[None, None].to(x).expand(B, new, h, w, c)], dim=1).contiguous()
        firsts = []
        for t in range(x.shape[1] - new, x.shape[1]):
            frame, first, _ = self.maskgit_generate(x, t, maskgit_steps=self.config.maskgit_steps, temperature=temperature,
                                                    action_ids=action_ids, domain=domain)
            x[:, t] = frame
            firsts.append(first)
        tokens = x.reshape(B, -1, c)
        return (tokens, torch.stack(firsts, dim=2)) if return_logits else tokens

    # ------------------------------------------------------------------------------------------ backward
    def _accum(self, p: nn.Parameter, g: torch.Tensor):
        g = g.reshape(p.shape)
        if p.grad is None:
            p.grad = g
        else:
            p.grad.add_(g)  # in place: with the own flat range (`_own_flat`) the gradient is a view into ONE buffer

    def _backward(self, grad_out: torch.Tensor):
        sv = self._saved
        B, T, H, W, Cc, p, S, A, dom = sv["dims"]
        d, SA, Fr, Mi, pc = self.config.d_model, S + A, B * T, B * T * S, Cc * p * p
        dev = sv["zl"].device
        stream = torch.cuda.current_stream(dev).cuda_stream
        eng = self._core._engine
        z0 = lambda *s: torch.zeros(*s, dtype=F32, device=dev)
        sv["inner"].backward(grad_out)                       # DiffLoss: parameter grads + d loss / d z
        dza = None
        if sv.get("act") is not None and self._act_scale is not None:
            # the action diffusion head (its own parameter gradients) and d action_loss / d (pooled action tokens)
            sv["act"]["inner"].backward(self._act_scale)
            dza = sv["act"]["za"].grad
        hooks = self.__dict__.get("_bwd_hooks") or {}         # (a data-parallel driver: MarTrainer, last micro-batch of a step)
        if hooks.get("after_head"):
            hooks["after_head"]()                             # the head's gradients are final: their all-reduce can start now
        dz = sv["zl"].grad.contiguous()
        dy, dpos2, dg_n, db_n = z0(Mi, d), z0(self.diffusion_pos_embed_learned.shape), z0(d), z0(d)
        _lib.call("hma_mar_readout_bwd", stream, ptr(dz), ptr(sv["yhat"]), ptr(sv["rstd_r"]), ptr(self.decoder_norm.weight), ptr(dy), ptr(dpos2),
                  ptr(dg_n), ptr(db_n), Mi, T, S)
        dWo, dBo = z0(d, d), z0(d)
        if not self._grads_live:
            eng.zero_grad()

        def fill_dx(ws):
            self._tn(stream, dY=ptr(dy), ldy=d, y_kind=A_F32, A=ws["x"].data_ptr(), lda=d, a_kind=A_F32, a_group=(S, SA), M=Mi, N=d, K=d,
                     dW=ptr(dWo), lddw=d, dBias=ptr(dBo))
            self._nt(stream, A=ptr(dy), lda=d, a_kind=A_F32, W=ptr(sv["wout_t"]), ldw=d, M=Mi, N=d, K=d, epi=EPI_F32,
                     Cp=ws["dx"].data_ptr(), ldc=d, c_group=(S, SA))
            if dza is not None:  # the mean over a frame's A action rows
                ws["dx"].view(Fr, SA, d)[:, S:].add_((dza / A)[:, None, :])

        dxtok, dpos, dg_z, db_z = z0(Mi, d), z0(self.pos_embed_TSC.shape), z0(d), z0(d)
        dWt, dpatch, dmask_tok = z0(d, _PAD), z0(Mi, _PAD), z0(8)

        def embed_bwd(ws):
            _lib.call("hma_mar_embed_bwd", stream, ws["dx"].data_ptr(), ptr(sv["xhat_e"]), ptr(sv["rstd_e"]), ptr(self.z_proj_ln.weight), ptr(dxtok),
                      ws["da_emb"].data_ptr() if A > 0 else None, ptr(dpos), sv["pos_stride"], ptr(dg_z), ptr(db_z), Fr, T, S, A)
            self._tn(stream, dY=ptr(dxtok), ldy=d, y_kind=A_F32, A=ptr(sv["patches"]), lda=_PAD, a_kind=A_BF16, M=Mi, N=d, K=_PAD, dW=ptr(dWt),
                     lddw=_PAD)
            self._nt(stream, A=ptr(dxtok), lda=d, a_kind=A_F32, W=ptr(sv["wtok_t"]), ldw=d, M=Mi, N=_PAD, K=d, epi=EPI_F32, Cp=ptr(dpatch), ldc=_PAD)
            _lib.call("hma_mar_mask_token_bwd", stream, ptr(dpatch), _PAD, ptr(sv["mk"]), ptr(dmask_tok), Fr, H, W, Cc, p)

        eng.trunk_train_backward(fill_dx, embed_bwd, on_segment=hooks.get("on_segment"), segment_layers=hooks.get("segment_layers", 0))
        self._grads_live = True
        with torch.no_grad():
            self._accum(self.out_x_proj.weight, dWo)
            self._accum(self.out_x_proj.bias, dBo)
            self._accum(self.decoder_norm.weight, dg_n)
            self._accum(self.decoder_norm.bias, db_n)
            self._accum(self.diffusion_pos_embed_learned, dpos2)
            self._accum(self.z_proj_ln.weight, dg_z)
            self._accum(self.z_proj_ln.bias, db_z)
            self._accum(self.pos_embed_TSC, dpos)
            self._accum(self.token_embed.weight, dWt[:, :pc].contiguous())
            self._accum(self.mask_token, dmask_tok[:Cc].clone())
            core = self._core
            active = set(eng.layout.regions) - {"frozen", "head", "tail"} - {f"dom:{x}" for x in eng.domains if x != dom}
            views = self.__dict__.get("_core_grad_views")
            if views is None or views[0] is not eng.G:  # (name -> view of the flat gradient buffer: built once per buffer)
                views = self.__dict__["_core_grad_views"] = (eng.G, {})
            for name, prm in self._core_named():  # trunk / action parameters: views into the engine's gradient buffer
                if eng.layout.entries[name].region in active:
                    v = views[1].get(name)
                    if v is None:
                        v = views[1][name] = eng.view(name, eng.G)
                    prm.grad = v
        self._saved = None

    def _save_pretrained(self, save_directory) -> None:
        """model.safetensors with the reference's tensor names (the parameters are views into flat buffers: cloned out first)."""
        import os

        from safetensors.torch import save_file

        state = {k: v.detach().to("cpu").contiguous().clone() for k, v in self.state_dict().items()}
        save_file(state, os.path.join(str(save_directory), "model.safetensors"))

    # ------------------------------------------------------------------------------------------ optimizer
    _OWN_ALIGN = 64

    # A walk over the module tree costs ~25 ms of host time at 30 domains (7 763 parameters); a train step made five of them and sat
    # at the edge of being host-bound (67 ms of GPU work; 100 ms per step on a slower host, round 6).  The lists are kept until the
    # parameter set changes (`init_action_projectors`, `_apply`).
    def _drop_param_lists(self) -> None:
        for k in ("_own_named_list", "_all_params_list", "_core_named_list", "_core_grad_views"):
            self.__dict__.pop(k, None)

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._drop_param_lists()
        return out

    def _all_params(self):
        ps = self.__dict__.get("_all_params_list")
        if ps is None:
            ps = self.__dict__["_all_params_list"] = list(self.parameters())
        return ps

    def _core_named(self):
        ps = self.__dict__.get("_core_named_list")
        if ps is None:
            ps = self.__dict__["_core_named_list"] = list(self._core.named_parameters())
        return ps

    def _own_named(self):
        """Parameters this class owns outside the engine's flat layout (input / output stages, the DiffLoss head), in a fixed
        order.  `action_diff_losses.*` and `action_mask_tokens` never receive a gradient (jointly_predict_actions is off): like
        the reference's grad-None parameters they are never stepped."""
        named = self.__dict__.get("_own_named_list")
        if named is None:
            named = self.__dict__["_own_named_list"] = [
                (n, p) for n, p in self.named_parameters()
                if not n.startswith(("decoder.", "action_mlp.", "action_diff_losses.")) and n != "action_mask_tokens"]
        return named

    def _make_flat(self, named, dev):
        """One fp32 buffer holding `named` parameters (64-element aligned) as views, with mirrored gradient and Adam-moment buffers and
        decay flags per 64 elements (train_multi.py:907-918: only names containing "bias" are exempt)."""
        A = self._OWN_ALIGN
        offs, total = [], 0
        for _, p in named:
            offs.append(total)
            total += (p.numel() + A - 1) // A * A
        P = torch.zeros(total, dtype=F32, device=dev)
        flags = torch.zeros(total // A, dtype=torch.uint8)
        pviews, gviews = [], []
        G = torch.zeros_like(P)
        with torch.no_grad():
            for (n, p), o in zip(named, offs):
                v = P[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                pviews.append(v)
                gviews.append(G[o:o + p.numel()].view(p.shape))
                flags[o // A:(o + p.numel() + A - 1) // A] = 1 if "bias" in n else 2
        return dict(P=P, G=G, M=torch.zeros_like(P), V=torch.zeros_like(P), flags=flags.to(dev), pviews=pviews, gviews=gviews,
                    names=[n for n, _ in named], steps=torch.zeros(2, dtype=torch.int32, device=dev), calls=0)

    def _own_flat(self, dev=None):
        """A SECOND flat range next to the engine's: the own parameters become views into one fp32 buffer with mirrored gradient
        and Adam-moment buffers, so they are all-reduced with ONE collective and stepped by ONE fused clip + AdamW launch
        (hma_adamw_counted) instead of tensor by tensor."""
        named = self._own_named()
        dev = torch.device(dev) if dev is not None else named[0][1].device
        own = self.__dict__.get("_own")
        if own is not None and own["P"].device == dev and all(p.data_ptr() == v.data_ptr() for (_, p), v in zip(named, own["pviews"])):
            return own
        own = self._make_flat(named, dev)
        self.__dict__["_own"] = own
        return own

    def _act_named(self, dom: str):
        return [(f"action_diff_losses.{dom}.{n}", p) for n, p in self.action_diff_losses[dom].named_parameters()]

    def _act_flat(self, dom: str, dev=None):
        """jointly_predict_actions: one more flat range PER DOMAIN for its action diffusion head -- stepped (and its update counted)
        only when the domain is active, like the engine's per-domain blocks and like the reference's grad-None parameters."""
        named = self._act_named(dom)
        dev = torch.device(dev) if dev is not None else named[0][1].device
        cache = self.__dict__.setdefault("_actf", {})
        af = cache.get(dom)
        if af is not None and af["P"].device == dev and all(p.data_ptr() == v.data_ptr() for (_, p), v in zip(named, af["pviews"])):
            return af
        af = cache[dom] = self._make_flat(named, dev)
        return af

    @staticmethod
    def _gather(named, flat) -> None:
        for (n, p), gv in zip(named, flat["gviews"]):
            if p.grad is None:
                gv.zero_()
            elif p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)
            p.grad = gv

    def _own_head_range(self, own):
        """(start, stop) elements of the diffusion head's parameters inside the own flat range (they are consecutive in it)."""
        idx = [i for i, n in enumerate(own["names"]) if n.startswith("diffloss.")]
        assert idx == list(range(idx[0], idx[-1] + 1)), "the diffusion head's parameters are not consecutive"
        a = (own["gviews"][idx[0]].data_ptr() - own["G"].data_ptr()) // 4
        b = (own["gviews"][idx[-1]].data_ptr() - own["G"].data_ptr()) // 4 + own["gviews"][idx[-1]].numel()
        return a, b

    def _own_gather_grads(self, own, prefix: Optional[str] = None, skip: Optional[str] = None) -> None:
        """Make the flat gradient buffer hold every own gradient (a caller may have produced fresh .grad tensors).  `prefix` /
        `skip`: only the parameters whose names start with it / all but those."""
        for (n, p), gv in zip(self._own_named(), own["gviews"]):
            if (prefix is not None and not n.startswith(prefix)) or (skip is not None and n.startswith(skip)):
                continue
            if p.grad is None:
                gv.zero_()
            elif p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)
            p.grad = gv

    def optimizer_step(self, lr: float, domain, betas=(0.9, 0.95), eps: float = 1e-8, weight_decay: float = 0.05,
                       max_norm: Optional[float] = 1.0) -> None:
        """Global-norm clip + AdamW over everything that received a gradient (train_multi.py:593-598): the trunk and the active
        domains' blocks through the engine's flat ranges, the parameters this class owns through their own flat range, with one
        clip coefficient over both.  `domain`: a name or the list of domains active on some rank this step.  Weight decay
        grouping as the reference (train_multi.py:907-918): only names containing "bias" are exempt."""
        eng = self._core._engine
        own = self._own_flat(eng.device)
        self._own_gather_grads(own)
        domains = [domain] if isinstance(domain, str) else list(domain)
        flats = [own]
        if self.config.jointly_predict_actions:  # the active domains' action heads
            for dom in domains:
                af = self._act_flat(dom, eng.device)
                self._gather(self._act_named(dom), af)
                flats.append(af)
        eng.optimizer_step(lr, domains, betas, eps, weight_decay, max_norm, extra_grads=[f["G"] for f in flats])
        stream = torch.cuda.current_stream().cuda_stream
        for f in flats:
            _lib.call("hma_adamw_counted", stream, ptr(f["P"]), ptr(f["G"]), ptr(f["M"]), ptr(f["V"]), None, f["P"].numel(), lr,
                      betas[0], betas[1], eps, weight_decay, ptr(f["steps"]), f["calls"] & 1, eng.sqnorm.data_ptr(),
                      float(max_norm or 0.0), ptr(f["flags"]))
            f["calls"] += 1

    def zero_grad(self, set_to_none: bool = True, active_domains=None):
        """`active_domains`: zero only the engine's dense range and these domains' blocks (a step driver that knows which domains
        the step touches; the other blocks are never read -- their parameters' .grad is None)."""
        own = self.__dict__.get("_own")
        for prm in self._all_params():
            prm.grad = None
        if own is not None:  # own gradients accumulate in place in the flat buffer
            own["G"].zero_()
            for (_, p), gv in zip(self._own_named(), own["gviews"]):
                p.grad = gv
        for dom, af in self.__dict__.get("_actf", {}).items():
            af["G"].zero_()
            for (_, p), gv in zip(self._act_named(dom), af["gviews"]):
                p.grad = gv
        if self._core._engine is not None:
            self._core._engine.zero_grad(active_domains)
        self._grads_live = False
