"""STMaskGIT with the reference's module API, state-dict and checkpoint layout
(hma/model/st_mask_git.py:150-789), executing on the MI355X engine (hma_amd/engine.py).

Kept from the reference: constructor from a GenieConfig; `forward(input_ids,
labels, action_ids, domain, **kw) -> ModelOutput(loss, acc, logits)`; `compute_logits`;
`maskgit_generate` (mutates `prompt_THW`, returns the FIRST pass's logits); `generate`;
`init_action_projectors`; `set_mup_shapes`; `from_pretrained` / `save_pretrained` (config.json +
model.safetensors through huggingface_hub's mixin); attributes `.config .mask_token_id .decoder
.action_preprocessor .h .w`.  `domain` is a list of str, element 0 is used (:648, :669).

Deliberate differences (all documented in DESIGN.md): no `.cuda()` hard-coding (:710); `generate`
defaults h/w to the model's (the reference raises UnboundLocalError without `w=`, :283-289).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch
import torch.nn as nn
from huggingface_hub import PyTorchModelHubMixin
from transformers.utils import ModelOutput

from .. import _lib
from ..config import GenieConfig
from ..engine import STEngine
from ..ops import ptr, stream_ptr
from .factorization_utils import FactorizedEmbedding
from .st_transformer import STTransformerDecoder


def cosine_schedule(u):
    """u in [0, 1] (st_mask_git.py:116-125)."""
    if isinstance(u, torch.Tensor):
        return torch.cos(u * torch.pi / 2)
    if isinstance(u, float):
        return math.cos(u * math.pi / 2)
    raise NotImplementedError(f"Unexpected {type(u)=} {u=}")


class ModulateLayer(nn.Module):
    """Parameter holder for the per-layer, per-domain action modulation (st_mask_git.py:51-87)."""

    def __init__(self, model_channels: int, out_channels: int):
        super().__init__()
        self.norm_final = nn.LayerNorm(out_channels, elementwise_affine=False, eps=1e-6)
        self.linear_out = nn.Linear(out_channels, out_channels, bias=True)
        self.adaLN_modulation = nn.Sequential(nn.Linear(model_channels, model_channels), nn.SiLU(),
                                              nn.Linear(model_channels, 2 * out_channels, bias=True))
        for m in self.modules():
            if isinstance(m, nn.Linear):
                torch.nn.init.xavier_uniform_(m.weight, gain=0.1)
                nn.init.constant_(m.bias, 0)

    @torch.no_grad()
    def forward(self, x: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
        """linear_out(LN(x) * (1 + scale) + shift), (shift, scale) = adaLN(c) (st_mask_git.py:66-76), inference only; in the
        model the same arithmetic runs fused inside the trunk plans.  x ((b s), t, d), c (b, T, d)."""
        from .. import ops
        from .._lib import EPI_F32, EPI_SILU2
        shp = x.shape
        b, t, d = c.shape[0], shp[1], shp[2]
        s_ = shp[0] // b
        bf = lambda w: w.detach().to(torch.bfloat16).contiguous()
        cc = c[:, :t].reshape(b * t, d).float().contiguous()
        pre = torch.empty(b * t, d, dtype=torch.bfloat16, device=x.device)
        act = torch.empty_like(pre)
        ops.linear(cc, bf(self.adaLN_modulation[0].weight), self.adaLN_modulation[0].bias, epi=EPI_SILU2, out=pre, out2=act)
        ss = ops.linear(act, bf(self.adaLN_modulation[2].weight), self.adaLN_modulation[2].bias, epi=EPI_F32)   # (b t, [shift | scale])
        # rows in (b, t, s) order: one (shift, scale) per run of s rows
        xr = x.reshape(b, s_, t, d).permute(0, 2, 1, 3).reshape(b * t * s_, d).float().contiguous()
        xhat = torch.empty(xr.shape, dtype=torch.bfloat16, device=x.device)
        xm = torch.empty_like(xhat)
        rstd = torch.empty(xr.shape[0], dtype=torch.float32, device=x.device)
        _lib.call("hma_modln_fwd", stream_ptr(), ptr(xr), ptr(ss), ptr(xhat), ptr(xm), ptr(rstd), b * t, s_, 1e-6)
        y = ops.linear(xm, bf(self.linear_out.weight), self.linear_out.bias, epi=EPI_F32)
        return y.reshape(b, t, s_, d).permute(0, 2, 1, 3).reshape(shp).to(x.dtype)


class BasicMLP(nn.Module):
    """Action stem Linear-LN-ReLU-Linear (st_mask_git.py:90-113); runs inside hma_action_stem_fwd."""

    def __init__(self, d_action: int, d_model: int):
        super().__init__()
        self.model = nn.Sequential(nn.Linear(d_action, d_model, bias=True), nn.LayerNorm(d_model), nn.ReLU(),
                                   nn.Linear(d_model, d_model, bias=True))
        for m in self.modules():
            if isinstance(m, nn.Linear):
                torch.nn.init.xavier_uniform_(m.weight, gain=0.01)
                nn.init.constant_(m.bias, 0)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Linear -> LayerNorm -> ReLU -> Linear (st_mask_git.py:101-102) on already normalised actions, through the action
        stem kernel (fp32; identity statistics, skip_norm = 1).  Inference only."""
        d_a, d = self.model[0].weight.shape[1], self.model[0].weight.shape[0]
        rows = x.numel() // d_a
        dev = x.device
        a = x.reshape(rows, d_a).float().contiguous()
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        zero, one = torch.zeros(d_a, device=dev), torch.ones(d_a, device=dev)
        an, xhat, rstd, h, out = e(rows, d_a), e(rows, d), e(rows), e(rows, d), e(rows, d)
        _lib.call("hma_action_stem_fwd", stream_ptr(), ptr(a), ptr(zero), ptr(one), d_a, ptr(self.model[0].weight), ptr(self.model[0].bias),
                  ptr(self.model[1].weight), ptr(self.model[1].bias), ptr(self.model[3].weight), ptr(self.model[3].bias), ptr(an),
                  ptr(xhat), ptr(rstd), ptr(h), ptr(out), rows, d_a, 1)
        return out.reshape(*x.shape[:-1], d).to(x.dtype)


class ActionStat(nn.Module):
    """Per-domain action mean/std buffers (st_mask_git.py:128-147)."""

    def __init__(self, input_info):
        super().__init__()
        self.register_buffer("mean", torch.tensor(input_info[0], dtype=torch.float32))
        self.register_buffer("std", torch.tensor(input_info[1], dtype=torch.float32))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        d = self.mean.numel()
        return ((x.reshape(*x.shape[:-1], -1, d) - self.mean) / (self.std + 1e-10)).reshape(x.shape)

    def unnormalize(self, actions: torch.Tensor) -> torch.Tensor:
        d = self.mean.numel()
        return (actions.reshape(*actions.shape[:-1], -1, d) * (self.std + 1e-10) + self.mean).reshape(actions.shape)

    def extra_repr(self):
        return f"mean={self.mean}, std={self.std}"


class FixedMuReadout(nn.Linear):
    """Readout Linear.  mup's MuReadout scales its input by output_mult / width_mult with
    width_mult = d_model / 256 (st_mask_git.py:772-789); at d_model = 256 -- the only width the
    kernels are built for -- that factor is exactly 1."""

    output_mult = 1.0

    def __init__(self, d_input: int, d_output: int):
        super().__init__(d_input, d_output)
        torch.nn.init.xavier_uniform_(self.weight, gain=0.01)
        nn.init.constant_(self.bias, 0)

    def width_mult(self) -> float:
        return self.in_features / 256


class _EngineLoss(torch.autograd.Function):
    """Bridges `loss.backward()` to the engine's hand-written backward."""

    @staticmethod
    def forward(ctx, anchor: torch.Tensor, owner, loss_value: torch.Tensor) -> torch.Tensor:
        ctx.owner = owner
        return loss_value.clone()

    @staticmethod
    def backward(ctx, grad_out):
        ctx.owner.__dict__["_bwd_ran"] = True
        ctx.owner._engine_backward(grad_out)
        return None, None, None


class _ActionLoss(torch.autograd.Function):
    """The action loss of jointly_predict_actions as a second autograd leaf: its backward only records d total / d action_loss.  It
    is created AFTER the video loss' node, so autograd runs it first and `_EngineLoss.backward` -- the one engine backward of the
    step -- finds the scale (0 when the caller left the action loss out of the objective).  An objective WITHOUT the video loss
    (`out.action_loss.backward()` alone) never reaches that node: a callback at the end of the autograd pass then runs the engine
    backward with a zero video-loss scale, so the action loss' gradients exist either way."""

    @staticmethod
    def forward(ctx, anchor: torch.Tensor, owner, value: torch.Tensor) -> torch.Tensor:
        ctx.owner = owner
        return value.clone()

    @staticmethod
    def backward(ctx, grad_out):
        owner = ctx.owner
        owner._engine.act_scale = float(grad_out)
        owner.__dict__["_bwd_ran"] = False

        def after_pass():
            if not owner.__dict__.get("_bwd_ran", False):
                owner.__dict__["_bwd_ran"] = True
                owner._engine_backward(torch.zeros((), device=grad_out.device))

        torch.autograd.Variable._execution_engine.queue_callback(after_pass)
        return None, None, None


class STMaskGIT(nn.Module, PyTorchModelHubMixin):
    def __init__(self, config: GenieConfig):
        super().__init__()
        if isinstance(config, dict):  # huggingface_hub hands back the parsed config.json
            config = GenieConfig.from_dict(config)
        self.h = self.w = math.isqrt(config.S)
        assert self.h ** 2 == config.S, "Expected S to be square"
        self.decoder = STTransformerDecoder(
            num_layers=config.num_layers, num_heads=config.num_heads, d_model=config.d_model, qkv_bias=config.qkv_bias,
            proj_bias=config.proj_bias, qk_norm=config.qk_norm, use_mup=config.use_mup, attn_drop=config.attn_drop,
            mlp_ratio=config.mlp_ratio, mlp_bias=config.mlp_bias, mlp_drop=config.mlp_drop,
            action_processing=config.action_network, random_dummy_action=config.random_dummy_action,
            jointly_predict_actions=config.jointly_predict_actions, mask_token_id=config.image_vocab_size)
        self.pos_embed_TSC = nn.Parameter(torch.zeros(1, config.T, config.S + config.action_token_size, config.d_model))
        self.mask_token_id = config.image_vocab_size
        self.seq_len = config.S
        self.relevant_action_mask = None
        self.token_embed = FactorizedEmbedding(factored_vocab_size=config.factored_vocab_size,
                                               num_factored_vocabs=config.num_factored_vocabs, d_model=config.d_model,
                                               mask_token_id=self.mask_token_id)
        cls = FixedMuReadout if config.use_mup else nn.Linear
        self.out_x_proj = cls(config.d_model, config.factored_vocab_size * config.num_factored_vocabs)
        self.config = config
        self.action_mask_tokens = nn.Parameter(torch.zeros(1, config.T, 1, config.d_model))
        self._engine: Optional[STEngine] = None
        self._anchor: Optional[torch.Tensor] = None
        self._grads_live = False
        self.touched_domains: set = set()
        self.decoder._bind(self)
        if (config.init_actions or config.use_actions) and config.action_domains is not None:
            self.init_action_projectors(config.action_domains, config.d_actions, config.action_stats, config.action_network)

    # ------------------------------------------------------------------ construction
    def init_action_projectors(self, domains: List[str], d_actions: List[int], action_stats, action_network: str = "mlp",
                               use_diffusion: bool = False):
        """Per-domain action stems and per-layer projectors (st_mask_git.py:201-251)."""
        self.config.init_actions = True
        self.config.action_domains = list(domains)
        self.config.d_actions = list(d_actions)
        self.config.action_stats = action_stats
        assert len(domains) == len(d_actions) == len(action_stats), f"{len(domains)=} {len(d_actions)=} {len(action_stats)=}"
        self.action_preprocessor = nn.ModuleDict()
        self.action_mlp = nn.ModuleDict()
        self.action_out_projectors = nn.ModuleDict()
        cls = FixedMuReadout if self.config.use_mup else nn.Linear
        for domain, d_action, stat in zip(domains, d_actions, action_stats):
            self.action_preprocessor[domain] = ActionStat(stat)
            self.action_mlp[domain] = BasicMLP(d_action, self.config.d_model)
            if not use_diffusion:
                self.action_out_projectors[domain] = cls(self.config.d_model, d_action)
        for layer in self.decoder.layers:
            layer.action_projectors = nn.ModuleDict()
            for domain in domains:
                if "modulate" in action_network:
                    layer.action_projectors[domain] = ModulateLayer(self.config.d_model, self.config.d_model)
                elif "mlp" in action_network:
                    layer.action_projectors[domain] = nn.Identity()
                else:
                    raise NotImplementedError(f"action_network={action_network!r}: only 'modulate' heads are built")
        self._engine = None  # parameter set changed: re-flatten on next use

    def init_weights(self):
        std = 0.02
        for module in self.modules():
            if isinstance(module, nn.Linear):
                module.weight.data.normal_(mean=0.0, std=std)
                if module.bias is not None:
                    module.bias.data.zero_()
            elif isinstance(module, nn.Embedding):
                module.weight.data.normal_(mean=0.0, std=std)

    def set_mup_shapes(self, rescale_params: bool = False):
        """muP base shape is d_model = 256 (st_mask_git.py:755-760): identity for the supported width."""
        if self.config.d_model != 256:
            raise NotImplementedError("muP rescaling for d_model != 256 is unpinned (no reference test) and not built")

    @classmethod
    def from_pretrained(cls, *args, **kwargs):
        model = super().from_pretrained(*args, **kwargs)
        if model.config.use_mup:
            model.set_mup_shapes(rescale_params=False)
        return model

    # ------------------------------------------------------------------ engine plumbing
    def _get_engine(self, device) -> STEngine:
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.HmaKernelError("hma_amd runs on an MI355X only (no CPU path): move the model and inputs to 'cuda'")
        sentinel = self.out_x_proj.weight
        eng = self._engine
        if eng is not None and eng.device == device and sentinel.data_ptr() == eng.view("out_x_proj.weight").data_ptr():
            return eng
        cfg = self.config
        domains = list(cfg.action_domains or []) if hasattr(self, "action_mlp") else []
        d_actions = list(cfg.d_actions or []) if domains else []
        adims = [len(s[0]) for s in (cfg.action_stats or [])] if domains else []
        eng = STEngine(cfg, domains, d_actions, adims, device)
        names = {n for n, _ in self.named_parameters()}
        if names != set(eng.layout.entries):
            raise RuntimeError(f"parameter set does not match the flat layout: {sorted(names ^ set(eng.layout.entries))[:8]}")
        with torch.no_grad():
            for name, p in self.named_parameters():
                v = eng.view(name)
                v.copy_(p.data)
                p.data = v
            for dom in domains:
                st = self.action_preprocessor[dom]
                st.mean.data = st.mean.data.to(device).contiguous()
                st.std.data = st.std.data.to(device).contiguous()
                eng.buffers[dom] = (st.mean, st.std)
        eng.weights_changed()
        self._engine = eng
        self._anchor = torch.zeros((), device=device, requires_grad=True)
        self._grads_live = False
        return eng

    @staticmethod
    def _version_sum(params) -> int:
        """Sum of the parameters' version counters.  (The named parameters are re-pointed at views of the engine's flat buffer through
        `.data`, which keeps every parameter's OWN counter: one representative per storage is not enough -- an optimizer's `p.add_()`
        bumps only that parameter's -- so all of them are read: ~0.5 ms of host time per module-level forward for the 362 M model's
        8 000 tensors; `Trainer` / `MarTrainer` steps update the weights themselves and tell the engine.)"""
        import operator
        return sum(map(operator.attrgetter("_version"), params))

    def _check_versions(self, eng) -> None:
        """In-place writes through the named parameters (load_state_dict, a torch optimizer) bump the parameters' version counters:
        re-derive the bf16 / packed weight copies when ANY parameter moved -- not four sentinel tensors, which missed an optimizer that
        only steps, say, one domain's action projectors."""
        ps = getattr(self, "_ver_params", None)
        if ps is None or getattr(self, "_ver_engine", None) is not eng:
            ps = self._ver_params = list(self.parameters())
            self._ver_engine = eng
        ver = self._version_sum(ps)
        if ver != getattr(self, "_seen_versions", None):
            self._seen_versions = ver
            eng.weights_changed()

    def _domain_key(self, domain) -> Optional[str]:
        if domain is None:
            return None
        return domain if isinstance(domain, str) else domain[0]

    def _hw(self, kwargs):
        h, w = self.h, self.w
        if "h" in kwargs:
            assert "w" in kwargs
            h, w = int(kwargs["h"][0]), int(kwargs["w"][0])
        return h, w

    def _run(self, x_THW: torch.Tensor, labels, action_ids, domain, train: bool, kwargs):
        eng = self._get_engine(x_THW.device)
        self._check_versions(eng)
        B, T = x_THW.shape[:2]
        ids = x_THW.reshape(B, T, -1).contiguous()
        dom = self._domain_key(domain) if (action_ids is not None or (self.config.jointly_predict_actions and not train)) else None
        ws = eng.forward(ids, labels, action_ids, dom, train, skip_normalization=bool(kwargs.get("skip_normalization", False)),
                         action_mask=kwargs.get("action_mask"))
        return eng, ws

    # ------------------------------------------------------------------ reference API
    def compute_logits(self, x_THW: torch.Tensor, action_ids: torch.Tensor = None, domain=None, **kwargs):
        """(B,T,H,W) ids -> (logits (B, C, T, H, W), None)  (st_mask_git.py:632-686)."""
        h, w = self._hw(kwargs)
        B, T = x_THW.shape[:2]
        eng, ws = self._run(x_THW, None, action_ids, domain, False, kwargs)
        logits = ws["logits"].view(B, T, h, w, -1).clone().permute(0, 4, 1, 2, 3)  # (own storage: valid after the next forward)
        actions = eng._act["out"].view(B, T, -1).clone() if eng._act is not None else None  # jointly_predict_actions (:676-678)
        return logits, actions

    def forward(self, input_ids, labels, action_ids=None, domain="default", **kwargs):
        """input_ids / labels (B, T*H*W) int64; action_ids (B, T, Da) (st_mask_git.py:688-735)."""
        T = self.config.T
        H, W = self._hw(kwargs)
        B = input_ids.shape[0]
        x_THW = input_ids.reshape(B, T, H, W)
        train = torch.is_grad_enabled() and self.training
        if self.config.jointly_predict_actions and action_ids is not None and kwargs.get("action_mask") is None:
            # between 0 (fully unmasked, as in video prediction) and 1 (fully masked, as in policies), drawn per sample (:704-710);
            # pass `action_mask` (B, T) to fix it
            drop_ratio = torch.rand(len(action_ids), 1, 1)
            kwargs = dict(kwargs, action_mask=(torch.rand(len(action_ids), T, 1) < drop_ratio)[..., 0].to(input_ids.device))
        eng, ws = self._run(x_THW, labels, action_ids, domain, train, kwargs)
        stats = ws["stats"]
        loss_value = stats[0] / stats[2]
        acc = stats[1] / stats[2]
        # (a copy: the engine's logits buffer is overwritten by the next forward, and reference callers -- evaluate.py -- keep
        # `outputs.logits` across steps.  The fused Trainer path never comes through here.)
        logits = ws["logits"].view(B, T, H, W, -1).clone().permute(0, 4, 1, 2, 3)
        if train:
            if action_ids is not None:
                self.touched_domains.add(self._domain_key(domain))
            loss = _EngineLoss.apply(self._anchor, self, loss_value)
        else:
            loss = loss_value.clone()
        if eng._act is not None and eng._act["loss"] is None:  # policy mode (no action ids): predicted actions, no action loss
            return ModelOutput(loss=loss, acc=acc.clone(), logits=logits, actions=eng._act["out"].view(B, T, -1).clone())
        if eng._act is not None:  # jointly_predict_actions: (:724-733)
            a = eng._act
            action_loss = _ActionLoss.apply(self._anchor, self, a["loss"]) if train else a["loss"].clone()
            return ModelOutput(loss=loss, acc=acc.clone(), logits=logits, action_loss=action_loss, actions=a["out"].view(B, T, -1).clone())
        return ModelOutput(loss=loss, acc=acc.clone(), logits=logits)

    def compute_video_loss_and_acc(self, logits_CTHW, targets_THW, relevant_mask_THW):
        """Loss / accuracy of given logits (st_mask_git.py:603-630) through the CE kernel (no gradient)."""
        B, Cc, T, H, W = logits_CTHW.shape
        S = H * W
        dev = logits_CTHW.device
        lg = logits_CTHW.permute(0, 2, 3, 4, 1).reshape(B * T * S, Cc).contiguous().float()
        labels = targets_THW.reshape(B, T * S).contiguous()
        ids = torch.zeros(B, T, S, dtype=torch.int64, device=dev)
        ids[:, 1:] = torch.where(relevant_mask_THW.reshape(B, T - 1, S).bool(), self.mask_token_id, 0)
        stats = torch.zeros(8, dtype=torch.float32, device=dev)  # HMA_CE_STATS_FLOATS
        _lib.call("hma_count_masked", stream_ptr(), ptr(ids), ptr(stats), B, T, S, self.mask_token_id)
        _lib.call("hma_ce_fwd_bwd", stream_ptr(), ptr(lg), ptr(ids), ptr(labels), ptr(stats), None, None, 1.0, B, T, S,
                  self.mask_token_id, 0.01)
        return stats[0] / stats[2], stats[1] / stats[2]

    def _engine_backward(self, grad_out: torch.Tensor) -> None:
        eng = self._engine
        sentinel = self.out_x_proj.weight
        if not self._grads_live or sentinel.grad is None:
            eng.zero_grad()
        eng.gscale.copy_(grad_out.reshape(1).to(torch.float32))
        B, T, S, A, dom = eng._last
        ws = eng._ws
        ws["stats"].zero_()
        eng._loss_plan(B, T, S, True).run(torch.cuda.current_stream().cuda_stream)
        eng.backward(grad_scale=eng.grad_scale.value)
        self._grads_live = True
        active = set(eng.layout.regions) - {"frozen"} - {f"dom:{d}" for d in eng.domains if d != dom}
        with torch.no_grad():
            for name, p in self.named_parameters():
                if eng.layout.entries[name].region in active:
                    p.grad = eng.view(name, eng.G)

    def _trunk_grads_begin(self) -> None:
        """Before a block-level autograd backward (st_transformer._TrunkFn): the flat gradient buffer starts from zero unless this
        model's `.grad`s are already live views of it (gradient accumulation, or an earlier block of the same autograd pass)."""
        if not self._grads_live or all(p.grad is None for p in self.parameters()):
            self._engine.zero_grad()
            self._grads_live = True

    def _trunk_grads_publish(self, stamp) -> None:
        """Point `.grad` of the parameters of layers [l0, l1) (the action projectors of the step's domain only) at their views of
        the gradient buffer."""
        eng = self._engine
        dom, l0, l1 = stamp[4], stamp[5], stamp[6]
        with torch.no_grad():
            for l in range(l0, l1):
                pre = f"decoder.layers.{l}."
                for name, p in self.decoder.layers[l].named_parameters():
                    if ".action_projectors." in "." + name and (dom is None or f"action_projectors.{dom}." not in name):
                        continue
                    p.grad = eng.view(pre + name, eng.G)

    def _save_pretrained(self, save_directory) -> None:
        """model.safetensors with the reference's tensor names (the views share one flat storage, which
        safetensors' save_model would treat as aliased tensors, so they are cloned out first)."""
        import os

        from safetensors.torch import save_file

        state = {k: v.detach().to("cpu").contiguous().clone() for k, v in self.state_dict().items()}
        save_file(state, os.path.join(str(save_directory), "model.safetensors"))

    def zero_grad(self, set_to_none: bool = True):
        for p in self.parameters():
            p.grad = None
        if self._engine is not None:
            self._engine.zero_grad()
        self._grads_live = False
        self.touched_domains = set()

    # ------------------------------------------------------------------ generation
    def init_mask(self, prompt_THW, t=1):
        return torch.zeros(prompt_THW.size(0), t * self.seq_len, dtype=torch.bool, device=prompt_THW.device)

    @torch.no_grad()
    def maskgit_generate(self, prompt_THW: torch.LongTensor, out_t: int, maskgit_steps: int = 1, temperature: float = 0.0,
                         unmask_mode: str = "random", action_ids=None, domain="default", **kwargs):
        """MaskGIT decode of frame `out_t` (st_mask_git.py:338-467).  Returns (samples_HW, first-pass
        factored logits (B, V, NV, H, W), None) and writes the samples into `prompt_THW[:, out_t]`."""
        assert out_t, "maskgit_generate requires out_t > 0"
        assert torch.all(prompt_THW[:, out_t:] == self.mask_token_id), \
            f"when generating z{out_t}, frames {out_t} and later must be masked"
        if unmask_mode not in ("greedy", "random"):
            raise NotImplementedError(f"Expected `unmask_mode` to be one of ['greedy', 'random'], got {unmask_mode}")
        bs, t, h, w = prompt_THW.shape
        S = h * w
        cfg = self.config
        rand_draws = kwargs.pop("rand_draws", None)  # replay hook for tests: one (B, H, W) draw per step
        # Categorical branch (temperature > 1e-8, :411-416; the temperature cancels inside Categorical's normalisation, so it
        # only toggles argmax vs sampling).  torch.multinomial draws its single sample as argmax_k p_k / q_k, q ~ Exp(1):
        # `sample_draws` replays those q, one (B, H*W, num_factored_vocabs, vocab) tensor per step; otherwise they are drawn here.
        sample_draws = kwargs.pop("sample_draws", None)
        sampled = temperature > 1e-8
        work = prompt_THW.reshape(bs, t, S).contiguous()
        unmasked = torch.zeros(bs, S, dtype=torch.uint8, device=prompt_THW.device)
        first = None
        actions = None
        for step in range(maskgit_steps):
            eng, ws = self._run(work.view(bs, t, h, w), None, action_ids, domain, False, kwargs)
            if eng._act is not None:  # jointly_predict_actions: the step's predicted actions (:382, :392; the last step's are returned)
                actions = eng._act["out"].view(bs, t, -1).clone()
            if first is None:
                first = ws["logits"].view(bs, t, S, -1)[:, out_t].clone()
            last = step == maskgit_steps - 1
            n = 0 if last else math.ceil(cosine_schedule((step + 1) / maskgit_steps) * S)
            override = None
            if not last and unmask_mode == "random":
                if rand_draws is not None:
                    override = rand_draws[step].reshape(bs, S).contiguous().float()
                else:
                    override = torch.rand(bs, h, w, device=prompt_THW.device).reshape(bs, S).contiguous()  # torch.rand_like, :435
            noise = None
            if sampled:
                NV_, V_ = cfg.num_factored_vocabs, cfg.factored_vocab_size
                if (NV_, V_) != (2, 512):
                    raise NotImplementedError("the categorical MaskGIT step is built for the 2 x 512 factorised vocabulary")
                noise = (sample_draws[step].to(prompt_THW.device, torch.float32).reshape(bs, S, NV_, V_).contiguous()
                         if sample_draws is not None else
                         torch.empty(bs, S, NV_, V_, device=prompt_THW.device, dtype=torch.float32).exponential_())
                noise = noise.clamp_min(1e-30)  # (the draws divide the probabilities: an exact 0 would make inf / nan)
            eng.maskgit_step(work, unmasked, out_t, n, last, override, sample_noise=noise)
        prompt_THW.copy_(work.view(bs, t, h, w))
        samples_HW = work[:, out_t].view(bs, h, w).clone()
        V, NV = cfg.factored_vocab_size, cfg.num_factored_vocabs
        factored = first.view(bs, h, w, NV, V).permute(0, 4, 3, 1, 2)
        return samples_HW, factored, actions

    @torch.no_grad()
    def _generate_cached(self, input_ids, num_new_frames, h, w, maskgit_steps, temperature, action_ids, domain, kwargs):
        """`generate` with a per-layer temporal K/V cache: exactly the reference's arithmetic per frame (frame t's
        logits depend on frames <= t only), but each MaskGIT iteration moves the 320 rows of ONE frame through the
        trunk instead of the whole window; one extra pass per finished frame stores its final K/V."""
        unmask_mode = kwargs.pop("unmask_mode", "random")
        rand_draws = kwargs.pop("rand_draws", None)
        sample_draws = kwargs.pop("sample_draws", None)  # (see maskgit_generate) one tensor per MaskGIT step of the rollout
        sampled = temperature > 1e-8
        sdraw = 0
        step_hook = kwargs.pop("step_hook", None)  # tests: called before each sampling step with (t, step, window ids, frame logits)
        skip_norm = bool(kwargs.get("skip_normalization", False))
        S = h * w
        B = input_ids.size(0)
        dev = input_ids.device
        eng = self._get_engine(dev)
        prompt = input_ids.reshape(B, -1, S).contiguous()
        P = prompt.size(1)
        T_total = P + num_new_frames
        dom = self._domain_key(domain) if action_ids is not None else None
        acts = None if action_ids is None else action_ids.float().contiguous()
        eng.decode_prefill(prompt, acts, dom, T_total, skip_norm)
        out = torch.cat([prompt, torch.full((B, num_new_frames, S), self.mask_token_id, dtype=torch.long, device=dev)], dim=1)
        draw = 0
        for t in range(P, T_total):
            frame = out[:, t : t + 1].contiguous()  # (B, 1, S) view the sampling kernel updates in place
            unmasked = torch.zeros(B, S, dtype=torch.uint8, device=dev)
            a_t = None if acts is None else acts[:, t]
            for step in range(maskgit_steps):
                # (the frame's action embedding and adaLN rows are computed by its first pass and reused by the later ones)
                logits = eng.decode_frame(frame.view(B, S), a_t, dom, t, T_total, same_actions=step > 0)
                if step_hook is not None:
                    win = out.clone()
                    win[:, t] = frame[:, 0]
                    step_hook(t, step, win, logits)
                last = step == maskgit_steps - 1
                n = 0 if last else math.ceil(cosine_schedule((step + 1) / maskgit_steps) * S)
                override = None
                if not last and unmask_mode == "random":
                    if rand_draws is not None:
                        override = rand_draws[draw].reshape(B, S).contiguous().float()
                        draw += 1
                    else:
                        override = torch.rand(B, h, w, device=dev).reshape(B, S).contiguous()
                noise = None
                if sampled:
                    if sample_draws is not None:
                        noise = sample_draws[sdraw].to(dev, torch.float32).reshape(B, S, 2, 512).contiguous()
                        sdraw += 1
                    else:
                        noise = torch.empty(B, S, 2, 512, device=dev, dtype=torch.float32).exponential_()
                eng.maskgit_step(frame, unmasked, 0, n, last, override, logits_T=1, logits_t=0, logits=logits, sample_noise=noise)
            out[:, t] = frame[:, 0]
            if t + 1 < T_total:  # store the finished frame's K/V (its tokens changed after the last pass)
                eng.decode_frame(frame.view(B, S), a_t, dom, t, T_total, readout=False, same_actions=True)
        return out.reshape(B, -1)

    @torch.no_grad()
    def generate(self, input_ids: torch.LongTensor, attention_mask: torch.LongTensor = None, max_new_tokens: int = 0,
                 min_new_tokens: int = None, return_logits: bool = False, return_with_actions: bool = False,
                 maskgit_steps: int = 1, temperature: float = 0.0, action_ids: torch.Tensor = None, domain: str = "default",
                 **kwargs):
        """Frame-by-frame rollout over `maskgit_generate` (st_mask_git.py:253-329)."""
        assert min_new_tokens in (None, max_new_tokens), "Expecting `min_new_tokens`, if specified, to match `max_new_tokens`."
        h, w = self._hw(kwargs)
        S = h * w
        num_new_frames = max_new_tokens // S
        B = input_ids.size(0)
        # (jointly_predict_actions without input actions -- the policy mode, :663-666 -- appends mask tokens to every frame: only the
        # window path builds that; the K/V-cached loop is for the conditioned / unconditioned rollouts)
        policy = self.config.jointly_predict_actions and action_ids is None
        if kwargs.pop("use_cache", True) and not return_logits and not return_with_actions and num_new_frames > 0 and not policy:
            return self._generate_cached(input_ids, num_new_frames, h, w, maskgit_steps, temperature, action_ids, domain,
                                         kwargs)
        inputs_THW = input_ids.clone().reshape(B, -1, h, w)
        masked = torch.cat([inputs_THW, torch.full((B, num_new_frames, h, w), self.mask_token_id, dtype=torch.long,
                                                   device=input_ids.device)], dim=1)
        all_logits = []
        for timestep in range(inputs_THW.size(1), inputs_THW.size(1) + num_new_frames):
            sample_HW, factored_logits, actions = self.maskgit_generate(masked, timestep, maskgit_steps=maskgit_steps,
                                                                        temperature=temperature, action_ids=action_ids,
                                                                        domain=domain, **kwargs)
            masked[:, timestep] = sample_HW
            all_logits.append(factored_logits)
        tokens = masked.reshape(B, -1)
        if return_with_actions:  # (:319-322) the last frame's predicted actions, un-normalised
            if actions is None:
                raise ValueError("return_with_actions needs a model built with jointly_predict_actions=True")
            return tokens, self.action_preprocessor[self._domain_key(domain)].unnormalize(actions)
        if return_logits:
            return tokens, torch.stack(all_logits, dim=3)
        return tokens
