"""The MaskGIT sampling step and the K/V-cached incremental decode of STEngine (st_mask_git.py:253-467 as driven by
hma/generate.py:139-184): methods of STEngine, kept in their own file (a mix-in) so that the training plans in engine.py and the
decode cache / frame-pass plans here can be read separately."""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import (A_BF16, A_BF16_AFFINE, A_BF16_FRAG32, A_BF16_HEADBLK, A_F32, EPI_ATOMIC_F32, EPI_BF16, EPI_DGELU, EPI_DSILU, EPI_F32, EPI_GELU2,
                   EPI_RESID, EPI_SILU2)
from .plan import Plan

BF16, F32 = torch.bfloat16, torch.float32


class DecodeMixin:
    def maskgit_step(self, prompt_BTS: torch.Tensor, unmasked: torch.Tensor, out_t: int, n_mask: int, last: bool,
                     conf_override: Optional[torch.Tensor] = None, conf_out: Optional[torch.Tensor] = None,
                     logits_T: int = 0, logits_t: int = 0, logits: Optional[torch.Tensor] = None,
                     sample_noise: Optional[torch.Tensor] = None) -> None:
        """One sampling step on the logits of the last forward (st_mask_git.py:397-453); updates in place.  `sample_noise`
        (f32 [B, S, 2, 512], Exp(1) draws) selects the categorical branch (:411-416)."""
        B, T, S = prompt_BTS.shape
        assert prompt_BTS.is_contiguous() and prompt_BTS.dtype == torch.int64 and unmasked.dtype == torch.uint8
        stream = torch.cuda.current_stream().cuda_stream
        lg = self._ws["logits"] if logits is None else logits
        co = None if conf_override is None else conf_override.data_ptr()
        cout = None if conf_out is None else conf_out.data_ptr()
        if sample_noise is not None:
            assert sample_noise.is_contiguous() and sample_noise.dtype == F32 and sample_noise.numel() == B * S * 1024
        if B * S <= 64 * 256:
            # few samples (a decode step: 64): sampling as one wave per token over the whole chip, then the per-sample ranking
            sc = getattr(self, "_mg_scratch", None)
            if sc is None or sc[0].numel() < B * S or sc[0].device != prompt_BTS.device:
                sc = self._mg_scratch = (torch.empty(B * S, dtype=F32, device=prompt_BTS.device),
                                         torch.empty(B * S, dtype=torch.int32, device=prompt_BTS.device))
            _lib.call("hma_maskgit_step_wide", stream, lg.data_ptr(), prompt_BTS.data_ptr(), unmasked.data_ptr(), co,
                      cout if cout is not None else sc[0].data_ptr(), sc[1].data_ptr(),
                      None if sample_noise is None else sample_noise.data_ptr(), B, T, S, out_t, n_mask, int(last),
                      self.cfg.image_vocab_size, logits_T, logits_t)
            return
        if sample_noise is None:
            _lib.call("hma_maskgit_step", stream, lg.data_ptr(), prompt_BTS.data_ptr(), unmasked.data_ptr(), co, cout,
                      B, T, S, out_t, n_mask, int(last), self.cfg.image_vocab_size, logits_T, logits_t)
        else:
            assert sample_noise.is_contiguous() and sample_noise.dtype == F32 and sample_noise.numel() == B * S * 1024
            _lib.call("hma_maskgit_step_sampled", stream, lg.data_ptr(), prompt_BTS.data_ptr(), unmasked.data_ptr(), co, cout,
                      sample_noise.data_ptr(), B, T, S, out_t, n_mask, int(last), self.cfg.image_vocab_size, logits_T, logits_t)

    # ------------------------------------------------------------------------------ incremental decode
    # Frame t of the trunk depends on frames <= t only (spatial attention, modulation, MLP, positions and the
    # action tokens are per frame; temporal attention is causal), so per-layer temporal K/V of finished frames
    # are constants of the rollout: only the 320 rows of the frame being decoded flow through the layers.
    def decode_begin(self, B: int, T_total: int, S: int, A: int) -> Dict[str, torch.Tensor]:
        key = (B, T_total, S, A)
        if getattr(self, "_dws_key", None) == key:
            return self._dws
        self._dws, self._dplans, self._dgraphs, self._dseen = {}, {}, {}, {}
        L = self.cfg.num_layers
        SA, M1 = S + A, B * (S + A)
        dev = self.device
        d: Dict[str, torch.Tensor] = {}

        def buf(name, shape, dtype):
            d[name] = torch.empty(shape, dtype=dtype, device=dev)

        buf("ids", (B, 1, S), torch.int64)
        buf("x", (M1, 256), F32)
        for nm in ("xh1", "o_s", "x2b", "o_t", "xh2", "xhm", "xm"):
            buf(nm, (M1, 256), BF16)
        buf("qkv_s", (M1, 768), BF16)
        buf("lse_s", (M1, 8), F32)
        for nm in ("rstd1", "rstd2", "rstdm"):
            buf(nm, (M1,), F32)
        if not self._use_fused(M1, False):
            buf("u", (M1, 1024), BF16)
            buf("hg", (M1, 1024), BF16)
        buf("logits", (B * S, 1024), F32)
        buf("cache", (L, B * T_total * SA, 768), BF16)
        if A > 0:
            buf("actions", (B * self.max_d_a,), F32)
            buf("an", (B * self.max_d_a,), F32)
            buf("sxhat", (B, 256), F32)
            buf("srstd", (B,), F32)
            buf("sh", (B, 256), F32)
            buf("a_emb", (B, 256), F32)
            buf("ada_pre", (L, B, 256), BF16)
            buf("ada_act", (L, B, 256), BF16)
            buf("ss", (L, B, 512), F32)
        self._dws, self._dws_key = d, key
        return d

    def _decode_plan(self, B: int, T_total: int, S: int, A: int, domain: Optional[str], t: int, readout: bool,
                     same_actions: bool = False) -> Plan:
        """`same_actions`: the frame's actions are those of the previous pass (the 2nd .. last MaskGIT iteration of a frame and its
        K / V refresh): the action stem and the adaLN stacks -- a_emb and the per-layer shift / scale rows -- are reused, not re-run."""
        key = (B, T_total, S, A, domain, t, readout, self._skip_norm, same_actions)
        if key in self._dplans:
            return self._dplans[key]
        cfg, d = self.cfg, self._dws
        L = cfg.num_layers
        SA, M1 = S + A, B * (S + A)
        pl = Plan(self._dev_index)
        use_mod = A > 0 and self.modulate
        if A > 0 and not same_actions:
            am = f"action_mlp.{domain}.model"
            pl.add("hma_action_stem_fwd", d["actions"].data_ptr(), self.buffers[domain][0].data_ptr(),
                   self.buffers[domain][1].data_ptr(), self.action_dims[domain], self._p(f"{am}.0.weight"), self._p(f"{am}.0.bias"),
                   self._p(f"{am}.1.weight"), self._p(f"{am}.1.bias"), self._p(f"{am}.3.weight"), self._p(f"{am}.3.bias"),
                   d["an"].data_ptr(), d["sxhat"].data_ptr(), d["srstd"].data_ptr(), d["sh"].data_ptr(), d["a_emb"].data_ptr(), B,
                   self.d_actions[domain], self._skip_norm)
            if use_mod:
                ap = f"decoder.layers.0.action_projectors.{domain}"
                pl.gemm_nt(A=d["a_emb"].data_ptr(), lda=256, a_kind=A_F32, W=self._wb(f"{ap}.adaLN_modulation.0.weight"), ldw=256,
                           M=B, N=256, K=256, epi=EPI_SILU2, Cp=d["ada_pre"].data_ptr(), ldc=256,
                           bias=self._p(f"{ap}.adaLN_modulation.0.bias"), C2=d["ada_act"].data_ptr(), ldc2=256, batch=L, sA=0,
                           sW=self.layout.dom_layer_stride, sBias=self.layout.dom_layer_stride, sC=B * 256, sC2=B * 256)
                pl.gemm_nt(A=d["ada_act"].data_ptr(), lda=256, a_kind=A_BF16, W=self._wb(f"{ap}.adaLN_modulation.2.weight"),
                           ldw=256, M=B, N=512, K=256, epi=EPI_F32, Cp=d["ss"].data_ptr(), ldc=512,
                           bias=self._p(f"{ap}.adaLN_modulation.2.bias"), batch=L, sA=B * 256, sW=self.layout.dom_layer_stride,
                           sBias=self.layout.dom_layer_stride, sC=B * 512)
        pfr = cfg.S + cfg.action_token_size
        pl.add("hma_embed_fwd", d["ids"].data_ptr(), self._p("token_embed.factored_embeds.0.weight"),
               self._p("token_embed.factored_embeds.1.weight"), self._p("token_embed.mask_token_embed"),
               self._p("pos_embed_TSC") + 4 * t * pfr * 256, d["a_emb"].data_ptr() if A > 0 else None, d["x"].data_ptr(), B, 1, S, A,
               pfr, cfg.factored_vocab_size, cfg.image_vocab_size)
        x = d["x"].data_ptr()
        fused = self._use_fused(M1, False)
        names = ("xh1", "rstd1", "qkv_s", "o_s", "lse_s", "x2b", "o_t", "xh2", "rstd2", "xhm", "xm", "rstdm") + (
            () if fused else ("u", "hg"))
        for l in range(L):
            bufs = {k: d[k].data_ptr() for k in names}
            bufs["qkv_t"] = None
            if use_mod:
                bufs["ss"] = d["ss"].data_ptr() + l * B * 512 * 4
            kv = {"cache": d["cache"][l].data_ptr(), "row_off": t * SA, "c_group": (SA, T_total * SA), "t_query": t,
                  "T_cache": T_total}
            cb = self.chain_b_ok and self._use_chain(M1, SA)
            self._emit_layer(pl, l, x, bufs, M1, B, B, t + 1, SA, use_mod, domain, kv=kv, have_ln1=fused and l > 0,
                             ln_next=(d["xh1"].data_ptr(), d["rstd1"].data_ptr()) if fused and l + 1 < L else None, fused=fused,
                             have_qkv_s=cb and l > 0, chain_b=cb, next_qkv_s=bufs["qkv_s"] if (cb and l + 1 < L) else None)
        if readout:
            pl.gemm_nt(A=x, lda=256, a_kind=A_F32, a_group=(S, SA), W=self._wb("out_x_proj.weight"), ldw=256, M=B * S, N=1024,
                       K=256, epi=EPI_F32, Cp=d["logits"].data_ptr(), ldc=1024, bias=self._p("out_x_proj.bias"))
        self._dplans[key] = pl
        return pl

    def decode_prefill(self, ids_BPS: torch.Tensor, actions: Optional[torch.Tensor], domain: Optional[str], T_total: int,
                       skip_normalization: bool = False) -> None:
        """Run the prompt frames through the trunk, filling the per-layer temporal qkv cache."""
        B, P, S = ids_BPS.shape
        A = self.cfg.action_token_size if actions is not None else 0
        d = self.decode_begin(B, T_total, S, A)
        skip = 1 if skip_normalization else 0
        if skip != self._skip_norm:
            self._skip_norm, self._plans, self._dplans = skip, {}, {}
        ws = self._workspace(B, P, S, A, False)
        stream = torch.cuda.current_stream().cuda_stream
        self.refresh_weights(domain if actions is not None else None, stream)
        ws["ids"].copy_(ids_BPS, non_blocking=True)
        if actions is not None:
            d_a = self.d_actions[domain]
            ws["actions"][: B * P * d_a].copy_(actions[:, :P].reshape(-1), non_blocking=True)
        pl = self._forward_plan(B, P, S, A, False, domain if A > 0 else None, readout=False, kv_cache=d["cache"], T_cache=T_total)
        if self.jpa and A > 0:
            # jointly_predict_actions: the plan's embedding reads `a_tok` (what the concatenated action tokens carry).  Prompt frames
            # are never action-masked in a rollout (st_mask_git.py:656-660 with no relevant_action_mask): a_tok = the embedded actions.
            pl.run(stream, 0, pl.marks["post_stem"])
            ws["a_tok"].copy_(ws["a_emb"])
            pl.run(stream, pl.marks["post_stem"], None)
        else:
            pl.run(stream)

    def decode_frame(self, ids_BS: torch.Tensor, actions_t: Optional[torch.Tensor], domain: Optional[str], t: int, T_total: int,
                     readout: bool = True, same_actions: bool = False) -> torch.Tensor:
        """One pass of frame t (tokens ids_BS, possibly partly masked) against the cached frames < t; refreshes
        frame t's own cache rows.  Returns the (B*S, 1024) fp32 logits buffer of that frame."""
        B, S = ids_BS.shape
        A = self.cfg.action_token_size if actions_t is not None else 0
        d = self.decode_begin(B, T_total, S, A)
        stream = torch.cuda.current_stream().cuda_stream
        d["ids"].view(B, S).copy_(ids_BS, non_blocking=True)
        same_actions = same_actions and actions_t is not None
        if actions_t is not None and not same_actions:
            d_a = self.d_actions[domain]
            d["actions"][: B * d_a].copy_(actions_t.reshape(-1), non_blocking=True)
        pl = self._decode_plan(B, T_total, S, A, domain if A > 0 else None, t, readout, same_actions)
        if not self.decode_graphs:
            pl.run(stream)
            return d["logits"]
        # ~420 launches of M = B * 320 rows each: replayed as one hipGraph per (frame index, readout) after two eager runs
        key = (B, T_total, S, A, domain if A > 0 else None, t, readout, self._skip_norm, same_actions)
        g = self._dgraphs.get(key)
        if g is None:
            pl.run(stream)
            n = self._dseen.get(key, 0) + 1
            self._dseen[key] = n
            if n >= 2:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    pl.run(torch.cuda.current_stream().cuda_stream)
                self._dgraphs[key] = g
        else:
            g.replay()
        return d["logits"]
