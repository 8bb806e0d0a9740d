"""STEngine -- runs the STMaskGIT hot path (forward, loss, backward, clip+AdamW, MaskGIT step) as
replayable plans of libhma_hip.so launches over static device buffers.

Host side of the drop-in boundary: Python here only sequences C-ABI calls (include/hma_hip.h) on the
current HIP stream; all arithmetic is in the HIP kernels.  Reference path being replaced:
STMaskGIT.compute_logits / forward (hma/model/st_mask_git.py:632-735), STTransformerDecoder / STBlock
(hma/model/st_transformer.py:79-114, 172-177) and the update at hma/train_multi.py:593-598.

Data layout in HBM: the residual stream x is fp32 [B*T*(S+A), 256], rows
(b, t, s) with s fastest -- spatial attention reads 320-row frames contiguously, temporal attention
gathers the T rows of a column at stride (S+A) rows; every saved activation is bf16 in the same row
order; weights live in one flat fp32 buffer (params.py) with bf16 + transposed bf16 copies.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import (A_BF16, A_BF16_AFFINE, A_BF16_FRAG32, A_BF16_HEADBLK, A_F32, EPI_ATOMIC_F32, EPI_BF16, EPI_DGELU, EPI_DSILU, EPI_F32, EPI_GELU2,
                   EPI_RESID, EPI_SILU2)
from .ops import make_chain_a_bwd, make_chain_a_fwd, make_chain_b_fwd, make_chain_s_bwd, make_readout_ce, make_gemm_nt, make_gemm_tn, make_mlp_bwd, make_mlp_fwd
from .engine_decode import DecodeMixin
from .params import ALIGN, ParamLayout
from .plan import LaunchTimer, Plan  # noqa: F401  (re-exported: bench.py / tests import them from here)

BF16, F32 = torch.bfloat16, torch.float32


class STEngine(DecodeMixin):
    def __init__(self, cfg, domains: Sequence[str], d_actions: Sequence[int], action_dims: Sequence[int], device):
        if cfg.d_model != 256 or cfg.num_heads != 8:
            raise NotImplementedError("the gfx950 kernels are specialised for d_model=256, 8 heads of 32 (HMA-base)")
        if cfg.num_factored_vocabs != 2 or cfg.factored_vocab_size != 512:
            raise NotImplementedError("readout/loss kernels are built for the 2 x 512 factorised vocabulary")
        # jointly_predict_actions (st_mask_git.py:656-660, 676-678, 724-733): the action tokens of masked frames are
        # `action_mask_tokens[t]`, a frame's action tokens are mean-pooled after the trunk and read out per domain, and an action
        # loss trains that head.  The trunk kernels are unchanged; the three small per-frame pieces ([B*T, 256] mixes, a 256 x d_a
        # Linear, its MSE) are fp32 torch ops on the GPU between plan segments.
        # qk_norm=True (the GenieConfig dataclass default; no shipped JSON uses it): norm1 / norm2 are identities and q, k get a per-head
        # LayerNorm (attention.py:31-35,44-48; st_transformer.py:55,62).  Runs on the GEMM-per-Linear plans (the chain / fused-MLP
        # kernels have the block's LayerNorms built in) with hma_qknorm_fwd / _bwd behind the qkv projections.
        self.qkn = bool(cfg.qk_norm)
        # the per-domain adaLN stacks' backward runs per group of this many layers (a data-parallel driver sets it to its layers per
        # gradient bucket, so that a domain's slice of a bucket is final when the bucket is)
        self.ada_group = 8
        self.jpa = bool(cfg.jointly_predict_actions)
        self._act: Optional[dict] = None
        self.act_scale = 0.0   # d total / d action_loss of the backward in flight (0: the action loss is not part of the objective)
        _lib.load()
        self.cfg = cfg
        self.device = torch.device(device)
        self.domains = list(domains)
        self.d_actions = dict(zip(domains, d_actions))
        self.action_dims = dict(zip(domains, action_dims))
        self.layout = ParamLayout(cfg, domains, d_actions, action_dims)
        n = self.layout.total
        self.P = torch.zeros(n, dtype=F32, device=self.device)
        self.G = torch.zeros(n, dtype=F32, device=self.device)
        self.Wb = torch.zeros(n, dtype=BF16, device=self.device)
        self.M: Optional[torch.Tensor] = None  # Adam moments, allocated on first optimizer step
        self.V: Optional[torch.Tensor] = None
        self.flags = self.layout.decay_flags().to(self.device)
        self.sqnorm = torch.zeros(1, dtype=F32, device=self.device)
        # Update counts per flat range (0 = dense, 1 + i = action domain i) live ON THE DEVICE (hma_adamw_counted): a step whose
        # all-reduced gradient norm is not finite is skipped identically on every rank and does not count, without a host sync.
        # Slot [r, calls & 1] is current after `calls` launches on range r.
        self.steps_dev = torch.zeros(1 + len(self.domains), 2, dtype=torch.int32, device=self.device)
        self._step_calls = [0] * (1 + len(self.domains))
        self.drop_seed = torch.zeros(1, dtype=torch.int32, device=self.device)  # bumped per training forward (dropout masks)
        self._drop_counter = 0
        L, d = cfg.num_layers, cfg.d_model
        hid = int(d * cfg.mlp_ratio)
        mk = lambda *s: torch.zeros(*s, dtype=BF16, device=self.device)
        # transposed bf16 weight copies for the input-gradient GEMMs
        self.WT = {"qkv_s": mk(L, d, 3 * d), "proj_s": mk(L, d, d), "qkv_t": mk(L, d, 3 * d), "proj_t": mk(L, d, d),
                   "fc1": mk(L, d, hid), "fc2": mk(L, hid, d), "out": mk(d, 1024)}
        # forward weights of the two Linears that follow a LayerNorm, with the LayerNorm's affine folded in
        # (hma_fold_ln_bf16): the forward GEMM then reads the saved xhat as a plain bf16 operand
        self.WF = {"qkv_s": mk(L, 3 * d, d), "fc1": mk(L, hid, d)}
        self.BF = {"qkv_s": torch.zeros(L, 3 * d, dtype=F32, device=self.device), "fc1": torch.zeros(L, hid, dtype=F32, device=self.device)}
        # Fused MLP block (csrc/mlp.hip; the four MFMA-fragment-ordered weight streams per layer, 512 KB each): training and
        # large inference passes.  Measured in situ on MI355X: forward 256 us per layer at M = 163840
        # against 418 us for fc1 + fc2 + the next LayerNorm; backward 535-570 us against 379 us for dfc2 + dfc1 + LayerNorm
        # backward; the step as a whole comes out 1-2 % ahead with 1.4 GB less HBM traffic per layer.  Passes with fewer
        # rows than half a 128-row tile per CU keep the two GEMM launches.
        # With mlp_drop > 0 a TRAINING pass is fused only where chain B runs its forward: chain B and hma_mlp_bwd carry the two
        # nn.Dropout masks, hma_mlp_fwd does not.
        self.mlp_drop = float(getattr(cfg, "mlp_drop", 0.0) or 0.0)
        self.fused_mlp = hid == 1024
        self.fused_mlp_train = self.fused_mlp
        self.fused_mlp_min_rows = 128 * 128   # (the decode frame pass at B = 64, 20 480 rows = 160 tiles, is 17 % faster with it)
        if self.fused_mlp:
            self.MP = {k: mk(L, 512 * 512) for k in ("w1p", "w2p", "w2tp", "w1tp")}
        self.modulate = "modulate" in cfg.action_network
        if self.modulate:
            for dom in self.domains:
                self.WT[f"lin:{dom}"] = mk(L, d, d)
                self.WT[f"ada0:{dom}"] = mk(L, d, d)
                self.WT[f"ada2:{dom}"] = mk(L, d, 2 * d)
        # Row-local chain kernels (csrc/chain.hip): everything between the spatial attention and the temporal attention of a
        # block in one launch, forward and backward.  Their weights are streamed as 16 KB bundles (hma_chain_pack, 8192 bf16
        # each): per layer proj_s (8 bundles), qkv_t (24), and the same transposed for the backward chain; per domain
        # linear_out (8) and its transpose.
        self.use_chain = True
        self.chain_min_rows = 0
        # Which fused launches a pass takes is decided by its SHAPE alone (the `_use_*` predicates and the conditions at the call sites):
        # the round-5 environment switches that selected the older launch sequences inside one build are gone -- a same-box A / B of a
        # kernel change runs two builds (`bench.py --lib variants/...`).  The older sequences remain as what the shapes outside the
        # fused kernels' reach run: T != 16 (chain A / temporal attention / chain B as three launches), T <= 8 (projection dgrad +
        # hma_attn_temporal_bwd), qk_norm, frames that are not a multiple of 32 rows (row-major dqkv), passes below the LDS-DMA
        # weight-gradient path (pair launches, hma_ln_bwd).
        self.chain_s = True      # spatial qkv dgrad + norm1 backward + residual in one launch (hma_chain_s_bwd)
        self.attn_hb = True      # the spatial attention backward's dqkv head-blocked (hma_attn_spatial_bwd_blocked)
        self.chain_ab = True     # chain A + causal temporal attention + chain B in one launch (hma_chain_ab_fwd)
        self.chain_t = True      # temporal projection dgrad + temporal attention backward in one launch (hma_chain_t_bwd)
        self.wgrad_multi = True  # the block's seven weight gradients in one launch at the end of its backward (hma_gemm_tn_multi)
        # ... and the weight gradients of this many consecutive blocks in ONE launch (2: every workgroup streams twice the rows of its
        # 256 x 256 block, so the launch writes and its reduction reads half the partials per layer, and the launch's fixed cost is
        # paid once per two layers; the operands of the block in waiting live in a second set of buffers)
        self.wgrad_layers = 2
        BUN = 8192
        self.CP = {"proj_s": mk(L, 8 * BUN), "qkv_t": mk(L, 24 * BUN), "proj_s_T": mk(L, 8 * BUN), "qkv_t_T": mk(L, 24 * BUN),
                   "qkv_s_T": mk(L, 24 * BUN), "proj_t_T": mk(L, 8 * BUN)}  # (qkv_s_T: norm1's gamma folded into its output rows, for chain S backward)
        # readout + cross-entropy in one launch (hma_readout_ce) for training steps that do not hand the logits to the caller
        self.fused_ce = True
        self.CP["out"] = mk(32 * BUN)
        # chain B (proj_t + norm2 + MLP + the next block's norm1 + qkv_s in one launch) for passes that save nothing
        self.chain_b_ok = hid == 1024
        self.chain_b_train = True
        if self.chain_b_ok:
            self.CP.update({"proj_t": mk(L, 8 * BUN), "mlp": mk(L, 64 * BUN), "qkv_s": mk(L, 24 * BUN)})
        if self.modulate:
            for dom in self.domains:
                self.CP[f"lin:{dom}"] = mk(L, 8 * BUN)
                self.CP[f"lin_T:{dom}"] = mk(L, 8 * BUN)
        if self.qkn:
            self.use_chain = self.fused_mlp = self.fused_mlp_train = self.chain_b_ok = self.fused_ce = False
        self.buffers: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}  # domain -> (mean, std) of ActionStat
        self._pack_jobs = {}          # None | domain -> the hma_*_multi job arrays of refresh_weights
        self._seen_version = -1       # P._version at the last refresh (detects external in-place updates)
        self._wb_ok = False           # flat bf16 copy current
        self._wt_ok = False           # dense transposed copies current
        self._dom_fresh: set = set()  # domains whose transposed copies are current
        self.max_d_a = max(list(d_actions) + [1])
        self.gscale = torch.ones(1, dtype=F32, device=self.device)  # device-side loss-gradient scale
        self._ws: Dict[str, torch.Tensor] = {}
        self._ws_key = None
        self._dws: Dict[str, torch.Tensor] = {}
        self._dws_key = None
        self._dplans: Dict[tuple, Plan] = {}
        self._dgraphs: Dict[tuple, "torch.cuda.CUDAGraph"] = {}
        self._dseen: Dict[tuple, int] = {}
        self.decode_graphs = True  # a decode frame pass is replayed as one hipGraph per (frame index, readout) after two eager runs
        self._plans: Dict[tuple, Plan] = {}
        self.scale = (8.0 / 32.0) if cfg.use_mup else 32.0 ** -0.5  # attention.py:27
        self.grad_scale = C.c_float(1.0)
        self.timer: Optional[LaunchTimer] = None
        # One workspace per GPU for the life of the process: recorded plans (and captured graphs) of EVERY engine hold
        # its raw pointer, so it must never be re-allocated.  (It used to be when `device` came without an index --
        # torch.device("cuda") != tensor.device -- and a second engine's creation then freed the block under the first
        # engine's plans, whose wgrad partials landed in whatever the allocator put there next.)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._dev_index = idx
        if idx not in Plan._tn_workspaces:
            Plan._tn_workspaces[idx] = torch.empty(256 * (65536 + 256), dtype=F32, device=torch.device("cuda", idx))  # 64 MB + bias partials

    # ------------------------------------------------------------------------------ Adam update counts (device-resident)
    def _steps_now(self) -> List[int]:
        vals = self.steps_dev.cpu()
        return [int(vals[i, self._step_calls[i] & 1]) for i in range(len(self._step_calls))]

    @property
    def opt_step(self) -> int:
        """Updates applied to the dense range (reads the device counter: synchronises)."""
        return self._steps_now()[0]

    @property
    def dom_steps(self) -> Dict[str, int]:
        """Updates applied to each action domain's block (a head is only stepped when its domain had a gradient on some rank)."""
        cur = self._steps_now()
        return {d: cur[1 + i] for i, d in enumerate(self.domains) if cur[1 + i] > 0}

    def set_steps(self, opt_step: int, dom_steps: Dict[str, int]) -> None:
        vals = self.steps_dev.cpu()
        vals[0, self._step_calls[0] & 1] = int(opt_step)
        for i, d in enumerate(self.domains):
            vals[1 + i, self._step_calls[1 + i] & 1] = int(dom_steps.get(d, 0))
        self.steps_dev.copy_(vals)

    # ------------------------------------------------------------------------------ pointers
    def _p(self, name: str) -> int:
        return self.P.data_ptr() + 4 * self.layout.off(name)

    def _g(self, name: str) -> int:
        return self.G.data_ptr() + 4 * self.layout.off(name)

    def _wb(self, name: str) -> int:
        return self.Wb.data_ptr() + 2 * self.layout.off(name)

    def view(self, name: str, buf: Optional[torch.Tensor] = None) -> torch.Tensor:
        e = self.layout.entries[name]
        buf = self.P if buf is None else buf
        return buf[e.offset:e.offset + e.numel].view(e.shape)

    # ------------------------------------------------------------------------------ weights
    def refresh_weights(self, domain: Optional[str], stream: int) -> None:
        """(Re)build the bf16 and transposed-bf16 copies if the fp32 masters changed."""
        lay, cfg = self.layout, self.cfg
        L, d = cfg.num_layers, cfg.d_model
        hid = int(d * cfg.mlp_ratio)
        if self._seen_version != self.P._version:  # load_state_dict / an external optimizer wrote the views
            self._seen_version = self.P._version
            self._wb_ok = self._wt_ok = False
            self._dom_fresh = set()
        if not self._wb_ok:
            _lib.call("hma_cast_bf16", stream, self.P.data_ptr(), self.Wb.data_ptr(), lay.total)
            self._wb_ok = True
        if not self._wt_ok:
            jobs = self._pack_jobs.get(None)
            if jobs is None:
                jobs = self._pack_jobs[None] = self._layer_pack_jobs()
            self._run_pack_jobs(jobs, stream)
            if not self.qkn:
                ls = lay.layer_stride
                for key, lin, norm, rows, has_bias in (("qkv_s", "spatial_attn.qkv", "norm1", 3 * d, cfg.qkv_bias),
                                                       ("fc1", "mlp.fc1", "norm2", hid, cfg.mlp_bias)):
                    pre = f"decoder.layers.{L - 1}."
                    _lib.call("hma_fold_ln_bf16", stream, self._p(pre + lin + ".weight"), self._p(pre + norm + ".weight"),
                              self._p(pre + norm + ".bias"), self._p(pre + lin + ".bias") if has_bias else None,
                              self.WF[key][L - 1].data_ptr(), self.BF[key][L - 1].data_ptr(), rows, d, L, ls, -rows * d, -rows)
            self._wt_ok = True
        if domain is not None and self.modulate and domain not in self._dom_fresh:
            jobs = self._pack_jobs.get(domain)
            if jobs is None:
                jobs = self._pack_jobs[domain] = self._domain_pack_jobs(domain)
            self._run_pack_jobs(jobs, stream)
            self._dom_fresh.add(domain)

    @staticmethod
    def _run_pack_jobs(jobs, stream: int) -> None:
        """One launch per kind of copy (hma_*_multi): transposed bf16, fused-MLP fragments, chain bundles."""
        for fn, arr in jobs:
            if len(arr):
                _lib.call(fn, stream, arr, len(arr))

    def _layer_pack_jobs(self):
        """The job lists of the per-layer weight copies (built once: every pointer is into the flat buffers, which never move)."""
        lay, cfg = self.layout, self.cfg
        L, d = cfg.num_layers, cfg.d_model
        hid = int(d * cfg.mlp_ratio)
        # layers are stored L-1 .. 0 at a constant stride: batch index j reads layer L-1-j and writes slot L-1-j
        ls = lay.layer_stride
        pre = f"decoder.layers.{L - 1}."
        tj, mj, cj = [], [], []
        for key, suffix, rows, cols in (("qkv_s", "spatial_attn.qkv.weight", 3 * d, d), ("proj_s", "spatial_attn.proj.weight", d, d),
                                        ("qkv_t", "temporal_attn.qkv.weight", 3 * d, d), ("proj_t", "temporal_attn.proj.weight", d, d),
                                        ("fc1", "mlp.fc1.weight", hid, d), ("fc2", "mlp.fc2.weight", d, hid)):
            tj.append(dict(src=self._p(pre + suffix), dst=self.WT[key][L - 1].data_ptr(), rows=rows, cols=cols, batch=L,
                           src_batch_stride=ls, dst_batch_stride=-rows * cols))
        tj.append(dict(src=self._p("out_x_proj.weight"), dst=self.WT["out"].data_ptr(), rows=1024, cols=d, batch=1))
        if self.fused_mlp:
            w1, w2, g2 = self._p(pre + "mlp.fc1.weight"), self._p(pre + "mlp.fc2.weight"), self._p(pre + "norm2.weight")
            n = 512 * 512
            for key, src, rs, cs, rsc, csc, kind in (("w1p", w1, d, 1, None, g2, 0), ("w2p", w2, hid, 1, None, None, 1),
                                                      ("w2tp", w2, 1, hid, None, None, 0), ("w1tp", w1, 1, d, g2, None, 1)):
                mj.append(dict(src=src, row_stride=rs, col_stride=cs, row_scale=rsc, col_scale=csc, dst=self.MP[key][L - 1].data_ptr(),
                               kind=kind, batch=L, src_batch_stride=ls, dst_batch_stride=-n))
        if self.use_chain:
            BUN = 8192

            def cp(src, rs, cs, rsc, csc, dst, kind, rows, cols, batch=L, sstride=ls, dstride=0, bstride=1):
                cj.append(dict(src=src, row_stride=rs, col_stride=cs, row_scale=rsc, col_scale=csc, dst=dst, kind=kind, rows=rows,
                               cols=cols, batch=batch, src_batch_stride=sstride, dst_batch_stride=dstride, bundle_stride=bstride))

            wp, wq = self._p(pre + "spatial_attn.proj.weight"), self._p(pre + "temporal_attn.qkv.weight")
            cp(wp, d, 1, None, None, self.CP["proj_s"][L - 1].data_ptr(), 0, d, d, dstride=-8 * BUN)
            cp(wp, 1, d, None, None, self.CP["proj_s_T"][L - 1].data_ptr(), 0, d, d, dstride=-8 * BUN)
            cp(wq, d, 1, None, None, self.CP["qkv_t"][L - 1].data_ptr(), 0, 3 * d, d, dstride=-24 * BUN)
            cp(self._p("out_x_proj.weight"), d, 1, None, None, self.CP["out"].data_ptr(), 0, 1024, d, batch=1, sstride=0)
            cp(self._p(pre + "temporal_attn.proj.weight"), 1, d, None, None, self.CP["proj_t_T"][L - 1].data_ptr(), 0, d, d,
               dstride=-8 * BUN)  # (chain T backward)
            for c in range(3):  # input gradient: A[n][k] = W[256 c + k][n], one 8-bundle group per k-chunk
                cp(wq + 4 * c * d * d, 1, d, None, None, self.CP["qkv_t_T"][L - 1].data_ptr() + 2 * c * 8 * BUN, 0, d, d, dstride=-24 * BUN)
            wqs_, g1_ = self._p(pre + "spatial_attn.qkv.weight"), self._p(pre + "norm1.weight")
            for c in range(3):  # the same for the spatial qkv, times norm1's gamma per OUTPUT row (d xhat = gamma . (dqkv W))
                cp(wqs_ + 4 * c * d * d, 1, d, g1_, None, self.CP["qkv_s_T"][L - 1].data_ptr() + 2 * c * 8 * BUN, 0, d, d, dstride=-24 * BUN)
            if self.chain_b_ok:
                # chain B (inference): proj_t, the fc1 (norm2's gamma folded in) / fc2 bundles interleaved per hidden block, and the
                # spatial qkv with norm1's gamma folded in (consumed by the PREVIOUS block's chain)
                wpt, w1, w2 = self._p(pre + "temporal_attn.proj.weight"), self._p(pre + "mlp.fc1.weight"), self._p(pre + "mlp.fc2.weight")
                g2 = self._p(pre + "norm2.weight")
                cp(wpt, d, 1, None, None, self.CP["proj_t"][L - 1].data_ptr(), 0, d, d, dstride=-8 * BUN)
                cp(w1, d, 1, None, g2, self.CP["mlp"][L - 1].data_ptr(), 0, hid, d, dstride=-64 * BUN, bstride=2)
                cp(w2, hid, 1, None, None, self.CP["mlp"][L - 1].data_ptr() + 2 * BUN, 1, d, hid, dstride=-64 * BUN, bstride=2)
                cp(wqs_, d, 1, None, g1_, self.CP["qkv_s"][L - 1].data_ptr(), 0, 3 * d, d, dstride=-24 * BUN)
        return [("hma_transpose_cast_bf16_multi", _lib.pack_jobs(tj)), ("hma_mlp_pack_multi", _lib.pack_jobs(mj)),
                ("hma_chain_pack_multi", _lib.pack_jobs(cj))]

    def _domain_pack_jobs(self, domain: str):
        lay, cfg = self.layout, self.cfg
        L, d = cfg.num_layers, cfg.d_model
        pre = f"decoder.layers.0.action_projectors.{domain}"
        ds = lay.dom_layer_stride  # (a domain's block is layer-major: the batch stride of every per-layer stack in it)
        tj = [dict(src=self._p(f"{pre}.{name}.weight"), dst=self.WT[f"{key}:{domain}"].data_ptr(), rows=rows, cols=d, batch=L,
                   src_batch_stride=ds, dst_batch_stride=rows * d)
              for key, name, rows in (("lin", "linear_out", d), ("ada0", "adaLN_modulation.0", d), ("ada2", "adaLN_modulation.2", 2 * d))]
        cj = []
        if self.use_chain:
            wl = self._p(f"{pre}.linear_out.weight")
            for key, rs, cs in ((f"lin:{domain}", d, 1), (f"lin_T:{domain}", 1, d)):
                cj.append(dict(src=wl, row_stride=rs, col_stride=cs, dst=self.CP[key].data_ptr(), kind=0, rows=d, cols=d, batch=L,
                               src_batch_stride=ds, dst_batch_stride=8 * 8192, bundle_stride=1))
        return [("hma_transpose_cast_bf16_multi", _lib.pack_jobs(tj)), ("hma_chain_pack_multi", _lib.pack_jobs(cj))]

    def weights_changed(self) -> None:
        self._wb_ok = self._wt_ok = False
        self._dom_fresh = set()

    # ------------------------------------------------------------------------------ workspace
    def _use_fused(self, rows: int, train: bool, SA: int = 0) -> bool:
        """Whether a pass over `rows` token rows (frames of SA rows) runs the fused MLP block (see __init__)."""
        if not self.fused_mlp:
            return False
        if train and self.mlp_drop > 0.0 and not (self.chain_b_ok and self.chain_b_train and SA > 0 and self._use_chain(rows, SA)):
            return False
        if train and rows % 32:  # hma_mlp_bwd hands gelu(u) / dL/du to the weight gradients in the HMA_A_BF16_FRAG32 order (LDS-DMA path only)
            return False
        return rows >= self.fused_mlp_min_rows and (self.fused_mlp_train or not train)

    def _use_chain(self, rows: int, SA: int) -> bool:
        """Whether a pass over `rows` token rows (frames of SA rows) runs the row-local chain kernels."""
        return self.use_chain and rows % 16 == 0 and SA % 16 == 0 and rows >= self.chain_min_rows

    def _workspace(self, B: int, T: int, S: int, A: int, train: bool) -> Dict[str, torch.Tensor]:
        key = (B, T, S, A, train, self._use_fused(B * T * (S + A), train, S + A), self._use_chain(B * T * (S + A), S + A))
        if self._ws_key == key:
            return self._ws
        self._ws, self._plans = {}, {}
        self.ws_generation = getattr(self, "ws_generation", 0) + 1  # captured graphs point into the old buffers
        torch.cuda.empty_cache()
        L = self.cfg.num_layers
        Ls = L if train else 1
        SA = S + A
        M, Mi, Fr = B * T * SA, B * T * S, B * T
        dev = self.device
        ws: Dict[str, torch.Tensor] = {}

        def buf(name, shape, dtype):
            ws[name] = torch.empty(shape, dtype=dtype, device=dev)

        buf("ids", (B, T, S), torch.int64)
        buf("labels", (B, T * S), torch.int64)
        buf("x", (M, 256), F32)
        for nm in ("xh1", "o_s", "x2b", "o_t", "xh2"):
            buf(nm, (Ls, M, 256), BF16)
        buf("qkv_s", (Ls, M, 768), BF16)
        buf("qkv_t", (Ls, M, 768), BF16)
        buf("lse_s", (Ls, M, 8), F32)
        buf("rstd1", (Ls, M), F32)
        buf("rstd2", (Ls, M), F32)
        if self.qkn and train:
            buf("qraw_s", (Ls, M, 512), BF16)  # q | k in front of their LayerNorm, for its backward
            buf("qraw_t", (Ls, M, 512), BF16)
        fused = self._use_fused(M, train, SA)
        if not fused:  # the fused MLP block never materialises the hidden activation
            buf("u", (Ls, M, 1024), BF16)
            buf("hg", (Ls, M, 1024), BF16)
        if A > 0:
            buf("xhm", (Ls, M, 256), BF16)
            buf("xm", (Ls, M, 256), BF16)
            buf("rstdm", (Ls, M), F32)
            buf("ada_pre", (L, Fr, 256), BF16)
            buf("ada_act", (L, Fr, 256), BF16)
            buf("ss", (L, Fr, 512), F32)
            buf("actions", (Fr * self.max_d_a,), F32)
            buf("an", (Fr * self.max_d_a,), F32)
            buf("sxhat", (Fr, 256), F32)
            buf("srstd", (Fr,), F32)
            buf("sh", (Fr, 256), F32)
            buf("a_emb", (Fr, 256), F32)
            if self.jpa:
                buf("a_tok", (Fr, 256), F32)    # what the concatenated action tokens carry: a_emb, or action_mask_tokens[t] on masked frames
                buf("amask", (Fr, 1), F32)
                buf("da_tok", (Fr, 256), F32)
        buf("logits", (Mi, 1024), F32)
        buf("stats", (8,), F32)  # HMA_CE_STATS_FLOATS: loss / hits / masked rows + the kernels' ticket and fixed-point accumulator
        if train:
            buf("dlogits", (Mi, 1024), BF16)
            buf("dx", (M, 256), F32)
            buf("dxb", (M, 256), BF16)  # bf16 copy of dx: operand of the dgrad / wgrad GEMMs that read it
            if fused:
                buf("dxb2", (M, 256), BF16)   # hma_mlp_bwd writes the new copy while the fc2 weight gradient still reads the old one
                Mt = (M + 127) // 128 * 128    # gelu(u) and dL/du of the layer in flight (operands of its two weight gradients),
                buf("hg1", (Mt, 1024), BF16)   # in hma_mlp_bwd's fragment order (HMA_A_BF16_FRAG32: whole 128-row tiles)
                buf("du1", (Mt, 1024), BF16)
            if float(getattr(self.cfg, "mlp_drop", 0.0) or 0.0) > 0.0:
                buf("dxm", (M, 256), BF16)  # dx behind the Dropout that follows fc2
            buf("t256", (M, 256), BF16)
            if self._use_chain(M, SA) and A > 0 and self.modulate:
                buf("dx2b", (M, 256), BF16)  # bf16(dx) in front of the modulate block: dY of linear_out's weight gradient
            buf("dqkv", (M, 768), BF16)
            if self.wgrad_multi and fused and self._use_chain(M, SA):
                # the deferred weight gradients read bf16(dx) as it was at three points of the block and both attentions' dqkv: every
                # producer of bf16(dx) writes the next of four buffers, the spatial attention gets its own dqkv
                buf("dxb3", (M, 256), BF16)
                buf("dxb4", (M, 256), BF16)
                buf("dqkv_s", (M, 768), BF16)
                # the second set: operands of the block whose weight gradients wait for the next block's (wgrad_layers = 2)
                for nm in ("dxb5", "dxb6", "dxb7", "dxb8"):
                    buf(nm, (M, 256), BF16)
                buf("dqkv_s_b", (M, 768), BF16)
                buf("dqkv_b", (M, 768), BF16)
                buf("hg1_b", (Mt, 1024), BF16)
                buf("du1_b", (Mt, 1024), BF16)
                if "dx2b" in ws:
                    buf("dx2b_b", (M, 256), BF16)
                if "dxm" in ws:
                    buf("dxm_b", (M, 256), BF16)
            buf("delta", (M, 8), F32)
            if A > 0:
                buf("dss", (L, Fr, 512), F32)
                buf("dpre", (L, Fr, 256), BF16)
                buf("da_emb", (Fr, 256), F32)
                buf("stem_scratch", (Fr, 256), F32)
        self._ws, self._ws_key = ws, key
        return ws

    # ------------------------------------------------------------------------------ plans
    def _lw(self, l: int, suffix: str, kind: str = "wb") -> int:
        name = f"decoder.layers.{l}.{suffix}"
        return {"wb": self._wb, "p": self._p, "g": self._g}[kind](name)

    def _has(self, name: str) -> bool:
        return name in self.layout.entries

    def _emit_layer(self, pl: Plan, l: int, x: int, b: Dict[str, int], M: int, Fr: int, B: int, T: int, SA: int,
                    use_mod: bool, domain: Optional[str], kv: Optional[dict] = None, train: bool = False,
                    have_ln1: bool = False, ln_next: Optional[Tuple[int, int]] = None, fused: bool = False,
                    have_qkv_s: bool = False, chain_b: bool = False, next_qkv_s: Optional[int] = None,
                    next_ln1: Optional[Tuple[int, int]] = None) -> None:
        """One STBlock forward (st_transformer.py:79-114) on M rows = Fr frames of SA tokens.  `kv` redirects the
        temporal qkv into the per-layer decode cache: {"cache": ptr, "row_off": rows, "c_group": (rows, stride),
        "t_query": -1 | t, "T_cache": frames}."""
        cfg = self.cfg
        dp_ = lambda t, i: t[i].data_ptr()
        qb = lambda a: self._lw(l, f"{a}.qkv.bias", "p") if cfg.qkv_bias else None
        pb = lambda a: self._lw(l, f"{a}.proj.bias", "p") if cfg.proj_bias else None
        # spatial: x += proj(attn(qkv(LN1 x)))          st_transformer.py:85-86
        qkn = self.qkn
        qn = lambda a: (self._lw(l, f"{a}.norm.weight", "p"), self._lw(l, f"{a}.norm.bias", "p"))
        if qkn:
            # norm1 is the identity: the qkv Linear reads bf16(x); then the per-head LayerNorm of q and k, in place
            pl.add("hma_cast_bf16", x, b["xh1"], M * 256)
            pl.gemm_nt(A=b["xh1"], lda=256, a_kind=A_BF16, W=self._lw(l, "spatial_attn.qkv.weight"), ldw=256, M=M, N=768, K=256,
                       epi=EPI_BF16, Cp=b["qkv_s"], ldc=768, bias=qb("spatial_attn"))
            pl.add("hma_qknorm_fwd", b["qkv_s"], 768, b.get("qraw_s"), *qn("spatial_attn"), 1e-5, M, 0, 0)
        elif not have_qkv_s:  # (otherwise the previous block's chain B already wrote this block's spatial qkv)
            if not have_ln1:  # (otherwise the previous block's fused MLP already wrote this block's LN1 output)
                pl.add("hma_ln_fwd", x, b["xh1"], b["rstd1"], M, 1e-5)
            pl.gemm_nt(A=b["xh1"], lda=256, a_kind=A_BF16, W=self.WF["qkv_s"][l].data_ptr(), ldw=256, M=M, N=768, K=256,
                       epi=EPI_BF16, Cp=b["qkv_s"], ldc=768, bias=self.BF["qkv_s"][l].data_ptr())  # norm1 folded into W / bias
        pl.add("hma_attn_spatial_fwd", b["qkv_s"], b["o_s"], b["lse_s"], Fr, SA, self.scale, flops=4.0 * Fr * SA * SA * 256,
               nbytes=(1536.0 + 512 + 32) * Fr * SA)
        qkv_dst = b["qkv_t"] if kv is None else kv["cache"] + kv["row_off"] * 768 * 2
        qkv_grp = (0, 0) if kv is None else kv["c_group"]
        # (T == 16 only: a column tile is 16 frame lanes and the kernel's time is per tile -- at T = 12 its 421 us of compute per 10 240-column
        # launch would serve 3 / 4 of the rows, more than the fusion saves: DESIGN.md section 6.0; dropout needs the modulated form)
        if (self.chain_ab and train and chain_b and kv is None and T == 16 and (use_mod or self.mlp_drop <= 0.0) and not qkn
                and self._use_chain(M, SA)):
            # ---- chain A + causal temporal attention + chain B over columns of 16 frames, ONE launch (csrc/chain.hip): the residual row
            # stays in registers between the chains, the attention is wave-local
            kwm = {}
            seg_lin = (None, 0)
            if use_mod:
                ap = f"decoder.layers.{l}.action_projectors.{domain}"
                seg_lin = (dp_(self.CP[f"lin:{domain}"], l), 8)
                kwm = dict(ss=b["ss"], xhat_m=b["xhm"], xm=b["xm"], rstd_m=b["rstdm"], b_lin=self._p(f"{ap}.linear_out.bias"))
            segs = [(dp_(self.CP["proj_s"], l), 8), seg_lin, (dp_(self.CP["qkv_t"], l), 24),
                    (self.CP["proj_t"][l].data_ptr(), 8), (self.CP["mlp"][l].data_ptr(), 64),
                    (self.CP["qkv_s"][l + 1].data_ptr(), 24) if next_qkv_s is not None else (None, 0)]
            kwq = {}
            if next_qkv_s is not None:
                kwq = dict(xhat1n=next_ln1[0], rstd1n=next_ln1[1], qkv_s=next_qkv_s, b_qkv_s=self.BF["qkv_s"][l + 1].data_ptr())
            pl.chain_ab_fwd(M, next_qkv_s is not None, B=B, SA=SA, segs=segs, o_s=b["o_s"], x=x, b1=self.BF["fc1"][l].data_ptr(),
                            x2b=b["x2b"], qkv_t=b["qkv_t"], o_t=b["o_t"], xhat2=b["xh2"],
                            rstd2=b["rstd2"], attn_scale=self.scale, b_proj_s=pb("spatial_attn"),
                            b_qkv_t=qb("temporal_attn"), b_proj_t=pb("temporal_attn"),
                            b2=self._lw(l, "mlp.fc2.bias", "p") if cfg.mlp_bias else None, **kwm, **kwq, **self._drop_fused(train, l))
            return
        if self._use_chain(M, SA):
            # ---- chain A (csrc/chain.hip): proj + residual -> modulate-LN -> linear_out + residual -> temporal qkv, one launch
            ap = f"decoder.layers.{l}.action_projectors.{domain}"
            segs = [(dp_(self.CP["proj_s"], l), 8)]
            if use_mod:
                segs.append((dp_(self.CP[f"lin:{domain}"], l), 8))
            segs.append((dp_(self.CP["qkv_t"], l), 24))
            sv = dict(xhat=b["xhm"], xm=b["xm"], rstd=b["rstdm"]) if (train and use_mod) else {}
            pl.chain_a_fwd(M, use_mod, train, segs=segs, o=b["o_s"], x=x, qkv=qkv_dst, q_group=qkv_grp,
                           ss=b["ss"] if use_mod else None, b_proj=pb("spatial_attn"),
                           b_lin=self._p(f"{ap}.linear_out.bias") if use_mod else None, b_qkv=qb("temporal_attn"),
                           x_bf16=b["x2b"] if train else None, rows_per_frame=SA, **sv)
        else:
            # (the LayerNorm / modulate prologue of the NEXT sub-block is fused into this projection's epilogue)
            fuse = dict(ln_xhat=b["xhm"], ln_rstd=b["rstdm"], ln_eps=1e-6, ln_ss=b["ss"], ln_xm=b["xm"], ln_rows_per_frame=SA) if use_mod else {}
            pl.gemm_nt(A=b["o_s"], lda=256, a_kind=A_BF16, W=self._lw(l, "spatial_attn.proj.weight"), ldw=256, M=M, N=256, K=256,
                       epi=EPI_RESID, Cp=x, ldc=256, bias=pb("spatial_attn"), C2=None if use_mod else b["x2b"], ldc2=256, **fuse)
            # action modulation: x += Lin(LN0(x) (1 + scale) + shift)   st_mask_git.py:66-76
            if use_mod:
                ap = f"decoder.layers.{l}.action_projectors.{domain}"
                pl.gemm_nt(A=b["xm"], lda=256, a_kind=A_BF16, W=self._wb(f"{ap}.linear_out.weight"), ldw=256, M=M, N=256, K=256,
                           epi=EPI_RESID, Cp=x, ldc=256, bias=self._p(f"{ap}.linear_out.bias"), C2=b["x2b"], ldc2=256)
            # temporal (causal, un-normed input)                st_transformer.py:111
            pl.gemm_nt(A=b["x2b"], lda=256, a_kind=A_BF16, W=self._lw(l, "temporal_attn.qkv.weight"), ldw=256, M=M, N=768,
                       K=256, epi=EPI_BF16, Cp=qkv_dst, ldc=768, c_group=qkv_grp, bias=qb("temporal_attn"))
        if qkn:
            pl.add("hma_qknorm_fwd", qkv_dst, 768, b.get("qraw_t"), *qn("temporal_attn"), 1e-5, M, qkv_grp[0], qkv_grp[1])
        if kv is None:
            pl.add("hma_attn_temporal_fwd", b["qkv_t"], b["o_t"], B, T, SA, self.scale, flops=4.0 * M * T * 256, nbytes=2048.0 * M)
        else:
            pl.add("hma_attn_temporal_cached", kv["cache"], b["o_t"], B, T, kv["t_query"], kv["T_cache"], SA, self.scale)
        if chain_b:
            # ---- chain B (csrc/chain.hip, passes that save nothing): proj_t + residual -> norm2 -> MLP + residual -> the next
            # block's norm1 + spatial qkv, one launch
            segs = [(self.CP["proj_t"][l].data_ptr(), 8), (self.CP["mlp"][l].data_ptr(), 64)]
            if next_qkv_s is not None:
                segs.append((self.CP["qkv_s"][l + 1].data_ptr(), 24))
            sv = {}
            if train:  # the two LayerNorm outputs are saved for the backward (norm2 of this block, norm1 of the next one)
                sv = dict(xhat2=b["xh2"], rstd2=b["rstd2"])
                if next_qkv_s is not None:
                    sv.update(xhat1n=next_ln1[0], rstd1n=next_ln1[1])
            pl.chain_b_fwd(M, next_qkv_s is not None, segs=segs, o=b["o_t"], x=x, b_proj=pb("temporal_attn"),
                           b1=self.BF["fc1"][l].data_ptr(), b2=self._lw(l, "mlp.fc2.bias", "p") if cfg.mlp_bias else None,
                           b_qkv=self.BF["qkv_s"][l + 1].data_ptr() if next_qkv_s is not None else None, qkv=next_qkv_s, **sv,
                           **self._drop_fused(train, l))
            return
        if qkn:  # norm2 is the identity: the MLP reads bf16 of the new x (the residual epilogue's second output)
            pl.gemm_nt(A=b["o_t"], lda=256, a_kind=A_BF16, W=self._lw(l, "temporal_attn.proj.weight"), ldw=256, M=M, N=256, K=256,
                       epi=EPI_RESID, Cp=x, ldc=256, bias=pb("temporal_attn"), C2=b["xh2"], ldc2=256)
        else:
            pl.gemm_nt(A=b["o_t"], lda=256, a_kind=A_BF16, W=self._lw(l, "temporal_attn.proj.weight"), ldw=256, M=M, N=256, K=256,
                       epi=EPI_RESID, Cp=x, ldc=256, bias=pb("temporal_attn"), ln_xhat=b["xh2"], ln_rstd=b["rstd2"], ln_eps=1e-5)
        # MLP (its LayerNorm: fused above)                   st_transformer.py:112
        if fused:
            pl.mlp_fwd(M, xhat=b["xh2"], x=x, w1p=self.MP["w1p"][l].data_ptr(), w2p=self.MP["w2p"][l].data_ptr(),
                       b1=self.BF["fc1"][l].data_ptr(), b2=self._lw(l, "mlp.fc2.bias", "p") if cfg.mlp_bias else None,
                       ln_xhat=ln_next[0] if ln_next else None, ln_rstd=ln_next[1] if ln_next else None, ln_eps=1e-5)
            return
        pl.gemm_nt(A=b["xh2"], lda=256, a_kind=A_BF16, W=self._lw(l, "mlp.fc1.weight") if qkn else self.WF["fc1"][l].data_ptr(), ldw=256,
                   M=M, N=1024, K=256, epi=EPI_GELU2, Cp=b["u"], ldc=1024, C2=b["hg"], ldc2=1024,
                   bias=(self._lw(l, "mlp.fc1.bias", "p") if cfg.mlp_bias else None) if qkn else self.BF["fc1"][l].data_ptr(),
                   **self._drop(train, l, 0))  # norm2 folded into W / bias
        pl.gemm_nt(A=b["hg"], lda=1024, a_kind=A_BF16, W=self._lw(l, "mlp.fc2.weight"), ldw=1024, M=M, N=256, K=1024,
                   epi=EPI_RESID, Cp=x, ldc=256, bias=self._lw(l, "mlp.fc2.bias", "p") if cfg.mlp_bias else None,
                   **self._drop(train, l, 1))

    def bump_dropout(self) -> None:
        """New dropout masks for the next training forward + backward (a device-side seed: plans / graphs stay valid)."""
        if float(getattr(self.cfg, "mlp_drop", 0.0) or 0.0) > 0.0:
            self._drop_counter += 1
            self.drop_seed.fill_((self._drop_counter * 2654435761) % (1 << 31))

    def _drop(self, train: bool, l: int, which: int) -> dict:
        """Dropout arguments of the MLP's two nn.Dropout sites (st_transformer.py:24-27): training only, mlp_drop > 0."""
        p = float(getattr(self.cfg, "mlp_drop", 0.0) or 0.0)
        if not train or p <= 0.0:
            return {}
        return dict(drop_p=p, drop_salt=2 * l + which, drop_seed=self.drop_seed.data_ptr())

    def _drop_fused(self, train: bool, l: int) -> dict:
        """The same two sites for the kernels that hold both (chain B forward, hma_mlp_bwd): salts 2 l (activation), 2 l + 1 (output)."""
        if not train or self.mlp_drop <= 0.0:
            return {}
        return dict(drop_p=self.mlp_drop, drop_salt=2 * l, drop_seed=self.drop_seed.data_ptr())

    def _forward_plan(self, B, T, S, A, train, domain, embed=True, l0=0, l1=None, readout=True, kv_cache=None,
                      T_cache=0) -> Plan:
        l1 = self.cfg.num_layers if l1 is None else l1
        # (T_cache is part of the key: a re-allocated cache of another length can land on the old one's address)
        key = ("fwd", B, T, S, A, train, domain, embed, l0, l1, readout, None if kv_cache is None else (kv_cache.data_ptr(), T_cache),
               self._use_fused(B * T * (S + A), train, S + A), self._use_chain(B * T * (S + A), S + A), self.chain_ab)
        if key in self._plans:
            return self._plans[key]
        cfg, ws = self.cfg, self._ws
        L = cfg.num_layers
        SA, M, Mi, Fr = S + A, B * T * (S + A), B * T * S, B * T
        fused = self._use_fused(M, train, SA)
        pl = Plan(self._dev_index)
        dp = lambda t, l=0, per=0: t.data_ptr() + (l * per) * t.element_size()
        sl = (lambda l: l) if train else (lambda l: 0)

        def kv_for_layer(l):
            if kv_cache is None:
                return None
            return {"cache": kv_cache[l].data_ptr(), "row_off": 0, "c_group": (T * (S + A), T_cache * (S + A)), "t_query": -1,
                    "T_cache": T_cache}

        if A > 0 and domain is not None:  # (domain None with A > 0: jointly_predict_actions' policy mode -- mask tokens, no stem, no modulation)
            dom = domain
            am = f"action_mlp.{dom}.model"
            if embed:
              pl.add("hma_action_stem_fwd", ws["actions"].data_ptr(), self.buffers[dom][0].data_ptr(),
                   self.buffers[dom][1].data_ptr(), self.action_dims[dom], self._p(f"{am}.0.weight"),
                   self._p(f"{am}.0.bias"), self._p(f"{am}.1.weight"), self._p(f"{am}.1.bias"), self._p(f"{am}.3.weight"),
                   self._p(f"{am}.3.bias"), ws["an"].data_ptr(), ws["sxhat"].data_ptr(), ws["srstd"].data_ptr(),
                   ws["sh"].data_ptr(), ws["a_emb"].data_ptr(), Fr, self.d_actions[dom], self._skip_norm)
              pl.mark("post_stem")
            if self.modulate:
                ap = f"decoder.layers.0.action_projectors.{dom}"
                pl.gemm_nt(A=ws["a_emb"].data_ptr(), lda=256, a_kind=A_F32, W=self._wb(f"{ap}.adaLN_modulation.0.weight"),
                           ldw=256, M=Fr, N=256, K=256, epi=EPI_SILU2, Cp=ws["ada_pre"].data_ptr(), ldc=256,
                           bias=self._p(f"{ap}.adaLN_modulation.0.bias"), C2=ws["ada_act"].data_ptr(), ldc2=256, batch=L,
                           sA=0, sW=self.layout.dom_layer_stride, sBias=self.layout.dom_layer_stride, sC=Fr * 256, sC2=Fr * 256)
                pl.gemm_nt(A=ws["ada_act"].data_ptr(), lda=256, a_kind=A_BF16, W=self._wb(f"{ap}.adaLN_modulation.2.weight"),
                           ldw=256, M=Fr, N=512, K=256, epi=EPI_F32, Cp=ws["ss"].data_ptr(), ldc=512,
                           bias=self._p(f"{ap}.adaLN_modulation.2.bias"), batch=L, sA=Fr * 256, sW=self.layout.dom_layer_stride,
                           sBias=self.layout.dom_layer_stride, sC=Fr * 512)
        if embed:
            pl.add("hma_embed_fwd", ws["ids"].data_ptr(), self._p("token_embed.factored_embeds.0.weight"),
                   self._p("token_embed.factored_embeds.1.weight"), self._p("token_embed.mask_token_embed"),
                   self._p("pos_embed_TSC"), (ws["a_tok"] if self.jpa else ws["a_emb"]).data_ptr() if A > 0 else None,
                   ws["x"].data_ptr(), B, T, S, A, cfg.S + cfg.action_token_size, cfg.factored_vocab_size, cfg.image_vocab_size)
        x = ws["x"].data_ptr()
        for l in range(l0, l1):
            s = sl(l)
            names = ("xh1", "rstd1", "qkv_s", "o_s", "lse_s", "x2b", "qkv_t", "o_t", "xh2", "rstd2") + (() if fused else ("u", "hg"))
            if self.qkn and train:
                names = names + ("qraw_s", "qraw_t")
            bufs = {k: dp(ws[k], s, ws[k][0].numel()) for k in names}
            if A > 0 and self.modulate and domain is not None:
                bufs.update({k: dp(ws[k], s, ws[k][0].numel()) for k in ("xhm", "xm", "rstdm")})
                bufs["ss"] = dp(ws["ss"], l, Fr * 512)
            ln_next = None
            if fused and l + 1 < l1:  # the MLP kernel also emits the next block's LN1 output
                s1 = sl(l + 1)
                ln_next = (dp(ws["xh1"], s1, ws["xh1"][0].numel()), dp(ws["rstd1"], s1, ws["rstd1"][0].numel()))
            # chain B: every pass that saves nothing, and training passes that would take the fused MLP block (no dropout)
            cb = self.chain_b_ok and self._use_chain(M, SA) and ((not train) or (fused and self.chain_b_train))
            nxt = None
            if cb and l + 1 < l1:
                s1 = sl(l + 1)
                nxt = (dp(ws["qkv_s"], s1, ws["qkv_s"][0].numel()), dp(ws["xh1"], s1, ws["xh1"][0].numel()),
                       dp(ws["rstd1"], s1, ws["rstd1"][0].numel()))
            self._emit_layer(pl, l, x, bufs, M, Fr, B, T, SA, A > 0 and self.modulate and domain is not None, domain, kv=kv_for_layer(l), train=train,
                             have_ln1=fused and l > l0, ln_next=ln_next, fused=fused, have_qkv_s=cb and l > l0, chain_b=cb,
                             next_qkv_s=nxt[0] if nxt else None, next_ln1=(nxt[1], nxt[2]) if nxt else None)
        # readout on the image tokens only                      st_mask_git.py:681-683
        if readout:
          pl.gemm_nt(A=x, lda=256, a_kind=A_F32, a_group=(S, SA), W=self._wb("out_x_proj.weight"), ldw=256, M=Mi, N=1024, K=256,
                   epi=EPI_F32, Cp=ws["logits"].data_ptr(), ldc=1024, bias=self._p("out_x_proj.bias"))
        self._plans[key] = pl
        return pl

    def _use_fused_ce(self, B: int, T: int, S: int) -> bool:
        return self.fused_ce and self.use_chain and (B * T * S) % 16 == 0

    def _loss_plan(self, B, T, S, with_grad: bool, fused: bool = False, A: int = 0) -> Plan:
        """Masked-row count + cross-entropy (+ dlogits) from ws["logits"]; `fused`: readout and cross-entropy from ws["x"] in one launch
        (hma_readout_ce: the logits are never written; the scale is fixed when the plan is built, hence part of the key)."""
        key = ("loss", B, T, S, with_grad, fused, A if fused else 0, self.grad_scale.value if fused else None)
        if key in self._plans:
            return self._plans[key]
        ws, cfg = self._ws, self.cfg
        pl = Plan(self._dev_index)
        pl.add("hma_count_masked", ws["ids"].data_ptr(), ws["stats"].data_ptr(), B, T, S, cfg.image_vocab_size)
        if fused:
            pl.readout_ce(B * T * S, segs=[(self.CP["out"].data_ptr(), 32)], x=ws["x"].data_ptr(), bias=self._p("out_x_proj.bias"),
                          input_ids=ws["ids"].data_ptr(), labels=ws["labels"].data_ptr(), stats=ws["stats"].data_ptr(),
                          dlogits=ws["dlogits"].data_ptr() if with_grad else None, grad_scale_dev=self.gscale.data_ptr(),
                          grad_scale=self.grad_scale.value, S=S, SA=S + A, T=T, mask_id=cfg.image_vocab_size, label_smoothing=0.01)
        else:
            pl.add("hma_ce_fwd_bwd", ws["logits"].data_ptr(), ws["ids"].data_ptr(), ws["labels"].data_ptr(), ws["stats"].data_ptr(),
                   ws["dlogits"].data_ptr() if with_grad else None, self.gscale.data_ptr(), self.grad_scale, B, T, S,
                   cfg.image_vocab_size, 0.01)
        self._plans[key] = pl
        return pl

    def _backward_plan(self, B, T, S, A, domain) -> Plan:
        key = ("bwd", B, T, S, A, domain, self._use_fused(B * T * (S + A), True, S + A), self._use_chain(B * T * (S + A), S + A), self.ada_group,
               self.chain_s, self.attn_hb, self.wgrad_multi, self.chain_t, self.wgrad_layers)
        if key in self._plans:
            return self._plans[key]
        cfg, ws = self.cfg, self._ws
        L = cfg.num_layers
        SA, M, Mi, Fr = S + A, B * T * (S + A), B * T * S, B * T
        pl = Plan(self._dev_index)
        dp = lambda t, l=0, per=0: t.data_ptr() + (l * per) * t.element_size()
        x, dx, t256, dqkv = ws["x"].data_ptr(), ws["dx"].data_ptr(), ws["t256"].data_ptr(), ws["dqkv"].data_ptr()
        dxb = ws["dxb"].data_ptr()
        # chain S leaves norm1's dgamma / dbeta to the qkv weight gradient's reduction, which exists on the LDS-DMA path only: that path
        # needs whole 32-row stages and at least two 64-row slabs (a split-M launch with a workspace); smaller passes keep hma_ln_bwd
        chain_s_ok = self.chain_s and (not self.qkn) and self._use_chain(M, SA) and M % 32 == 0 and M >= 128
        multi = self.wgrad_multi and "dqkv_s" in ws and self._use_fused(M, True, SA) and chain_s_ok
        dqkv_s = ws["dqkv_s"].data_ptr() if multi else dqkv
        # `pair`: the weight gradients of two consecutive blocks in one launch -- the first block's operands (its three bf16(dx) states,
        # both dqkv, gelu(u) / dU, the dropout and modulate copies) stay alive in the "_b" set while the second block runs
        pair = multi and self.wgrad_layers > 1
        ring_names = ("dxb", "dxb2", "dxb3", "dxb4") + (("dxb5", "dxb6", "dxb7", "dxb8") if pair else ())
        ring = [ws[k].data_ptr() for k in ring_names] if multi else []
        nxt = lambda cur: ring[(ring.index(cur) + 1) % len(ring)]  # the next bf16(dx) buffer (deferred weight gradients still read the earlier ones)
        pend: list = []   # weight gradients waiting for their launch (`multi`)
        pend_layers = 0
        alt = lambda name, odd: ws[name + "_b" if (pair and odd) else name].data_ptr()
        # readout
        pl.gemm_tn(dY=ws["dlogits"].data_ptr(), ldy=1024, y_kind=A_BF16, A=x, lda=256, a_kind=A_F32, a_group=(S, SA), M=Mi,
                   N=1024, K=256, dW=self._g("out_x_proj.weight"), lddw=256, dBias=self._g("out_x_proj.bias"))
        pl.gemm_nt(A=ws["dlogits"].data_ptr(), lda=1024, a_kind=A_BF16, W=self.WT["out"].data_ptr(), ldw=1024, M=Mi, N=256,
                   K=1024, epi=EPI_F32, Cp=dx, ldc=256, c_group=(S, SA))
        # Every producer of dx below also writes its bf16 rounding (dxb): that is what autocast hands a linear's
        # backward, and the GEMMs that consume it read half the bytes (the fp32 dx stays the accumulator).
        pl.mark("post_readout")  # (STMAR drives its own readout and enters the plan here)
        pl.add("hma_cast_bf16", dx, dxb, M * 256)
        use_mod = A > 0 and self.modulate
        if use_mod and self._use_chain(M, SA):
            pl.add("hma_zero_f32", ws["dss"].data_ptr(), ws["dss"].numel())  # the backward chains add the frames' sums with atomics
        ada_g, ada_hi = max(1, min(int(self.ada_group), L)), L
        pl.dxb_in = {}  # layer -> the bf16(dx) buffer its backward reads first (an entry into the middle of the plan must fill it)
        for l in reversed(range(L)):
            pl.dxb_in[l] = dxb
            odd = bool((L - 1 - l) & 1)  # which buffer set this block's deferred operands use
            if multi:
                dqkv, dqkv_s = alt("dqkv", odd), alt("dqkv_s", odd)
            xh1, rstd1 = dp(ws["xh1"], l, M * 256), dp(ws["rstd1"], l, M)
            qkv_s, o_s, lse_s = dp(ws["qkv_s"], l, M * 768), dp(ws["o_s"], l, M * 256), dp(ws["lse_s"], l, M * 8)
            x2b = dp(ws["x2b"], l, M * 256)
            qkv_t, o_t = dp(ws["qkv_t"], l, M * 768), dp(ws["o_t"], l, M * 256)
            xh2, rstd2 = dp(ws["xh2"], l, M * 256), dp(ws["rstd2"], l, M)
            gw = lambda suffix: self._lw(l, suffix, "g")
            gb = lambda suffix, on=True: self._lw(l, suffix, "g") if on else None
            wt = lambda k: dp(self.WT[k], l, self.WT[k][0].numel())
            if self._use_fused(M, True, SA):
                # ---- MLP, fused: u recomputed from xhat2, dU / gelu(u) written once for the two weight gradients, the
                # LayerNorm backward applied in the same kernel (its dgamma / dbeta come out of the fc1 weight-gradient
                # reduction).  The new bf16 copy of dx goes to the other dxb buffer: fc2's weight gradient still reads the old.
                dxb_new = nxt(dxb) if multi else (ws["dxb2"].data_ptr() if dxb == ws["dxb"].data_ptr() else ws["dxb"].data_ptr())
                hg, du = (alt("hg1", odd), alt("du1", odd)) if multi else (ws["hg1"].data_ptr(), ws["du1"].data_ptr())
                dkw = self._drop_fused(True, l)
                dy2 = dxb  # dY of the fc2 weight gradient: behind the output Dropout when there is one (written by hma_mlp_bwd)
                if dkw:
                    dy2 = dkw["dy_drop"] = alt("dxm", odd) if multi else ws["dxm"].data_ptr()
                pl.mlp_bwd(M, xhat=xh2, rstd=rstd2, dy=dxb, dx=dx, dx_bf16=dxb_new, w1p=dp(self.MP["w1p"], l, 512 * 512),
                           w2tp=dp(self.MP["w2tp"], l, 512 * 512), w1tp=dp(self.MP["w1tp"], l, 512 * 512),
                           b1=self.BF["fc1"][l].data_ptr(), hg=hg, du=du, **dkw)
                kw_mlp = (dict(dY=dy2, ldy=256, y_kind=A_BF16, A=hg, lda=1024, a_kind=A_BF16_FRAG32, M=M, N=256, K=1024,
                               dW=gw("mlp.fc2.weight"), lddw=1024, dBias=gb("mlp.fc2.bias", cfg.mlp_bias)),
                          dict(dY=du, ldy=1024, y_kind=A_BF16_FRAG32, A=xh2, lda=256, a_kind=A_BF16_AFFINE,
                               gamma=self._lw(l, "norm2.weight", "p"), beta=self._lw(l, "norm2.bias", "p"), M=M, N=1024,
                               K=256, dW=gw("mlp.fc1.weight"), lddw=256, dBias=gb("mlp.fc1.bias", cfg.mlp_bias),
                               w_master=self._lw(l, "mlp.fc1.weight", "p"), dgamma=gw("norm2.weight"),
                               dbeta=gw("norm2.bias")))
                if multi:
                    pend.extend(kw_mlp)
                else:
                    pl.gemm_tn_pair(*kw_mlp)
                dxb = dxb_new
            else:
                u, hg = dp(ws["u"], l, M * 1024), dp(ws["hg"], l, M * 1024)
                # ---- MLP (with mlp_drop: the gradient first passes the Dropout behind fc2, then the one behind the GELU)
                dmlp = dxb
                if self._drop(True, l, 1):
                    dmlp = ws["dxm"].data_ptr()
                    pl.add("hma_dropout_bf16", dx, dmlp, M, 256, float(cfg.mlp_drop), self.drop_seed.data_ptr(), 2 * l + 1)
                pl.gemm_nt(A=dmlp, lda=256, a_kind=A_BF16, W=wt("fc2"), ldw=256, M=M, N=1024, K=256, epi=EPI_DGELU, Cp=u, ldc=1024,
                           U=u, ldu=1024, **self._drop(True, l, 0))  # dU overwrites u in place
                # the two MLP weight gradients in one launch (dmlp / hg and dU / xhat2 are all live here)
                aff2 = dict(a_kind=A_BF16) if self.qkn else dict(a_kind=A_BF16_AFFINE, gamma=self._lw(l, "norm2.weight", "p"),
                                                                 beta=self._lw(l, "norm2.bias", "p"))
                pl.gemm_tn_pair(dict(dY=dmlp, ldy=256, y_kind=A_BF16, A=hg, lda=1024, a_kind=A_BF16, M=M, N=256, K=1024,
                                     dW=gw("mlp.fc2.weight"), lddw=1024, dBias=gb("mlp.fc2.bias", cfg.mlp_bias)),
                                dict(dY=u, ldy=1024, y_kind=A_BF16, A=xh2, lda=256, M=M, N=1024,
                                     K=256, dW=gw("mlp.fc1.weight"), lddw=256, dBias=gb("mlp.fc1.bias", cfg.mlp_bias), **aff2))
                if self.qkn:  # norm2 is the identity: the fc1 input gradient goes straight into the residual gradient
                    pl.gemm_nt(A=u, lda=1024, a_kind=A_BF16, W=wt("fc1"), ldw=1024, M=M, N=256, K=1024, epi=EPI_RESID, Cp=dx, ldc=256,
                               C2=dxb, ldc2=256)
                else:
                    pl.gemm_nt(A=u, lda=1024, a_kind=A_BF16, W=wt("fc1"), ldw=1024, M=M, N=256, K=1024, epi=EPI_BF16, Cp=t256, ldc=256)
                    pl.add("hma_ln_bwd", t256, xh2, rstd2, self._lw(l, "norm2.weight", "p"), dx, gw("norm2.weight"), gw("norm2.bias"), M,
                           dxb)
            # ---- temporal attention
            if self.chain_t and 8 < T <= 16 and not self.qkn and self._use_chain(M, SA):  # (a column tile is 16 frame lanes: at T <= 8 half of them would idle)
                # chain T backward (csrc/chain.hip): the projection's input gradient and the attention backward of a column, one launch
                pl.chain_t_bwd(B, SA, segs=[(self.CP["proj_t_T"][l].data_ptr(), 8)], dy_bf16=dxb, qkv=qkv_t, dqkv=dqkv, attn_scale=self.scale, T=T)
            else:
                pl.gemm_nt(A=dxb, lda=256, a_kind=A_BF16, W=wt("proj_t"), ldw=256, M=M, N=256, K=256, epi=EPI_BF16, Cp=t256, ldc=256)
                pl.add("hma_attn_temporal_bwd", qkv_t, o_t, t256, dqkv, B, T, SA, self.scale, flops=10.0 * M * T * 256, nbytes=3584.0 * M)
            if self.qkn:  # through the per-head LayerNorm of q and k (dqkv in place; the affine's gradients by atomics)
                pl.add("hma_qknorm_bwd", dqkv, 768, dp(ws["qraw_t"], l, M * 512), self._lw(l, "temporal_attn.norm.weight", "p"), 1e-5,
                       gw("temporal_attn.norm.weight"), gw("temporal_attn.norm.bias"), M)
            # projection and qkv weight gradients in one launch (dxb is not updated before the dqkv dgrad below)
            kw_t = (dict(dY=dxb, ldy=256, y_kind=A_BF16, A=o_t, lda=256, a_kind=A_BF16, M=M, N=256, K=256,
                         dW=gw("temporal_attn.proj.weight"), lddw=256, dBias=gb("temporal_attn.proj.bias", cfg.proj_bias)),
                    dict(dY=dqkv, ldy=768, y_kind=A_BF16, A=x2b, lda=256, a_kind=A_BF16, M=M, N=768, K=256,
                         dW=gw("temporal_attn.qkv.weight"), lddw=256, dBias=gb("temporal_attn.qkv.bias", cfg.qkv_bias)))
            if multi:
                pend.extend(kw_t)
            else:
                pl.gemm_tn_pair(*kw_t)
            if self._use_chain(M, SA):
                # ---- chain A backward (csrc/chain.hip): temporal qkv dgrad + residual -> linear_out dgrad -> modulate-LN backward
                # -> spatial projection dgrad, one launch; dss accumulates by atomics (zeroed at the top of the plan)
                segs = [(self.CP["qkv_t_T"][l].data_ptr(), 24)]
                kwm = {}
                if use_mod:
                    xhm, xm, rstdm = dp(ws["xhm"], l, M * 256), dp(ws["xm"], l, M * 256), dp(ws["rstdm"], l, M)
                    ap = f"decoder.layers.{l}.action_projectors.{domain}"
                    segs.append((self.CP[f"lin_T:{domain}"][l].data_ptr(), 8))
                    dx2b = alt("dx2b", odd) if multi else ws["dx2b"].data_ptr()
                    kwm = dict(xhat=xhm, rstd=rstdm, ss=dp(ws["ss"], l, Fr * 512), dx2_bf16=dx2b, dss=dp(ws["dss"], l, Fr * 512))
                segs.append((self.CP["proj_s_T"][l].data_ptr(), 8))
                if multi:
                    dxb = nxt(dxb)  # (the deferred temporal projection gradient still reads the previous buffer)
                pl.chain_a_bwd(M, use_mod, segs=segs, dqkv=dqkv, dx=dx, dx1_bf16=dxb, d_o=t256, rows_per_frame=SA, **kwm)
                if use_mod:
                    kw_lin = dict(dY=dx2b, ldy=256, y_kind=A_BF16, A=xm, lda=256, a_kind=A_BF16, M=M, N=256, K=256,
                                  dW=self._g(f"{ap}.linear_out.weight"), lddw=256, dBias=self._g(f"{ap}.linear_out.bias"))
                    if multi:
                        pend.append(kw_lin)
                    else:
                        pl.gemm_tn(**kw_lin)
            else:
                pl.gemm_nt(A=dqkv, lda=768, a_kind=A_BF16, W=wt("qkv_t"), ldw=768, M=M, N=256, K=768, epi=EPI_RESID, Cp=dx, ldc=256,
                           C2=dxb, ldc2=256)
                # ---- action modulation
                if use_mod:
                    xhm, xm, rstdm = dp(ws["xhm"], l, M * 256), dp(ws["xm"], l, M * 256), dp(ws["rstdm"], l, M)
                    ap = f"decoder.layers.{l}.action_projectors.{domain}"
                    pl.gemm_tn(dY=dxb, ldy=256, y_kind=A_BF16, A=xm, lda=256, a_kind=A_BF16, M=M, N=256, K=256,
                               dW=self._g(f"{ap}.linear_out.weight"), lddw=256, dBias=self._g(f"{ap}.linear_out.bias"))
                    pl.gemm_nt(A=dxb, lda=256, a_kind=A_BF16, W=dp(self.WT[f"lin:{domain}"], l, 256 * 256), ldw=256, M=M, N=256, K=256,
                               epi=EPI_BF16, Cp=t256, ldc=256)
                    pl.add("hma_modln_bwd", t256, xhm, rstdm, dp(ws["ss"], l, Fr * 512), dx, dp(ws["dss"], l, Fr * 512), Fr, SA, dxb)
                # ---- spatial attention
                pl.gemm_nt(A=dxb, lda=256, a_kind=A_BF16, W=wt("proj_s"), ldw=256, M=M, N=256, K=256, epi=EPI_BF16, Cp=t256, ldc=256)
            # (on the chain path the attention backward writes dqkv HEAD-BLOCKED -- whole contiguous 2 KB tiles instead of 32-byte pieces
            # of 32 cache lines per store instruction -- and its two consumers, the qkv weight gradient and chain S, read that order)
            chain_s = chain_s_ok
            hb = chain_s and self.attn_hb and SA % 32 == 0
            pl.add("hma_attn_spatial_bwd_blocked" if hb else "hma_attn_spatial_bwd", qkv_s, o_s, t256, lse_s, ws["delta"].data_ptr(), dqkv_s, Fr, SA, self.scale,
                   flops=10.0 * Fr * SA * SA * 256,  # 5 products of 2 n^2 d per head (the recomputed S is not counted)
                   nbytes=(1536.0 + 512 + 512 + 32 + 1536) * Fr * SA)  # qkv, o, dO, lse read once; dqkv written
            # projection and qkv weight gradients in one launch (dxb is next updated by the LayerNorm backward below)
            if self.qkn:
                pl.add("hma_qknorm_bwd", dqkv_s, 768, dp(ws["qraw_s"], l, M * 512), self._lw(l, "spatial_attn.norm.weight", "p"), 1e-5,
                       gw("spatial_attn.norm.weight"), gw("spatial_attn.norm.bias"), M)
            # (on the chain path norm1's dgamma / dbeta come out of the qkv weight gradient's reduction, as norm2's do out of fc1's)
            aff1 = dict(a_kind=A_BF16) if self.qkn else dict(a_kind=A_BF16_AFFINE, gamma=self._lw(l, "norm1.weight", "p"),
                                                             beta=self._lw(l, "norm1.bias", "p"))
            if chain_s:
                aff1.update(w_master=self._lw(l, "spatial_attn.qkv.weight", "p"), dgamma=gw("norm1.weight"), dbeta=gw("norm1.bias"))
            kw_s = (dict(dY=dxb, ldy=256, y_kind=A_BF16, A=o_s, lda=256, a_kind=A_BF16, M=M, N=256, K=256,
                         dW=gw("spatial_attn.proj.weight"), lddw=256, dBias=gb("spatial_attn.proj.bias", cfg.proj_bias)),
                    dict(dY=dqkv_s, ldy=768, y_kind=A_BF16_HEADBLK if hb else A_BF16, y_group=(SA, 0) if hb else (0, 0), A=xh1, lda=256,
                         M=M, N=768, K=256, dW=gw("spatial_attn.qkv.weight"), lddw=256,
                         dBias=gb("spatial_attn.qkv.bias", cfg.qkv_bias), **aff1))
            if multi:
                pend.extend(kw_s)
            else:
                pl.gemm_tn_pair(*kw_s)
            if self.qkn:  # norm1 is the identity
                pl.gemm_nt(A=dqkv_s, lda=768, a_kind=A_BF16, W=wt("qkv_s"), ldw=768, M=M, N=256, K=768, epi=EPI_RESID, Cp=dx, ldc=256,
                           C2=dxb, ldc2=256)
            elif chain_s:
                # ---- chain S backward (csrc/chain.hip): spatial qkv dgrad -> norm1 backward -> residual, one launch
                if multi:
                    dxb = nxt(dxb)  # (the deferred spatial projection gradient still reads the previous buffer)
                pl.chain_s_bwd(M, segs=[(self.CP["qkv_s_T"][l].data_ptr(), 24)], dqkv=dqkv_s, dx=dx, xhat=xh1, rstd=rstd1, dx_bf16=dxb,
                               hb_rows=SA if hb else 0)
                if multi:
                    # ---- the seven (six without the modulation) weight gradients of this block and, `pair`, of the one before it: one
                    # launch, one reduction.  Pairs never straddle a gradient bucket (layer marks at even distances from the top)
                    # ... nor a group of `ada_group` layers: a data-parallel driver sets that to its layers per bucket, and a bucket's
                    # gradients have to be complete at its layer mark (the all-reduce of the bucket is issued there)
                    pend_layers += 1
                    if pend_layers == (2 if pair else 1) or l == 0 or (L - l) % ada_g == 0:
                        pl.gemm_tn_multi(pend)
                        pend, pend_layers = [], 0
            else:
                pl.gemm_nt(A=dqkv_s, lda=768, a_kind=A_BF16, W=wt("qkv_s"), ldw=768, M=M, N=256, K=768, epi=EPI_BF16, Cp=t256, ldc=256)
                pl.add("hma_ln_bwd", t256, xh1, rstd1, self._lw(l, "norm1.weight", "p"), dx, gw("norm1.weight"), gw("norm1.bias"), M,
                       dxb)
            # ---- the adaLN stacks of a finished group of layers (batched over the group): their weight gradients are final with the
            # group, so a domain's slice of a gradient bucket can be all-reduced with it; d a_emb accumulates (atomics) for the stem
            if use_mod and ((L - l) % ada_g == 0 or l == 0):
                n = ada_hi - l
                ap = f"decoder.layers.{l}.action_projectors.{domain}"
                dsl = self.layout.dom_layer_stride
                o = lambda name, per: dp(ws[name], l, per)
                pl.gemm_tn(dY=o("dss", Fr * 512), ldy=512, y_kind=A_F32, A=o("ada_act", Fr * 256), lda=256, a_kind=A_BF16,
                           M=Fr, N=512, K=256, dW=self._g(f"{ap}.adaLN_modulation.2.weight"), lddw=256,
                           dBias=self._g(f"{ap}.adaLN_modulation.2.bias"), batch=n, sY=Fr * 512, sA=Fr * 256, sdW=dsl, sdBias=dsl)
                pl.gemm_nt(A=o("dss", Fr * 512), lda=512, a_kind=A_F32, W=dp(self.WT[f"ada2:{domain}"], l, 256 * 512), ldw=512, M=Fr,
                           N=256, K=512, epi=EPI_DSILU, Cp=o("dpre", Fr * 256), ldc=256, U=o("ada_pre", Fr * 256), ldu=256,
                           batch=n, sA=Fr * 512, sW=256 * 512, sC=Fr * 256, sU=Fr * 256)
                pl.gemm_tn(dY=o("dpre", Fr * 256), ldy=256, y_kind=A_BF16, A=ws["a_emb"].data_ptr(), lda=256, a_kind=A_F32,
                           M=Fr, N=256, K=256, dW=self._g(f"{ap}.adaLN_modulation.0.weight"), lddw=256,
                           dBias=self._g(f"{ap}.adaLN_modulation.0.bias"), batch=n, sY=Fr * 256, sA=0, sdW=dsl, sdBias=dsl)
                pl.gemm_nt(A=o("dpre", Fr * 256), lda=256, a_kind=A_BF16, W=dp(self.WT[f"ada0:{domain}"], l, 256 * 256), ldw=256, M=Fr,
                           N=256, K=256, epi=EPI_ATOMIC_F32, Cp=ws["da_emb"].data_ptr(), ldc=256, batch=n, sA=Fr * 256,
                           sW=256 * 256, sC=0)
                ada_hi = l
            pl.mark(f"layer{l}")
        # ---- embedding, action stem
        pl.mark("embed")
        pl.add("hma_embed_bwd", ws["ids"].data_ptr(), dx, self._g("token_embed.factored_embeds.0.weight"),
               self._g("token_embed.factored_embeds.1.weight"), self._g("token_embed.mask_token_embed"), self._g("pos_embed_TSC"),
               (ws["da_tok"] if self.jpa else ws["da_emb"]).data_ptr() if A > 0 else None, B, T, S, A,
               cfg.S + cfg.action_token_size, cfg.factored_vocab_size, cfg.image_vocab_size)
        pl.mark("post_embed")
        if A > 0:
            dom = domain
            am = f"action_mlp.{dom}.model"
            pl.add("hma_action_stem_bwd", ws["da_emb"].data_ptr(), ws["an"].data_ptr(), ws["sxhat"].data_ptr(),
                   ws["srstd"].data_ptr(), ws["sh"].data_ptr(), self._p(f"{am}.1.weight"), self._p(f"{am}.3.weight"),
                   self._g(f"{am}.0.weight"), self._g(f"{am}.0.bias"), self._g(f"{am}.1.weight"), self._g(f"{am}.1.bias"),
                   self._g(f"{am}.3.weight"), self._g(f"{am}.3.bias"), ws["stem_scratch"].data_ptr(), Fr, self.d_actions[dom])
        pl.mark("end")
        self._plans[key] = pl
        return pl

    # ------------------------------------------------------------------------------ public ops
    _skip_norm = 0

    def stats_buffers(self) -> Dict[str, torch.Tensor]:
        return self._ws

    def forward(self, ids_BTS: torch.Tensor, labels: Optional[torch.Tensor], actions: Optional[torch.Tensor],
                domain: Optional[str], train: bool, skip_normalization: bool = False, loss_grad: bool = False,
                need_logits: bool = True, action_mask: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """Embeds, runs the trunk and the readout; with `labels` also the loss (and, with `loss_grad`, dlogits
        scaled by `self.grad_scale * self.gscale`).  `need_logits=False` (a training step that only wants the loss and the
        gradient): readout and cross-entropy run as one launch and ws["logits"] is NOT written."""
        B, T, S = ids_BTS.shape
        A = self.cfg.action_token_size if (actions is not None and "concat" in self.cfg.action_network) else 0
        # jointly_predict_actions without input actions (st_mask_git.py:663-666, "as in policies"): every action token is a mask token,
        # the decoder runs unconditioned, the domain only selects the action read-out.  Inference only (the reference's forward has no
        # action labels in this case).
        policy = self.jpa and actions is None and domain in self.d_actions and not train and "concat" in self.cfg.action_network
        if policy:
            A = self.cfg.action_token_size
        if actions is not None and A == 0:
            raise NotImplementedError("only the 'concat+modulate' action network is built")
        if actions is not None and domain not in self.d_actions:
            raise KeyError(f"unknown action domain {domain!r}")
        d_a = self.d_actions[domain] if actions is not None else 0
        if T > 16:
            raise NotImplementedError("temporal attention kernel handles T <= 16 frames")
        skip = 1 if skip_normalization else 0
        if skip != self._skip_norm:
            self._skip_norm, self._plans = skip, {}
        ws = self._workspace(B, T, S, A, train)
        stream = torch.cuda.current_stream().cuda_stream
        self.refresh_weights(domain if actions is not None else None, stream)
        ws["ids"].copy_(ids_BTS, non_blocking=True)
        if actions is not None:
            if actions.shape[-1] != d_a:
                raise ValueError(f"action_ids last dim {actions.shape[-1]} != d_action {d_a} of domain {domain}")
            ws["actions"][: B * T * d_a].copy_(actions[:, :T].reshape(-1), non_blocking=True)
        if train:
            self.bump_dropout()
            self._invalidate_trunk_stamps()
        fce = labels is not None and train and loss_grad and not need_logits and self._use_fused_ce(B, T, S)
        pl = self._forward_plan(B, T, S, A, train, domain if (A > 0 and not policy) else None, readout=not fce)
        self._act, self.act_scale = None, 0.0
        if policy:
            ws["a_tok"].copy_(self.view("action_mask_tokens").reshape(-1, 256)[:T].repeat(B, 1))
            pl.run(stream, timer=self.timer)
            pooled = ws["x"].view(B * T, S + A, 256)[:, S:].mean(dim=1)
            W, bias = self.view(f"action_out_projectors.{domain}.weight"), self.view(f"action_out_projectors.{domain}.bias")
            self._act = dict(out=torch.addmm(bias, pooled, W.t()), loss=None, dom=domain)
        elif self.jpa and A > 0:
            # action stem -> mix in the mask tokens -> everything else (the modulation keeps the embedded actions, :672)
            pl.run(stream, 0, pl.marks["post_stem"], timer=self.timer)
            m = ws["amask"]
            m.zero_() if action_mask is None else m.copy_(action_mask.reshape(B * T, 1).to(self.device, F32))
            mt = self.view("action_mask_tokens").reshape(-1, 256)[:T]
            torch.where(m.bool(), mt.repeat(B, 1), ws["a_emb"], out=ws["a_tok"])
            pl.run(stream, pl.marks["post_stem"], None, timer=self.timer)
            # action read-out of the pooled action tokens + the reference's action loss (:676-678, :725-726)
            SA = S + A
            pooled = ws["x"].view(B * T, SA, 256)[:, S:].mean(dim=1)
            W, bias = self.view(f"action_out_projectors.{domain}.weight"), self.view(f"action_out_projectors.{domain}.bias")
            out = torch.addmm(bias, pooled, W.t())
            labels_a = actions[:, :T].reshape(B * T, d_a).to(self.device, F32)
            mfrac = m.mean()
            self._act = dict(pooled=pooled, out=out, labels=labels_a, mfrac=mfrac, dom=domain,
                             loss=((labels_a - out) ** 2).mean() * mfrac)
        else:
            pl.run(stream, timer=self.timer)
        self._last = (B, T, S, A, domain if (A > 0 and not policy) else None)
        self._last_fce = fce
        if labels is not None:
            ws["labels"].copy_(labels.reshape(B, T * S), non_blocking=True)
            ws["stats"].zero_()
            self._loss_plan(B, T, S, train and loss_grad, fused=fce, A=A).run(stream, timer=self.timer)
        return ws

    def _invalidate_trunk_stamps(self, keep: Optional[range] = None) -> None:
        """A training forward is about to overwrite saved per-layer activations: block-level autograd nodes (trunk_autograd_forward)
        whose backward has not run yet must not use them.  `keep`: the layers the caller re-stamps itself."""
        stamps = getattr(self, "_trunk_stamps", None)
        if stamps is None:
            return
        for l in range(len(stamps)):
            if keep is None or l not in keep:
                stamps[l] = 0

    # ------------------------------------------------------------------------------ trunk-only training (STMAR)
    def trunk_train_forward(self, B: int, T: int, S: int, actions: Optional[torch.Tensor], domain: Optional[str],
                            build_x: Callable[[dict], None], train: bool = True) -> dict:
        """Action stem -> `build_x(ws)` writes the residual stream ws["x"] ([B*T*(S+A), 256] fp32, may read ws["a_emb"])
        -> the L ST-blocks with every activation saved.  No token embedding, readout or loss: the caller owns those
        (hma/model/st_mar.py keeps the trunk of st_mask_git.py and replaces what is around it).  `actions` None (st_mar.py:154,
        the unconditioned model): no action tokens (A = 0), no modulation -- the blocks alone."""
        if actions is None:
            ws = self._workspace(B, T, S, 0, train)
            stream = torch.cuda.current_stream().cuda_stream
            self.refresh_weights(None, stream)
            build_x(ws)
            if train:
                self.bump_dropout()
                self._invalidate_trunk_stamps()
            self._forward_plan(B, T, S, 0, train, None, embed=False, readout=False).run(stream, timer=self.timer)
            if train:
                self._last = (B, T, S, 0, None)
            return ws
        A = self.cfg.action_token_size
        d_a = self.d_actions[domain]
        if actions.shape[-1] != d_a:
            raise ValueError(f"action_ids last dim {actions.shape[-1]} != d_action {d_a} of domain {domain}")
        ws = self._workspace(B, T, S, A, train)
        stream = torch.cuda.current_stream().cuda_stream
        self.refresh_weights(domain, stream)
        ws["actions"][: B * T * d_a].copy_(actions[:, :T].reshape(-1), non_blocking=True)
        am = f"action_mlp.{domain}.model"
        _lib.call("hma_action_stem_fwd", stream, ws["actions"].data_ptr(), self.buffers[domain][0].data_ptr(),
                  self.buffers[domain][1].data_ptr(), self.action_dims[domain], self._p(f"{am}.0.weight"), self._p(f"{am}.0.bias"),
                  self._p(f"{am}.1.weight"), self._p(f"{am}.1.bias"), self._p(f"{am}.3.weight"), self._p(f"{am}.3.bias"),
                  ws["an"].data_ptr(), ws["sxhat"].data_ptr(), ws["srstd"].data_ptr(), ws["sh"].data_ptr(), ws["a_emb"].data_ptr(),
                  B * T, d_a, self._skip_norm)
        build_x(ws)
        if train:
            self.bump_dropout()
            self._invalidate_trunk_stamps()
        self._forward_plan(B, T, S, A, train, domain, embed=False, readout=False).run(stream, timer=self.timer)
        if train:
            self._last = (B, T, S, A, domain)
        return ws

    def trunk_train_backward(self, fill_dx: Callable[[dict], None], embed_bwd: Callable[[dict], None],
                             on_segment: Optional[Callable[[str], None]] = None, segment_layers: int = 0) -> None:
        """`fill_dx(ws)` writes d loss / d x_out into ws["dx"] (zeroed before) -> backward of the ST-blocks -> `embed_bwd(ws)`
        consumes ws["dx"] (= d loss / d x_in) and adds the action rows' sum into ws["da_emb"] -> adaLN stacks and action stem.
        `on_segment(label)` as in `backward`: called after every `segment_layers` enqueued layers ('layer<l>', then 'end')."""
        B, T, S, A, domain = self._last
        ws = self._ws
        stream = torch.cuda.current_stream().cuda_stream
        ws["dx"].zero_()
        if A > 0:
            ws["da_emb"].zero_()
        fill_dx(ws)
        if on_segment is not None and segment_layers > 0:
            self.ada_group = segment_layers  # (the plan completes every deferred launch of a segment before its mark)
        pl = self._backward_plan(B, T, S, A, domain)
        start = pl.marks["post_readout"]
        if on_segment is not None and segment_layers > 0:
            L = self.cfg.num_layers
            for l in reversed(range(L)):
                if (L - l) % segment_layers == 0 or l == 0:
                    stop = pl.marks[f"layer{l}"]
                    pl.run(stream, start, stop, timer=self.timer)
                    start = stop
                    on_segment(f"layer{l}")
        pl.run(stream, start, pl.marks["embed"], timer=self.timer)
        embed_bwd(ws)
        pl.run(stream, pl.marks["post_embed"], None, timer=self.timer)
        if on_segment is not None and segment_layers > 0:
            on_segment("end")

    # ------------------------------------------------------------------------------ blocks under autograd (STBlock / STTransformerDecoder)
    def trunk_autograd_forward(self, x_BTSD: torch.Tensor, a_emb: Optional[torch.Tensor], domain: Optional[str], l0: int = 0,
                               l1: Optional[int] = None) -> Tuple[torch.Tensor, tuple]:
        """Layers [l0, l1) on a given residual stream with every activation saved (st_transformer.py:79-114, 172-177 as autograd
        modules).  Returns the new stream and a stamp `trunk_autograd_backward` checks: the saved activations of a layer belong to
        its LAST forward."""
        B, T, SA, D = x_BTSD.shape
        L = self.cfg.num_layers
        l1 = L if l1 is None else l1
        A = self.cfg.action_token_size if a_emb is not None else 0
        S = SA - A
        if a_emb is not None and not self.modulate:
            raise NotImplementedError("only the 'modulate' action projector is built")
        if T > 16:
            raise NotImplementedError("temporal attention kernel handles T <= 16 frames")
        dom = domain if A > 0 else None
        ws = self._workspace(B, T, S, A, True)
        stream = torch.cuda.current_stream().cuda_stream
        self.refresh_weights(dom, stream)
        ws["x"].view(B, T, SA, D).copy_(x_BTSD, non_blocking=True)
        if A > 0:
            ws["a_emb"].view(B, T, D).copy_(a_emb[:, :T], non_blocking=True)
        self.bump_dropout()
        self._forward_plan(B, T, S, A, True, dom, embed=False, l0=l0, l1=l1, readout=False).run(stream, timer=self.timer)
        self._last = (B, T, S, A, dom)
        stamps = getattr(self, "_trunk_stamps", None)
        if stamps is None or len(stamps) != L:
            stamps = self._trunk_stamps = [0] * L
        if A > 0:
            # the adaLN stacks above ran for ALL layers from this call's action embeddings (one batched GEMM pair): the saved shift /
            # scale rows of layers outside [l0, l1) now belong to this call's a_emb -- unless it is the tensor the previous call had
            # ("the tensor the previous call had": address + version only mean that while the previous tensor is ALIVE -- a new tensor the
            # caching allocator puts at a freed address starts at version 0 too -- so the engine keeps a reference to it until the next call)
            akey = (a_emb.data_ptr(), a_emb._version, tuple(a_emb.shape), tuple(a_emb.stride()), dom)
            if akey != getattr(self, "_trunk_aemb_key", None):
                self._invalidate_trunk_stamps(keep=range(l0, l1))
            self._trunk_aemb_key = akey
            self._trunk_aemb_ref = a_emb
        self._trunk_counter = getattr(self, "_trunk_counter", 0) + 1
        for l in range(l0, l1):
            stamps[l] = self._trunk_counter
        return ws["x"].view(B, T, SA, D).clone(), (B, T, S, A, dom, l0, l1, self.ws_generation, self._trunk_counter, self._drop_counter)

    def trunk_autograd_backward(self, dy_BTSD: torch.Tensor, stamp: tuple) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """d loss / d (stream out) -> (d loss / d (stream in), d loss / d a_emb); the layers' weight gradients are ADDED into the flat
        gradient buffer.  Runs the layers' part of the recorded backward plan (one adaLN group per layer:
        a range of layers is then a contiguous run of the plan)."""
        B, T, S, A, dom, l0, l1, gen, count, drops = stamp
        L = self.cfg.num_layers
        if self.mlp_drop > 0.0 and drops != self._drop_counter:
            raise RuntimeError("another training forward drew new Dropout masks since this forward (one device seed per engine): run "
                               "backward before the next forward when mlp_drop > 0")
        if gen != self.ws_generation or any(self._trunk_stamps[l] != count for l in range(l0, l1)):
            raise RuntimeError("the saved activations of these layers were overwritten by a later forward (the engine keeps ONE set "
                               "per layer): run backward before the next forward through the same layers")
        ws = self._ws
        SA, M, Fr = S + A, B * T * (S + A), B * T
        stream = torch.cuda.current_stream().cuda_stream
        saved = self.ada_group, self.wgrad_layers
        self.ada_group, self.wgrad_layers = 1, 1  # (a range of layers is a self-contained run of the plan: nothing deferred across layers)
        try:
            pl = self._backward_plan(B, T, S, A, dom)
        finally:
            self.ada_group, self.wgrad_layers = saved
        ws["dx"].view(B, T, SA, 256).copy_(dy_BTSD, non_blocking=True)
        if A > 0:
            ws["da_emb"].zero_()
        if l1 == L:
            start = pl.marks["post_readout"]
        else:
            start = pl.marks[f"layer{l1}"]
            _lib.call("hma_cast_bf16", stream, ws["dx"].data_ptr(), pl.dxb_in[l1 - 1], M * 256)
            if "dss" in ws:
                ws["dss"].zero_()
        pl.run(stream, start, pl.marks[f"layer{l0}"], timer=self.timer)
        return ws["dx"].view(B, T, SA, 256).clone(), (ws["da_emb"].view(B, T, 256).clone() if A > 0 else None)

    def run_trunk(self, x_BTSD: torch.Tensor, a_emb: Optional[torch.Tensor], domain: Optional[str], l0: int = 0,
                  l1: Optional[int] = None) -> torch.Tensor:
        """Layers [l0, l1) of the decoder on a given residual stream (inference; STTransformerDecoder.forward)."""
        B, T, SA, D = x_BTSD.shape
        A = self.cfg.action_token_size if a_emb is not None else 0
        S = SA - A
        if a_emb is not None and not self.modulate:
            raise NotImplementedError("only the 'modulate' action projector is built")
        ws = self._workspace(B, T, S, A, False)
        stream = torch.cuda.current_stream().cuda_stream
        self.refresh_weights(domain if a_emb is not None else None, stream)
        ws["x"].view(B, T, SA, D).copy_(x_BTSD, non_blocking=True)
        if a_emb is not None:
            ws["a_emb"].view(B, T, D).copy_(a_emb[:, :T], non_blocking=True)
        self._forward_plan(B, T, S, A, False, domain if A > 0 else None, embed=False, l0=l0, l1=l1, readout=False).run(stream)
        return ws["x"].view(B, T, SA, D).clone()

    @staticmethod
    def zero_dx_action_rows(ws: dict, frames: int, S: int, A: int) -> None:
        """Before the backward plan: the read-out's input gradient is STORED into the image rows of dx (hma_gemm_nt, EPI_F32 with
        c_group = (S, S + A)), so only the A action rows of each frame have to start at zero (a fifth of the buffer)."""
        if A > 0:
            ws["dx"].view(frames, S + A, -1)[:, S:].zero_()

    def zero_grad(self, active_domains: Optional[Sequence[str]] = None) -> None:
        """Zero the whole gradient buffer, or only the ranges that can receive gradients this step."""
        if active_domains is None:
            self.G.zero_()
        else:
            for a, b in self.layout.trainable_ranges(active_domains):
                self.G[a:b].zero_()

    def zero_grad_domains(self, domains: Sequence[str]) -> None:
        """Zero only the blocks of `domains` (a domain that first appears in a later micro-batch of an accumulation window)."""
        for dom in domains:
            a, b = self.layout.regions[f"dom:{dom}"]
            self.G[a:b].zero_()

    def backward(self, grad_scale: float = 1.0, on_segment: Optional[Callable[[str], None]] = None,
                 segment_layers: int = 0) -> None:
        """Accumulates parameter gradients of the last `forward(train=True)` into the flat G buffer.

        `on_segment(label)` is called after every `segment_layers` layers have been enqueued (labels
        'layer<l>' then 'end') so a data-parallel driver can start all-reducing finished buckets."""
        B, T, S, A, domain = self._last
        ws = self._ws
        stream = torch.cuda.current_stream().cuda_stream
        if grad_scale != self.grad_scale.value:
            # dlogits were produced with the scale captured at forward time: redo the (cheap) CE pass
            self.grad_scale.value = grad_scale
            ws["stats"].zero_()
            self._loss_plan(B, T, S, True, fused=getattr(self, "_last_fce", False), A=A).run(stream)
        lay = self.layout
        self.zero_dx_action_rows(ws, B * T, S, A)
        if A > 0:
            ws["da_emb"].zero_()
        jpa = self.jpa and A > 0 and self._act is not None
        if jpa:
            ws["da_tok"].zero_()
            if self.act_scale != 0.0:
                # backward of the action loss: the read-out's own gradients, and d loss / d (the action rows of the final x)
                a = self._act
                d_out = (a["out"] - a["labels"]) * (2.0 * self.act_scale / a["out"].numel()) * a["mfrac"]
                W = self.view(f"action_out_projectors.{a['dom']}.weight")
                self.view(f"action_out_projectors.{a['dom']}.weight", self.G).addmm_(d_out.t(), a["pooled"])
                self.view(f"action_out_projectors.{a['dom']}.bias", self.G).add_(d_out.sum(dim=0))
                ws["dx"].view(B * T, S + A, 256)[:, S:].add_((d_out @ W).div_(A)[:, None, :])
        if on_segment is not None and segment_layers > 0:
            self.ada_group = segment_layers  # (the plan completes every deferred launch of a segment before its mark)
        pl = self._backward_plan(B, T, S, A, domain)

        def tail(start):
            if not jpa:
                pl.run(stream, start, None, timer=self.timer)
                return
            # the embedding backward left the action rows' sums in da_tok: masked frames feed action_mask_tokens[t], the others the
            # action stem (through da_emb, to which the adaLN stacks add afterwards)
            pl.run(stream, start, pl.marks["post_embed"], timer=self.timer)
            m = ws["amask"]
            ws["da_emb"].add_(ws["da_tok"] * (1.0 - m))
            self.view("action_mask_tokens", self.G).reshape(-1, 256)[:T].add_((ws["da_tok"] * m).view(B, T, 256).sum(dim=0))
            pl.run(stream, pl.marks["post_embed"], None, timer=self.timer)

        if on_segment is None or segment_layers <= 0:
            tail(0)
            return
        L = self.cfg.num_layers
        start = 0
        for l in reversed(range(L)):
            if (L - l) % segment_layers == 0 or l == 0:
                stop = pl.marks[f"layer{l}"]
                pl.run(stream, start, stop, timer=self.timer)
                start = stop
                on_segment(f"layer{l}")
        tail(start)
        on_segment("end")

    def optimizer_step(self, lr: float, active_domains: Sequence[str], betas=(0.9, 0.95), eps: float = 1e-8,
                       weight_decay: float = 0.05, max_norm: Optional[float] = 1.0, extra_grads: Sequence[torch.Tensor] = ()) -> None:
        """Global-norm clip + AdamW over the ranges that received gradients (hma/train_multi.py:593-598)."""
        if self.M is None:
            self.M = torch.zeros_like(self.P)
            self.V = torch.zeros_like(self.P)
        stream = torch.cuda.current_stream().cuda_stream
        ranges = self.layout.trainable_ranges(active_domains)
        # Adam's bias correction uses the number of updates EACH parameter has received (torch.optim.AdamW keeps
        # `state["step"]` per parameter and skips parameters whose grad is None): the dense range is stepped every
        # time, a domain's block only when that domain was active on some rank.  The counts live on the device.
        idxs = [0] + [1 + i for i, dom in enumerate(self.layout.domains) if dom in active_domains]
        assert len(idxs) == len(ranges)
        # The square norm is always taken: besides the clip coefficient it carries the non-finite check (a NaN / Inf loss on
        # any rank makes the all-reduced gradients, hence this norm, non-finite on every rank -> every rank skips the update).
        self.sqnorm.zero_()
        for a, b in ranges:
            _lib.call("hma_sqnorm", stream, self.G.data_ptr() + 4 * a, b - a, self.sqnorm.data_ptr())
        for g in extra_grads:  # gradients held outside the flat buffer (STMAR's own range) count towards the global norm
            _lib.call("hma_sqnorm", stream, g.data_ptr(), g.numel(), self.sqnorm.data_ptr())
        sq = self.sqnorm.data_ptr()
        for (a, b), ri in zip(ranges, idxs):
            _lib.call("hma_adamw_counted", stream, self.P.data_ptr() + 4 * a, self.G.data_ptr() + 4 * a, self.M.data_ptr() + 4 * a,
                      self.V.data_ptr() + 4 * a, self.Wb.data_ptr() + 2 * a, b - a, lr, betas[0], betas[1], eps, weight_decay,
                      self.steps_dev.data_ptr() + 8 * ri, self._step_calls[ri] & 1, sq, float(max_norm or 0.0),
                      self.flags.data_ptr() + a // ALIGN)
            self._step_calls[ri] += 1
        # bf16 copies were emitted by the update itself; only the transposed copies are stale
        self._wt_ok = False
        self._dom_fresh = set()

    def grad_norm(self) -> torch.Tensor:
        return self.sqnorm.sqrt()
