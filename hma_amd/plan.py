"""Launch plans: recorded sequences of C-ABI calls with fixed arguments, replayed on a stream or captured into
hipGraphs by the step drivers, plus the per-launch HIP-event timer bench.py uses for the kernel families' roofline numbers."""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import A_F32, EPI_ATOMIC_F32, EPI_BF16, EPI_DGELU, EPI_DSILU, EPI_F32, EPI_GELU2, EPI_RESID, EPI_SILU2
from .ops import (make_chain_ab_fwd, make_chain_a_bwd, make_chain_a_fwd, make_chain_b_fwd, make_chain_s_bwd, make_chain_t_bwd, make_gemm_nt, make_gemm_tn, make_mlp_bwd,
                  make_mlp_fwd, make_readout_ce)


class Plan:
    """A recorded sequence of C-ABI launches with fixed arguments; `run` replays it on a stream."""

    # Per GPU, for the life of the process (recorded plans and captured graphs hold the raw pointers): the scratch of the two-stage
    # weight-gradient reduction.  Keyed by device index, so that engines on different devices of one process do not share it.
    _tn_workspaces: Dict[object, torch.Tensor] = {}

    @property
    def tn_workspace(self) -> Optional[torch.Tensor]:
        return Plan._tn_workspaces.get(self.dev)

    def __init__(self, dev: Optional[int] = None):
        self.dev = torch.cuda.current_device() if (dev is None and torch.cuda.is_available()) else dev
        self.calls: List[Tuple[Callable, str, tuple]] = []
        self.flops: List[float] = []   # algorithmic FLOPs of each call (0 for non-GEMM launches)
        self.bytes: List[float] = []   # algorithmic HBM bytes of each call: every operand read once, every result written once
        self.marks: Dict[str, int] = {}
        self.keep: list = []

    def add(self, name: str, *args, flops: float = 0.0, nbytes: float = 0.0) -> None:
        self.calls.append((getattr(_lib.load(), name), name, args))
        self.flops.append(flops)
        self.bytes.append(nbytes)

    @staticmethod
    def _nt_bytes(g) -> float:
        """A once + the weight + what the epilogue reads and writes."""
        b = max(g.batch, 1)
        mn = float(g.M) * g.N * b
        out = {EPI_BF16: 2, EPI_F32: 4, EPI_RESID: 8, EPI_GELU2: 4, EPI_SILU2: 4, EPI_DGELU: 4, EPI_DSILU: 4, EPI_ATOMIC_F32: 8}[g.epi]
        extra = (2 if (g.epi == EPI_RESID and g.C2) else 0) + (2 if g.ln_xhat else 0) + (2 if g.ln_xm else 0)
        return float(g.M) * g.K * b * (4 if g.a_kind == A_F32 else 2) + 2.0 * g.N * g.K * b + mn * (out + extra)

    @staticmethod
    def _tn_bytes(g) -> float:
        b = max(g.batch, 1)
        return b * (float(g.M) * g.N * (4 if g.y_kind == A_F32 else 2) + float(g.M) * g.K * (4 if g.a_kind == A_F32 else 2) + 4.0 * g.N * g.K)

    def gemm_nt(self, **kw) -> None:
        g = make_gemm_nt(**kw)
        self.keep.append(g)
        self.add("hma_gemm_nt", C.byref(g), flops=2.0 * g.M * g.N * g.K * max(g.batch, 1), nbytes=self._nt_bytes(g))

    def gemm_tn(self, **kw) -> None:
        wsb = self.tn_workspace
        if wsb is not None:
            kw.setdefault("ws", wsb.data_ptr())
            kw.setdefault("ws_elems", wsb.numel())
        g = make_gemm_tn(**kw)
        self.keep.append(g)
        self.add("hma_gemm_tn", C.byref(g), flops=2.0 * g.M * g.N * g.K * max(g.batch, 1), nbytes=self._tn_bytes(g))

    def gemm_tn_pair(self, kw0: dict, kw1: dict) -> None:
        """Two weight gradients whose operands are live at the same time, in one launch (hma_gemm_tn_pair)."""
        gs = []
        wsb = self.tn_workspace
        for kw in (kw0, kw1):
            if wsb is not None:
                kw.setdefault("ws", wsb.data_ptr())
                kw.setdefault("ws_elems", wsb.numel())
            gs.append(make_gemm_tn(**kw))
        self.keep.extend(gs)
        self.add("hma_gemm_tn_pair", C.byref(gs[0]), C.byref(gs[1]),
                 flops=sum(2.0 * g.M * g.N * g.K * max(g.batch, 1) for g in gs), nbytes=sum(self._tn_bytes(g) for g in gs))

    def gemm_tn_multi(self, kws: Sequence[dict]) -> None:
        """Up to 16 weight gradients whose operands are all live, in one launch + one reduction (hma_gemm_tn_multi)."""
        gs = []
        wsb = self.tn_workspace
        for kw in kws:
            if wsb is not None:
                kw.setdefault("ws", wsb.data_ptr())
                kw.setdefault("ws_elems", wsb.numel())
            gs.append(make_gemm_tn(**kw))
        arr = (C.POINTER(type(gs[0])) * len(gs))(*[C.pointer(g) for g in gs])
        self.keep.extend(gs)
        self.keep.append(arr)
        self.add("hma_gemm_tn_multi", arr, len(gs),
                 flops=sum(2.0 * g.M * g.N * g.K * max(g.batch, 1) for g in gs), nbytes=sum(self._tn_bytes(g) for g in gs))

    def mlp_fwd(self, M: int, **kw) -> None:
        g = make_mlp_fwd(M=M, **kw)
        self.keep.append(g)
        self.add("hma_mlp_fwd", C.byref(g), flops=2.0 * M * 256 * 1024 * 2, nbytes=3072.0 * M)  # xhat 512 + x 1024 in, x 1024 + xhat 512 out

    def mlp_bwd(self, M: int, **kw) -> None:
        g = make_mlp_bwd(M=M, **kw)
        self.keep.append(g)
        # algorithmic dgrad FLOPs (the recompute is not counted); xhat 2 x 512 + dy 512 + dx 1024 in, dx 1024 + dxb 512 + hg / du 2 x 2048 out
        self.add("hma_mlp_bwd", C.byref(g), flops=2.0 * M * 256 * 1024 * 2, nbytes=8192.0 * M)

    def chain_a_fwd(self, M: int, use_mod: bool, save: bool, **kw) -> None:
        g = make_chain_a_fwd(M=M, use_mod=use_mod, **kw)
        self.keep.append(g)
        n_out = 256 * (2 if use_mod else 1) + 768
        # o 512 + x 1024 in; x 1024 + qkv 1536 out (+ xhat, xm, bf16(x): 512 each when they are saved)
        nbytes = (512 + 1024 + 1024 + 1536 + (512 * (3 if use_mod else 1) if save else 0)) * float(M)
        self.add("hma_chain_a_fwd", C.byref(g), flops=2.0 * M * 256 * n_out, nbytes=nbytes)

    def chain_b_fwd(self, M: int, with_qkv: bool, **kw) -> None:
        g = make_chain_b_fwd(M=M, **kw)
        self.keep.append(g)
        # o 512 + x 1024 in; x 1024 (+ qkv 1536) out (+ the saved LayerNorm outputs, 512 each, in training)
        saved = 512 * ((1 if kw.get("xhat2") else 0) + (1 if kw.get("xhat1n") else 0))
        self.add("hma_chain_b_fwd", C.byref(g), flops=2.0 * M * 256 * (256 + 2048 + (768 if with_qkv else 0)),
                 nbytes=(512 + 1024 + 1024 + (1536 if with_qkv else 0) + saved) * float(M))

    def chain_ab_fwd(self, M: int, with_qkv: bool, **kw) -> None:
        g = make_chain_ab_fwd(**kw)
        self.keep.append(g)
        mod = g.bundles[1] > 0
        # o_s 512 + x 1024 in; [xhat_m, xm,] bf16(x2) 512 each + qkv_t 1536 + o_t 512 + x 1024 + xhat2 512 (+ xhat1' 512 + qkv_s 1536) out
        nbytes = (512 + 1024 + (1536 if mod else 512) + 1536 + 512 + 1024 + 512 + ((512 + 1536) if with_qkv else 0)) * float(M)
        flops = 2.0 * M * 256 * (256 * (2 if mod else 1) + 768 + 256 + 2048 + (768 if with_qkv else 0)) + 4.0 * M * 16 * 256
        self.add("hma_chain_ab_fwd", C.byref(g), flops=flops, nbytes=nbytes)

    def chain_a_bwd(self, M: int, use_mod: bool, **kw) -> None:
        g = make_chain_a_bwd(M=M, use_mod=use_mod, **kw)
        self.keep.append(g)
        n_in = 768 + 256 * (2 if use_mod else 1)
        # dqkv 1536 + dx 1024 (+ xhat 512) in; dx 1024 + bf16(dx1) 512 + d_o 512 (+ bf16(dx2) 512) out
        nbytes = (1536 + 1024 + 1024 + 512 + 512 + (1024 if use_mod else 0)) * float(M)
        self.add("hma_chain_a_bwd", C.byref(g), flops=2.0 * M * 256 * n_in, nbytes=nbytes)

    def chain_s_bwd(self, M: int, **kw) -> None:
        g = make_chain_s_bwd(M=M, **kw)
        self.keep.append(g)
        # dqkv 1536 + xhat 512 + dx 1024 in; dx 1024 + bf16(dx) 512 out
        self.add("hma_chain_s_bwd", C.byref(g), flops=2.0 * M * 256 * 768, nbytes=(1536 + 512 + 1024 + 1024 + 512) * float(M))

    def chain_t_bwd(self, B: int, SA: int, **kw) -> None:
        g = make_chain_t_bwd(B=B, SA=SA, **kw)
        self.keep.append(g)
        M = float(g.T) * B * SA
        # bf16(dx) 512 + qkv 1536 in; dqkv 1536 out.  Projection dgrad + the attention's five products per head (as hma_attn_temporal_bwd)
        self.add("hma_chain_t_bwd", C.byref(g), flops=2.0 * M * 256 * 256 + 10.0 * M * g.T * 256, nbytes=(512 + 1536 + 1536) * M)

    def readout_ce(self, rows: int, **kw) -> None:
        g = make_readout_ce(rows=rows, **kw)
        self.keep.append(g)
        # x 1024 in, dlogits 2048 out per image row (+ ids / labels); the logits themselves stay in registers
        self.add("hma_readout_ce", C.byref(g), flops=2.0 * rows * 256 * 1024, nbytes=(1024.0 + (2048 if kw.get("dlogits") else 0) + 16) * rows)

    def mark(self, label: str) -> None:
        self.marks[label] = len(self.calls)

    def run(self, stream: int, start: int = 0, stop: Optional[int] = None, timer: "Optional[LaunchTimer]" = None) -> None:
        if timer is None:
            for fn, name, args in self.calls[start:stop]:
                rc = fn(stream, *args)
                if rc != 0:
                    raise _lib.HmaKernelError(f"{name} failed with code {rc}")
            return
        stop = len(self.calls) if stop is None else stop
        for i in range(start, stop):
            fn, name, args = self.calls[i]
            if name in timer.names:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = fn(stream, *args)
                e1.record()
                timer.pairs.append((name, self.flops[i], e0, e1, self.bytes[i]))
            else:
                rc = fn(stream, *args)
            if rc != 0:
                raise _lib.HmaKernelError(f"{name} failed with code {rc}")


class LaunchTimer:
    """HIP-event brackets around chosen launches, recorded on the stream the kernels run on."""

    def __init__(self, names: Sequence[str]):
        self.names = set(names)
        self.pairs: list = []

    def summary(self) -> Dict[str, Dict[str, float]]:
        out: Dict[str, Dict[str, float]] = {}
        for name, flops, e0, e1, nbytes in self.pairs:
            d = out.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out
