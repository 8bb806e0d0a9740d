"""Diffusion head: `DiffLoss` and its `SimpleMLPAdaLN` network on hand-written HIP kernels.

Mirror of hma/model/diffloss.py: same constructor, parameter names / shapes (`net.time_embed.mlp.{0,2}`, `net.cond_embed`,
`net.input_proj`, `net.res_blocks.{i}.{in_ln,mlp.0,mlp.2,adaLN_modulation.1}`, `net.final_layer.{adaLN_modulation.1,linear}`),
`forward(target, z, mask) -> loss` (cosine schedule, 1000 steps, epsilon prediction, learned-range variance, MSE + VLB
-- hma/diffusion/__init__.py:13-41) and `sample(z, temperature, cfg=1.0)` (respaced `num_sampling_steps` chain).
Extra keyword arguments replay the random draws (`t`, `noise`; `noise0`, `step_noises`) for parity tests.

Execution: every Linear is an `hma_gemm_nt` (forward / input gradient) or `hma_gemm_tn` (weight gradient) call in bf16
with fp32 accumulation, everything else one of the row kernels of csrc/diffusion.hip.  `forward` runs the whole
forward AND backward eagerly (the loss is a scalar, its gradients are linear in the incoming grad), keeps the
parameter / z gradients, and hands them to autograd through a thin Function -- no autograd graph over the kernels,
no CPU or eager-PyTorch fallback.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from ..ops import make_gemm_nt, make_gemm_tn, ptr
from .._lib import A_BF16, A_F32, EPI_BF16, EPI_F32, EPI_RESID, EPI_SILU2, EPI_DSILU

import ctypes as C

BF16, F32 = torch.bfloat16, torch.float32


# ------------------------------------------------------------------------------------------------ schedule (host, fp64)
def _cosine_betas(n: int = 1000, max_beta: float = 0.999) -> np.ndarray:
    ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2  # gaussian_diffusion.py:112-137
    return np.array([min(1 - ab((i + 1) / n) / ab(i / n), max_beta) for i in range(n)], dtype=np.float64)


def _space_timesteps(num_timesteps: int, section_counts) -> List[int]:
    """respace.py:8-61 (no 'ddimN' strings: DiffLoss passes plain counts)."""
    if isinstance(section_counts, str):
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per, extra = num_timesteps // len(section_counts), num_timesteps % len(section_counts)
    start, steps = 0, []
    for i, cnt in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < cnt:
            raise ValueError(f"cannot divide section of {size} steps into {cnt}")
        stride = 1 if cnt <= 1 else (size - 1) / (cnt - 1)
        cur = 0.0
        for _ in range(cnt):
            steps.append(start + round(cur))
            cur += stride
        start += size
    return sorted(set(steps))


class _Schedule:
    """GaussianDiffusion.__init__ tables (gaussian_diffusion.py:150-189) as fp32 device arrays for the kernels."""

    def __init__(self, betas: np.ndarray, timestep_map: Optional[List[int]] = None):
        betas = np.asarray(betas, dtype=np.float64)
        self.n = len(betas)
        ac = np.cumprod(1.0 - betas)
        ac_prev = np.append(1.0, ac[:-1])
        pv = betas * (1.0 - ac_prev) / (1.0 - ac)
        self.host = {
            "sqrt_ac": np.sqrt(ac), "sqrt_1mac": np.sqrt(1.0 - ac),
            "t6": np.stack([np.sqrt(1.0 / ac), np.sqrt(1.0 / ac - 1), betas * np.sqrt(ac_prev) / (1.0 - ac),
                            (1.0 - ac_prev) * np.sqrt(1.0 - betas) / (1.0 - ac), np.log(np.append(pv[1], pv[1:])), np.log(betas)]),
        }
        self.timestep_map = timestep_map
        self._dev: Dict[str, torch.Tensor] = {}

    def on(self, device) -> Dict[str, torch.Tensor]:
        key = str(device)
        if self._dev.get("_key") != key:
            self._dev = {k: torch.from_numpy(v.astype(np.float32)).contiguous().to(device) for k, v in self.host.items()}
            self._dev["tmap"] = None if self.timestep_map is None else torch.tensor(self.timestep_map, dtype=torch.int32, device=device)
            self._dev["_key"] = key
        return self._dev

    @staticmethod
    def train() -> "_Schedule":
        return _Schedule(_cosine_betas())

    @staticmethod
    def sampling(num_sampling_steps) -> "_Schedule":
        base = np.cumprod(1.0 - _cosine_betas())
        use = _space_timesteps(1000, str(num_sampling_steps))
        last, nb = 1.0, []
        for i in use:  # SpacedDiffusion.__init__, respace.py:71-85
            nb.append(1 - base[i] / last)
            last = base[i]
        return _Schedule(np.array(nb), use)


# ------------------------------------------------------------------------------------------------ parameter containers
class TimestepEmbedder(nn.Module):
    def __init__(self, hidden_size, frequency_embedding_size=256):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(frequency_embedding_size, hidden_size), nn.SiLU(), nn.Linear(hidden_size, hidden_size))
        self.frequency_embedding_size = frequency_embedding_size


class ResBlock(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.channels = channels
        self.in_ln = nn.LayerNorm(channels, eps=1e-6)
        self.mlp = nn.Sequential(nn.Linear(channels, channels), nn.SiLU(), nn.Linear(channels, channels))
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(channels, 3 * channels))


class FinalLayer(nn.Module):
    def __init__(self, model_channels, out_channels):
        super().__init__()
        self.norm_final = nn.LayerNorm(model_channels, elementwise_affine=False, eps=1e-6)
        self.linear = nn.Linear(model_channels, out_channels)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(model_channels, 2 * model_channels))


class SimpleMLPAdaLN(nn.Module):
    """Parameter container with the reference's names and init (diffloss.py:152-210); executed by DiffLoss."""

    def __init__(self, in_channels, model_channels, out_channels, z_channels, num_res_blocks, grad_checkpointing=False):
        super().__init__()
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks = num_res_blocks
        self.time_embed = TimestepEmbedder(model_channels)
        self.cond_embed = nn.Linear(z_channels, model_channels)
        self.input_proj = nn.Linear(in_channels, model_channels)
        self.res_blocks = nn.ModuleList([ResBlock(model_channels) for _ in range(num_res_blocks)])
        self.final_layer = FinalLayer(model_channels, out_channels)
        self.initialize_weights()

    def initialize_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight, gain=0.1)
                nn.init.constant_(m.bias, 0)
        nn.init.normal_(self.time_embed.mlp[0].weight, std=0.02)
        nn.init.normal_(self.time_embed.mlp[2].weight, std=0.02)
        for block in self.res_blocks:  # zero-out the modulation and output layers
            nn.init.constant_(block.adaLN_modulation[-1].weight, 0)
            nn.init.constant_(block.adaLN_modulation[-1].bias, 0)
        nn.init.constant_(self.final_layer.adaLN_modulation[-1].weight, 0)
        nn.init.constant_(self.final_layer.adaLN_modulation[-1].bias, 0)
        nn.init.constant_(self.final_layer.linear.weight, 0)
        nn.init.constant_(self.final_layer.linear.bias, 0)


class _Grads(torch.autograd.Function):
    """loss (already computed, with d loss / d params and d loss / d z in hand) -> autograd."""

    @staticmethod
    def forward(ctx, loss, z, dz, arena, grads, *params):
        ctx.save_for_backward(dz, arena, *grads)
        ctx.z_needs = z.requires_grad
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        # the parameter gradients are views of ONE buffer (`arena`): scaled by the incoming gradient in one launch, not one per tensor
        dz, arena, *grads = ctx.saved_tensors
        scaled = arena * g
        base, off0 = arena.untyped_storage().data_ptr(), arena.storage_offset()
        out = [scaled.as_strided(gr.shape, gr.stride(), gr.storage_offset() - off0) if gr.untyped_storage().data_ptr() == base else gr * g
               for gr in grads]
        return (None, dz * g if ctx.z_needs else None, None, None, None, *out)


_PAD = 128  # zero-padded columns of x_t (GEMM K) and of the output Linear (GEMM N)


class DiffLoss(nn.Module):
    """Diffusion Loss (hma/model/diffloss.py:10-59)."""

    def __init__(self, target_channels, z_channels, depth, width, num_sampling_steps, grad_checkpointing=False):
        super().__init__()
        if width % 256 or not 256 <= width <= 2048:
            raise NotImplementedError("the adaLN row kernels handle widths that are multiples of 256 up to 2048")
        if z_channels % 128 or target_channels > 64:
            raise NotImplementedError("z_channels must be a multiple of 128 and target_channels <= 64")
        self.in_channels = target_channels
        self.z_channels, self.width, self.depth = z_channels, width, depth
        self.net = SimpleMLPAdaLN(in_channels=target_channels, model_channels=width, out_channels=target_channels * 2,
                                  z_channels=z_channels, num_res_blocks=depth, grad_checkpointing=grad_checkpointing)
        self._train = _Schedule.train()
        self._gen = _Schedule.sampling(num_sampling_steps)

    # ---------------------------------------------------------------------------------------- plumbing
    @staticmethod
    def _nt(stream, A, W, bias, out, epi, *, a_kind=A_BF16, out2=None, aux=None):
        M, K = A.shape
        N = W.shape[0]
        g = make_gemm_nt(A=ptr(A), lda=K, a_kind=a_kind, W=ptr(W), ldw=K, M=M, N=N, K=K, epi=epi, Cp=ptr(out), ldc=N,
                         bias=ptr(bias), C2=ptr(out2), ldc2=N, U=ptr(aux), ldu=N)
        _lib.call("hma_gemm_nt", stream, C.byref(g))

    def _tn(self, stream, dY, A, dW, dB, *, y_kind=A_BF16, a_kind=A_BF16):
        """Weight / bias gradient of one Linear.  With the workspace the bf16 x bf16 shapes of the head (1024 x 1024, 3072 x 1024:
        2 x 65536 x 1024 x 1024 FLOP each) run on the LDS-DMA ring kernel with a two-stage reduction (`hma_gemm_tn`); without it
        they ran on the register-staged kernel at 186 TFLOP/s."""
        M, N = dY.shape
        K = A.shape[1]
        ws = getattr(self, "_tn_ws", None)
        if ws is None or ws.device != dY.device:
            ws = torch.empty(256 * (65536 + 256), dtype=F32, device=dY.device)
            object.__setattr__(self, "_tn_ws", ws)  # scratch, not a buffer of the module
        g = make_gemm_tn(dY=ptr(dY), ldy=N, y_kind=y_kind, A=ptr(A), lda=K, a_kind=a_kind, M=M, N=N, K=K, dW=ptr(dW), lddw=K,
                         dBias=ptr(dB), ws=ptr(ws), ws_elems=ws.numel())
        _lib.call("hma_gemm_tn", stream, C.byref(g))

    def _weights(self, stream, need_t: bool):
        """bf16 copies of every Linear weight ([out, in], the NT layout) and, for backward, of its transpose."""
        dev = self.net.cond_embed.weight.device
        lin = self._linears()
        Wb, Wt, bias = {}, {}, {}
        for name, (w, b, pad_out, pad_in) in lin.items():
            n_out, n_in = w.shape
            wp = w.detach()
            if pad_out or pad_in:  # zero rows / columns up to the GEMM tile multiples
                wp = torch.zeros(n_out + pad_out, n_in + pad_in, dtype=F32, device=dev)
                wp[:n_out, :n_in] = w.detach()
            wp = wp.contiguous()
            wb = torch.empty(wp.shape, dtype=BF16, device=dev)
            _lib.call("hma_cast_bf16", stream, ptr(wp), ptr(wb), wp.numel())
            Wb[name] = wb
            bb = b.detach()
            if pad_out:
                bb = torch.zeros(n_out + pad_out, dtype=F32, device=dev)
                bb[:n_out] = b.detach()
            bias[name] = bb.contiguous()
            if need_t:
                wt = torch.empty(wp.shape[1], wp.shape[0], dtype=BF16, device=dev)
                _lib.call("hma_transpose_cast_bf16", stream, ptr(wp), ptr(wt), wp.shape[0], wp.shape[1], 1, 0, 0)
                Wt[name] = wt
        return Wb, Wt, bias

    def _linears(self):
        n, Cc = self.net, self.in_channels
        d = {"t0": (n.time_embed.mlp[0].weight, n.time_embed.mlp[0].bias, 0, 0),
             "t2": (n.time_embed.mlp[2].weight, n.time_embed.mlp[2].bias, 0, 0),
             "cond": (n.cond_embed.weight, n.cond_embed.bias, 0, 0),
             "in": (n.input_proj.weight, n.input_proj.bias, 0, _PAD - Cc),
             "fada": (n.final_layer.adaLN_modulation[1].weight, n.final_layer.adaLN_modulation[1].bias, 0, 0),
             "lin": (n.final_layer.linear.weight, n.final_layer.linear.bias, _PAD - 2 * Cc, 0)}
        for i, b in enumerate(n.res_blocks):
            d[f"ada{i}"] = (b.adaLN_modulation[1].weight, b.adaLN_modulation[1].bias, 0, 0)
            d[f"m0_{i}"] = (b.mlp[0].weight, b.mlp[0].bias, 0, 0)
            d[f"m2_{i}"] = (b.mlp[2].weight, b.mlp[2].bias, 0, 0)
        return d

    def _net_forward(self, stream, Wb, bias, xt_pad, tfreq, z, keep: bool):
        """SimpleMLPAdaLN.forward (diffloss.py:212-233) -> out fp32 [N, 128] = [eps | v | 0]; `keep` saves what backward reads."""
        N, W, dev = xt_pad.shape[0], self.width, xt_pad.device
        e = lambda *s, dt=BF16: torch.empty(*s, dtype=dt, device=dev)
        sv: Dict[str, torch.Tensor] = {}
        u_t, a_t = e(N, W), e(N, W)
        self._nt(stream, tfreq, Wb["t0"], bias["t0"], u_t, EPI_SILU2, out2=a_t)
        y = e(N, W, dt=F32)
        self._nt(stream, z, Wb["cond"], bias["cond"], y, EPI_F32, a_kind=A_F32 if z.dtype == F32 else A_BF16)
        self._nt(stream, a_t, Wb["t2"], bias["t2"], y, EPI_RESID)           # y = cond_embed(c) + time_embed(t)
        sy = e(N, W)
        _lib.call("hma_silu_cast", stream, ptr(y), ptr(sy), N * W)
        x = e(N, W, dt=F32)
        self._nt(stream, xt_pad, Wb["in"], bias["in"], x, EPI_F32)
        sv.update(u_t=u_t, a_t=a_t, y=y, sy=sy)
        for i, blk in enumerate(self.net.res_blocks):
            mod = e(N, 3 * W)
            self._nt(stream, sy, Wb[f"ada{i}"], bias[f"ada{i}"], mod, EPI_BF16)    # shift | scale | gate
            hm = e(N, W)
            _lib.call("hma_adaln_fwd", stream, ptr(x), ptr(mod), 3 * W, 0, W, ptr(blk.in_ln.weight), ptr(blk.in_ln.bias), 1e-6, ptr(hm), N, W)
            u1, a1 = e(N, W), e(N, W)
            self._nt(stream, hm, Wb[f"m0_{i}"], bias[f"m0_{i}"], u1, EPI_SILU2, out2=a1)
            h2 = e(N, W)
            self._nt(stream, a1, Wb[f"m2_{i}"], bias[f"m2_{i}"], h2, EPI_BF16)
            if keep:
                sv[f"x{i}"] = x.clone()
                sv.update({f"mod{i}": mod, f"hm{i}": hm, f"u1_{i}": u1, f"a1_{i}": a1, f"h2_{i}": h2})
            _lib.call("hma_gate_fwd", stream, ptr(x), ptr(mod), 3 * W, 2 * W, ptr(h2), N, W)
        modf = e(N, 2 * W)
        self._nt(stream, sy, Wb["fada"], bias["fada"], modf, EPI_BF16)
        hf = e(N, W)
        _lib.call("hma_adaln_fwd", stream, ptr(x), ptr(modf), 2 * W, 0, W, None, None, 1e-6, ptr(hf), N, W)
        out = e(N, _PAD, dt=F32)
        self._nt(stream, hf, Wb["lin"], bias["lin"], out, EPI_F32)
        if keep:
            sv.update(xf=x, modf=modf, hf=hf)
        return out, sv

    # ---------------------------------------------------------------------------------------- training
    def forward(self, target, z, mask=None, *, t: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None):
        dev = z.device
        if dev.type != "cuda":
            raise RuntimeError("DiffLoss runs on the GPU kernels only")
        N, Cc, W = target.shape[0], self.in_channels, self.width
        sch = self._train.on(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        target = target.detach().to(F32).contiguous()
        zc = z.detach().contiguous()
        zc = zc if zc.dtype in (F32, BF16) else zc.float()
        t = torch.randint(0, self._train.n, (N,), device=dev) if t is None else t.to(dev)
        noise = torch.randn(N, Cc, device=dev) if noise is None else noise.to(dev, F32).contiguous()
        maskf = None if mask is None else mask.detach().to(dev, F32).contiguous()
        need_grad = torch.is_grad_enabled() and (z.requires_grad or any(p.requires_grad for p in self.parameters()))
        Wb, Wt, bias = self._weights(stream, need_grad)

        xt = torch.empty(N, Cc, dtype=F32, device=dev)
        xt_pad = torch.empty(N, _PAD, dtype=BF16, device=dev)
        tfreq = torch.empty(N, 256, dtype=BF16, device=dev)
        _lib.call("hma_diff_prepare", stream, ptr(target), ptr(noise), ptr(t), ptr(sch["sqrt_ac"]), ptr(sch["sqrt_1mac"]), None, ptr(xt),
                  ptr(xt_pad), ptr(tfreq), N, Cc, _PAD)
        out, sv = self._net_forward(stream, Wb, bias, xt_pad, tfreq, zc, keep=need_grad)
        denom = (maskf.sum() + 1e-8).reshape(1) if maskf is not None else torch.full((1,), float(N), device=dev)
        stats = torch.zeros(8, dtype=F32, device=dev)  # HMA_CE_STATS_FLOATS
        dout = torch.zeros(N, _PAD, dtype=F32, device=dev) if need_grad else None
        _lib.call("hma_diff_loss", stream, ptr(out), _PAD, ptr(target), ptr(xt), ptr(noise), ptr(t), ptr(sch["t6"]), self._train.n, ptr(maskf),
                  ptr(denom), 1.0, ptr(stats), None, ptr(dout), N, Cc)
        loss = (stats[:1] / denom).reshape(())
        self.last_net_out = out[:, : 2 * Cc]
        if not need_grad:
            return loss
        names, grads, dz, arena = self._backward(stream, Wb, Wt, sv, dout, xt_pad, tfreq, zc)
        named = dict(self.named_parameters())
        params = [named[n] for n in names]
        return _Grads.apply(loss, z, dz, arena, grads, *params)

    def _backward(self, stream, Wb, Wt, sv, dout, xt_pad, tfreq, z):
        N, W, dev, Cc = dout.shape[0], self.width, dout.device, self.in_channels
        e = lambda *s, dt=BF16: torch.empty(*s, dtype=dt, device=dev)
        lin = self._linears()
        G: Dict[str, torch.Tensor] = {}

        # every gradient accumulator of this backward out of ONE zero-filled buffer (46 fills of a few KB .. 4 MB were 46 launches)
        al = lambda n: (n + 63) // 64 * 64
        need = sum(al((w.shape[0] + po) * (w.shape[1] + pi)) + al(w.shape[0] + po) for w, _, po, pi in lin.values()) + 2 * self.depth * al(W)
        arena, used = torch.zeros(need, dtype=F32, device=dev), [0]

        def take(*shape):
            n = 1
            for d_ in shape:
                n *= d_
            t = arena[used[0]:used[0] + n].view(*shape)
            used[0] += al(n)
            return t

        def wgrad(key, dY, A, y_kind=A_BF16, a_kind=A_BF16):
            w, b, pad_out, pad_in = lin[key]
            dW = take(w.shape[0] + pad_out, w.shape[1] + pad_in)
            dB = take(w.shape[0] + pad_out)
            self._tn(stream, dY, A, dW, dB, y_kind=y_kind, a_kind=a_kind)
            G[key + ".w"], G[key + ".b"] = dW[: w.shape[0], : w.shape[1]], dB[: w.shape[0]]  # (views: the padding is not part of them)

        # final layer: out = linear(adaln(x)); dout is fp32 [N, 128]
        wgrad("lin", dout, sv["hf"], y_kind=A_F32)
        dhf = e(N, W)
        self._nt(stream, dout, Wt["lin"], None, dhf, EPI_BF16, a_kind=A_F32)
        # (dx and dsy are 1 GB each at the C4 shape: their first producers WRITE them -- no zero-fill, no read of zeros)
        dx = e(N, W, dt=F32)
        dmodf = e(N, 2 * W)
        _lib.call("hma_adaln_bwd_acc", stream, ptr(dhf), ptr(sv["xf"]), ptr(sv["modf"]), 2 * W, 0, W, None, None, 1e-6, ptr(dx), ptr(dmodf),
                  None, None, N, W, 0)
        wgrad("fada", dmodf, sv["sy"])
        dsy = e(N, W, dt=F32)
        self._nt(stream, dmodf, Wt["fada"], None, dsy, EPI_F32)
        for i in reversed(range(self.depth)):
            blk = self.net.res_blocks[i]
            mod, dmod = sv[f"mod{i}"], e(N, 3 * W)
            dh2 = e(N, W)
            _lib.call("hma_gate_bwd", stream, ptr(dx), ptr(mod), 3 * W, 2 * W, ptr(sv[f"h2_{i}"]), ptr(dh2), ptr(dmod), N, W)
            wgrad(f"m2_{i}", dh2, sv[f"a1_{i}"])
            du1 = e(N, W)
            self._nt(stream, dh2, Wt[f"m2_{i}"], None, du1, EPI_DSILU, aux=sv[f"u1_{i}"])
            wgrad(f"m0_{i}", du1, sv[f"hm{i}"])
            dhm = e(N, W)
            self._nt(stream, du1, Wt[f"m0_{i}"], None, dhm, EPI_BF16)
            dg, db = take(W), take(W)
            _lib.call("hma_adaln_bwd", stream, ptr(dhm), ptr(sv[f"x{i}"]), ptr(mod), 3 * W, 0, W, ptr(blk.in_ln.weight), ptr(blk.in_ln.bias),
                      1e-6, ptr(dx), ptr(dmod), ptr(dg), ptr(db), N, W)
            G[f"ln{i}.w"], G[f"ln{i}.b"] = dg, db
            wgrad(f"ada{i}", dmod, sv["sy"])
            self._nt(stream, dmod, Wt[f"ada{i}"], None, dsy, EPI_RESID)
        wgrad("in", dx, xt_pad, y_kind=A_F32)
        dy = e(N, W, dt=F32)
        _lib.call("hma_silu_bwd", stream, ptr(sv["y"]), ptr(dsy), ptr(dy), N * W)
        wgrad("cond", dy, z, y_kind=A_F32, a_kind=A_F32 if z.dtype == F32 else A_BF16)
        dz = e(N, self.z_channels, dt=F32)
        self._nt(stream, dy, Wt["cond"], None, dz, EPI_F32, a_kind=A_F32)
        wgrad("t2", dy, sv["a_t"], y_kind=A_F32)
        dut = e(N, W)
        self._nt(stream, dy, Wt["t2"], None, dut, EPI_DSILU, a_kind=A_F32, aux=sv["u_t"])
        wgrad("t0", dut, tfreq)
        # map to parameter names
        pn = {"t0": "net.time_embed.mlp.0", "t2": "net.time_embed.mlp.2", "cond": "net.cond_embed", "in": "net.input_proj",
              "fada": "net.final_layer.adaLN_modulation.1", "lin": "net.final_layer.linear"}
        for i in range(self.depth):
            pn.update({f"ada{i}": f"net.res_blocks.{i}.adaLN_modulation.1", f"m0_{i}": f"net.res_blocks.{i}.mlp.0",
                       f"m2_{i}": f"net.res_blocks.{i}.mlp.2"})
        names, grads = [], []
        for k, base in pn.items():
            names += [base + ".weight", base + ".bias"]
            grads += [G[k + ".w"], G[k + ".b"]]
        for i in range(self.depth):
            names += [f"net.res_blocks.{i}.in_ln.weight", f"net.res_blocks.{i}.in_ln.bias"]
            grads += [G[f"ln{i}.w"], G[f"ln{i}.b"]]
        return names, grads, dz, arena

    # ---------------------------------------------------------------------------------------- sampling
    @torch.no_grad()
    def sample(self, z, temperature=1.0, cfg=1.0, clip_denoised=False, *, noise0: Optional[torch.Tensor] = None,
               step_noises: Optional[torch.Tensor] = None):
        """p_sample_loop over the respaced chain (diffloss.py:37-59, gaussian_diffusion.py:396-493).  cfg != 1 is the
        reference's guidance branch (:39-43, forward_with_cfg :235-243): z = [cond | uncond], the start noise (`noise0`:
        [N/2, C]) is duplicated, the network sees the first half of x twice and eps is mixed inside the reverse-step kernel."""
        dev = z.device
        N, Cc = z.shape[0], self.in_channels
        guided = float(cfg) != 1.0
        if guided and N % 2:
            raise ValueError("classifier-free guidance needs an even batch: z = [cond | uncond]")
        sch = self._gen.on(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        Wb, _, bias = self._weights(stream, False)
        zc = z.detach().contiguous()
        zc = zc if zc.dtype in (F32, BF16) else zc.float()
        if guided:
            half0 = torch.randn(N // 2, Cc, device=dev) if noise0 is None else noise0.to(dev, F32)
            x = torch.cat([half0, half0], dim=0).contiguous()
        else:
            x = (torch.randn(N, Cc, device=dev) if noise0 is None else noise0.to(dev, F32)).contiguous().clone()
        xt_pad = torch.empty(N, _PAD, dtype=BF16, device=dev)
        tfreq = torch.empty(N, 256, dtype=BF16, device=dev)
        for k, i in enumerate(reversed(range(self._gen.n))):
            t = torch.full((N,), i, dtype=torch.long, device=dev)
            xin = torch.cat([x[: N // 2], x[: N // 2]], dim=0) if guided else x
            _lib.call("hma_diff_prepare", stream, ptr(xin), None, ptr(t), None, None, ptr(sch["tmap"]), None, ptr(xt_pad), ptr(tfreq), N, Cc, _PAD)
            out, _ = self._net_forward(stream, Wb, bias, xt_pad, tfreq, zc, keep=False)
            nz = torch.randn(N, Cc, device=dev) if step_noises is None else step_noises[k].to(dev, F32).contiguous()
            if guided:
                _lib.call("hma_diff_p_sample_cfg", stream, ptr(out), _PAD, ptr(x), ptr(nz), ptr(sch["t6"]), self._gen.n, i,
                          float(temperature), 1 if clip_denoised else 0, N, Cc, float(cfg))
            else:
                _lib.call("hma_diff_p_sample", stream, ptr(out), _PAD, ptr(x), ptr(nz), ptr(sch["t6"]), self._gen.n, i,
                          float(temperature), 1 if clip_denoised else 0, N, Cc)
        return x
