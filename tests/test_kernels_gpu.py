"""Op-level parity of every C-ABI entry point against plain fp32 math on the CPU (GPU box only).

Inputs are rounded to bf16 first wherever the kernel consumes bf16, so the tolerances below only
cover accumulation order and the bf16 rounding of OUTPUTS (2^-9 relative)."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hma_amd import _lib, ops  # noqa: E402
from hma_amd._lib import (A_BF16, A_BF16_AFFINE, A_BF16_FRAG32, A_F32, EPI_ATOMIC_F32, EPI_BF16, EPI_DGELU, EPI_DSILU, EPI_F32, EPI_GELU2, EPI_RESID,
                          EPI_SILU2)  # noqa: E402
from oracle import st_maskgit_ref as R  # noqa: E402

DEV = "cuda"


def rb(t):  # round to bf16 and back (fp32 values exactly representable in bf16)
    return t.to(torch.bfloat16).float()


def g(seed):
    return torch.Generator().manual_seed(seed)


def close(a, b, rtol, what=""):
    a = a.float().cpu().double()
    b = b.float().cpu().double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item() + 1e-12
    assert err <= rtol * ref, f"{what}: max err {err:.3e} vs ref scale {ref:.3e} (rtol {rtol})"


BF = 2.0 ** -8  # one bf16 ulp of the output scale, with margin


# ------------------------------------------------------------------------------------------ GEMM NT
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 256, 256), (1000, 768, 256), (384, 256, 1024)])
def test_gemm_nt_plain_and_bias(M, N, K):
    x = rb(torch.randn(M, K, generator=g(1)))
    w = rb(torch.randn(N, K, generator=g(2)) * 0.1)
    b = torch.randn(N, generator=g(3))
    ref = x @ w.t() + b
    y = ops.linear(x.to(DEV).bfloat16(), w.to(DEV).bfloat16(), b.to(DEV), epi=EPI_F32)
    close(y, ref, 1e-5, "f32 out")
    y = ops.linear(x.to(DEV).bfloat16(), w.to(DEV).bfloat16(), None, epi=EPI_BF16)
    close(y, x @ w.t(), BF, "bf16 out")
    # asymmetric check: catches a transposed tile
    assert (y.float().cpu() - (x @ w.t())).abs().max() < (y.float().cpu() - (x @ w.t()).flip(0)).abs().max()


def test_gemm_nt_f32_and_affine_inputs():
    M, N, K = 300, 256, 256
    x = torch.randn(M, K, generator=g(4))
    w = rb(torch.randn(N, K, generator=g(5)) * 0.1)
    y = ops.linear(x.to(DEV), w.to(DEV).bfloat16(), None, epi=EPI_F32)
    close(y, rb(x) @ w.t(), 1e-5, "f32 A")
    gam = torch.randn(K, generator=g(6)) * 0.2 + 1.0
    bet = torch.randn(K, generator=g(7)) * 0.2
    xh = rb(torch.randn(M, K, generator=g(8)))
    y = ops.linear(xh.to(DEV).bfloat16(), w.to(DEV).bfloat16(), None, epi=EPI_F32, gamma=gam.to(DEV), beta=bet.to(DEV))
    close(y, rb(xh * gam + bet) @ w.t(), 2e-4, "affine A")  # fma vs mul+add can flip one bf16 rounding of an input


def test_gemm_nt_epilogues():
    M, N, K = 260, 256, 128
    x = rb(torch.randn(M, K, generator=g(9)))
    w = rb(torch.randn(N, K, generator=g(10)) * 0.2)
    b = torch.randn(N, generator=g(11)) * 0.5
    xd, wd, bd = x.to(DEV).bfloat16(), w.to(DEV).bfloat16(), b.to(DEV)
    lin = x @ w.t() + b
    # residual (+ bf16 copy)
    res = torch.randn(M, N, generator=g(12))
    out = res.to(DEV).clone()
    out2 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.linear(xd, wd, bd, epi=EPI_RESID, out=out, out2=out2)
    close(out, res + lin, 1e-5, "resid")
    close(out2, res + lin, BF, "resid bf16 copy")
    # GELU2 / SILU2: u (rounded) and act(u)
    for epi, act in ((EPI_GELU2, F.gelu), (EPI_SILU2, F.silu)):
        u = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        h = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        ops.linear(xd, wd, bd, epi=epi, out=u, out2=h)
        close(u, lin, BF, "pre-activation")
        close(h, act(u.float().cpu()), BF, "activation of the rounded u")
    # DGELU / DSILU
    uu = rb(torch.randn(M, N, generator=g(13)))
    for epi, act in ((EPI_DGELU, F.gelu), (EPI_DSILU, F.silu)):
        ur = uu.clone().requires_grad_(True)
        act(ur).backward(lin - b)
        y = ops.linear(xd, wd, None, epi=epi, aux=uu.to(DEV).bfloat16())
        close(y, ur.grad, 2 * BF, "d-activation")
    # atomic accumulate
    acc = torch.ones(M, N, device=DEV)
    ops.linear(xd, wd, None, epi=EPI_ATOMIC_F32, out=acc)
    close(acc, 1 + (lin - b), 1e-5, "atomic")


@pytest.mark.parametrize("M,K", [(256, 1024), (384, 768), (1000, 1024), (8192 + 40, 768), (70000, 512)])
def test_gemm_nt_ring(M, K):
    """Deep-K, N = 256 GEMMs with plain bf16 operands take the LDS-DMA ring kernel (gemm_nt_ring_kernel): bf16 output,
    residual accumulate with bias, bf16 copy and a remapped output; partial tiles and uneven row runs."""
    N = 256
    x = rb(torch.randn(M, K, generator=g(40)))
    w = rb(torch.randn(N, K, generator=g(41)) * 0.1)
    b = torch.randn(N, generator=g(42))
    xd, wd = x.to(DEV).bfloat16(), w.to(DEV).bfloat16()
    lin = (x.double() @ w.double().t()).float()
    y = ops.linear(xd, wd, None, epi=EPI_BF16)
    close(y, lin, BF, "ring bf16 out")
    assert (y.float().cpu() - lin).abs().max() < (y.float().cpu() - lin.flip(0)).abs().max()
    res = torch.randn(M, N, generator=g(43))
    out = res.to(DEV).clone()
    out2 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.linear(xd, wd, b.to(DEV), epi=EPI_RESID, out=out, out2=out2)
    close(out, res + lin + b, 1e-5, "ring resid")
    close(out2, res + lin + b, BF, "ring resid bf16 copy")
    # output rows remapped into (S + A)-row frames
    if M % 64 == 0:
        S, SA = 64, 80
        frames = M // S
        outr = torch.zeros(frames * SA, N, device=DEV)
        gm = ops.make_gemm_nt(A=xd.data_ptr(), lda=K, a_kind=A_BF16, W=wd.data_ptr(), ldw=K, M=M, N=N, K=K, epi=EPI_RESID,
                              Cp=outr.data_ptr(), ldc=N, c_group=(S, SA))
        _lib.call("hma_gemm_nt", ops.stream_ptr(), C.byref(gm))
        full = torch.zeros(frames, SA, N)
        full[:, :S] = lin.reshape(frames, S, N)
        close(outr, full.reshape(-1, N), 1e-5, "ring c remap")


def test_fold_ln_into_linear():
    """hma_fold_ln_bf16: Linear(xhat * gamma + beta) == xhat @ Wf^T + bf, batched over per-layer blocks of one flat buffer."""
    L, N, K = 3, 768, 256
    stride = N * K + 2 * K + N + 64                      # [W | gamma | beta | bias | pad] per layer, layers at a constant stride
    flat = torch.randn(L * stride, generator=g(50))
    flat_d = flat.to(DEV)
    Wf = torch.zeros(L, N, K, dtype=torch.bfloat16, device=DEV)
    bf = torch.zeros(L, N, device=DEV)
    base = flat_d.data_ptr()
    _lib.call("hma_fold_ln_bf16", ops.stream_ptr(), base, base + 4 * N * K, base + 4 * (N * K + K), base + 4 * (N * K + 2 * K),
              Wf.data_ptr(), bf.data_ptr(), N, K, L, stride, N * K, N)
    for l in range(L):
        blk = flat[l * stride:(l + 1) * stride]
        W, gam, bet, bias = blk[:N * K].view(N, K), blk[N * K:N * K + K], blk[N * K + K:N * K + 2 * K], blk[N * K + 2 * K:N * K + 2 * K + N]
        close(Wf[l], W * gam, BF, "folded weight")
        close(bf[l], bias + W @ bet, 1e-5, "folded bias")
    bf.fill_(7.0)
    _lib.call("hma_fold_ln_bf16", ops.stream_ptr(), base, base + 4 * N * K, base + 4 * (N * K + K), None,
              Wf.data_ptr(), bf.data_ptr(), N, K, 1, 0, 0, 0)
    close(bf[0], flat[:N * K].view(N, K) @ flat[N * K + K:N * K + 2 * K], 1e-5, "folded bias, no Linear bias")


def test_gemm_nt_row_remap_and_batch():
    # A rows sliced out of (S + A)-row frames, C written back remapped
    frames, S, SA, K, N = 3, 64, 80, 256, 128
    xa = torch.randn(frames * SA, K, generator=g(14))
    w = rb(torch.randn(N, K, generator=g(15)) * 0.1)
    ref = rb(xa).reshape(frames, SA, K)[:, :S].reshape(-1, K) @ w.t()
    out = torch.empty(frames * S, N, device=DEV)
    xa_d, w_d = xa.to(DEV), w.to(DEV).bfloat16()
    gm = ops.make_gemm_nt(A=xa_d.data_ptr(), lda=K, a_kind=A_F32, W=w_d.data_ptr(), ldw=K, M=frames * S, N=N, K=K,
                          epi=EPI_F32, Cp=out.data_ptr(), ldc=N, a_group=(S, SA))
    _lib.call("hma_gemm_nt", ops.stream_ptr(), C.byref(gm))
    close(out, ref, 1e-5, "a remap")
    out2 = torch.zeros(frames * SA, N, device=DEV)
    gm.C, gm.c_group_rows, gm.c_group_stride = out2.data_ptr(), S, SA
    _lib.call("hma_gemm_nt", ops.stream_ptr(), C.byref(gm))
    full = torch.zeros(frames, SA, N)
    full[:, :S] = ref.reshape(frames, S, N)
    close(out2, full.reshape(-1, N), 1e-5, "c remap")
    # batched: shared A, per-batch W / bias / C
    Bz, M = 3, 70
    x = rb(torch.randn(M, K, generator=g(16)))
    ws = rb(torch.randn(Bz, N, K, generator=g(17)) * 0.1)
    bs = torch.randn(Bz, N, generator=g(18))
    x_d, ws_d, bs_d = x.to(DEV).bfloat16(), ws.to(DEV).bfloat16(), bs.to(DEV)
    outb = torch.empty(Bz, M, N, device=DEV)
    gb = ops.make_gemm_nt(A=x_d.data_ptr(), lda=K, a_kind=A_BF16, W=ws_d.data_ptr(), ldw=K, M=M, N=N, K=K, epi=EPI_F32,
                          Cp=outb.data_ptr(), ldc=N, bias=bs_d.data_ptr(), batch=Bz, sA=0, sW=N * K, sBias=N, sC=M * N)
    _lib.call("hma_gemm_nt", ops.stream_ptr(), C.byref(gb))
    close(outb, torch.einsum("mk,bnk->bmn", x, ws) + bs[:, None], 1e-5, "batched")


# ------------------------------------------------------------------------------------------ GEMM TN
@pytest.mark.parametrize("M,N,K,splits", [(64, 128, 128, 1), (1000, 256, 256, 0), (1920, 768, 256, 7), (333, 256, 1024, 3)])
def test_gemm_tn(M, N, K, splits):
    dy = rb(torch.randn(M, N, generator=g(20)))
    x = rb(torch.randn(M, K, generator=g(21)))
    dW = torch.ones(N, K, device=DEV)
    db = torch.ones(N, device=DEV)
    ops.linear_wgrad(dy.to(DEV).bfloat16(), x.to(DEV).bfloat16(), dW, db, splits=splits)
    close(dW, 1 + dy.t() @ x, 2e-5, "dW")
    close(db, 1 + dy.sum(0), 2e-5, "dbias")
    # fp32 sources are rounded to bf16 on the way in; bias grad uses the fp32 values
    dyf = torch.randn(M, N, generator=g(22))
    xf = torch.randn(M, K, generator=g(23))
    dW.zero_(); db.zero_()
    ops.linear_wgrad(dyf.to(DEV), xf.to(DEV), dW, db, splits=splits)
    close(dW, rb(dyf).t() @ rb(xf), 2e-5, "dW f32 sources")
    close(db, dyf.sum(0), 2e-5, "dbias f32")
    # LN affine on the activation operand
    gam = torch.randn(K, generator=g(24)) * 0.2 + 1
    bet = torch.randn(K, generator=g(25)) * 0.2
    dW.zero_()
    ops.linear_wgrad(dy.to(DEV).bfloat16(), x.to(DEV).bfloat16(), dW, None, gamma=gam.to(DEV), beta=bet.to(DEV), splits=splits)
    close(dW, dy.t() @ rb(x * gam + bet), 2e-5, "dW affine")
    # two-stage (workspace) reduction instead of atomics: the per-workgroup partials are bf16 (summed in fp32),
    # the precision class of the reference's autocast backward, which returns the whole dW in bf16
    # (F.linear under torch.autocast, hma/train_multi.py mixed_precision="bf16"); bias partials stay fp32
    ws = torch.full((256 * (65536 + 256),), float("nan"), device=DEV)
    dW.fill_(1.0); db.fill_(1.0)
    ops.linear_wgrad(dy.to(DEV).bfloat16(), x.to(DEV).bfloat16(), dW, db, ws=ws)
    close(dW, 1 + dy.t() @ x, BF, "dW via workspace")
    close(db, 1 + dy.sum(0), 2e-5, "dbias via workspace")
    # a workspace without room for the bias partials falls back to atomics for them
    ws2 = torch.full((256 * 65536,), float("nan"), device=DEV)
    dW.fill_(1.0); db.fill_(1.0)
    ops.linear_wgrad(dy.to(DEV).bfloat16(), x.to(DEV).bfloat16(), dW, db, ws=ws2)
    close(dW, 1 + dy.t() @ x, BF, "dW via small workspace")
    close(db, 1 + dy.sum(0), 2e-5, "dbias via small workspace")


@pytest.mark.parametrize("M,N,K", [(32, 256, 256), (96, 256, 256), (4096, 256, 256), (2080, 768, 256), (8192, 1024, 256),
                                   (4000 * 32, 256, 1024), (16384, 256, 768)])
def test_gemm_tn_ring(M, N, K):
    """bf16 x bf16 wgrads with a workspace and M % 32 == 0 take the LDS-DMA ring kernel (gemm_tn_dma_kernel): plain,
    with the bias gradient, and with the LayerNorm affine that the reduction kernel applies afterwards."""
    dy = rb(torch.randn(M, N, generator=g(30)) * 0.5)
    x = rb(torch.randn(M, K, generator=g(31)))
    dyd, xd = dy.to(DEV).bfloat16(), x.to(DEV).bfloat16()
    ws = torch.full((256 * (65536 + 256),), float("nan"), device=DEV)
    ref = (dy.double().t() @ x.double()).float()
    dW = torch.ones(N, K, device=DEV)
    db = torch.ones(N, device=DEV)
    ops.linear_wgrad(dyd, xd, dW, db, ws=ws)
    close(dW, 1 + ref, BF, "dW ring")
    close(db, 1 + dy.double().sum(0).float(), 2e-5, "dbias ring")
    dW.fill_(-2.0)
    ops.linear_wgrad(dyd, xd, dW, None, ws=ws)
    close(dW, ref - 2, BF, "dW ring, no bias")
    gam = torch.randn(K, generator=g(32)) * 0.2 + 1
    bet = torch.randn(K, generator=g(33)) * 0.2
    refa = (dy.double().t() @ (x.double() * gam.double() + bet.double())).float()
    dW.fill_(0.5); db.fill_(0.25)
    ops.linear_wgrad(dyd, xd, dW, db, gamma=gam.to(DEV), beta=bet.to(DEV), ws=ws)
    close(dW, 0.5 + refa, BF, "dW ring affine")
    close(db, 0.25 + dy.double().sum(0).float(), 2e-5, "dbias ring affine")
    dW.fill_(0.5)
    ops.linear_wgrad(dyd, xd, dW, None, gamma=gam.to(DEV), beta=bet.to(DEV), ws=ws)
    close(dW, 0.5 + refa, BF, "dW ring affine, no bias")


@pytest.mark.parametrize("M", [4096, 32 * 130, 40, 163840])   # (163 840: the headline step's rows -- every problem split over several M-slices)
def test_gemm_tn_pair(M):
    """hma_gemm_tn_pair: two weight gradients in one launch (MLP pair, fc1 with the deferred LayerNorm affine, and a
    projection / qkv pair without a qkv bias) give what two hma_gemm_tn calls give; M = 40 is not eligible for the ring
    kernel and must fall back to two calls."""
    ws = torch.full((256 * (65536 + 256),), float("nan"), device=DEV)

    def prob(N, K, seed, bias=True, affine=False):
        dy = rb(torch.randn(M, N, generator=g(seed)) * 0.5)
        x = rb(torch.randn(M, K, generator=g(seed + 1)))
        gam = torch.randn(K, generator=g(seed + 2)) * 0.2 + 1 if affine else None
        bet = torch.randn(K, generator=g(seed + 3)) * 0.2 if affine else None
        xe = x.double() * gam.double() + bet.double() if affine else x.double()
        t = dict(dy=dy.to(DEV).bfloat16(), x=x.to(DEV).bfloat16(), dW=torch.full((N, K), 0.5, device=DEV),
                 db=torch.full((N,), 0.25, device=DEV) if bias else None, gam=None if gam is None else gam.to(DEV),
                 bet=None if bet is None else bet.to(DEV), refW=0.5 + (dy.double().t() @ xe).float(),
                 refb=0.25 + dy.double().sum(0).float())
        t["g"] = ops.make_gemm_tn(dY=ops.ptr(t["dy"]), ldy=N, y_kind=A_BF16, A=ops.ptr(t["x"]), lda=K,
                                  a_kind=A_BF16_AFFINE if affine else A_BF16, M=M, N=N, K=K, dW=ops.ptr(t["dW"]), lddw=K,
                                  dBias=ops.ptr(t["db"]), gamma=ops.ptr(t["gam"]), beta=ops.ptr(t["bet"]), ws=ops.ptr(ws),
                                  ws_elems=ws.numel())
        return t

    for pa, pb in ((prob(256, 1024, 60), prob(1024, 256, 64, affine=True)), (prob(256, 256, 70), prob(768, 256, 74, bias=False)),
                   (prob(768, 256, 80, bias=False, affine=True), prob(256, 256, 84))):
        _lib.call("hma_gemm_tn_pair", ops.stream_ptr(), C.byref(pa["g"]), C.byref(pb["g"]))
        for t in (pa, pb):
            close(t["dW"], t["refW"], BF, "paired dW")
            if t["db"] is not None:
                close(t["db"], t["refb"], 2e-5, "paired dbias")


def test_gemm_tn_multi_seven_problems():
    """hma_gemm_tn_multi: the seven weight gradients of an STBlock (MLP pair with the fragment-ordered fc operands left out here: plain
    bf16 operands of the same shapes; fc1 / qkv_s with the deferred LayerNorm affine + its dgamma / dbeta; qkv_s's dY head-blocked) in one
    launch give what seven hma_gemm_tn calls give."""
    from hma_amd._lib import A_BF16_HEADBLK
    n, frames = 320, 8
    M = frames * n
    ws = torch.full((256 * (65536 + 256),), float("nan"), device=DEV)
    shapes = [(256, 1024, True, False, False), (1024, 256, True, True, False), (256, 256, True, False, False), (768, 256, False, False, False),
              (256, 256, True, False, False), (256, 256, True, False, False), (768, 256, False, True, True)]
    res = {}
    for mode in ("multi", "single"):
        gs, outs = [], []
        for i, (N, K, bias, affine, hb) in enumerate(shapes):
            dy = rb(torch.randn(M, N, generator=g(200 + 10 * i)) * 0.5).to(DEV).bfloat16()
            x = rb(torch.randn(M, K, generator=g(201 + 10 * i))).to(DEV).bfloat16()
            gam = (torch.randn(K, generator=g(202 + 10 * i)) * 0.2 + 1).to(DEV) if affine else None
            bet = (torch.randn(K, generator=g(203 + 10 * i)) * 0.2).to(DEV) if affine else None
            wm = torch.randn(N, K, generator=g(204 + 10 * i)).to(DEV) if affine else None
            dW, db = torch.zeros(N, K, device=DEV), (torch.zeros(N, device=DEV) if bias else None)
            dg, dbt = (torch.zeros(K, device=DEV), torch.zeros(K, device=DEV)) if affine else (None, None)
            dyb = ops.rows_to_headblk(dy, n) if hb else dy
            gs.append(ops.make_gemm_tn(dY=ops.ptr(dyb), ldy=N, y_kind=A_BF16_HEADBLK if hb else A_BF16, y_group=(n, 0) if hb else (0, 0),
                                       A=ops.ptr(x), lda=K, a_kind=A_BF16_AFFINE if affine else A_BF16, M=M, N=N, K=K, dW=ops.ptr(dW), lddw=K,
                                       dBias=ops.ptr(db), gamma=ops.ptr(gam), beta=ops.ptr(bet), w_master=ops.ptr(wm), dgamma=ops.ptr(dg),
                                       dbeta=ops.ptr(dbt), ws=ops.ptr(ws), ws_elems=ws.numel()))
            outs.append((dW, db, dg, dbt, dy, x, gam, bet, wm, dyb))  # (wm / dyb: kept alive until the launches below have run)
        if mode == "multi":
            arr = (C.POINTER(type(gs[0])) * len(gs))(*[C.pointer(q) for q in gs])
            _lib.call("hma_gemm_tn_multi", ops.stream_ptr(), arr, len(gs))
        else:
            for q in gs:
                _lib.call("hma_gemm_tn", ops.stream_ptr(), C.byref(q))
        torch.cuda.synchronize()
        res[mode] = outs
    for i, ((dW, db, dg, dbt, dy, x, gam, bet, *_keep), (dW1, db1, dg1, dbt1, *_rest)) in enumerate(zip(res["multi"], res["single"])):
        xe = x.double() * gam.double() + bet.double() if gam is not None else x.double()
        close(dW, (dy.double().t() @ xe).float(), BF, f"multi dW, problem {i}")
        close(dW, dW1, BF, f"multi vs single dW, problem {i}")
        if db is not None:
            close(db, dy.double().sum(0).float(), 2e-5, f"multi dbias, problem {i}")
        if dg is not None:
            close(dg, dg1, 2e-2, f"multi dgamma, problem {i}")
            close(dbt, dbt1, 2e-2, f"multi dbeta, problem {i}")


def test_gemm_tn_remap_and_batch():
    frames, S, SA, N, K = 3, 64, 80, 128, 256
    dy = rb(torch.randn(frames * S, N, generator=g(26)))
    xa = torch.randn(frames * SA, K, generator=g(27))
    dW = torch.zeros(N, K, device=DEV)
    dy_d, xa_d = dy.to(DEV).bfloat16(), xa.to(DEV)
    gt = ops.make_gemm_tn(dY=dy_d.data_ptr(), ldy=N, y_kind=A_BF16, A=xa_d.data_ptr(), lda=K, a_kind=A_F32, M=frames * S,
                          N=N, K=K, dW=dW.data_ptr(), lddw=K, a_group=(S, SA), splits=2)
    _lib.call("hma_gemm_tn", ops.stream_ptr(), C.byref(gt))
    close(dW, dy.t() @ rb(xa).reshape(frames, SA, K)[:, :S].reshape(-1, K), 2e-5, "tn remap")
    Bz, M = 3, 70
    dys = torch.randn(Bz, M, N, generator=g(28))
    x = torch.randn(M, K, generator=g(29))
    dWs = torch.zeros(Bz, N, K, device=DEV)
    dbs = torch.zeros(Bz, N, device=DEV)
    dys_d, x_d = dys.to(DEV), x.to(DEV)
    gt = ops.make_gemm_tn(dY=dys_d.data_ptr(), ldy=N, y_kind=A_F32, A=x_d.data_ptr(), lda=K, a_kind=A_F32, M=M, N=N, K=K,
                          dW=dWs.data_ptr(), lddw=K, dBias=dbs.data_ptr(), batch=Bz, sY=M * N, sA=0, sdW=N * K, sdBias=N)
    _lib.call("hma_gemm_tn", ops.stream_ptr(), C.byref(gt))
    close(dWs, torch.einsum("bmn,mk->bnk", rb(dys), rb(x)), 2e-5, "tn batched")
    close(dbs, dys.sum(1), 2e-5, "tn batched bias")


# ------------------------------------------------------------------------------------------ norms
def test_layernorm_fwd_bwd():
    rows = 777
    x = torch.randn(rows, 256, generator=g(30)) * 2 + 0.5
    xhat, rstd = ops.ln_fwd(x.to(DEV), 1e-5)
    ref = F.layer_norm(x, (256,), None, None, 1e-5)
    close(xhat, ref, BF, "xhat")
    close(rstd, 1 / torch.sqrt(x.var(1, unbiased=False) + 1e-5), 1e-5, "rstd")
    gam = torch.randn(256, generator=g(31)) * 0.3 + 1
    dxn = rb(torch.randn(rows, 256, generator=g(32)))
    xr = x.clone().requires_grad_(True)
    gr = gam.clone().requires_grad_(True)
    br = torch.zeros(256, requires_grad=True)
    F.layer_norm(xr, (256,), gr, br, 1e-5).backward(dxn)
    dx = torch.ones(rows, 256, device=DEV)
    dg = torch.zeros(256, device=DEV)
    db = torch.zeros(256, device=DEV)
    dxb = torch.empty(rows, 256, dtype=torch.bfloat16, device=DEV)
    dxn_d, gam_d = dxn.to(DEV).bfloat16(), gam.to(DEV)  # kept alive: a temporary's block can be re-used by the next one
    _lib.call("hma_ln_bwd", ops.stream_ptr(), dxn_d.data_ptr(), xhat.data_ptr(), rstd.data_ptr(),
              gam_d.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), rows, dxb.data_ptr())
    close(dx, 1 + xr.grad, 2 * BF, "ln dx")
    assert torch.equal(dxb, dx.bfloat16()), "bf16 copy of dx"
    close(dg, gr.grad, 2 * BF, "dgamma")
    close(db, br.grad, 1e-4, "dbeta")


def test_modulate_ln_fwd_bwd():
    frames, rpf = 5, 320
    x = torch.randn(frames * rpf, 256, generator=g(33)) * 1.5
    ss = torch.randn(frames, 512, generator=g(34)) * 0.3
    xd, ssd = x.to(DEV), ss.to(DEV)
    xhat = torch.empty(frames * rpf, 256, dtype=torch.bfloat16, device=DEV)
    xm = torch.empty_like(xhat)
    rstd = torch.empty(frames * rpf, device=DEV)
    _lib.call("hma_modln_fwd", ops.stream_ptr(), xd.data_ptr(), ssd.data_ptr(), xhat.data_ptr(), xm.data_ptr(),
              rstd.data_ptr(), frames, rpf, 1e-6)
    xr = x.clone().requires_grad_(True)
    sr = ss.clone().requires_grad_(True)
    shift, scale = sr[:, None, :256], sr[:, None, 256:]
    y = F.layer_norm(xr.reshape(frames, rpf, 256), (256,), None, None, 1e-6) * (1 + scale) + shift
    close(xm, y.detach().reshape(-1, 256), BF, "xm")
    dxm = rb(torch.randn(frames * rpf, 256, generator=g(35)))
    y.backward(dxm.reshape(frames, rpf, 256))
    dx = torch.zeros(frames * rpf, 256, device=DEV)
    dss = torch.empty(frames, 512, device=DEV)
    dxb = torch.empty(frames * rpf, 256, dtype=torch.bfloat16, device=DEV)
    dxm_d = dxm.to(DEV).bfloat16()
    _lib.call("hma_modln_bwd", ops.stream_ptr(), dxm_d.data_ptr(), xhat.data_ptr(), rstd.data_ptr(),
              ssd.data_ptr(), dx.data_ptr(), dss.data_ptr(), frames, rpf, dxb.data_ptr())
    close(dx, xr.grad, 2 * BF, "modln dx")
    assert torch.equal(dxb, dx.bfloat16()), "bf16 copy of dx"
    close(dss, sr.grad, 2 * BF, "dss")


# ------------------------------------------------------------------------------------------ attention
def _ref_attn(qkv, n_seq, n, scale, causal):
    """fp32 attention on packed (rows, 768) qkv, sequences of n rows."""
    q, k, v = qkv.reshape(n_seq, n, 3, 8, 32).permute(2, 0, 3, 1, 4)
    att = (q * scale) @ k.transpose(-1, -2)
    if causal:
        att = att.masked_fill(~torch.ones(n, n, dtype=torch.bool).tril(), -torch.finfo(att.dtype).max)
    return (att.softmax(-1) @ v).transpose(1, 2).reshape(n_seq * n, 256)


@pytest.mark.parametrize("frames,n", [(3, 320), (9, 256), (2, 64)])
def test_attn_spatial(frames, n):
    scale = 0.25
    qkv = rb(torch.randn(frames * n, 768, generator=g(40)))
    qr = qkv.clone().requires_grad_(True)
    ref = _ref_attn(qr, frames, n, scale, False)
    o, lse = ops.attn_spatial_fwd(qkv.to(DEV).bfloat16(), frames, n, scale)
    close(o, ref.detach(), 2 * BF, "spatial fwd")
    d_o = rb(torch.randn(frames * n, 256, generator=g(41)))
    ref.backward(d_o)
    dqkv = ops.attn_spatial_bwd(qkv.to(DEV).bfloat16(), o, d_o.to(DEV).bfloat16(), lse, frames, n, scale)
    for name, sl in (("dq", slice(0, 256)), ("dk", slice(256, 512)), ("dv", slice(512, 768))):
        close(dqkv[:, sl], qr.grad[:, sl], 4 * BF, f"spatial {name}")


@pytest.mark.parametrize("frames,n", [(3, 320), (20, 320), (9, 256), (2, 64)])
def test_attn_spatial_bwd_blocked_equals_row_major(frames, n):
    """hma_attn_spatial_bwd_blocked writes the same gradients in the head-blocked order (HMA_A_BF16_HEADBLK): bit-identical values,
    [frame][head][q | k | v][n][32] instead of [row][768]."""
    scale = 0.25
    qkv = rb(torch.randn(frames * n, 768, generator=g(140))).to(DEV).bfloat16()
    d_o = rb(torch.randn(frames * n, 256, generator=g(141))).to(DEV).bfloat16()
    o, lse = ops.attn_spatial_fwd(qkv, frames, n, scale)
    rows = ops.attn_spatial_bwd(qkv, o, d_o, lse, frames, n, scale)
    blk = ops.attn_spatial_bwd(qkv, o, d_o, lse, frames, n, scale, blocked=True)
    torch.cuda.synchronize()
    assert torch.equal(ops.headblk_to_rows(blk, n), rows)
    assert torch.equal(ops.rows_to_headblk(rows, n), blk)


@pytest.mark.parametrize("frames,n", [(4, 320), (7, 64)])
def test_gemm_tn_headblocked_operand(frames, n):
    """hma_gemm_tn / _pair with dY (the spatial qkv gradient, W = 3) or A (W = 1) in the head-blocked order: the weight gradient the
    row-major operand gives, bit for bit (only the LDS-DMA source addresses differ)."""
    from hma_amd._lib import A_BF16_HEADBLK
    M = frames * n
    ws = torch.full((256 * (65536 + 256),), float("nan"), device=DEV)
    dy = rb(torch.randn(M, 768, generator=g(150)) * 0.5).to(DEV).bfloat16()
    x = rb(torch.randn(M, 256, generator=g(151))).to(DEV).bfloat16()
    gam, bet = (torch.randn(256, generator=g(152)) * 0.2 + 1).to(DEV), (torch.randn(256, generator=g(153)) * 0.2).to(DEV)
    out = {}
    for hb in (False, True):
        dW, db = torch.zeros(768, 256, device=DEV), torch.zeros(768, device=DEV)
        dW2, db2 = torch.zeros(256, 256, device=DEV), torch.zeros(256, device=DEV)
        dyb = ops.rows_to_headblk(dy, n) if hb else dy
        xb = ops.rows_to_headblk(x, n) if hb else x
        ga = ops.make_gemm_tn(dY=ops.ptr(dyb), ldy=768, y_kind=A_BF16_HEADBLK if hb else A_BF16, y_group=(n, 0) if hb else (0, 0), A=ops.ptr(x), lda=256,
                              a_kind=A_BF16_AFFINE, gamma=ops.ptr(gam), beta=ops.ptr(bet), M=M, N=768, K=256, dW=ops.ptr(dW), lddw=256,
                              dBias=ops.ptr(db), ws=ops.ptr(ws), ws_elems=ws.numel())
        dy256 = dy[:, :256].contiguous()  # (bound to a name: the launch reads it after this statement)
        gb = ops.make_gemm_tn(dY=ops.ptr(dy256), ldy=256, y_kind=A_BF16, A=ops.ptr(xb), lda=256,
                              a_kind=A_BF16_HEADBLK if hb else A_BF16, a_group=(n, 0) if hb else (0, 0), M=M, N=256, K=256, dW=ops.ptr(dW2),
                              lddw=256, dBias=ops.ptr(db2), ws=ops.ptr(ws), ws_elems=ws.numel())
        _lib.call("hma_gemm_tn_pair", ops.stream_ptr(), C.byref(ga), C.byref(gb))
        torch.cuda.synchronize()
        out[hb] = (dW, db, dW2, db2)
    for a, b in zip(out[False], out[True]):
        assert torch.equal(a, b)
    ref = dy.double().t() @ (x.double() * gam.double() + bet.double())
    close(out[True][0], ref.float(), BF, "head-blocked dW")


@pytest.mark.parametrize("B,T,n_s", [(2, 16, 5), (3, 3, 7), (1, 1, 4), (2, 12, 5)])
def test_attn_temporal(B, T, n_s):
    scale = 0.25
    rows = B * T * n_s
    qkv = rb(torch.randn(rows, 768, generator=g(42)))
    # rows are (b, t, s): gather each column's T rows for the reference
    idx = torch.arange(rows).reshape(B, T, n_s).permute(0, 2, 1).reshape(-1)
    qr = qkv.clone().requires_grad_(True)
    ref_cols = _ref_attn(qr[idx], B * n_s, T, scale, True)
    ref = torch.empty(rows, 256)
    ref[idx] = ref_cols.detach()
    o = ops.attn_temporal_fwd(qkv.to(DEV).bfloat16(), B, T, n_s, scale)
    close(o, ref, 2 * BF, "temporal fwd")
    d_o = rb(torch.randn(rows, 256, generator=g(43)))
    ref_cols.backward(d_o[idx])
    dqkv = ops.attn_temporal_bwd(qkv.to(DEV).bfloat16(), o, d_o.to(DEV).bfloat16(), B, T, n_s, scale)
    close(dqkv, qr.grad, 4 * BF, "temporal dqkv")


# ------------------------------------------------------------------------------------------ embedding / stem
@pytest.mark.parametrize("B,T,id_range", [(2, 3, 512 * 512), (5, 7, 512 * 512), (9, 4, 8192)])
def test_embed_fwd_bwd(B, T, id_range):
    # (the larger cases take the LDS-atomic table-gradient kernel; ids < 8192 puts every token on 16 rows of table 1)
    S, A, V = 256, 64, 512
    mask_id = V * V
    ids = torch.randint(0, id_range, (B, T, S), generator=g(50))
    ids[torch.rand(B, T, S, generator=g(51)) < 0.4] = mask_id
    E0 = torch.randn(V, 256, generator=g(52)); E1 = torch.randn(V, 256, generator=g(53))
    me = torch.randn(1, 256, generator=g(54)); pos = torch.randn(T + 1, S + A, 256, generator=g(55))
    a_emb = torch.randn(B, T, 256, generator=g(56))
    P = {k: v.clone().requires_grad_(True) for k, v in dict(E0=E0, E1=E1, me=me, pos=pos, a=a_emb).items()}
    sd = {"token_embed.factored_embeds.0.weight": P["E0"], "token_embed.factored_embeds.1.weight": P["E1"],
          "token_embed.mask_token_embed": P["me"]}
    cfg = R.RefConfig(num_layers=1, num_heads=8, d_model=256, T=T)
    ref = torch.cat([R.token_embed(sd, cfg, ids), P["a"][:, :, None].expand(B, T, A, 256)], 2) + P["pos"][None, :T]
    x = torch.empty(B, T, S + A, 256, device=DEV)
    d = lambda t: t.to(DEV).contiguous()
    ids_d, E0d, E1d, med, posd, ad = d(ids), d(E0), d(E1), d(me), d(pos), d(a_emb)
    _lib.call("hma_embed_fwd", ops.stream_ptr(), ids_d.data_ptr(), E0d.data_ptr(), E1d.data_ptr(), med.data_ptr(),
              posd.data_ptr(), ad.data_ptr(), x.data_ptr(), B, T, S, A, S + A, V, mask_id)
    assert torch.equal(x.cpu(), ref.detach())
    dx = torch.randn(B, T, S + A, 256, generator=g(57))
    ref.backward(dx)
    G = {k: torch.zeros_like(v, device=DEV) for k, v in dict(E0=E0, E1=E1, me=me, pos=pos, a=a_emb).items()}
    dxd = d(dx)
    _lib.call("hma_embed_bwd", ops.stream_ptr(), ids_d.data_ptr(), dxd.data_ptr(), G["E0"].data_ptr(), G["E1"].data_ptr(),
              G["me"].data_ptr(), G["pos"].data_ptr(), G["a"].data_ptr(), B, T, S, A, S + A, V, mask_id)
    for k in G:
        close(G[k], P[k].grad, 1e-5, f"embed grad {k}")


@pytest.mark.parametrize("d_a,adim", [(7, 7), (14, 7), (70, 7)])
def test_action_stem(d_a, adim):
    rows = 6
    cfg = R.RefConfig(num_layers=1, num_heads=8, d_model=256, T=3)
    gen = g(60)
    sd = {"action_preprocessor.d.mean": torch.randn(adim, generator=gen), "action_preprocessor.d.std": torch.rand(adim, generator=gen) + 0.5,
          "action_mlp.d.model.0.weight": torch.randn(256, d_a, generator=gen) * 0.3, "action_mlp.d.model.0.bias": torch.randn(256, generator=gen) * 0.1,
          "action_mlp.d.model.1.weight": torch.randn(256, generator=gen) * 0.2 + 1, "action_mlp.d.model.1.bias": torch.randn(256, generator=gen) * 0.1,
          "action_mlp.d.model.3.weight": torch.randn(256, 256, generator=gen) * 0.1, "action_mlp.d.model.3.bias": torch.randn(256, generator=gen) * 0.1}
    a = torch.randn(2, 3, d_a, generator=gen)
    leaf = {k: v.clone().requires_grad_(k.startswith("action_mlp")) for k, v in sd.items()}
    ref = R.action_stem(leaf, cfg, a, "d")
    dd = {k: v.to(DEV).contiguous() for k, v in sd.items()}
    an = torch.empty(rows, d_a, device=DEV); xh = torch.empty(rows, 256, device=DEV); rs = torch.empty(rows, device=DEV)
    h = torch.empty(rows, 256, device=DEV); out = torch.empty(rows, 256, device=DEV)
    a_d = a.to(DEV).contiguous()
    p = "action_mlp.d.model"
    _lib.call("hma_action_stem_fwd", ops.stream_ptr(), a_d.data_ptr(), dd["action_preprocessor.d.mean"].data_ptr(),
              dd["action_preprocessor.d.std"].data_ptr(), adim, dd[f"{p}.0.weight"].data_ptr(), dd[f"{p}.0.bias"].data_ptr(),
              dd[f"{p}.1.weight"].data_ptr(), dd[f"{p}.1.bias"].data_ptr(), dd[f"{p}.3.weight"].data_ptr(),
              dd[f"{p}.3.bias"].data_ptr(), an.data_ptr(), xh.data_ptr(), rs.data_ptr(), h.data_ptr(), out.data_ptr(), rows, d_a, 0)
    close(out, ref.detach().reshape(rows, 256), 1e-5, "stem fwd")
    dout = torch.randn(rows, 256, generator=gen)
    ref.backward(dout.reshape(2, 3, 256))
    Gd = {k: torch.zeros_like(v) for k, v in dd.items() if k.startswith(p)}
    scratch = torch.empty(rows, 256, device=DEV)
    dout_d = dout.to(DEV)
    _lib.call("hma_action_stem_bwd", ops.stream_ptr(), dout_d.data_ptr(), an.data_ptr(), xh.data_ptr(), rs.data_ptr(),
              h.data_ptr(), dd[f"{p}.1.weight"].data_ptr(), dd[f"{p}.3.weight"].data_ptr(), Gd[f"{p}.0.weight"].data_ptr(),
              Gd[f"{p}.0.bias"].data_ptr(), Gd[f"{p}.1.weight"].data_ptr(), Gd[f"{p}.1.bias"].data_ptr(),
              Gd[f"{p}.3.weight"].data_ptr(), Gd[f"{p}.3.bias"].data_ptr(), scratch.data_ptr(), rows, d_a)
    for k in Gd:
        close(Gd[k], leaf[k].grad, 2e-5, f"stem grad {k}")


# ------------------------------------------------------------------------------------------ loss / sampling
def test_ce_loss_acc_and_grad():
    B, T, S = 2, 3, 256
    mask_id = 262144
    gen = g(70)
    logits = torch.randn(B, T, S, 1024, generator=gen) * 2
    labels = torch.randint(0, 262144, (B, T * S), generator=gen)
    inputs = labels.clone().reshape(B, T, S)
    inputs[:, 1:][torch.rand(B, T - 1, S, generator=gen) < 0.5] = mask_id
    # make some rows "correct" so the accuracy is not trivially zero
    fl = R.factorize_token_ids(labels.reshape(B, T, S), 2, 512)
    for b, t, s in [(0, 1, 3), (1, 2, 100), (0, 2, 255)]:
        logits[b, t, s, fl[b, t, s, 0]] = 30.0
        logits[b, t, s, 512 + fl[b, t, s, 1]] = 30.0
        inputs[b, t, s] = mask_id
    cfg = R.RefConfig(num_layers=1, num_heads=8, d_model=256, T=T)
    lr = logits.clone().requires_grad_(True)
    loss, acc = R.video_loss_and_acc(cfg, lr.reshape(B, T, 16, 16, 1024).permute(0, 4, 1, 2, 3), labels,
                                     inputs.reshape(B, T, 16, 16))
    (loss * 3.0).backward()
    stats = torch.zeros(8, device=DEV)  # HMA_CE_STATS_FLOATS
    dl = torch.empty(B * T * S, 1024, dtype=torch.bfloat16, device=DEV)
    ld, idd, lbd = logits.to(DEV), inputs.to(DEV), labels.to(DEV)
    _lib.call("hma_count_masked", ops.stream_ptr(), idd.data_ptr(), stats.data_ptr(), B, T, S, mask_id)
    _lib.call("hma_ce_fwd_bwd", ops.stream_ptr(), ld.data_ptr(), idd.data_ptr(), lbd.data_ptr(), stats.data_ptr(),
              dl.data_ptr(), None, 3.0, B, T, S, mask_id, 0.01)
    st = stats.cpu()
    assert st[2].item() == float((inputs[:, 1:] == mask_id).sum())
    assert abs(st[0].item() / st[2].item() - loss.item()) < 1e-5 * abs(loss.item())
    assert abs(st[1].item() / st[2].item() - acc.item()) < 1e-7
    close(dl, lr.grad.reshape(-1, 1024), 2 * BF, "dlogits")


def test_maskgit_step_bit_exact():
    B, T, S = 3, 4, 256
    mask_id = 262144
    gen = g(80)
    logits = torch.randn(B, T, S, 1024, generator=gen) * 3
    out_t = 2
    prompt = torch.randint(0, 262144, (B, T, S), generator=gen)
    prompt[:, out_t:] = mask_id
    unmasked = torch.zeros(B, S, dtype=torch.uint8)
    ld = logits.to(DEV)
    pd, ud = prompt.to(DEV), unmasked.to(DEV)
    conf = torch.empty(B, S, device=DEV)
    steps = 4
    p_ref, u_ref = prompt.clone(), unmasked.bool().clone()
    for step in range(steps):
        last = step == steps - 1
        n = math.ceil(R.cosine_schedule((step + 1) / steps) * S)
        rnd = torch.rand(B, S, generator=gen) if step % 2 else None  # alternate greedy / "random" confidences
        rd = None if rnd is None else rnd.to(DEV)
        _lib.call("hma_maskgit_step", ops.stream_ptr(), ld.data_ptr(), pd.data_ptr(), ud.data_ptr(),
                  None if rd is None else rd.data_ptr(), conf.data_ptr(), B, T, S, out_t, n, int(last), mask_id, 0, 0)
        # oracle: argmax ids are bit-exact; ranks use the kernel's own confidences (fp32 summation order differs from torch)
        fl = logits[:, out_t].reshape(B, S, 2, 512)
        a = fl.argmax(-1)
        samples = a[..., 1] * 512 + a[..., 0]
        pr = fl.softmax(-1).max(-1).values
        close(conf, pr[..., 1] * pr[..., 0], 1e-5, "confidence")
        c = conf.cpu() if rnd is None else rnd
        new, u_ref = R.maskgit_select(c, samples, u_ref, p_ref[:, out_t], n, mask_id, last)
        p_ref[:, out_t] = new
        assert torch.equal(pd.cpu(), p_ref), f"step {step}"
        if not last:
            assert torch.equal(ud.cpu().bool(), u_ref), f"step {step}"
        # a new forward would change the logits; emulate by perturbing them
        logits = logits + torch.randn(B, T, S, 1024, generator=gen) * 0.5
        ld = logits.to(DEV)
    assert (p_ref[:, out_t] != mask_id).all()


# ------------------------------------------------------------------------------------------ optimizer
def test_sqnorm_adamw_cast_transpose():
    n = 100_003 + 61
    gen = g(90)
    p = torch.randn(n, generator=gen); gr = torch.randn(n, generator=gen) * 3
    m = torch.randn(n, generator=gen) * 0.1; v = torch.rand(n, generator=gen) * 0.1
    pd, gd, md, vd = p.to(DEV), gr.to(DEV), m.to(DEV), v.to(DEV)
    sq = torch.zeros(1, device=DEV)
    _lib.call("hma_sqnorm", ops.stream_ptr(), gd.data_ptr(), n, sq.data_ptr())
    assert abs(sq.item() - (gr.double() ** 2).sum().item()) < 1e-4 * sq.item()
    pb = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    params, grads, ms, vs = {"w": p.clone()}, {"w": gr.clone()}, {"w": m.clone()}, {"w": v.clone()}
    R.clip_and_adamw(params, grads, ms, vs, step=3, lr=1e-2, max_norm=1.0)
    _lib.call("hma_adamw", ops.stream_ptr(), pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), pb.data_ptr(), n,
              1e-2, 0.9, 0.95, 1e-8, 0.05, 3, sq.data_ptr(), 1.0, None)
    close(pd, params["w"], 1e-6, "adamw p")
    close(md, ms["w"], 1e-6, "adamw m")
    close(vd, vs["w"], 1e-6, "adamw v")
    assert torch.equal(pb.cpu(), pd.cpu().bfloat16())
    src = torch.randn(3, 40, 70, generator=gen).to(DEV)
    dst = torch.empty(3, 70, 40, dtype=torch.bfloat16, device=DEV)
    _lib.call("hma_transpose_cast_bf16", ops.stream_ptr(), src.data_ptr(), dst.data_ptr(), 40, 70, 3, 40 * 70, 40 * 70)
    assert torch.equal(dst.cpu(), src.cpu().transpose(1, 2).bfloat16())


# ------------------------------------------------------------------------------------------ dropout in the GEMM epilogues
def test_gemm_dropout_masks_are_consistent_between_forward_and_backward():
    """nn.Dropout of the MLP (st_transformer.py:24-27) inside the epilogues: GELU2 drops the activation (not the saved
    pre-activation), DGELU re-creates the same mask, RESID drops the branch output, hma_dropout_bf16 the gradient behind it."""
    M, p = 1000, 0.25
    x = rb(torch.randn(M, 256, generator=g(70)))
    w1 = rb(torch.randn(1024, 256, generator=g(71)) * 0.1)
    seed = torch.tensor([12345], dtype=torch.int32, device=DEV)
    xd, w1d = x.to(DEV).bfloat16(), w1.to(DEV).bfloat16()
    u, h = (torch.empty(M, 1024, dtype=torch.bfloat16, device=DEV) for _ in range(2))
    gm = ops.make_gemm_nt(A=xd.data_ptr(), lda=256, a_kind=A_BF16, W=w1d.data_ptr(), ldw=256, M=M, N=1024, K=256, epi=EPI_GELU2,
                          Cp=u.data_ptr(), ldc=1024, C2=h.data_ptr(), ldc2=1024, drop_p=p, drop_salt=3, drop_seed=seed.data_ptr())
    _lib.call("hma_gemm_nt", ops.stream_ptr(), C.byref(gm))
    uf = x @ w1.t()
    close(u, uf, BF, "u is not dropped")
    ref = F.gelu(u.float().cpu())
    keep = h.float().cpu() != 0
    frac = 1 - keep.float().mean().item()
    assert abs(frac - p) < 0.01, frac
    close(h.float().cpu()[keep], (ref / (1 - p))[keep], 2 * BF, "kept activations are scaled by 1 / (1 - p)")
    # backward: dh * gelu'(u) * mask / (1 - p), same seed / salt / indices
    dh = rb(torch.randn(M, 256, generator=g(72)))
    w2t = rb(torch.randn(1024, 256, generator=g(73)) * 0.1)   # [N = 1024, K = 256]: dgrad weight of fc2
    du = torch.empty(M, 1024, dtype=torch.bfloat16, device=DEV)
    dhd, w2d = dh.to(DEV).bfloat16(), w2t.to(DEV).bfloat16()
    gd = ops.make_gemm_nt(A=dhd.data_ptr(), lda=256, a_kind=A_BF16, W=w2d.data_ptr(), ldw=256, M=M, N=1024, K=256, epi=EPI_DGELU,
                          Cp=du.data_ptr(), ldc=1024, U=u.data_ptr(), ldu=1024, drop_p=p, drop_salt=3, drop_seed=seed.data_ptr())
    _lib.call("hma_gemm_nt", ops.stream_ptr(), C.byref(gd))
    uu = u.float().cpu().requires_grad_(True)
    F.gelu(uu).backward(torch.ones_like(uu))
    want = (dh @ w2t.t()) * uu.grad * keep.float() / (1 - p)
    close(du, want, 2 * BF, "du with the forward's mask")
    assert torch.equal(du.float().cpu() != 0, keep & (want != 0)) or ((du.float().cpu() != 0) ^ keep).float().mean().item() < 1e-3
    # a different seed gives a different mask
    seed.fill_(777)
    _lib.call("hma_gemm_nt", ops.stream_ptr(), C.byref(gm))
    assert ((h.float().cpu() != 0) ^ keep).float().mean().item() > 0.2
    # RESID (K = 1024) drops the branch output; hma_dropout_bf16 re-creates that mask on a gradient
    hh = rb(torch.randn(M, 1024, generator=g(74)) * 0.3)
    w2 = rb(torch.randn(256, 1024, generator=g(75)) * 0.05)
    xres = torch.zeros(M, 256, device=DEV)
    hd, w2b = hh.to(DEV).bfloat16(), w2.to(DEV).bfloat16()
    gr = ops.make_gemm_nt(A=hd.data_ptr(), lda=1024, a_kind=A_BF16, W=w2b.data_ptr(), ldw=1024, M=M, N=256, K=1024, epi=EPI_RESID,
                          Cp=xres.data_ptr(), ldc=256, drop_p=p, drop_salt=8, drop_seed=seed.data_ptr())
    _lib.call("hma_gemm_nt", ops.stream_ptr(), C.byref(gr))
    y = hh @ w2.t()
    k2 = xres.cpu() != 0
    assert abs(1 - k2.float().mean().item() - p) < 0.01
    close(xres.cpu()[k2], (y / (1 - p))[k2], 1e-4, "kept branch outputs")
    gsrc = torch.randn(M, 256, generator=g(76))
    gdst = torch.empty(M, 256, dtype=torch.bfloat16, device=DEV)
    _lib.call("hma_dropout_bf16", ops.stream_ptr(), gsrc.to(DEV).data_ptr(), gdst.data_ptr(), M, 256, p, seed.data_ptr(), 8)
    assert torch.equal(gdst.float().cpu() != 0, k2 & (rb(gsrc / (1 - p)) != 0))
    close(gdst, gsrc * k2.float() / (1 - p), BF, "masked gradient")


# ------------------------------------------------------------------------------------------ fused MLP block
def _mlp_weights(seed):
    w1 = torch.randn(1024, 256, generator=g(seed)) * 0.06
    b1 = torch.randn(1024, generator=g(seed + 1)) * 0.3
    w2 = torch.randn(256, 1024, generator=g(seed + 2)) * 0.04
    b2 = torch.randn(256, generator=g(seed + 3)) * 0.2
    gam = torch.randn(256, generator=g(seed + 4)) * 0.2 + 1.0
    bet = torch.randn(256, generator=g(seed + 5)) * 0.2
    return w1, b1, w2, b2, gam, bet


def _mlp_pack(src, rs, cs, rscale, cscale, kind):
    dst = torch.empty(512 * 512, dtype=torch.bfloat16, device=DEV)
    _lib.call("hma_mlp_pack", ops.stream_ptr(), ops.ptr(src), rs, cs, ops.ptr(rscale), ops.ptr(cscale), ops.ptr(dst), kind, 1, 0, 0)
    return dst


def _mlp_packed(w1, w2, gam):
    w1d, w2d, gd = w1.to(DEV), w2.to(DEV), gam.to(DEV)
    return dict(w1p=_mlp_pack(w1d, 256, 1, None, gd, 0), w2p=_mlp_pack(w2d, 1024, 1, None, None, 1),
                w2tp=_mlp_pack(w2d, 1, 1024, None, None, 0), w1tp=_mlp_pack(w1d, 1, 256, gd, None, 1), keep=(w1d, w2d, gd))


def test_mlp_pack_layout():
    # fragment f, lane (rho, hi), element i  <->  logical (row, col) as documented in include/hma_hip.h / csrc/mlp.hip
    w = torch.arange(1024 * 256, dtype=torch.float32).reshape(1024, 256) % 251.0  # exactly representable in bf16
    rowmap = lambda r: (r & 3) + 4 * (r >> 3) + 16 * ((r >> 2) & 1)
    p0 = _mlp_pack(w.to(DEV), 256, 1, None, None, 0).float().cpu().reshape(512, 64, 8)
    for f, lane, i in [(0, 0, 0), (17, 5, 3), (300, 37, 7), (511, 63, 7), (123, 32, 0)]:
        mb, j, rho, hi = f >> 4, f & 15, lane & 31, lane >> 5
        assert p0[f, lane, i] == w[32 * mb + rowmap(rho), 32 * (j >> 1) + 16 * hi + 8 * (j & 1) + i]
    wt = torch.arange(256 * 1024, dtype=torch.float32).reshape(256, 1024) % 241.0
    p1 = _mlp_pack(wt.to(DEV), 1024, 1, None, None, 1).float().cpu().reshape(512, 64, 8)
    for f, lane, i in [(0, 0, 0), (17, 5, 3), (300, 37, 7), (511, 63, 7)]:
        s, mb, j, rho, hi = f >> 4, (f >> 1) & 7, f & 1, lane & 31, lane >> 5
        assert p1[f, lane, i] == wt[32 * mb + rowmap(rho), 32 * s + 16 * hi + 8 * j + i]


def test_pack_multi_equals_single_calls():
    # hma_{chain_pack,mlp_pack,transpose_cast_bf16}_multi: several jobs (mixed shapes, batches, negative batch strides, scales,
    # more than one launch's worth) in one call leave exactly the bytes of the single calls
    L, d = 3, 256
    gen = g(811)
    W = torch.randn(L, 1024 * 256, generator=gen).to(DEV)           # L matrices, 1024*256 floats apart
    sc = torch.rand(L, 1024 * 256, generator=gen).to(DEV) + 0.5     # scales move with the same batch stride
    sp = ops.stream_ptr()
    ss = 1024 * 256

    def bufs(n):
        return [torch.zeros(n, dtype=torch.bfloat16, device=DEV) for _ in range(2)]

    # chain bundles: (rs, cs, row_scale, col_scale, kind, rows, cols, batch, reverse, bundle_stride)
    cases = [(256, 1, False, False, 0, 256, 256, L, True, 1), (1, 256, True, False, 0, 256, 256, L, False, 1),
             (256, 1, False, True, 0, 1024, 256, L, True, 2), (1024, 1, False, False, 1, 256, 1024, L, True, 2),
             (256, 1, False, False, 0, 768, 256, 1, False, 1)] * 6      # 30 jobs: two launches
    jobs, singles = [], []
    for rs, cs, rsc, csc, kind, rows, cols, batch, rev, bstr in cases:
        n = (rows if kind == 0 else cols) // 32 * 8192 * bstr
        a, b = bufs(L * n)
        src = ops.ptr(W) + (4 * ss * (L - 1) if rev else 0)
        rp = (ops.ptr(sc) + (4 * ss * (L - 1) if rev else 0)) if rsc else None
        cpn = (ops.ptr(sc) + (4 * ss * (L - 1) if rev else 0)) if csc else None
        sst = -ss if rev else ss
        common = dict(src=src, row_stride=rs, col_stride=cs, row_scale=rp, col_scale=cpn, kind=kind, rows=rows, cols=cols, batch=batch,
                      src_batch_stride=sst, dst_batch_stride=n, bundle_stride=bstr)
        jobs.append(dict(common, dst=ops.ptr(a)))
        _lib.call("hma_chain_pack", sp, src, rs, cs, rp, cpn, ops.ptr(b), kind, rows, cols, batch, sst, n, bstr)
        singles.append((a, b))
    arr = _lib.pack_jobs(jobs)
    _lib.call("hma_chain_pack_multi", sp, arr, len(arr))
    for i, (a, b) in enumerate(singles):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), f"chain job {i}"
        assert b.float().abs().sum() > 0
    # fused-MLP fragments
    jobs, singles = [], []
    for rs, cs, rsc, csc, kind, batch in [(256, 1, False, True, 0, L), (1024, 1, False, False, 1, L), (1, 1024, False, False, 0, 2),
                                          (1, 256, True, False, 1, 1)]:
        a, b = bufs(L * 512 * 512)
        rp, cpn = (ops.ptr(sc) if rsc else None), (ops.ptr(sc) if csc else None)
        jobs.append(dict(src=ops.ptr(W), row_stride=rs, col_stride=cs, row_scale=rp, col_scale=cpn, dst=ops.ptr(a), kind=kind, batch=batch,
                         src_batch_stride=ss, dst_batch_stride=512 * 512))
        _lib.call("hma_mlp_pack", sp, ops.ptr(W), rs, cs, rp, cpn, ops.ptr(b), kind, batch, ss, 512 * 512)
        singles.append((a, b))
    arr = _lib.pack_jobs(jobs)
    _lib.call("hma_mlp_pack_multi", sp, arr, len(arr))
    for i, (a, b) in enumerate(singles):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), f"mlp job {i}"
    # transposed bf16 copies (ragged 32 x 32 edge tiles included)
    jobs, singles = [], []
    for rows, cols, batch in [(768, 256, L), (256, 1024, L), (1024, 256, 1), (70, 45, 2), (256, 256, 0)]:
        a, b = bufs(L * rows * cols)
        jobs.append(dict(src=ops.ptr(W), dst=ops.ptr(a), rows=rows, cols=cols, batch=batch, src_batch_stride=ss, dst_batch_stride=rows * cols))
        _lib.call("hma_transpose_cast_bf16", sp, ops.ptr(W), ops.ptr(b), rows, cols, batch, ss, rows * cols)
        singles.append((a, b))
    arr = _lib.pack_jobs(jobs)
    _lib.call("hma_transpose_cast_bf16_multi", sp, arr, len(arr))
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(singles):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), f"transpose job {i}"
    # argument errors are reported, not launched
    bad = _lib.pack_jobs([dict(src=ops.ptr(W), dst=None, kind=0, rows=256, cols=256, batch=1, bundle_stride=1)])
    for fn in ("hma_chain_pack_multi", "hma_mlp_pack_multi", "hma_transpose_cast_bf16_multi"):
        with pytest.raises(_lib.HmaKernelError):
            _lib.call(fn, sp, bad, 1)
        _lib.call(fn, sp, None, 0)


@pytest.mark.parametrize("M,with_ln", [(128, True), (1000, True), (5 * 128 + 37, False), (40960, True)])
def test_mlp_fwd_fused(M, with_ln):
    w1, b1, w2, b2, gam, bet = _mlp_weights(100)
    x = torch.randn(M, 256, generator=g(7)) * 1.5 + 0.3
    xh = rb(F.layer_norm(x, (256,), eps=1e-5))
    pk = _mlp_packed(w1, w2, gam)
    b1f = (b1 + w1 @ bet).to(DEV)
    xd = x.to(DEV).clone()
    ln_x = torch.empty(M, 256, dtype=torch.bfloat16, device=DEV) if with_ln else None
    ln_r = torch.empty(M, dtype=torch.float32, device=DEV) if with_ln else None
    xhd = xh.to(DEV).bfloat16()
    a = ops.make_mlp_fwd(M=M, xhat=ops.ptr(xhd), x=ops.ptr(xd), w1p=ops.ptr(pk["w1p"]), w2p=ops.ptr(pk["w2p"]), b1=ops.ptr(b1f),
                         b2=ops.ptr(b2.to(DEV)), ln_xhat=ops.ptr(ln_x), ln_rstd=ops.ptr(ln_r), ln_eps=1e-5)
    b2d = b2.to(DEV)
    a.b2 = ops.ptr(b2d)
    _lib.call("hma_mlp_fwd", ops.stream_ptr(), C.byref(a))
    torch.cuda.synchronize()
    # reference on the bf16-rounded operands the kernel multiplies
    u = xh @ rb(w1 * gam).t() + (b1 + w1 @ bet)
    hgr = rb(F.gelu(u))
    ref = x + hgr @ rb(w2).t() + b2
    close(xd, ref, 3e-3, "mlp fwd x")  # bf16 rounding of the hidden activation flips with the GELU's 1e-7 differences
    err = (xd.cpu() - ref).pow(2).mean().sqrt() / (ref - x).pow(2).mean().sqrt()
    assert err < 4e-3, err
    if with_ln:
        lr = F.layer_norm(ref, (256,), eps=1e-5)
        close(ln_x, lr, 2 * BF, "fused next LN")
        var = ref.var(dim=1, unbiased=False)
        close(ln_r, torch.rsqrt(var + 1e-5), 2e-3, "rstd")


def _from_frag32(t, M):
    """[Mt, ld] in the HMA_A_BF16_FRAG32 order (include/hma_hip.h) -> row-major [M, ld]."""
    Mt, ld = t.shape
    v = t.reshape(Mt // 128, ld // 32, 4, 2, 2, 32, 8)        # tile, block, row group, (c / 8) % 2, (c / 16) % 2, row, c % 8
    return v.permute(0, 2, 5, 1, 4, 3, 6).reshape(Mt, ld)[:M]  # tile, group, row | block, c/16, c/8, c%8


@pytest.mark.parametrize("M", [128, 1000, 40960, 163840])
def test_mlp_bwd_fused(M):
    _mlp_bwd_case(M, 0.0)


@pytest.mark.parametrize("M", [128, 4096 + 32])
def test_mlp_bwd_fused_dropout(M):
    """mlp_drop > 0: hma_mlp_bwd re-creates the forward's two masks (the hash of hma_dropout_bf16 / the GEMM epilogues): dy is masked
    (and written to dy_drop for the fc2 weight gradient), hg and the gradient entering gelu' carry the activation mask."""
    _mlp_bwd_case(M, 0.1)


def _drop_mask(M, cols, p, seed, salt):
    """keep / (1 - p) per element, from hma_dropout_bf16 applied to ones (1 / 0.9 is not exact in bf16: normalise)"""
    ones = torch.ones(M, cols, device=DEV)
    out = torch.empty(M, cols, dtype=torch.bfloat16, device=DEV)
    _lib.call("hma_dropout_bf16", ops.stream_ptr(), ones.data_ptr(), out.data_ptr(), M, cols, p, seed.data_ptr(), salt)
    return (out.float() > 0).float().cpu() / (1.0 - p)


def _mlp_bwd_case(M, p_drop):
    w1, b1, w2, b2, gam, bet = _mlp_weights(200)
    x = torch.randn(M, 256, generator=g(17)) * 1.5 + 0.3
    mean, var = x.mean(1, keepdim=True), x.var(1, unbiased=False, keepdim=True)
    rstd = torch.rsqrt(var + 1e-5)
    xh = rb((x - mean) * rstd)
    dy = rb(torch.randn(M, 256, generator=g(18)) * 0.02)
    dx0 = torch.randn(M, 256, generator=g(19)) * 0.02
    pk = _mlp_packed(w1, w2, gam)
    b1f = (b1 + w1 @ bet).to(DEV)
    xhd, dyd, rsd = xh.to(DEV).bfloat16(), dy.to(DEV).bfloat16(), rstd.reshape(-1).to(DEV)
    dxd = dx0.to(DEV).clone()
    dxb = torch.empty(M, 256, dtype=torch.bfloat16, device=DEV)
    Mt = (M + 127) // 128 * 128
    hg = torch.zeros(Mt, 1024, dtype=torch.bfloat16, device=DEV)  # HMA_A_BF16_FRAG32 order
    du = torch.zeros(Mt, 1024, dtype=torch.bfloat16, device=DEV)
    dkw, m0, m1 = {}, 1.0, 1.0
    if p_drop > 0:
        seed = torch.tensor([12345], dtype=torch.int32, device=DEV)
        dyo = torch.zeros(M, 256, dtype=torch.bfloat16, device=DEV)
        dkw = dict(drop_p=p_drop, drop_salt=6, drop_seed=seed.data_ptr(), dy_drop=ops.ptr(dyo))
        m0, m1 = _drop_mask(M, 1024, p_drop, seed, 6), _drop_mask(M, 256, p_drop, seed, 7)
        assert 0.85 < (m0 > 0).float().mean() < 0.95 and 0.85 < (m1 > 0).float().mean() < 0.95
    a = ops.make_mlp_bwd(M=M, xhat=ops.ptr(xhd), rstd=ops.ptr(rsd), dy=ops.ptr(dyd), dx=ops.ptr(dxd), dx_bf16=ops.ptr(dxb),
                         w1p=ops.ptr(pk["w1p"]), w2tp=ops.ptr(pk["w2tp"]), w1tp=ops.ptr(pk["w1tp"]), b1=ops.ptr(b1f),
                         hg=ops.ptr(hg), du=ops.ptr(du), **dkw)
    _lib.call("hma_mlp_bwd", ops.stream_ptr(), C.byref(a))
    torch.cuda.synchronize()
    # reference (fp32 math on the rounded operands)
    w1f = rb(w1 * gam)
    u = xh @ w1f.t() + (b1 + w1 @ bet)
    cdf = 0.5 * (1.0 + torch.erf(u / math.sqrt(2.0)))
    hgr = u * cdf * m0
    if p_drop > 0:
        dy = rb(dy * m1)  # the gradient behind the output Dropout
        close(dyo, dy, BF, "dy_drop")
        assert torch.equal(dyo.float().cpu() == 0, (m1 == 0) | (dy == 0))
        dyd = dyo
    dhg = (dy @ rb(w2)) * m0
    dur = dhg * (cdf + u * torch.exp(-0.5 * u * u) / math.sqrt(2.0 * math.pi))
    close(_from_frag32(hg, M), hgr, 2 * BF, "hg")
    close(_from_frag32(du, M), dur, 2 * BF, "du")
    if M % 32 == 0:  # the two weight gradients read that order directly (LDS-DMA source addressing)
        dW2 = torch.zeros(256, 1024, device=DEV)
        dW1 = torch.zeros(1024, 256, device=DEV)
        dB1 = torch.zeros(1024, device=DEV)
        ws = torch.empty(256 * (65536 + 256), device=DEV)
        one, zero = torch.ones(256, device=DEV), torch.zeros(256, device=DEV)
        ta = ops.make_gemm_tn(dY=ops.ptr(dyd), ldy=256, y_kind=A_BF16, A=ops.ptr(hg), lda=1024, a_kind=A_BF16_FRAG32, M=M, N=256,
                              K=1024, dW=ops.ptr(dW2), lddw=1024, ws=ops.ptr(ws), ws_elems=ws.numel())
        tb = ops.make_gemm_tn(dY=ops.ptr(du), ldy=1024, y_kind=A_BF16_FRAG32, A=ops.ptr(xhd), lda=256, a_kind=A_BF16_AFFINE, M=M,
                              N=1024, K=256, dW=ops.ptr(dW1), lddw=256, dBias=ops.ptr(dB1), gamma=ops.ptr(one), beta=ops.ptr(zero),
                              ws=ops.ptr(ws), ws_elems=ws.numel())
        _lib.call("hma_gemm_tn_pair", ops.stream_ptr(), C.byref(ta), C.byref(tb))
        torch.cuda.synchronize()
        hb, db = _from_frag32(hg, M).float().cpu(), _from_frag32(du, M).float().cpu()
        close(dW2, dy.t() @ hb, 2e-2, "dW2 from FRAG32 hg")
        close(dW1, db.t() @ xh, 2e-2, "dW1 from FRAG32 du")
        close(dB1, db.sum(0), 2e-2, "db1 from FRAG32 du")
    gk = rb(dur) @ w1f  # dxhat with gamma folded (the kernel multiplies the bf16-rounded du it hands to the wgrad)
    s1 = gk.mean(1, keepdim=True)
    s2 = (gk * xh).mean(1, keepdim=True)
    ref = dx0 + rstd * (gk - s1 - xh * s2)
    close(dxd, ref, 2e-3, "dx")
    close(dxb, ref, 2 * BF, "dx bf16")
    assert (dxd.cpu() - ref).pow(2).mean().sqrt() / (ref - dx0).pow(2).mean().sqrt() < 3e-3


def test_gemm_tn_folded_affine_grads():
    # dgamma / dbeta of a LayerNorm folded into the Linear, from the weight-gradient reduction (hma_gemm_tn_t.w_master)
    M, N, K = 4096, 1024, 256
    dy = rb(torch.randn(M, N, generator=g(31)) * 0.05)
    xh = rb(torch.randn(M, K, generator=g(32)))
    w = torch.randn(N, K, generator=g(33)) * 0.1
    gam = torch.randn(K, generator=g(34)) * 0.2 + 1.0
    bet = torch.randn(K, generator=g(35)) * 0.2
    dW = torch.zeros(N, K, device=DEV)
    dB = torch.zeros(N, device=DEV)
    dG = torch.zeros(K, device=DEV)
    dBt = torch.zeros(K, device=DEV)
    ws = torch.empty(256 * (65536 + 256), device=DEV)
    wd, gd, bd = w.to(DEV), gam.to(DEV), bet.to(DEV)
    dyd, xhd = dy.to(DEV).bfloat16(), xh.to(DEV).bfloat16()
    t = ops.make_gemm_tn(dY=ops.ptr(dyd), ldy=N, y_kind=A_BF16, A=ops.ptr(xhd), lda=K, a_kind=A_BF16_AFFINE, M=M, N=N, K=K,
                         dW=ops.ptr(dW), lddw=K, dBias=ops.ptr(dB), gamma=ops.ptr(gd), beta=ops.ptr(bd), ws=ops.ptr(ws),
                         ws_elems=ws.numel(), w_master=ops.ptr(wd), dgamma=ops.ptr(dG), dbeta=ops.ptr(dBt))
    _lib.call("hma_gemm_tn", ops.stream_ptr(), C.byref(t))
    torch.cuda.synchronize()
    P = dy.t() @ xh
    cs = dy.sum(0)
    close(dW, P * gam + cs[:, None] * bet[None, :], 1e-2, "dW")  # bf16 partials
    close(dG, (w * P).sum(0), 1e-2, "dgamma")
    close(dBt, (w * cs[:, None]).sum(0), 1e-2, "dbeta")


@pytest.mark.parametrize("K", [64, 128, 192])
@pytest.mark.parametrize("M", [1000, 65536, 100000])
def test_gemm_nt_shallow_k_many_tiles(M, K):
    """K < 256 (fewer than four K-steps per tile: STMAR's token_embed has K = 128, hma/model/st_mar.py:160) at row counts where a
    persistent workgroup would own more than one tile (M > 128 x 256): round 3 found the first tile of every such workgroup lost."""
    x = rb(torch.randn(M, K, generator=g(31)))
    w = rb(torch.randn(256, K, generator=g(32)) * 0.2)
    y = ops.linear(x.to(DEV).bfloat16(), w.to(DEV).bfloat16(), None, epi=EPI_F32)
    close(y, x @ w.t(), 1e-5, "shallow K, f32 out")
    b = torch.randn(256, generator=g(33))
    y = ops.linear(x.to(DEV).bfloat16(), w.to(DEV).bfloat16(), b.to(DEV), epi=EPI_BF16)
    close(y, x @ w.t() + b, BF, "shallow K, bf16 out")


def _drop_hash_np(seed, salt, pair):
    """numpy replica of hma_common.h drop_hash (the murmur3 finaliser over seed, salt and the element-pair index).  A version on
    v_mad_u32_u24 (full-rate multiplies, 12 issue slots against 19) was measured in round 6: statistically indistinguishable, and the MAR
    step 0.5 ms SLOWER (its extra live register per hash costs the dropout forms of the chains more than the multiplies did)."""
    import numpy as np
    m32 = np.uint64(0xFFFFFFFF)
    pair = pair.astype(np.uint64)
    x = (pair & m32) ^ (((pair >> np.uint64(32)) * np.uint64(0x9E3779B9)) & m32)
    x ^= np.uint64((seed + 0x7F4A7C15 * (salt + 1)) & 0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & m32
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & m32
    x ^= x >> np.uint64(16)
    return x


@pytest.mark.parametrize("seed,salt,p", [(4242, 6, 0.05), (1, 0, 0.1), (506952113, 63, 0.5)])
def test_dropout_hash_statistics_and_replica(seed, salt, p):
    """The counter-based dropout mask (hma_common.h drop_hash: every kernel that masks -- GEMM epilogues, the chains, hma_mlp_bwd,
    hma_dropout_bf16 -- calls the one function): the kernel's keep mask equals the numpy replica bit for bit, the keep rate is 1 - p
    rounded down to 2^-16 within sampling noise, and neighbouring elements, rows and the two halves of a hash are uncorrelated."""
    import numpy as np
    M, cols = 4096, 1024
    ones = torch.ones(M, cols, device=DEV)
    out = torch.empty(M, cols, dtype=torch.bfloat16, device=DEV)
    sd = torch.tensor([seed], dtype=torch.int32, device=DEV)
    _lib.call("hma_dropout_bf16", ops.stream_ptr(), ones.data_ptr(), out.data_ptr(), M, cols, p, sd.data_ptr(), salt)
    keep = (out.float() > 0).cpu().numpy().reshape(-1)
    x = _drop_hash_np(seed, salt, np.arange(M * cols // 2))
    th = int(p * 65536)
    ref = np.empty(M * cols, dtype=bool)
    ref[0::2] = (x & np.uint64(0xFFFF)) >= th
    ref[1::2] = (x >> np.uint64(16)) >= th
    assert (keep == ref).all()
    n = keep.size
    want = 1.0 - th / 65536.0
    assert abs(keep.mean() - want) <= 5.0 * math.sqrt(want * (1 - want) / n) + 1e-9
    k = keep.astype(np.float64) - keep.mean()
    for lag in (1, 2, cols, cols + 1):
        c = (k[:-lag] * k[lag:]).mean() / k.var()
        assert abs(c) <= 6.0 / math.sqrt(n), (lag, c)
