"""Kernel-level parity of the row-local chain kernels (csrc/chain.hip) against fp32 math on the bf16-rounded operands
the kernels multiply (GPU box only).  Reference arithmetic: hma/model/st_transformer.py:85-112, hma/model/st_mask_git.py:66-76."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hma_amd import _lib, ops  # noqa: E402

DEV = "cuda"
BF = 2.0 ** -8


def rb(t):
    return t.to(torch.bfloat16).float()


def g(seed):
    return torch.Generator().manual_seed(seed)


def close(a, b, rtol, what=""):
    a = a.float().cpu().double()
    b = b.float().cpu().double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item() + 1e-12
    assert err <= rtol * ref, f"{what}: max err {err:.3e} vs ref scale {ref:.3e} (rtol {rtol})"


def rms(a, b):
    a = a.float().cpu().double()
    b = b.float().cpu().double()
    return ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30)).item()


def _weights(seed):
    wp = rb(torch.randn(256, 256, generator=g(seed)) * 0.06)
    wl = rb(torch.randn(256, 256, generator=g(seed + 1)) * 0.06)
    wq = rb(torch.randn(768, 256, generator=g(seed + 2)) * 0.06)
    bp = torch.randn(256, generator=g(seed + 3)) * 0.1
    bl = torch.randn(256, generator=g(seed + 4)) * 0.1
    bq = torch.randn(768, generator=g(seed + 5)) * 0.1
    return wp, wl, wq, bp, bl, bq


def _pack_nt(w):  # forward: A[n][k] = W[n][k]
    wd = w.to(DEV).contiguous()
    return ops.chain_pack(wd, kind=0, rows=w.shape[0], cols=256, row_stride=256, col_stride=1)


def _pack_t(w, c=0):  # input gradient: A[n][k] = W[256 c + k][n]
    wd = w.to(DEV).contiguous()
    return ops.chain_pack(wd[256 * c:], kind=0, rows=256, cols=256, row_stride=1, col_stride=256)


# (163 840 rows = the headline step's launch size, B = 32 x T = 16 frames of 320 rows: 5.7 tiles per workgroup)
@pytest.mark.parametrize("M,rpf,mod", [(112, 16, True), (1000 // 16 * 16, 16, True), (320 * 7, 320, True), (40960, 320, True), (163840, 320, True),
                                       (1008, 0, False), (2560, 0, False)])
def test_chain_a_fwd(M, rpf, mod):
    wp, wl, wq, bp, bl, bq = _weights(300)
    o = rb(torch.randn(M, 256, generator=g(1)))
    x = torch.randn(M, 256, generator=g(2)) * 1.5 + 0.2
    frames = M // rpf if mod else 1
    ss = torch.randn(frames, 512, generator=g(3)) * 0.3
    # reference
    x1 = x + o @ wp.t() + bp
    if mod:
        xh32 = F.layer_norm(x1, (256,), eps=1e-6)
        xh = rb(xh32)
        f = torch.arange(M) // rpf
        xm = rb(xh32 * (1 + ss[f, 256:]) + ss[f, :256])
        x2 = x1 + xm @ wl.t() + bl
        rstd = torch.rsqrt(x1.var(dim=1, unbiased=False) + 1e-6)
    else:
        x2 = x1
    qkv = rb(x2) @ wq.t() + bq
    # kernel
    segs = [(_pack_nt(wp), 8)] + ([(_pack_nt(wl), 8)] if mod else []) + [(_pack_nt(wq), 24)]
    xd = x.to(DEV).clone()
    od = o.to(DEV).bfloat16()
    ssd = ss.to(DEV)
    bpd, bld, bqd = bp.to(DEV), bl.to(DEV), bq.to(DEV)
    xhat_o = torch.zeros(M, 256, dtype=torch.bfloat16, device=DEV)
    xm_o = torch.zeros(M, 256, dtype=torch.bfloat16, device=DEV)
    rstd_o = torch.zeros(M, dtype=torch.float32, device=DEV)
    xb_o = torch.zeros(M, 256, dtype=torch.bfloat16, device=DEV)
    qkv_o = torch.zeros(M, 768, dtype=torch.bfloat16, device=DEV)
    a = ops.make_chain_a_fwd(M=M, segs=[(ops.ptr(t), n) for t, n in segs], o=ops.ptr(od), x=ops.ptr(xd), qkv=ops.ptr(qkv_o),
                             ss=ops.ptr(ssd) if mod else None, b_proj=ops.ptr(bpd), b_lin=ops.ptr(bld) if mod else None,
                             b_qkv=ops.ptr(bqd), xhat=ops.ptr(xhat_o), xm=ops.ptr(xm_o), rstd=ops.ptr(rstd_o),
                             x_bf16=ops.ptr(xb_o), rows_per_frame=rpf, use_mod=mod)
    _lib.call("hma_chain_a_fwd", ops.stream_ptr(), C.byref(a))
    torch.cuda.synchronize()
    close(xd, x2, 2e-3, "x2")
    assert rms(xd.cpu() - x, x2 - x) < 4e-3
    close(xb_o, x2, 2 * BF, "x2 bf16")
    close(qkv_o, qkv, 3 * BF, "qkv")
    assert rms(qkv_o, qkv) < 4e-3
    if mod:
        close(xhat_o, xh, 2 * BF, "xhat")
        close(xm_o, xm, 3 * BF, "xm")
        close(rstd_o, rstd, 1e-4, "rstd")
    # asymmetric check: a permuted row or column block would pass a symmetric statistic
    assert (qkv_o.float().cpu() - qkv).abs().max() < 0.2 * (qkv_o.float().cpu() - qkv.flip(1)).abs().max()


@pytest.mark.parametrize("M,rpf,mod", [(112, 16, True), (320 * 7, 320, True), (40960, 320, True), (163840, 320, True), (1008, 0, False)])
def test_chain_a_bwd(M, rpf, mod):
    wp, wl, wq, _, _, _ = _weights(400)
    dqkv = rb(torch.randn(M, 768, generator=g(11)) * 0.02)
    dx = torch.randn(M, 256, generator=g(12)) * 0.02
    x1 = torch.randn(M, 256, generator=g(13)) * 1.5 + 0.2
    frames = M // rpf if mod else 1
    ss = torch.randn(frames, 512, generator=g(14)) * 0.3
    # reference
    dx2 = dx + dqkv @ wq
    if mod:
        mean, var = x1.mean(1, keepdim=True), x1.var(1, unbiased=False, keepdim=True)
        rstd = torch.rsqrt(var + 1e-6)
        xh = rb((x1 - mean) * rstd)
        f = torch.arange(M) // rpf
        dxm = rb(dx2) @ wl
        gq = dxm * (1 + ss[f, 256:])
        dx1 = dx2 + rstd * (gq - gq.mean(1, keepdim=True) - xh * (gq * xh).mean(1, keepdim=True))
        dss = torch.zeros(frames, 512)
        dss[:, :256].index_add_(0, f, dxm)
        dss[:, 256:].index_add_(0, f, dxm * xh)
    else:
        dx1 = dx2
    d_o = rb(dx1) @ wp
    # kernel
    segs = [(_pack_t(wq, c), 8) for c in range(3)]
    wq_t = torch.cat([s[0] for s in segs])
    segs = [(wq_t, 24)] + ([(_pack_t(wl), 8)] if mod else []) + [(_pack_t(wp), 8)]
    dxd = dx.to(DEV).clone()
    dqd = dqkv.to(DEV).bfloat16()
    d2 = torch.zeros(M, 256, dtype=torch.bfloat16, device=DEV)
    d1 = torch.zeros(M, 256, dtype=torch.bfloat16, device=DEV)
    do = torch.zeros(M, 256, dtype=torch.bfloat16, device=DEV)
    kw = {}
    if mod:
        xhd, rsd, ssd = xh.to(DEV).bfloat16(), rstd.reshape(-1).to(DEV).contiguous(), ss.to(DEV)
        dssd = torch.zeros(frames, 512, dtype=torch.float32, device=DEV)
        kw = dict(xhat=ops.ptr(xhd), rstd=ops.ptr(rsd), ss=ops.ptr(ssd), dx2_bf16=ops.ptr(d2), dss=ops.ptr(dssd))
    a = ops.make_chain_a_bwd(M=M, segs=[(ops.ptr(t), n) for t, n in segs], dqkv=ops.ptr(dqd), dx=ops.ptr(dxd), dx1_bf16=ops.ptr(d1),
                             d_o=ops.ptr(do), rows_per_frame=rpf, use_mod=mod, **kw)
    _lib.call("hma_chain_a_bwd", ops.stream_ptr(), C.byref(a))
    torch.cuda.synchronize()
    close(dxd, dx1, 3e-3, "dx1")
    assert rms(dxd, dx1) < 3e-3
    close(d1, dx1, 2 * BF, "dx1 bf16")
    close(do, d_o, 3 * BF, "d_o")
    assert rms(do, d_o) < 5e-3
    if mod:
        close(d2, dx2, 2 * BF, "dx2 bf16")
        close(dssd, dss, 3e-3, "dss")
    assert (do.float().cpu() - d_o).abs().max() < 0.2 * (do.float().cpu() - d_o.flip(1)).abs().max()


@pytest.mark.parametrize("M", [16, 112, 1008, 320 * 7, 40960, 163840])
def test_chain_s_bwd(M):
    """Spatial qkv input gradient -> norm1 backward -> residual, one launch (what hma_gemm_nt + hma_ln_bwd did in two).  Reference:
    hma/model/st_transformer.py:85-86, hma/model/attention.py:39 (autograd mirror); gamma folded into the packed weight's output rows."""
    _, _, wq, _, _, _ = _weights(500)
    gamma = 1.0 + 0.2 * torch.randn(256, generator=g(20))
    dqkv = rb(torch.randn(M, 768, generator=g(21)) * 0.02)
    dx = torch.randn(M, 256, generator=g(22)) * 0.02
    x0 = torch.randn(M, 256, generator=g(23)) * 1.5 + 0.2
    mean, var = x0.mean(1, keepdim=True), x0.var(1, unbiased=False, keepdim=True)
    rstd = torch.rsqrt(var + 1e-5)
    xh = rb((x0 - mean) * rstd)
    # reference: the weight the kernel multiplies is bf16(gamma[k] W[n][k])
    wg = rb(wq * gamma[None, :])
    gq = dqkv @ wg
    dx1 = dx + rstd * (gq - gq.mean(1, keepdim=True) - xh * (gq * xh).mean(1, keepdim=True))
    # kernel
    gd = gamma.to(DEV)
    wd = wq.to(DEV).contiguous()
    wt = torch.cat([ops.chain_pack(wd[256 * c:], kind=0, rows=256, cols=256, row_stride=1, col_stride=256, row_scale=gd) for c in range(3)])
    dxd = dx.to(DEV).clone()
    dqd = dqkv.to(DEV).bfloat16()
    xhd, rsd = xh.to(DEV).bfloat16(), rstd.reshape(-1).to(DEV).contiguous()
    d1 = torch.zeros(M, 256, dtype=torch.bfloat16, device=DEV)
    a = ops.make_chain_s_bwd(M=M, segs=[(ops.ptr(wt), 24)], dqkv=ops.ptr(dqd), dx=ops.ptr(dxd), xhat=ops.ptr(xhd), rstd=ops.ptr(rsd),
                             dx_bf16=ops.ptr(d1))
    _lib.call("hma_chain_s_bwd", ops.stream_ptr(), C.byref(a))
    torch.cuda.synchronize()
    # the same from dqkv in the head-blocked order (what hma_attn_spatial_bwd_blocked writes): bit-identical
    for n in (16, 320):
        if M % n == 0:
            dxh = dx.to(DEV).clone()
            d1h = torch.zeros_like(d1)
            dqh = ops.rows_to_headblk(dqd, n)
            ah = ops.make_chain_s_bwd(M=M, segs=[(ops.ptr(wt), 24)], dqkv=ops.ptr(dqh), dx=ops.ptr(dxh), xhat=ops.ptr(xhd), rstd=ops.ptr(rsd),
                                      dx_bf16=ops.ptr(d1h), hb_rows=n)
            _lib.call("hma_chain_s_bwd", ops.stream_ptr(), C.byref(ah))
            torch.cuda.synchronize()
            assert torch.equal(dxh, dxd) and torch.equal(d1h, d1), f"head-blocked dqkv, n = {n}"
    close(dxd, dx1, 3e-3, "dx")
    assert rms(dxd, dx1) < 3e-3
    assert rms(dxd.cpu() - dx, dx1 - dx) < 4e-3  # (the part the kernel adds, not the residual it passes through)
    close(d1, dx1, 2 * BF, "dx bf16")
    assert (dxd.cpu() - dx1).abs().max() < 0.2 * (dxd.cpu() - dx1.flip(1)).abs().max()
    assert (dxd.cpu() - dx1).abs().max() < 0.2 * (dxd.cpu() - dx1.flip(0)).abs().max()


@pytest.mark.parametrize("B,SA,T", [(1, 1, 16), (1, 16, 16), (2, 48, 16), (3, 320, 16), (5, 77, 16), (32, 320, 16),
                                    (3, 320, 12), (2, 48, 9), (5, 77, 5), (1, 16, 1), (32, 320, 12)])
def test_chain_t_bwd(B, SA, T):
    """hma_chain_t_bwd -- the temporal projection's input gradient + the causal attention backward of every (sample, token) column in
    one launch -- against (a) the two launches it replaces (hma_gemm_nt, hma_attn_temporal_bwd) and (b) an fp32 autograd reference.
    Windows of T < 16 frames (the reference's default is 12, train_multi.py:78-83): the column tile's lanes T .. 15 are masked.
    Reference: hma/model/st_transformer.py:111, hma/model/attention.py:37-61 (autograd mirror)."""
    scale = 0.25
    M = B * T * SA
    wp = rb(torch.randn(256, 256, generator=g(900)) * 0.06)          # proj.weight [out][in]
    qkv = rb(torch.randn(M, 768, generator=g(901)))
    dy = rb(torch.randn(M, 256, generator=g(902)) * 0.05)
    qd, dyd = qkv.to(DEV).bfloat16(), dy.to(DEV).bfloat16()
    wt = _pack_t(wp)
    dq = torch.zeros(M, 768, dtype=torch.bfloat16, device=DEV)
    a = ops.make_chain_t_bwd(B=B, SA=SA, T=T, segs=[(ops.ptr(wt), 8)], dy_bf16=ops.ptr(dyd), qkv=ops.ptr(qd), dqkv=ops.ptr(dq), attn_scale=scale)
    _lib.call("hma_chain_t_bwd", ops.stream_ptr(), C.byref(a))
    torch.cuda.synchronize()
    # (a) two launches: d_o = dy Wproj as bf16, then the attention backward
    d_o = ops.linear(dyd, wp.t().contiguous().to(DEV).bfloat16(), epi=ops.EPI_BF16)
    o = ops.attn_temporal_fwd(qd, B, T, SA, scale)
    dq2 = ops.attn_temporal_bwd(qd, o, d_o, B, T, SA, scale)
    torch.cuda.synchronize()
    close(dq, dq2, 2 * BF, "chain T vs two launches")
    assert rms(dq, dq2) < 2e-3
    # (b) fp32 autograd on the same bf16 inputs (small cases)
    if M <= 16 * 3 * 320:
        idx = torch.arange(M).reshape(B, T, SA).permute(0, 2, 1).reshape(-1)
        qr = qkv.clone().requires_grad_(True)
        cols = qr[idx].reshape(B * SA, T, 3, 8, 32)
        q, k, v = (cols[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        att = (q @ k.transpose(-1, -2)) * scale
        att = att.masked_fill(torch.ones(T, T).triu(1).bool(), float("-inf")).softmax(-1)
        o_ref = (att @ v).permute(0, 2, 1, 3).reshape(B * SA * T, 256)
        o_ref.backward(rb(dy @ wp)[idx])
        close(dq, qr.grad, 4 * BF, "chain T vs autograd")
        assert rms(dq, qr.grad) < 1e-2
        assert (dq.float().cpu() - qr.grad).abs().max() < 0.2 * (dq.float().cpu() - qr.grad.flip(0)).abs().max()


@pytest.mark.parametrize("M,with_qkv,save", [(112, True, False), (1008, True, True), (20480, True, False), (40960, True, True), (163840, True, True),
                                             (2560, False, False), (2560, False, True)])
def test_chain_b_fwd(M, with_qkv, save):
    _chain_b_case(M, with_qkv, save, 0.0)


@pytest.mark.parametrize("M,with_qkv", [(1008, True), (4096, False)])
def test_chain_b_fwd_dropout(M, with_qkv):
    """mlp_drop > 0 (training): gelu(u) and the branch output pass nn.Dropout (st_transformer.py:25-26) with the counter-based masks
    of the GEMM epilogues / hma_dropout_bf16 (salts s and s + 1), so hma_mlp_bwd can re-create them."""
    _chain_b_case(M, with_qkv, True, 0.1)


def _drop_mask(M, cols, p, seed, salt):
    ones = torch.ones(M, cols, device=DEV)
    out = torch.empty(M, cols, dtype=torch.bfloat16, device=DEV)
    _lib.call("hma_dropout_bf16", ops.stream_ptr(), ones.data_ptr(), out.data_ptr(), M, cols, p, seed.data_ptr(), salt)
    return (out.float() > 0).float().cpu() / (1.0 - p)


def _chain_b_case(M, with_qkv, save, p_drop):
    """proj_t + residual -> norm2 -> fc1 -> GELU -> fc2 + residual -> the next block's norm1 -> qkv, one launch (inference form:
    nothing saved).  Reference: hma/model/st_transformer.py:111-112, :24-27, :85-86; hma/model/attention.py:39,60."""
    gq = lambda s_: torch.Generator().manual_seed(s_)
    wp = rb(torch.randn(256, 256, generator=gq(500)) * 0.06)
    w1 = torch.randn(1024, 256, generator=gq(501)) * 0.06
    w2 = rb(torch.randn(256, 1024, generator=gq(502)) * 0.04)
    wq = torch.randn(768, 256, generator=gq(503)) * 0.06
    bp, b1, b2, bq = (torch.randn(n, generator=gq(504 + i)) * 0.1 for i, n in enumerate((256, 1024, 256, 768)))
    g2, be2 = 1 + 0.1 * torch.randn(256, generator=gq(510)), 0.1 * torch.randn(256, generator=gq(511))
    g1, be1 = 1 + 0.1 * torch.randn(256, generator=gq(512)), 0.1 * torch.randn(256, generator=gq(513))
    o = rb(torch.randn(M, 256, generator=gq(1)))
    x = torch.randn(M, 256, generator=gq(2)) * 1.5 + 0.2
    # reference: the LayerNorm affines folded like the engine folds them (W diag(gamma) in bf16, bias + W beta in fp32)
    w1f, b1f = rb(w1 * g2), b1 + w1 @ be2
    wqf, bqf = rb(wq * g1), bq + wq @ be1
    m0, m1, dkw = 1.0, 1.0, {}
    if p_drop > 0:
        seed = torch.tensor([777], dtype=torch.int32, device=DEV)
        m0, m1 = _drop_mask(M, 1024, p_drop, seed, 10), _drop_mask(M, 256, p_drop, seed, 11)
        dkw = dict(drop_p=p_drop, drop_salt=10, drop_seed=seed.data_ptr())
    x1 = x + o @ wp.t() + bp
    u = rb(F.layer_norm(x1, (256,), eps=1e-5)) @ w1f.t() + b1f
    x2 = x1 + (rb(F.gelu(u) * m0) @ w2.t() + b2) * m1
    qkv = rb(F.layer_norm(x2, (256,), eps=1e-5)) @ wqf.t() + bqf
    # kernel
    d = lambda t: t.to(DEV).contiguous()
    pp = ops.chain_pack(d(wp), kind=0, rows=256, cols=256, row_stride=256, col_stride=1)
    mlp = torch.empty(64 * 8192, dtype=torch.bfloat16, device=DEV)
    ops.chain_pack(d(w1), kind=0, rows=1024, cols=256, row_stride=256, col_stride=1, col_scale=d(g2), out=mlp, bundle_stride=2)
    ops.chain_pack(d(w2), kind=1, rows=256, cols=1024, row_stride=1024, col_stride=1, out=mlp, bundle_stride=2, bundle_offset=1)
    pq = ops.chain_pack(d(wq), kind=0, rows=768, cols=256, row_stride=256, col_stride=1, col_scale=d(g1))
    segs = [(ops.ptr(pp), 8), (ops.ptr(mlp), 64)] + ([(ops.ptr(pq), 24)] if with_qkv else [])
    xd, od = d(x).clone(), d(o).bfloat16()
    qo = torch.zeros(M, 768, dtype=torch.bfloat16, device=DEV)
    bpd, b1d, b2d, bqd = d(bp), d(b1f), d(b2), d(bqf)
    kw = {}
    if save:
        xh2o, xh1o = (torch.zeros(M, 256, dtype=torch.bfloat16, device=DEV) for _ in range(2))
        rs2o, rs1o = torch.zeros(M, device=DEV), torch.zeros(M, device=DEV)
        kw = dict(xhat2=ops.ptr(xh2o), rstd2=ops.ptr(rs2o))
        if with_qkv:
            kw.update(xhat1n=ops.ptr(xh1o), rstd1n=ops.ptr(rs1o))
    a = ops.make_chain_b_fwd(M=M, segs=segs, o=ops.ptr(od), x=ops.ptr(xd), b_proj=ops.ptr(bpd), b1=ops.ptr(b1d), b2=ops.ptr(b2d),
                             b_qkv=ops.ptr(bqd) if with_qkv else None, qkv=ops.ptr(qo) if with_qkv else None, **kw, **dkw)
    _lib.call("hma_chain_b_fwd", ops.stream_ptr(), C.byref(a))
    torch.cuda.synchronize()
    if p_drop > 0:  # a dropped output element is the residual exactly; the masks do drop
        gone = m1 == 0
        assert 0.05 < gone.float().mean() < 0.15
        assert (xd.cpu()[gone] - x1[gone]).abs().max() < 1e-5
    close(xd, x2, 3e-3, "x2")
    assert rms(xd.cpu() - x, x2 - x) < 5e-3
    if with_qkv:
        close(qo, qkv, 4 * BF, "next qkv")
        assert rms(qo, qkv) < 6e-3
        assert (qo.float().cpu() - qkv).abs().max() < 0.2 * (qo.float().cpu() - qkv.flip(1)).abs().max()
    if save:  # what the backward re-reads
        close(xh2o, F.layer_norm(x1, (256,), eps=1e-5), 2 * BF, "xhat2")
        close(rs2o, torch.rsqrt(x1.var(dim=1, unbiased=False) + 1e-5), 1e-4, "rstd2")
        if with_qkv:
            close(xh1o, F.layer_norm(x2, (256,), eps=1e-5), 3 * BF, "next xhat1")
            close(rs1o, torch.rsqrt(x2.var(dim=1, unbiased=False) + 1e-5), 3e-3, "next rstd1")


@pytest.mark.parametrize("B,T,S,A", [(1, 2, 16, 0), (2, 3, 256, 64), (3, 4, 80, 16)])
def test_readout_ce_fused(B, T, S, A):
    """out_x_proj + the factorised cross-entropy in one launch (st_mask_git.py:681-683, :603-630): loss / accuracy sums and the bf16
    gradient of the logits against fp32 PyTorch on the same bf16-rounded operands; rows of frame 0 and unmasked rows do not count."""
    gq = lambda s_: torch.Generator().manual_seed(s_)
    SA, Mi = S + A, B * T * S
    x = torch.randn(B * T * SA, 256, generator=gq(1)) * 1.2
    w = rb(torch.randn(1024, 256, generator=gq(2)) * 0.08)
    bias = torch.randn(1024, generator=gq(3)) * 0.1
    mask_id = 262144
    labels = torch.randint(0, 262144, (Mi,), generator=gq(4))
    ids = torch.where(torch.rand(Mi, generator=gq(5)) < 0.6, torch.full((Mi,), mask_id), labels)
    # make some rows easy so that the accuracy count is not trivially zero: the label's two factors get a large logit
    xi = x.view(B * T, SA, 256)[:, :S].reshape(Mi, 256)
    logits = rb(xi) @ w.t() + bias
    easy = torch.arange(Mi) % 3 == 0
    f0, f1 = labels % 512, (labels // 512) % 512
    boost = torch.zeros(Mi, 1024)
    boost[torch.arange(Mi), f0] = 30.0
    boost[torch.arange(Mi), 512 + f1] = 30.0
    bias_rows = boost * easy[:, None]  # (folded into the reference only through x: emulate by editing labels instead)
    # simpler: for easy rows take the label FROM the arg-max of the logits
    am0, am1 = logits[:, :512].argmax(1), logits[:, 512:].argmax(1)
    labels = torch.where(easy, am1 * 512 + am0, labels)
    f0, f1 = labels % 512, (labels // 512) % 512
    frame_t = (torch.arange(Mi) // S) % T
    live = (frame_t >= 1) & (ids == mask_id)
    nmask = live.sum().item()
    assert nmask > 0
    eps, gs = 0.01, 0.37
    lp0, lp1 = F.log_softmax(logits[:, :512], 1), F.log_softmax(logits[:, 512:], 1)
    row = lambda lp, f: -(1 - eps) * lp[torch.arange(Mi), f] - eps * lp.mean(1)
    loss_rows = row(lp0, f0) + row(lp1, f1)
    ok = (am0 == f0) & (am1 == f1)
    oh = lambda f: F.one_hot(f, 512).float() * (1 - eps) + eps / 512
    gref = torch.cat([lp0.exp() - oh(f0), lp1.exp() - oh(f1)], 1) * (gs / nmask) * live[:, None]
    d = lambda t: t.to(DEV).contiguous()
    pw = ops.chain_pack(d(w), kind=0, rows=1024, cols=256, row_stride=256, col_stride=1)
    stats = torch.tensor([0.0, 0.0, float(nmask), 0.0, 0.0, 0.0, 0.0, 0.0], device=DEV)
    dl = torch.full((Mi, 1024), 7.0, dtype=torch.bfloat16, device=DEV)
    xd, bd, idd, lbd = d(x), d(bias), d(ids), d(labels)
    a = ops.make_readout_ce(rows=Mi, segs=[(ops.ptr(pw), 32)], x=ops.ptr(xd), bias=ops.ptr(bd), input_ids=ops.ptr(idd), labels=ops.ptr(lbd),
                            stats=ops.ptr(stats), dlogits=ops.ptr(dl), grad_scale_dev=None, grad_scale=gs, S=S, SA=SA, T=T, mask_id=mask_id,
                            label_smoothing=eps)
    _lib.call("hma_readout_ce", ops.stream_ptr(), C.byref(a))
    torch.cuda.synchronize()
    st = stats.cpu()
    assert abs(st[0].item() - loss_rows[live].sum().item()) <= 2e-4 * abs(loss_rows[live].sum().item())
    assert st[1].item() == float((ok & live).sum().item()) and st[1].item() > 0
    assert st[2].item() == float(nmask)
    g = dl.float().cpu()
    assert torch.equal(g[~live], torch.zeros_like(g[~live]))
    close(g, gref, 2 * BF, "dlogits")
    assert rms(g, gref) < 6e-3
    # and against the two kernels it replaces
    lg = torch.empty(Mi, 1024, device=DEV)
    ga = ops.make_gemm_nt(A=ops.ptr(xd), lda=256, a_kind=_lib.A_F32, a_group=(S, SA), W=ops.ptr(d(w).bfloat16()), ldw=256, M=Mi, N=1024, K=256,
                          epi=_lib.EPI_F32, Cp=ops.ptr(lg), ldc=1024, bias=ops.ptr(bd))
    _lib.call("hma_gemm_nt", ops.stream_ptr(), C.byref(ga))
    stats2 = torch.tensor([0.0, 0.0, float(nmask), 0.0, 0.0, 0.0, 0.0, 0.0], device=DEV)
    dl2 = torch.empty(Mi, 1024, dtype=torch.bfloat16, device=DEV)
    _lib.call("hma_ce_fwd_bwd", ops.stream_ptr(), ops.ptr(lg), ops.ptr(idd.view(B, T * S)), ops.ptr(lbd), ops.ptr(stats2), ops.ptr(dl2), None, gs, B, T, S,
              mask_id, eps)
    torch.cuda.synchronize()
    assert abs(stats2[0].item() - st[0].item()) <= 1e-5 * abs(st[0].item()) and stats2[1].item() == st[1].item()
    assert rms(dl2, dl) < 5e-3


@pytest.mark.parametrize("B,SA,with_qkv,use_mod,p_drop", [
    (1, 16, True, True, 0.0), (2, 48, True, True, 0.0), (3, 320, True, True, 0.0), (2, 320, False, True, 0.0), (32, 320, True, True, 0.0),
    # blocks without action tokens (no ModulateLayer), and the MAR configs' mlp_drop (mar_n32_h8_d256_action.json:17)
    (2, 48, True, False, 0.0), (3, 256, True, False, 0.0), (2, 256, False, False, 0.0), (16, 256, True, False, 0.0),
    (2, 48, True, True, 0.05), (3, 320, False, True, 0.1), (16, 320, True, True, 0.05)])
def test_chain_ab_fwd_equals_three_launches(B, SA, with_qkv, use_mod, p_drop):
    """hma_chain_ab_fwd (chain A + causal temporal attention + chain B over columns of T = 16 frames, one launch) against the three
    launches it replaces on the same inputs -- hma_chain_a_fwd (training form), hma_attn_temporal_fwd, hma_chain_b_fwd (training form):
    the same arithmetic in the same order, so every output agrees to bf16 rounding of identical fp32 values (bit-identical where no
    reduction order differs).  Also without the ModulateLayer (use_mod = False) and with the MLP's two nn.Dropout sites (p_drop > 0: the
    same counter-based masks in both paths).  Reference: st_transformer.py:86-112 and :85-86 of the next block, :24-27, attention.py:37-61."""
    T, scale = 16, 0.25
    M = B * T * SA
    gq = lambda s_: torch.Generator().manual_seed(s_)
    d = lambda t: t.to(DEV).contiguous()
    wps, wl, wqt, bps, bl, bqt = _weights(700)
    wpt = rb(torch.randn(256, 256, generator=gq(710)) * 0.06)
    w1 = torch.randn(1024, 256, generator=gq(711)) * 0.06
    w2 = rb(torch.randn(256, 1024, generator=gq(712)) * 0.04)
    wqs = torch.randn(768, 256, generator=gq(713)) * 0.06
    bpt, b1, b2, bqs = (torch.randn(n, generator=gq(714 + i)) * 0.1 for i, n in enumerate((256, 1024, 256, 768)))
    g2, g1 = 1 + 0.1 * torch.randn(256, generator=gq(720)), 1 + 0.1 * torch.randn(256, generator=gq(721))
    o_s = d(rb(torch.randn(M, 256, generator=gq(1)))).bfloat16()
    x0 = d(torch.randn(M, 256, generator=gq(2)) * 1.5 + 0.2)
    ss = d(torch.randn(B * T, 512, generator=gq(3)) * 0.3)
    p_ps, p_l, p_qt = _pack_nt(wps), _pack_nt(wl), _pack_nt(wqt)
    p_pt = _pack_nt(wpt)
    mlp = torch.empty(64 * 8192, dtype=torch.bfloat16, device=DEV)
    ops.chain_pack(d(w1), kind=0, rows=1024, cols=256, row_stride=256, col_stride=1, col_scale=d(g2), out=mlp, bundle_stride=2)
    ops.chain_pack(d(w2), kind=1, rows=256, cols=1024, row_stride=1024, col_stride=1, out=mlp, bundle_stride=2, bundle_offset=1)
    p_qs = ops.chain_pack(d(wqs), kind=0, rows=768, cols=256, row_stride=256, col_stride=1, col_scale=d(g1))
    bd = {k: d(v) for k, v in dict(bps=bps, bl=bl, bqt=bqt, bpt=bpt, b1=b1, b2=b2, bqs=bqs).items()}
    bf = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=DEV)
    f32 = lambda *s: torch.zeros(*s, dtype=torch.float32, device=DEV)

    def outs():
        return dict(xhat_m=bf(M, 256), xm=bf(M, 256), rstd_m=f32(M), x2b=bf(M, 256), qkv_t=bf(M, 768), o_t=bf(M, 256), xhat2=bf(M, 256),
                    rstd2=f32(M), xhat1n=bf(M, 256), rstd1n=f32(M), qkv_s=bf(M, 768))

    dkw = {}
    if p_drop > 0:
        seed = torch.tensor([4242], dtype=torch.int32, device=DEV)
        dkw = dict(drop_p=p_drop, drop_salt=6, drop_seed=ops.ptr(seed))
    # ---- three launches
    r, xr = outs(), x0.clone()
    if use_mod:
        a = ops.make_chain_a_fwd(M=M, segs=[(ops.ptr(p_ps), 8), (ops.ptr(p_l), 8), (ops.ptr(p_qt), 24)], o=ops.ptr(o_s), x=ops.ptr(xr),
                                 qkv=ops.ptr(r["qkv_t"]), ss=ops.ptr(ss), b_proj=ops.ptr(bd["bps"]), b_lin=ops.ptr(bd["bl"]),
                                 b_qkv=ops.ptr(bd["bqt"]), xhat=ops.ptr(r["xhat_m"]), xm=ops.ptr(r["xm"]), rstd=ops.ptr(r["rstd_m"]),
                                 x_bf16=ops.ptr(r["x2b"]), rows_per_frame=SA, use_mod=True)
    else:
        a = ops.make_chain_a_fwd(M=M, segs=[(ops.ptr(p_ps), 8), (ops.ptr(p_qt), 24)], o=ops.ptr(o_s), x=ops.ptr(xr),
                                 qkv=ops.ptr(r["qkv_t"]), ss=None, b_proj=ops.ptr(bd["bps"]), b_lin=None, b_qkv=ops.ptr(bd["bqt"]),
                                 x_bf16=ops.ptr(r["x2b"]), rows_per_frame=SA, use_mod=False)
    _lib.call("hma_chain_a_fwd", ops.stream_ptr(), C.byref(a))
    _lib.call("hma_attn_temporal_fwd", ops.stream_ptr(), ops.ptr(r["qkv_t"]), ops.ptr(r["o_t"]), B, T, SA, scale)
    kwq = dict(xhat1n=ops.ptr(r["xhat1n"]), rstd1n=ops.ptr(r["rstd1n"]), qkv=ops.ptr(r["qkv_s"]), b_qkv=ops.ptr(bd["bqs"])) if with_qkv else {}
    b = ops.make_chain_b_fwd(M=M, segs=[(ops.ptr(p_pt), 8), (ops.ptr(mlp), 64)] + ([(ops.ptr(p_qs), 24)] if with_qkv else []),
                             o=ops.ptr(r["o_t"]), x=ops.ptr(xr), b_proj=ops.ptr(bd["bpt"]), b1=ops.ptr(bd["b1"]), b2=ops.ptr(bd["b2"]),
                             xhat2=ops.ptr(r["xhat2"]), rstd2=ops.ptr(r["rstd2"]), **kwq, **dkw)
    _lib.call("hma_chain_b_fwd", ops.stream_ptr(), C.byref(b))
    # ---- one launch
    f, xf = outs(), x0.clone()
    kwq = dict(xhat1n=ops.ptr(f["xhat1n"]), rstd1n=ops.ptr(f["rstd1n"]), qkv_s=ops.ptr(f["qkv_s"]), b_qkv_s=ops.ptr(bd["bqs"])) if with_qkv else {}
    kwm = dict(ss=ops.ptr(ss), xhat_m=ops.ptr(f["xhat_m"]), xm=ops.ptr(f["xm"]), rstd_m=ops.ptr(f["rstd_m"]), b_lin=ops.ptr(bd["bl"])) if use_mod else {}
    ab = ops.make_chain_ab_fwd(B=B, SA=SA, segs=[(ops.ptr(p_ps), 8), (ops.ptr(p_l), 8) if use_mod else (None, 0), (ops.ptr(p_qt), 24),
                                                  (ops.ptr(p_pt), 8), (ops.ptr(mlp), 64),
                                                  (ops.ptr(p_qs) if with_qkv else None, 24 if with_qkv else 0)],
                               o_s=ops.ptr(o_s), x=ops.ptr(xf), b1=ops.ptr(bd["b1"]), x2b=ops.ptr(f["x2b"]), qkv_t=ops.ptr(f["qkv_t"]),
                               o_t=ops.ptr(f["o_t"]), xhat2=ops.ptr(f["xhat2"]), rstd2=ops.ptr(f["rstd2"]), attn_scale=scale,
                               b_proj_s=ops.ptr(bd["bps"]), b_qkv_t=ops.ptr(bd["bqt"]), b_proj_t=ops.ptr(bd["bpt"]),
                               b2=ops.ptr(bd["b2"]), **kwm, **kwq, **dkw)
    _lib.call("hma_chain_ab_fwd", ops.stream_ptr(), C.byref(ab))
    torch.cuda.synchronize()
    names = (["xhat_m", "xm", "rstd_m"] if use_mod else []) + ["x2b", "qkv_t", "o_t", "xhat2", "rstd2"] + (["xhat1n", "rstd1n", "qkv_s"] if with_qkv else [])
    if p_drop > 0:  # the masks did something: against the same launch without them
        assert (xf - x0).abs().max() > 0
    for k in names:
        assert torch.isfinite(f[k].float()).all(), k
        tol = 1e-5 if k.startswith("rstd") else 2 * BF
        close(f[k], r[k].float(), tol, k)
    close(xf, xr, 1e-5, "x")
    # the temporal attention's output is what the three-launch path's kernel gives on the fused path's own qkv, exactly or to a rounding
    assert (f["o_t"].float() - r["o_t"].float()).abs().max() <= BF * r["o_t"].float().abs().max()
