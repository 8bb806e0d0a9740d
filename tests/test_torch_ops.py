"""`torch.ops.hma.*` (hma_amd/torch_ops.py): registration, fake-tensor shapes (CPU) and values / gradients through the custom
ops against plain PyTorch fp32 math on the same bf16-rounded operands (GPU)."""
import math

import pytest
import torch

import hma_amd.torch_ops as T

BF = 2.0 ** -8


def test_ops_registered_with_fake_kernels():
    from torch._subclasses.fake_tensor import FakeTensorMode
    for name in T.OPS:
        assert hasattr(torch.ops.hma, name), name
    with FakeTensorMode():
        x = torch.empty(512, 256, dtype=torch.bfloat16, device="cuda")
        w = torch.empty(768, 256, dtype=torch.bfloat16, device="cuda")
        y = torch.ops.hma.linear(x, w, None)
        assert y.shape == (512, 768) and y.dtype == torch.bfloat16
        o, lse = torch.ops.hma.attn_spatial(y, 2, 256, 0.17)
        assert o.shape == (512, 256) and lse.shape == (512, 8)
        assert torch.ops.hma.attn_temporal(y, 2, 4, 64, 0.17).shape == (512, 256)
        xh, rs = torch.ops.hma.layer_norm(torch.empty(512, 256, device="cuda"), 1e-5)
        assert xh.dtype == torch.bfloat16 and rs.shape == (512,)


def test_cpu_tensors_are_refused():
    with pytest.raises(Exception):
        torch.ops.hma.linear(torch.zeros(16, 256, dtype=torch.bfloat16), torch.zeros(256, 256, dtype=torch.bfloat16), None)


def _close(a, b, tol, what):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs().max().item()
    assert err <= tol * (b.abs().max().item() + 1e-12), f"{what}: {err:.3e} vs scale {b.abs().max().item():.3e}"


@pytest.mark.gpu
def test_linear_forward_and_autograd():
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(640, 256, generator=g)).bfloat16()
    w = (torch.randn(768, 256, generator=g) * 0.05).bfloat16()
    b = torch.randn(768, generator=g) * 0.1
    xd = x.cuda().requires_grad_(True)
    wd = w.cuda().requires_grad_(True)
    bd = b.cuda().requires_grad_(True)
    y = torch.ops.hma.linear(xd, wd, bd)
    ref = x.float() @ w.float().t() + b
    _close(y, ref, 2 * BF, "linear")
    dy = (torch.randn(640, 768, generator=g) * 0.1).bfloat16()
    y.backward(dy.cuda())
    _close(xd.grad, dy.float() @ w.float(), 2 * BF, "dx")
    _close(wd.grad, dy.float().t() @ x.float(), 2e-2, "dW")
    _close(bd.grad, dy.float().sum(0), 2e-2, "dbias")


@pytest.mark.gpu
def test_attention_ops_and_autograd():
    g = torch.Generator().manual_seed(1)
    frames, n = 4, 320
    qkv = (torch.randn(frames * n, 768, generator=g) * 0.5).bfloat16()
    scale = 1.0 / math.sqrt(32)
    qd = qkv.cuda().requires_grad_(True)
    o, _ = torch.ops.hma.attn_spatial(qd, frames, n, scale)
    q, k, v = [t.reshape(frames, n, 8, 32).permute(0, 2, 1, 3) for t in qkv.float().requires_grad_(True).chunk(3, dim=1)]
    leaf = qkv.float().requires_grad_(True)
    q, k, v = [t.reshape(frames, n, 8, 32).permute(0, 2, 1, 3) for t in leaf.chunk(3, dim=1)]
    ref = torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1) @ v
    ref = ref.permute(0, 2, 1, 3).reshape(frames * n, 256)
    _close(o, ref, 3 * BF, "attn_spatial")
    d_o = (torch.randn(frames * n, 256, generator=g) * 0.1).bfloat16()
    o.backward(d_o.cuda())
    ref.backward(d_o.float())
    _close(qd.grad, leaf.grad, 3e-2, "attn_spatial dqkv")
    # temporal: rows (b, t, s), causal over t
    B, Tn, S = 2, 8, 40
    qkv = (torch.randn(B * Tn * S, 768, generator=g) * 0.5).bfloat16()
    qd = qkv.cuda().requires_grad_(True)
    o = torch.ops.hma.attn_temporal(qd, B, Tn, S, scale)
    leaf = qkv.float().requires_grad_(True)
    q, k, v = [t.reshape(B, Tn, S, 8, 32).permute(0, 2, 3, 1, 4) for t in leaf.chunk(3, dim=1)]  # b s h t c
    sc = q @ k.transpose(-1, -2) * scale
    sc = sc.masked_fill(torch.triu(torch.ones(Tn, Tn, dtype=torch.bool), 1), float("-inf"))
    ref = (torch.softmax(sc, dim=-1) @ v).permute(0, 3, 1, 2, 4).reshape(B * Tn * S, 256)
    _close(o, ref, 3 * BF, "attn_temporal")
    d_o = (torch.randn(B * Tn * S, 256, generator=g) * 0.1).bfloat16()
    o.backward(d_o.cuda())
    ref.backward(d_o.float())
    _close(qd.grad, leaf.grad, 3e-2, "attn_temporal dqkv")


@pytest.mark.gpu
def test_layer_norm_op():
    x = torch.randn(1000, 256, generator=torch.Generator().manual_seed(2)) * 2 + 0.5
    xh, rs = torch.ops.hma.layer_norm(x.cuda(), 1e-5)
    ref = torch.nn.functional.layer_norm(x, (256,), eps=1e-5)
    _close(xh, ref, 2 * BF, "xhat")
    _close(rs, torch.rsqrt(x.var(1, unbiased=False) + 1e-5), 1e-4, "rstd")


def _rms(a, b):
    a, b = a.float().cpu().double(), b.float().cpu().double()
    return ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30)).item()


@pytest.mark.gpu
def test_mlp_module_trains_through_autograd():
    """`Mlp(x)` under autograd returns gradients (reference: hma/model/st_transformer.py:24-27)."""
    from hma_amd.model import Mlp
    g = torch.Generator().manual_seed(3)
    ref = torch.nn.Sequential(torch.nn.Linear(256, 1024), torch.nn.GELU(), torch.nn.Linear(1024, 256))
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.dim() == 2 else 0.1))
    m = Mlp(256)
    m.load_state_dict({"fc1.weight": ref[0].weight, "fc1.bias": ref[0].bias, "fc2.weight": ref[2].weight, "fc2.bias": ref[2].bias})
    m = m.cuda().train()
    x = torch.randn(3, 200, 256, generator=g)
    r = torch.randn(3, 200, 256, generator=g) * 0.1
    xd = x.cuda().requires_grad_(True)
    y = m(xd)
    assert y.requires_grad and y.shape == x.shape
    (y * r.cuda()).sum().backward()
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    (yr * r).sum().backward()
    _close(y, yr, 1.5e-2, "Mlp forward")
    assert _rms(xd.grad, xr.grad) < 2e-2
    assert _rms(m.fc1.weight.grad, ref[0].weight.grad) < 2e-2
    assert _rms(m.fc2.weight.grad, ref[2].weight.grad) < 2e-2
    assert _rms(m.fc1.bias.grad, ref[0].bias.grad) < 2e-2
    assert _rms(m.fc2.bias.grad, ref[2].bias.grad) < 2e-2
    # without anything that requires a gradient in reach the module still runs (inference path, no graph)
    with torch.no_grad():
        assert not m(x.cuda()).requires_grad


@pytest.mark.gpu
@pytest.mark.parametrize("causal", [False, True])
def test_self_attention_module_trains_through_autograd(causal):
    """`SelfAttention(x, causal)` under autograd (reference: hma/model/attention.py:37-61, BasicSelfAttention)."""
    from hma_amd.model import SelfAttention
    g = torch.Generator().manual_seed(4)
    att = SelfAttention(num_heads=8, d_model=256, qkv_bias=False, proj_bias=True, qk_norm=False, use_mup=True)
    with torch.no_grad():
        att.qkv.weight.copy_(torch.randn(768, 256, generator=g) * 0.06)
        att.proj.weight.copy_(torch.randn(256, 256, generator=g) * 0.06)
        att.proj.bias.copy_(torch.randn(256, generator=g) * 0.1)
    wq, wp, bp = att.qkv.weight.detach().clone(), att.proj.weight.detach().clone(), att.proj.bias.detach().clone()
    att = att.cuda().train()
    Bn, N = (6, 16) if causal else (3, 320)
    x = torch.randn(Bn, N, 256, generator=g)
    r = torch.randn(Bn, N, 256, generator=g) * 0.1
    xd = x.cuda().requires_grad_(True)
    y = att(xd, causal=causal)
    (y * r.cuda()).sum().backward()
    # fp32 reference
    xr, wqr, wpr = x.clone().requires_grad_(True), wq.clone().requires_grad_(True), wp.clone().requires_grad_(True)
    q, k, v = [t.reshape(Bn, N, 8, 32).permute(0, 2, 1, 3) for t in (xr @ wqr.t()).chunk(3, dim=-1)]
    sc = (q * att.scale) @ k.transpose(-1, -2)
    if causal:
        sc = sc.masked_fill(torch.triu(torch.ones(N, N, dtype=torch.bool), 1), float("-inf"))
    o = (torch.softmax(sc, dim=-1) @ v).permute(0, 2, 1, 3).reshape(Bn, N, 256)
    yr = o @ wpr.t() + bp
    (yr * r).sum().backward()
    _close(y, yr, 2e-2, "SelfAttention forward")
    assert _rms(xd.grad, xr.grad) < 3e-2
    assert _rms(att.qkv.weight.grad, wqr.grad) < 3e-2
    assert _rms(att.proj.weight.grad, wpr.grad) < 3e-2


def _oracle_blocks(layers, x, a, dom, r):
    """The CPU oracle's st_block (pinned to the reference by G5) under torch autograd: output and the gradients of sum(y r)."""
    from oracle import st_maskgit_ref as R
    from tests.helpers import tiny_ref_config, tiny_state_dict
    cfg = tiny_ref_config()
    sd = tiny_state_dict(cfg)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("decoder.layers.")}
    full = dict(sd)
    full.update(params)
    xr = x.clone().requires_grad_(True)
    ar = None if a is None else a.clone().requires_grad_(True)
    y = xr
    for l in layers:
        y = R.st_block(full, cfg, l, y, ar, dom)
    (y * r).sum().backward()
    return y.detach(), xr.grad, (None if ar is None else ar.grad), {k: v.grad for k, v in params.items() if v.grad is not None}


def _check_block_grads(m, layers, dom, ref_grads, tag):
    from tests.helpers import rms_err
    worst, n = 0.0, 0
    for l in layers:
        for name, p in m.decoder.layers[l].named_parameters():
            key = f"decoder.layers.{l}.{name}"
            if "action_projectors." in name and (dom is None or f"action_projectors.{dom}." not in name):
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, f"{tag}: {key} belongs to another domain"
                continue
            assert p.grad is not None, f"{tag}: {key} has no gradient"
            ref = ref_grads[key]
            e = rms_err(p.grad, ref)
            tol = 6e-2 if float(ref.abs().max()) < 1e-3 or ref.numel() <= 512 else 3e-2  # (tiny tensors: bf16 rounding of the activations)
            assert e <= tol, f"{tag}: {key}: rms error {e:.3e}"
            worst, n = max(worst, e), n + 1
    assert n > 0
    return worst


@pytest.mark.gpu
def test_blocks_train_under_autograd():
    """STBlock / STTransformerDecoder are ordinary autograd modules in the reference (st_transformer.py:79-114, 172-177).  Here a
    forward with autograd on is one node over the engine's recorded plans: output, d x, d action embedding and every parameter
    gradient of the layers against the CPU oracle's blocks under torch autograd -- the whole stack with actions, ONE block in the
    middle of the stack with actions (an entry into the middle of the backward plan), one block without actions; and a backward
    whose saved activations were overwritten by a later forward fails loudly."""
    from tests.test_model_gpu import build_model
    from tests.helpers import golden, rel_err, rms_err
    g = golden("g5_stblock")
    m = build_model(train=True)
    L = len(m.decoder.layers)
    x, a = g["x"], g["a_emb"]
    gen = torch.Generator().manual_seed(11)
    r = torch.randn(x.shape, generator=gen)

    def run(mod, layers, xin, ain, dom, rr, tag):
        m.zero_grad()
        xd = xin.cuda().requires_grad_(True)
        ad = None if ain is None else ain.cuda().requires_grad_(True)
        y = mod(xd, action_ids=ad, domain=dom)
        assert y.requires_grad
        (y * rr.cuda()).sum().backward()
        yr, dxr, dar, pg = _oracle_blocks(layers, xin, ain, dom, rr)
        assert rel_err(y, yr) <= 1e-2, f"{tag}: forward"
        assert rms_err(xd.grad, dxr) <= 3e-2, f"{tag}: dx {rms_err(xd.grad, dxr):.3e}"
        if ain is not None:
            assert rms_err(ad.grad, dar) <= 3e-2, f"{tag}: d a_emb {rms_err(ad.grad, dar):.3e}"
        return _check_block_grads(m, layers, dom, pg, tag)

    run(m.decoder, list(range(L)), x, a, "domB", r, "decoder")
    mid = min(1, L - 1)
    run(m.decoder.layers[mid], [mid], x, a, "domA", r, "block with actions")
    x0 = x[:, :, :256].contiguous()
    run(m.decoder.layers[0], [0], x0, None, None, r[:, :, :256].contiguous(), "block without actions")
    # gradients accumulate over two backward passes, like autograd's
    m.zero_grad()
    for _ in range(2):
        xd = x.cuda().requires_grad_(True)
        (m.decoder.layers[0](xd, action_ids=a.cuda(), domain="domA") * r.cuda()).sum().backward()
    _, _, _, pg = _oracle_blocks([0], x, a, "domA", r)
    w = m.decoder.layers[0].mlp.fc1.weight
    assert rms_err(w.grad, 2 * pg["decoder.layers.0.mlp.fc1.weight"]) <= 3e-2
    # an external torch optimizer writes through the named parameters: the next forward runs on the new weights (the engine's bf16 /
    # packed copies are re-derived when the parameters' version counters move)
    from oracle import st_maskgit_ref as R
    from tests.helpers import tiny_ref_config, tiny_state_dict
    m.zero_grad()
    blk = m.decoder.layers[0]
    lr = 0.02 / max(float(v.abs().max()) for v in pg.values())
    opt = torch.optim.SGD([p for p in blk.parameters()], lr=lr)
    y0 = blk(x.cuda(), action_ids=a.cuda(), domain="domA")
    (y0 * r.cuda()).sum().backward()
    opt.step()
    with torch.no_grad():
        y1 = blk(x.cuda(), action_ids=a.cuda(), domain="domA")
    cfg_r = tiny_ref_config()
    sd2 = {k: (v - lr * pg[k] if k in pg else v) for k, v in tiny_state_dict(cfg_r).items()}
    y1_ref = R.st_block(sd2, cfg_r, 0, x, a, "domA")
    assert rel_err(y0.detach(), y1_ref) > 3e-2, "the update must move the output for this check to mean anything"
    assert rel_err(y1, y1_ref) <= 1e-2
    m.load_state_dict(tiny_state_dict(), strict=True)
    # one set of saved activations per layer: a backward after another forward through the same layer is refused
    m.zero_grad()
    xd = x.cuda().requires_grad_(True)
    y1 = m.decoder.layers[0](xd, action_ids=a.cuda(), domain="domA")
    m.decoder.layers[0](xd, action_ids=a.cuda(), domain="domA")
    with pytest.raises(RuntimeError, match="overwritten"):
        y1.sum().backward()
    with torch.no_grad():
        assert not m.decoder(x.cuda(), action_ids=a.cuda(), domain="domB").requires_grad


@pytest.mark.gpu
def test_embed_modulate_readout_ce_maskgit_step_adamw_ops():
    """The operator-level entry points SURVEY section 8(b) lists beside the GEMM / attention ops, as `torch.ops.hma.*`: values against
    plain PyTorch fp32 math (factorization_utils.py:31-54, st_mask_git.py:66-76, :603-630, :397-453, train_multi.py:593-598)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    dev = "cuda"
    V, MASK = 512, 262144
    # ---- embed
    B, T, S, A = 2, 3, 64, 16
    ids = torch.randint(0, V * V, (B, T, S), generator=g)
    ids[torch.rand(B, T, S, generator=g) < 0.3] = MASK
    e0, e1 = torch.randn(V, 256, generator=g), torch.randn(V, 256, generator=g)
    me, pos = torch.randn(1, 256, generator=g), torch.randn(1, T, S + A, 256, generator=g)
    a_emb = torch.randn(B, T, 256, generator=g)
    x = torch.ops.hma.embed(ids.to(dev), e0.to(dev), e1.to(dev), me.to(dev), pos.to(dev), a_emb.to(dev), A, MASK)
    tok = torch.where((ids == MASK)[..., None], me.expand(B, T, S, 256), e0[ids.clamp(max=V * V - 1) % V] + e1[ids.clamp(max=V * V - 1) // V])
    ref = torch.cat([tok, a_emb[:, :, None].expand(B, T, A, 256)], dim=2) + pos
    assert torch.allclose(x.cpu(), ref, atol=1e-6)
    # ---- modulate
    rows_pf, frames = 80, 6
    xr = torch.randn(frames * rows_pf, 256, generator=g) * 1.3 + 0.2
    ss = torch.randn(frames, 512, generator=g) * 0.3
    xhat, xm, rstd = torch.ops.hma.modulate(xr.to(dev), ss.to(dev), rows_pf, 1e-6)
    xh = F.layer_norm(xr, (256,), eps=1e-6)
    f = torch.arange(frames * rows_pf) // rows_pf
    _close(xhat, xh, 2 * BF, "modulate xhat")
    _close(xm, xh * (1 + ss[f, 256:]) + ss[f, :256], 3 * BF, "modulate xm")
    _close(rstd, torch.rsqrt(xr.var(1, unbiased=False) + 1e-6), 1e-4, "modulate rstd")
    # ---- readout_ce
    B, T, S = 2, 3, 16
    logits = torch.randn(B * T * S, 1024, generator=g) * 2
    labels = torch.randint(0, V * V, (B, T, S), generator=g)
    inp = labels.clone()
    inp[:, 1:][torch.rand(B, T - 1, S, generator=g) < 0.5] = MASK
    stats, dlog = torch.ops.hma.readout_ce(logits.to(dev), inp.to(dev), labels.to(dev), MASK, 0.01, 1.0)
    lr_ = logits.clone().requires_grad_(True)
    m = (inp == MASK).reshape(-1).float()
    lab = labels.reshape(-1)
    ce = sum(F.cross_entropy(lr_[:, k * V:(k + 1) * V], (lab // V ** k) % V, reduction="none", label_smoothing=0.01) for k in range(2))
    loss = (ce * m).sum() / m.sum()
    loss.backward()
    assert abs(stats[0].item() / stats[2].item() - loss.item()) <= 1e-4 * abs(loss.item()) and stats[2].item() == m.sum().item()
    _close(dlog, lr_.grad, 3 * BF, "dlogits")
    # ---- maskgit_step (last step: every token of the frame is written with the factor-wise arg-max)
    B, T, S = 3, 2, 32
    lg = torch.randn(B, T, S, 1024, generator=g)
    prompt = torch.full((B, T, S), MASK, dtype=torch.long)
    prompt[:, 0] = torch.randint(0, V * V, (B, S), generator=g)
    pd, um = prompt.to(dev), torch.zeros(B, S, dtype=torch.uint8, device=dev)
    conf = torch.ops.hma.maskgit_step(lg.to(dev), pd, um, 1, 0, True, MASK)
    want = lg[:, 1, :, :V].argmax(-1) + V * lg[:, 1, :, V:].argmax(-1)
    assert torch.equal(pd[:, 1].cpu(), want) and torch.equal(pd[:, 0].cpu(), prompt[:, 0]) and conf.shape == (B, S)
    # ---- adamw: one clipped step against torch.optim.AdamW
    n = 64 * 40
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 3
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pt], lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    pt.grad = gr.clone()
    torch.nn.utils.clip_grad_norm_([pt], 1.0)
    opt.step()
    pdv, mdv, vdv = p0.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    pb = torch.ops.hma.adamw(pdv, gr.to(dev), mdv, vdv, 1e-2, 0.9, 0.95, 1e-8, 0.05, 1, 1.0)
    assert torch.allclose(pdv.cpu(), pt.detach(), atol=2e-6, rtol=1e-5) and torch.equal(pb.cpu(), pdv.cpu().bfloat16())


@pytest.mark.gpu
def test_blocks_under_autograd_with_dropout():
    """mlp_drop > 0 in train() mode: the block's forward draws new masks per call (two calls differ), its backward runs on the masks of
    ITS forward (finite, and dominated by the residual path's identity: d sum(y r) / d x stays close to r), a backward after another
    training forward -- new masks on the device -- is refused, and eval() with gradients is refused (the saved-activation plans
    apply the Dropout).  (The masks and their backward are pinned at model / kernel level: the mlp_drop tests of tests/test_model_gpu.py and tests/test_chain_gpu.py.)"""
    from tests.test_model_gpu import build_model
    from tests.helpers import golden
    g = golden("g5_stblock")
    m = build_model(train=True, mlp_drop=0.1)
    blk = m.decoder.layers[0]
    x, a = g["x"].cuda(), g["a_emb"].cuda()
    gen = torch.Generator().manual_seed(3)
    r = torch.randn(x.shape, generator=gen).cuda()
    xd = x.clone().requires_grad_(True)
    y1 = blk(xd, action_ids=a, domain="domA")
    (y1 * r).sum().backward()
    assert torch.isfinite(xd.grad).all() and float(xd.grad.abs().max()) > 0
    y2 = blk(x, action_ids=a, domain="domA")
    assert float((y1.detach() - y2.detach()).abs().max()) > 1e-3, "two training forwards must draw different masks"
    # the residual path alone gives d (sum y r) / d x = r: the block's branches change it, Dropout must not break the identity part
    cos = torch.nn.functional.cosine_similarity(xd.grad.flatten(), r.flatten(), dim=0)
    assert float(cos) > 0.5
    xd2 = x.clone().requires_grad_(True)
    y3 = blk(xd2, action_ids=a, domain="domA")
    m.decoder.layers[1](x, action_ids=a, domain="domA")  # (another training forward: new masks)
    with pytest.raises(RuntimeError, match="Dropout masks"):
        y3.sum().backward()
    m.eval()
    with pytest.raises(NotImplementedError, match="eval"):
        blk(xd2, action_ids=a, domain="domA")
