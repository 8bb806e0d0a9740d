#!/usr/bin/env python3
"""Timing of the layer's weight-gradient launches as the engine issues them at the bench size (M = 163840): the three
hma_gemm_tn_pair calls of a block (fc2 + fc1 with the fragment-order operands of hma_mlp_bwd, projection + qkv of each
attention) and the single linear_out problem.  Operands rotate over 3 copies so the 256 MB MALL does not serve them.
TN_ROWMAJOR=1: the MLP pair with row-major gelu(u) / du (what the fragment order costs the DMA)."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hma_amd import _lib
if os.environ.get("HMA_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["HMA_LIB"])
from hma_amd import ops
from hma_amd._lib import A_BF16, A_BF16_AFFINE, A_BF16_FRAG32

dev = "cuda"
M = int(os.environ.get("GEMM_M", 163840))
NB = 3
WS = torch.empty(256 * (65536 + 256), device=dev)
ROWMAJOR = bool(os.environ.get("TN_ROWMAJOR"))


def prob(N, K, bias, affine, yfrag=False, afrag=False):
    ps = []
    dW = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev) if bias else None
    gam = torch.randn(K, device=dev) if affine else None
    bet = torch.randn(K, device=dev) if affine else None
    for _ in range(NB):
        dy = torch.randn(M, N, device=dev).bfloat16()
        x = torch.randn(M, K, device=dev).bfloat16()
        g = ops.make_gemm_tn(dY=ops.ptr(dy), ldy=N, y_kind=A_BF16_FRAG32 if yfrag and not ROWMAJOR else A_BF16, A=ops.ptr(x), lda=K,
                             a_kind=A_BF16_AFFINE if affine else (A_BF16_FRAG32 if afrag and not ROWMAJOR else A_BF16), M=M, N=N, K=K,
                             dW=ops.ptr(dW), lddw=K, dBias=ops.ptr(db), gamma=ops.ptr(gam), beta=ops.ptr(bet), ws=ops.ptr(WS),
                             ws_elems=WS.numel())
        ps.append((g, dy, x))
    return dict(ps=ps, keep=(dW, db, gam, bet), mb=(M * N + M * K) * 2 / 1e6, fl=2.0 * M * N * K)


def run(name, a, b=None):
    def fn(i):
        if b is None:
            _lib.call("hma_gemm_tn", ops.stream_ptr(), C.byref(a["ps"][i % NB][0]))
        else:
            _lib.call("hma_gemm_tn_pair", ops.stream_ptr(), C.byref(a["ps"][i % NB][0]), C.byref(b["ps"][i % NB][0]))
    for i in range(12): fn(i)
    torch.cuda.synchronize()
    reps = int(os.environ.get("TN_REPS", 60))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    mb = a["mb"] + (b["mb"] if b else 0)
    fl = a["fl"] + (b["fl"] if b else 0)
    print(f"{name:44s} {us:8.1f} us   {mb:6.0f} MB operands   {mb / us:6.2f} TB/s {fl / us / 1e6:7.1f} TFLOP/s", flush=True)


_w = torch.randn(8192, 8192, device=dev)
for _ in range(20): _w @ _w  # clocks up before the first shape
torch.cuda.synchronize()
which = (os.environ.get("TN_SHAPES") or "mlp,attn_t,attn_s,lin,fc2,fc1,qkv,proj").split(",")
if "mlp" in which:
    run("pair fc2 (A frag) + fc1 (dY frag, affine)", prob(256, 1024, True, False, afrag=True), prob(1024, 256, True, True, yfrag=True))
if "attn_t" in which:
    run("pair proj_t + qkv_t", prob(256, 256, True, False), prob(768, 256, False, False))
if "attn_s" in which:
    run("pair proj_s + qkv_s (affine)", prob(256, 256, True, False), prob(768, 256, False, True))
if "lin" in which:
    run("single linear_out N256 K256", prob(256, 256, True, False))
if "fc2" in which:
    run("single fc2 (A frag)", prob(256, 1024, True, False, afrag=True))
if "fc1" in which:
    run("single fc1 (dY frag, affine)", prob(1024, 256, True, True, yfrag=True))
if "qkv" in which:
    run("single qkv N768 K256", prob(768, 256, False, False))
if "proj" in which:
    run("single proj N256 K256", prob(256, 256, True, False))
if "same" in which:  # two copies of one problem in a pair: the half-chip behaviour of each shape
    run("pair fc2 + fc2 (frag)", prob(256, 1024, True, False, afrag=True), prob(256, 1024, True, False, afrag=True))
    run("pair fc1 + fc1 (frag, affine)", prob(1024, 256, True, True, yfrag=True), prob(1024, 256, True, True, yfrag=True))
    run("pair qkv + qkv", prob(768, 256, False, False), prob(768, 256, False, False))
    run("pair proj + proj", prob(256, 256, True, False), prob(256, 256, True, False))
    run("pair fc2 + fc1, no bias / affine", prob(256, 1024, False, False, afrag=True), prob(1024, 256, False, False, yfrag=True))
