#!/usr/bin/env python3
"""Per-shape timing of the layer GEMMs at the bench size (M = 163840): TFLOP/s of each C-ABI GEMM call."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hma_amd import _lib
if os.environ.get("HMA_DEBUG_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["HMA_DEBUG_LIB"])
from hma_amd import ops
from hma_amd._lib import *

dev = "cuda"
M = int(os.environ.get("GEMM_M", 163840))
def bf(*s): return torch.randn(*s, device=dev).bfloat16()
def f32(*s): return torch.randn(*s, device=dev)

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

rows = []
def nt(name, a, w, epi, bias=None, out=None, out2=None, aux=None, gamma=None, beta=None):
    fn = lambda: ops.linear(a, w, bias, epi=epi, out=out, out2=out2, aux=aux, gamma=gamma, beta=beta)
    ms = timeit(fn)
    fl = 2.0 * a.shape[0] * w.shape[0] * w.shape[1]
    rows.append((name, ms * 1e3, fl / ms / 1e9))
WS = torch.empty(256 * 65536, device=dev)
def tn(name, dy, x, N, K, bias=True, gamma=None, beta=None):
    dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev) if bias else None
    fn = lambda: ops.linear_wgrad(dy, x, dW, db, gamma=gamma, beta=beta, ws=None if os.environ.get('TN_ATOMIC') else WS)
    ms = timeit(fn)
    rows.append((name, ms * 1e3, 2.0 * M * N * K / ms / 1e9))

xh = bf(M, 256); x32 = f32(M, 256); g = f32(256); b = f32(256)
w768 = bf(768, 256); w256 = bf(256, 256); w1024 = bf(1024, 256); wk1024 = bf(256, 1024); wk768 = bf(256, 768)
b768 = f32(768); b256 = f32(256); b1024 = f32(1024)
qkv = torch.empty(M, 768, device=dev, dtype=torch.bfloat16)
o256 = torch.empty(M, 256, device=dev, dtype=torch.bfloat16)
u = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16); h = torch.empty_like(u)
xres = f32(M, 256); x2b = torch.empty(M, 256, device=dev, dtype=torch.bfloat16)
nt("fwd qkv  affine->bf16  N768 K256", xh, w768, EPI_BF16, out=qkv, gamma=g, beta=b)
nt("fwd proj bf16->resid   N256 K256", xh, w256, EPI_RESID, bias=b256, out=xres)
nt("fwd proj bf16->resid+c N256 K256", xh, w256, EPI_RESID, bias=b256, out=xres, out2=x2b)
nt("fwd qkvt bf16->bf16    N768 K256", xh, w768, EPI_BF16, out=qkv)
nt("fwd fc1  affine->gelu2 N1024 K256", xh, w1024, EPI_GELU2, bias=b1024, out=u, out2=h, gamma=g, beta=b)
nt("fwd fc2  bf16->resid   N256 K1024", h, wk1024, EPI_RESID, bias=b256, out=xres)
nt("bwd dfc2 f32->dgelu    N1024 K256", x32, w1024, EPI_DGELU, out=u, aux=u)
nt("bwd dfc1 bf16->bf16    N256 K1024", u, wk1024, EPI_BF16, out=o256)
nt("bwd dprj f32->bf16     N256 K256", x32, w256, EPI_BF16, out=o256)
nt("bwd dqkv bf16->resid   N256 K768", qkv, wk768, EPI_RESID, out=xres)
nt("bwd dqkv bf16->bf16    N256 K768", qkv, wk768, EPI_BF16, out=o256)
nt("ref  plain bf16->bf16  N256 K256", xh, w256, EPI_BF16, out=o256)
nt("ref  plain bf16->f32   N1024 K256", xh, w1024, EPI_F32)
tn("wgrad fc2  dy f32, a bf16    N256 K1024", x32, h, 256, 1024)
tn("wgrad fc1  dy bf16, a affine N1024 K256", u, xh, 1024, 256, gamma=g, beta=b)
tn("wgrad proj dy f32, a bf16    N256 K256", x32, xh, 256, 256)
tn("wgrad qkv  dy bf16, a bf16   N768 K256", qkv, xh, 768, 256, bias=False)
for r in rows:
    print(f"{r[0]:44s} {r[1]:9.1f} us  {r[2]:8.1f} TFLOP/s")
