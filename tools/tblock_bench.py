"""Isolated timing of the fused temporal block against the three launches it replaces, at the bench shape (B = 32)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hma_amd import _lib, ops  # noqa: E402
from hma_amd._lib import A_BF16, EPI_BF16, EPI_RESID  # noqa: E402

if os.environ.get("HMA_LIB"):
    _lib.LIB_PATH = os.environ["HMA_LIB"]
B, T, SA = int(os.environ.get("B", 32)), 16, 320
M = B * T * SA
dev = "cuda"
torch.manual_seed(0)
bf = torch.bfloat16
x = torch.randn(M, 256, device=dev)
xb = x.to(bf)
wqkv = torch.randn(768, 256, device=dev) * 0.08
wproj = torch.randn(256, 256, device=dev) * 0.08
bproj = torch.randn(256, device=dev) * 0.1
wq_p = torch.empty(384 * 512, dtype=bf, device=dev)
wp_p = torch.empty(128 * 512, dtype=bf, device=dev)
st = torch.cuda.current_stream().cuda_stream
_lib.call("hma_tblock_pack", st, ops.ptr(wqkv), ops.ptr(wq_p), 0, 1, 0, 0)
_lib.call("hma_tblock_pack", st, ops.ptr(wproj), ops.ptr(wp_p), 1, 1, 0, 0)
qkv = torch.empty(M, 768, dtype=bf, device=dev)
o = torch.empty(M, 256, dtype=bf, device=dev)
xh = torch.empty(M, 256, dtype=bf, device=dev)
rstd = torch.empty(M, device=dev)
a = ops.make_tblock_fwd(xb=ops.ptr(xb), x=ops.ptr(x), wqkvp=ops.ptr(wq_p), wprojp=ops.ptr(wp_p), bproj=ops.ptr(bproj), qkv=ops.ptr(qkv),
                        o=ops.ptr(o), ln_xhat=ops.ptr(xh), ln_rstd=ops.ptr(rstd), ln_eps=1e-5, scale=0.25, B=B, T=T, SA=SA)
wqb, wpb = wqkv.to(bf), wproj.to(bf)
g1 = ops.make_gemm_nt(A=ops.ptr(xb), lda=256, a_kind=A_BF16, W=ops.ptr(wqb), ldw=256, M=M, N=768, K=256, epi=EPI_BF16, Cp=ops.ptr(qkv), ldc=768)
g2 = ops.make_gemm_nt(A=ops.ptr(o), lda=256, a_kind=A_BF16, W=ops.ptr(wpb), ldw=256, M=M, N=256, K=256, epi=EPI_RESID, Cp=ops.ptr(x), ldc=256,
                      bias=ops.ptr(bproj), ln_xhat=ops.ptr(xh), ln_rstd=ops.ptr(rstd), ln_eps=1e-5)


def timeit(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def unfused():
    _lib.call("hma_gemm_nt", st, C.byref(g1))
    _lib.call("hma_attn_temporal_fwd", st, ops.ptr(qkv), ops.ptr(o), B, T, SA, 0.25)
    _lib.call("hma_gemm_nt", st, C.byref(g2))


tf = timeit(lambda: _lib.call("hma_tblock_fwd", st, C.byref(a)))
tu = timeit(unfused)
fl = 2.0 * M * 256 * 768 + 4.0 * M * T * 256 + 2.0 * M * 256 * 256
print(f"{os.environ.get('HMA_LIB', 'default'):>28s}  fused {tf:7.1f} us ({fl / tf / 1e6:6.0f} TFLOP/s)   three launches {tu:7.1f} us")
