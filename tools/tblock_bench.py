"""Isolated timing of the fused temporal-block probe (tools/probes/tblock_fwd_experiment.hip, built by tools/tblock_variants.sh into
variants/libtb_*.so) against the three launches it would replace, at the bench shape (B = 32).  HMA_TB=<path of the probe library>."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hma_amd import _lib, ops  # noqa: E402
from hma_amd._lib import A_BF16, EPI_BF16, EPI_RESID  # noqa: E402

c_vp, c_i64, c_i32 = C.c_void_p, C.c_int64, C.c_int32


class TBlockFwd(C.Structure):
    _fields_ = [("xb", c_vp), ("x", c_vp), ("wqkvp", c_vp), ("wprojp", c_vp), ("bqkv", c_vp), ("bproj", c_vp), ("qkv", c_vp), ("o", c_vp),
                ("ln_xhat", c_vp), ("ln_rstd", c_vp), ("ln_eps", C.c_float), ("scale", C.c_float), ("B", c_i64), ("T", c_i32), ("SA", c_i32)]


tb = C.CDLL(os.path.abspath(os.environ["HMA_TB"]))
tb.hma_tblock_pack.argtypes = [c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_i64]
tb.hma_tblock_fwd.argtypes = [c_vp, C.POINTER(TBlockFwd)]
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
assert tb.hma_tblock_pack(st, ops.ptr(wqkv), ops.ptr(wq_p), 0, 1, 0, 0) == 0
assert tb.hma_tblock_pack(st, ops.ptr(wproj), ops.ptr(wp_p), 1, 1, 0, 0) == 0
qkv = torch.empty(M, 768, dtype=bf, device=dev)
o = torch.empty(M, 256, dtype=bf, device=dev)
xh = torch.empty(M, 256, dtype=bf, device=dev)
rstd = torch.empty(M, device=dev)
a = TBlockFwd()
a.xb, a.x, a.wqkvp, a.wprojp, a.bqkv, a.bproj, a.qkv, a.o = ops.ptr(xb), ops.ptr(x), ops.ptr(wq_p), ops.ptr(wp_p), None, ops.ptr(bproj), ops.ptr(qkv), ops.ptr(o)
a.ln_xhat, a.ln_rstd, a.ln_eps, a.scale, a.B, a.T, a.SA = ops.ptr(xh), ops.ptr(rstd), 1e-5, 0.25, B, T, SA
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


def fused():
    assert tb.hma_tblock_fwd(st, C.byref(a)) == 0


def unfused():
    _lib.call("hma_gemm_nt", st, C.byref(g1))
    _lib.call("hma_attn_temporal_fwd", st, ops.ptr(qkv), ops.ptr(o), B, T, SA, 0.25)
    _lib.call("hma_gemm_nt", st, C.byref(g2))


tf = timeit(fused)
tu = timeit(unfused)
print(f"{os.environ['HMA_TB']:>28s}  fused {tf:7.1f} us   three launches {tu:7.1f} us")
