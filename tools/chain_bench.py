"""Isolated timing of the chain kernels at the bench shape (M = 163840) beside the launches they replace.
HMA_LIB=<path> selects a variant build."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hma_amd import _lib, ops  # noqa: E402
from hma_amd._lib import A_BF16, EPI_BF16, EPI_RESID  # noqa: E402

if os.environ.get("HMA_LIB"):
    _lib.LIB_PATH = os.environ["HMA_LIB"]
M = int(os.environ.get("M", 163840))
RPF = 320
dev = "cuda"
torch.manual_seed(0)
bf = torch.bfloat16
mk = lambda *s: torch.randn(*s, device=dev)
SKEW = int(os.environ.get("SKEW", 0))  # bytes of extra offset between consecutive activation arrays (0: back to back, as torch packs them)
_pool = torch.empty(4 << 30, dtype=torch.uint8, device=dev) if SKEW else None
_off = [0]


def carve(shape, dtype):
    if _pool is None:
        return torch.empty(*shape, device=dev, dtype=dtype)
    n = 1
    for d in shape:
        n *= d
    nbytes = n * torch.empty(0, dtype=dtype).element_size()
    t = _pool[_off[0]:_off[0] + nbytes].view(dtype).view(*shape)
    _off[0] += (nbytes + SKEW + 255) // 256 * 256
    return t


o = carve((M, 256), bf).copy_(mk(M, 256))
x = carve((M, 256), torch.float32).copy_(mk(M, 256))
ss = mk(M // RPF, 512) * 0.3
wp, wl, wq = mk(256, 256) * 0.05, mk(256, 256) * 0.05, mk(768, 256) * 0.05
bp, bl, bq = mk(256), mk(256), mk(768)
xhat, xm, xb = (carve((M, 256), bf) for _ in range(3))
rstd = torch.empty(M, device=dev)
qkv = carve((M, 768), bf)
pk = lambda w: ops.chain_pack(w.contiguous(), kind=0, rows=w.shape[0], cols=256, row_stride=256, col_stride=1)
pkt = lambda w, c=0: ops.chain_pack(w.contiguous()[256 * c:], kind=0, rows=256, cols=256, row_stride=1, col_stride=256)
pwp, pwl, pwq = pk(wp), pk(wl), pk(wq)
fa = ops.make_chain_a_fwd(M=M, segs=[(pwp.data_ptr(), 8), (pwl.data_ptr(), 8), (pwq.data_ptr(), 24)], o=o.data_ptr(), x=x.data_ptr(),
                          qkv=qkv.data_ptr(), ss=ss.data_ptr(), b_proj=bp.data_ptr(), b_lin=bl.data_ptr(), b_qkv=bq.data_ptr(),
                          xhat=xhat.data_ptr(), xm=xm.data_ptr(), rstd=rstd.data_ptr(), x_bf16=xb.data_ptr(), rows_per_frame=RPF)
dqkv = carve((M, 768), bf).copy_(mk(M, 768) * 0.02)
dx = carve((M, 256), torch.float32).copy_(mk(M, 256) * 0.02)
d2, d1, do = (carve((M, 256), bf) for _ in range(3))
dss = torch.zeros(M // RPF, 512, device=dev)
twq = torch.cat([pkt(wq, c) for c in range(3)])
twl, twp = pkt(wl), pkt(wp)
rstd.fill_(1.0)
ba = ops.make_chain_a_bwd(M=M, segs=[(twq.data_ptr(), 24), (twl.data_ptr(), 8), (twp.data_ptr(), 8)], dqkv=dqkv.data_ptr(),
                          dx=dx.data_ptr(), dx1_bf16=d1.data_ptr(), d_o=do.data_ptr(), xhat=xhat.data_ptr(), rstd=rstd.data_ptr(),
                          ss=ss.data_ptr(), dx2_bf16=d2.data_ptr(), dss=dss.data_ptr(), rows_per_frame=RPF)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


st = torch.cuda.current_stream().cuda_stream
B_ONLY = bool(os.environ.get("CHAIN_B_ONLY"))  # (measurement builds whose chain A does not fit LDS, e.g. -DCH_NS=8)
t_f = 1.0 if B_ONLY else timeit(lambda: _lib.call("hma_chain_a_fwd", st, C.byref(fa)))
t_b = 1.0 if B_ONLY else timeit(lambda: _lib.call("hma_chain_a_bwd", st, C.byref(ba)))
# what they replace
wpb, wlb, wqb = wp.to(bf), wl.to(bf), wq.to(bf)
g1 = ops.make_gemm_nt(A=o.data_ptr(), lda=256, a_kind=A_BF16, W=wpb.data_ptr(), ldw=256, M=M, N=256, K=256, epi=EPI_RESID, Cp=x.data_ptr(),
                      ldc=256, bias=bp.data_ptr(), ln_xhat=xhat.data_ptr(), ln_rstd=rstd.data_ptr(), ln_eps=1e-6, ln_ss=ss.data_ptr(),
                      ln_xm=xm.data_ptr(), ln_rows_per_frame=RPF)
g2 = ops.make_gemm_nt(A=xm.data_ptr(), lda=256, a_kind=A_BF16, W=wlb.data_ptr(), ldw=256, M=M, N=256, K=256, epi=EPI_RESID, Cp=x.data_ptr(),
                      ldc=256, bias=bl.data_ptr(), C2=xb.data_ptr(), ldc2=256)
g3 = ops.make_gemm_nt(A=xb.data_ptr(), lda=256, a_kind=A_BF16, W=wqb.data_ptr(), ldw=256, M=M, N=768, K=256, epi=EPI_BF16, Cp=qkv.data_ptr(),
                      ldc=768, bias=bq.data_ptr())
t_old = [timeit(lambda g=g: _lib.call("hma_gemm_nt", st, C.byref(g))) for g in (g1, g2, g3)]
by_f = M * 5632.0
print(f"skew {SKEW:8d} ", end="")
print(f"{os.environ.get('HMA_LIB', 'default'):>24s}  chain A fwd {t_f:7.1f} us ({by_f / t_f / 1e6:5.2f} TB/s)  [3 launches: "
      f"{' + '.join(f'{t:.0f}' for t in t_old)} = {sum(t_old):.0f} us]   chain A bwd {t_b:7.1f} us ({M * 5632.0 / t_b / 1e6:5.2f} TB/s)")

# ---- chain S backward (spatial qkv dgrad + norm1 backward + residual) beside the two launches it replaces
if hasattr(_lib.load(), "hma_chain_s_bwd") and not B_ONLY:
    gam = torch.rand(256, device=dev) + 0.5
    tws = torch.cat([ops.chain_pack(wq.contiguous()[256 * c:], kind=0, rows=256, cols=256, row_stride=1, col_stride=256, row_scale=gam)
                     for c in range(3)])
    sb = ops.make_chain_s_bwd(M=M, segs=[(tws.data_ptr(), 24)], dqkv=dqkv.data_ptr(), dx=dx.data_ptr(), xhat=xhat.data_ptr(),
                              rstd=rstd.data_ptr(), dx_bf16=d1.data_ptr())
    t_s = timeit(lambda: _lib.call("hma_chain_s_bwd", st, C.byref(sb)))
    wqt = wq.t().contiguous().to(bf)  # [256, 768]
    t256 = carve((M, 256), bf)
    gq_ = ops.make_gemm_nt(A=dqkv.data_ptr(), lda=768, a_kind=A_BF16, W=wqt.data_ptr(), ldw=768, M=M, N=256, K=768, epi=EPI_BF16,
                           Cp=t256.data_ptr(), ldc=256)
    dgam, dbet = torch.zeros(256, device=dev), torch.zeros(256, device=dev)
    t_g = timeit(lambda: _lib.call("hma_gemm_nt", st, C.byref(gq_)))
    t_l = timeit(lambda: _lib.call("hma_ln_bwd", st, t256.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), gam.data_ptr(), dx.data_ptr(),
                                   dgam.data_ptr(), dbet.data_ptr(), M, d1.data_ptr()))
    sbh = ops.make_chain_s_bwd(M=M, segs=[(tws.data_ptr(), 24)], dqkv=dqkv.data_ptr(), dx=dx.data_ptr(), xhat=xhat.data_ptr(),
                               rstd=rstd.data_ptr(), dx_bf16=d1.data_ptr(), hb_rows=RPF)
    t_sh = timeit(lambda: _lib.call("hma_chain_s_bwd", st, C.byref(sbh)))
    print(f"chain S bwd from head-blocked dqkv {t_sh:7.1f} us")
    print(f"{os.environ.get('HMA_LIB', 'default'):>24s}  chain S bwd {t_s:7.1f} us ({M * 4608.0 / t_s / 1e6:5.2f} TB/s)  [2 launches: {t_g:.0f} + {t_l:.0f} = "
          f"{t_g + t_l:.0f} us]")

if os.environ.get("CH_PROF"):
    lib = _lib.load()
    lib.hma_chain_debug_prof.argtypes = [C.c_void_p]
    buf = (C.c_ulonglong * 128)()
    _lib.call("hma_chain_a_fwd", st, C.byref(fa))
    torch.cuda.synchronize()
    lib.hma_chain_debug_prof(buf)
    print("fwd, workgroup 0, cycles.  compute waves: [barrier wait, stores, compute, loop/prefetch, end-of-tile wait]; loader (wave 7): [vmcnt wait, barrier wait, issue]")
    for w in range(8):
        v = [buf[w * 8 + i] for i in range(8)]
        print(f"  wave {w}: " + " ".join(f"{x:9d}" for x in v[:7]) + f"   sum {sum(v[:5])}")

# ---- chain B forward (training form: the two LayerNorm outputs saved) beside proj_t + LN GEMM, hma_mlp_fwd, qkv_s GEMM
w1, w2 = mk(1024, 256) * 0.05, mk(256, 1024) * 0.05
g2 = torch.ones(256, device=dev)
mlpw = torch.empty(64 * 8192, dtype=bf, device=dev)
ops.chain_pack(w1.contiguous(), kind=0, rows=1024, cols=256, row_stride=256, col_stride=1, col_scale=g2, out=mlpw, bundle_stride=2)
ops.chain_pack(w2.contiguous(), kind=1, rows=256, cols=1024, row_stride=1024, col_stride=1, out=mlpw, bundle_stride=2, bundle_offset=1)
b1v, b2v = mk(1024) * 0.1, mk(256) * 0.1
xh2, xh1n = (carve((M, 256), bf) for _ in range(2))
rs2, rs1n = torch.empty(M, device=dev), torch.empty(M, device=dev)
fb = ops.make_chain_b_fwd(M=M, segs=[(pwp.data_ptr(), 8), (mlpw.data_ptr(), 64), (pwq.data_ptr(), 24)], o=o.data_ptr(), x=x.data_ptr(),
                          b_proj=bp.data_ptr(), b1=b1v.data_ptr(), b2=b2v.data_ptr(), b_qkv=bq.data_ptr(), qkv=qkv.data_ptr(),
                          xhat2=xh2.data_ptr(), rstd2=rs2.data_ptr(), xhat1n=xh1n.data_ptr(), rstd1n=rs1n.data_ptr())
fbi = ops.make_chain_b_fwd(M=M, segs=[(pwp.data_ptr(), 8), (mlpw.data_ptr(), 64), (pwq.data_ptr(), 24)], o=o.data_ptr(), x=x.data_ptr(),
                           b_proj=bp.data_ptr(), b1=b1v.data_ptr(), b2=b2v.data_ptr(), b_qkv=bq.data_ptr(), qkv=qkv.data_ptr())
t_bt = timeit(lambda: _lib.call("hma_chain_b_fwd", st, C.byref(fb)))
t_bi = timeit(lambda: _lib.call("hma_chain_b_fwd", st, C.byref(fbi)))
from hma_amd._lib import A_BF16 as _A  # noqa: E402
gp = ops.make_gemm_nt(A=o.data_ptr(), lda=256, a_kind=A_BF16, W=wpb.data_ptr(), ldw=256, M=M, N=256, K=256, epi=EPI_RESID, Cp=x.data_ptr(),
                      ldc=256, bias=bp.data_ptr(), ln_xhat=xh2.data_ptr(), ln_rstd=rs2.data_ptr(), ln_eps=1e-5)
wpk = {k: (torch.randn(512 * 512, device=dev) * 0.05).to(bf) for k in ("w1p", "w2p")}
mf = ops.make_mlp_fwd(M=M, xhat=xh2.data_ptr(), x=x.data_ptr(), w1p=wpk["w1p"].data_ptr(), w2p=wpk["w2p"].data_ptr(), b1=b1v.data_ptr(),
                      b2=b2v.data_ptr(), ln_xhat=xh1n.data_ptr(), ln_rstd=rs1n.data_ptr(), ln_eps=1e-5)
gq = ops.make_gemm_nt(A=xh1n.data_ptr(), lda=256, a_kind=A_BF16, W=wqb.data_ptr(), ldw=256, M=M, N=768, K=256, epi=EPI_BF16,
                      Cp=qkv.data_ptr(), ldc=768, bias=bq.data_ptr())
t_o = [timeit(lambda: _lib.call("hma_gemm_nt", st, C.byref(gp))), timeit(lambda: _lib.call("hma_mlp_fwd", st, C.byref(mf))),
       timeit(lambda: _lib.call("hma_gemm_nt", st, C.byref(gq)))]
print(f"chain B fwd (train, saves xhat2 / xhat1') {t_bt:7.1f} us, (inference) {t_bi:7.1f} us   [3 launches: "
      f"{' + '.join(f'{t:.0f}' for t in t_o)} = {sum(t_o):.0f} us]   {2.0 * M * 256 * 3072 / t_bt / 1e6:6.0f} TFLOP/s")
