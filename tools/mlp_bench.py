"""Isolated timing of the fused MLP kernels at the bench shape (M = 163840).  HMA_LIB=<path> selects a variant build."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hma_amd import _lib, ops  # noqa: E402

if os.environ.get("HMA_LIB"):
    _lib.LIB_PATH = os.environ["HMA_LIB"]
M = int(os.environ.get("M", 163840))
dev = "cuda"
torch.manual_seed(0)
bf = torch.bfloat16
xh = torch.randn(M, 256, device=dev).to(bf)
x = torch.randn(M, 256, device=dev)
dy = (torch.randn(M, 256, device=dev) * 0.02).to(bf)
dx = torch.randn(M, 256, device=dev) * 0.02
dxb = torch.empty(M, 256, device=dev, dtype=bf)
rstd = torch.rand(M, device=dev) + 0.5
hg = torch.empty(M, 1024, device=dev, dtype=bf)
du = torch.empty(M, 1024, device=dev, dtype=bf)
w = {k: (torch.randn(512 * 512, device=dev) * 0.05).to(bf) for k in ("w1p", "w2p", "w2tp", "w1tp")}
b1 = torch.randn(1024, device=dev) * 0.1
b2 = torch.randn(256, device=dev) * 0.1
lnx = torch.empty(M, 256, device=dev, dtype=bf)
lnr = torch.empty(M, device=dev)
f = ops.make_mlp_fwd(M=M, xhat=xh.data_ptr(), x=x.data_ptr(), w1p=w["w1p"].data_ptr(), w2p=w["w2p"].data_ptr(), b1=b1.data_ptr(),
                     b2=b2.data_ptr(), ln_xhat=lnx.data_ptr(), ln_rstd=lnr.data_ptr(), ln_eps=1e-5)
b = ops.make_mlp_bwd(M=M, xhat=xh.data_ptr(), rstd=rstd.data_ptr(), dy=dy.data_ptr(), dx=dx.data_ptr(), dx_bf16=dxb.data_ptr(),
                     w1p=w["w1p"].data_ptr(), w2tp=w["w2tp"].data_ptr(), w1tp=w["w1tp"].data_ptr(), b1=b1.data_ptr(),
                     hg=hg.data_ptr(), du=du.data_ptr())


def timeit(name, arg, n=10):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _lib.call(name, st, C.byref(arg))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        _lib.call(name, st, C.byref(arg))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tf = timeit("hma_mlp_fwd", f)
tb = timeit("hma_mlp_bwd", b)
fl = 2.0 * M * 256 * 1024 * 2
print(f"{os.environ.get('HMA_LIB', 'default'):>28s}  fwd {tf:7.1f} us ({fl / tf / 1e6:6.0f} TFLOP/s)   bwd {tb:7.1f} us ({3 * fl / 2 / tb / 1e6:6.0f} TFLOP/s incl. recompute)")
if os.environ.get("MLP_PROF"):
    lib = _lib.load()
    lib.hma_mlp_debug_prof.argtypes = [C.c_void_p]
    buf = (C.c_ulonglong * 128)()
    _lib.call("hma_mlp_fwd", torch.cuda.current_stream().cuda_stream, C.byref(f))
    torch.cuda.synchronize()
    lib.hma_mlp_debug_prof(buf)
    names = ["loop", "sync", "dma-issue", "main(mfma)", "gelu/xch | -", "tail", "-", "-"]
    print("fwd phases (cycles of s_memtime = 100 MHz ticks? see sum), block 0:", names)
    for w in range(8):
        v = [buf[w * 8 + i] for i in range(8)]
        print(f"  wave {w} ({'producer' if w < 4 else 'consumer'}): " + " ".join(f"{x:9d}" for x in v[:6]) + f"   sum {sum(v)}")
    _lib.call("hma_mlp_bwd", torch.cuda.current_stream().cuda_stream, C.byref(b))
    torch.cuda.synchronize()
    lib.hma_mlp_debug_prof(buf)
    print("bwd phases, block 0.  producer: [loop, barrier, store-issue, tile-load, mfma, gelu/xch];  "
          "consumer: [loop, vmcnt-wait, barrier, first bundle part, MFMAs + bundle parts + row DMA, dx-add, epilogue]")
    for w in range(8):
        v = [buf[64 + w * 8 + i] for i in range(8)]
        print(f"  wave {w} ({'producer' if w < 4 else 'consumer'}): " + " ".join(f"{x:9d}" for x in v[:7]) + f"   sum {sum(v)}")
