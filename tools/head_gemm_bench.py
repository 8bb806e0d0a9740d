"""The diffusion head's GEMM shapes (M = 65 536 rows at the MAR bench shape, width 1024): hma_gemm_nt against torch.mm (hipBLASLt / rocBLAS)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hma_amd import _lib

if os.environ.get("HMA_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["HMA_LIB"])
from hma_amd import ops
from hma_amd._lib import EPI_BF16, EPI_F32, EPI_RESID, EPI_SILU2, EPI_DSILU

dev = "cuda"
M = int(os.environ.get("ROWS", 65536))


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for N, K in ((1024, 1024), (3072, 1024), (2048, 1024), (1024, 3072), (1024, 256), (1024, 128)):
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    us = timeit(lambda: ops.linear(a, w, None, epi=EPI_BF16, out=out))
    ref = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ut = timeit(lambda: torch.mm(a, w.t(), out=ref))
    err = (out.float() - ref.float()).abs().max().item()
    print(f"M {M} N {N} K {K}: hma_gemm_nt bf16 {us:7.1f} us {fl / us / 1e6:6.0f} TFLOP/s | torch.mm {ut:7.1f} us {fl / ut / 1e6:6.0f} TFLOP/s | max diff {err:.3g}")
    if N == 1024 and K in (1024, 3072):
        c = torch.randn(M, N, device=dev)
        us = timeit(lambda: ops.linear(a, w, None, epi=EPI_RESID, out=c))
        print(f"    residual epilogue (fp32 C += ): {us:7.1f} us {fl / us / 1e6:6.0f} TFLOP/s")
    if N == 1024 and K == 1024:
        o2 = torch.empty_like(out)
        us = timeit(lambda: ops.linear(a, w, None, epi=EPI_SILU2, out=out, out2=o2))
        print(f"    SiLU pair epilogue: {us:7.1f} us {fl / us / 1e6:6.0f} TFLOP/s")
