#!/usr/bin/env python3
"""Per-shape timing of the layer weight-gradient GEMMs (hma_gemm_tn) at the bench size M = 163840, bf16 operands as the
engine issues them.  Buffers rotate over 3 copies so the 256 MB MALL does not serve the operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hma_amd import _lib
if os.environ.get("HMA_DEBUG_LIB"):  # the -DHMA_PROF build of tools/phase_prof.py --build (ablation switches)
    _lib.LIB_PATH = os.path.abspath(os.environ["HMA_DEBUG_LIB"])
from hma_amd import ops

dev = "cuda"
M = int(os.environ.get("GEMM_M", 163840))
WS = torch.empty(256 * (65536 + 256), device=dev)
NB = 3


def run(name, N, K, bias, affine):
    dys = [torch.randn(M, N, device=dev).bfloat16() for _ in range(NB)]
    xs = [torch.randn(M, K, device=dev).bfloat16() for _ in range(NB)]
    dW = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev) if bias else None
    g = torch.randn(K, device=dev) if affine else None
    b = torch.randn(K, device=dev) if affine else None
    def fn(i):
        ops.linear_wgrad(dys[i % NB], xs[i % NB], dW, db, gamma=g, beta=b, ws=WS)
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    reps = int(os.environ.get("TN_REPS", 18))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if reps > 1000: print('START', flush=True)
    e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    mb = (M * N + M * K) * 2 / 1e6
    print(f"{name:34s} {us:8.1f} us   {mb:6.0f} MB operands   {mb / us:6.2f} TB/s {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)


SHAPES = {
    "proj": ("wgrad proj   N256  K256  bias", 256, 256, True, False),
    "qkv_t": ("wgrad qkv_t  N768  K256", 768, 256, False, False),
    "qkv_s": ("wgrad qkv_s  N768  K256  affine", 768, 256, False, True),
    "fc1": ("wgrad fc1    N1024 K256  affine bias", 1024, 256, True, True),
    "fc2": ("wgrad fc2    N256  K1024 bias", 256, 1024, True, False),
}
for k in (os.environ.get("TN_SHAPES") or ",".join(SHAPES)).split(","):
    run(*SHAPES[k])
