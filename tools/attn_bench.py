#!/usr/bin/env python3
"""Spatial attention forward / backward at the bench shape (512 frames of 320 tokens, 8 heads x 32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hma_amd import _lib
if os.environ.get("HMA_DEBUG_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["HMA_DEBUG_LIB"])
from hma_amd import ops

frames, n = int(os.environ.get("FRAMES", 512)), int(os.environ.get("NTOK", 320))
dev = "cuda"
NB = 3
qkvs = [(torch.randn(frames * n, 768, device=dev) * 0.5).bfloat16() for _ in range(NB)]
dos = [(torch.randn(frames * n, 256, device=dev) * 0.5).bfloat16() for _ in range(NB)]
scale = 0.25
outs = [ops.attn_spatial_fwd(q, frames, n, scale) for q in qkvs]


def timeit(fn, reps=12):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"fwd {timeit(lambda i: ops.attn_spatial_fwd(qkvs[i % NB], frames, n, scale)):8.1f} us")
print(f"bwd {timeit(lambda i: ops.attn_spatial_bwd(qkvs[i % NB], outs[i % NB][0], dos[i % NB], outs[i % NB][1], frames, n, scale)):8.1f} us")
if hasattr(_lib.load(), "hma_attn_spatial_bwd_blocked"):
    print(f"bwd, head-blocked dqkv {timeit(lambda i: ops.attn_spatial_bwd(qkvs[i % NB], outs[i % NB][0], dos[i % NB], outs[i % NB][1], frames, n, scale, blocked=True)):8.1f} us")
