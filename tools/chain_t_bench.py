"""Isolated timing of hma_chain_t_bwd at the bench shape (B = 32, SA = 320, T = 16) beside the two launches it replaces.
HMA_LIB=<path> selects a variant build."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hma_amd import _lib, ops  # noqa: E402

if os.environ.get("HMA_LIB"):
    _lib.LIB_PATH = os.environ["HMA_LIB"]
B, SA, T = int(os.environ.get("B", 32)), int(os.environ.get("SA", 320)), 16
M = B * T * SA
dev = "cuda"
torch.manual_seed(0)
bf = torch.bfloat16
qkv = torch.randn(M, 768, device=dev).to(bf)
dy = (torch.randn(M, 256, device=dev) * 0.05).to(bf)
wp = torch.randn(256, 256, device=dev) * 0.06
wt = ops.chain_pack(wp.contiguous(), kind=0, rows=256, cols=256, row_stride=1, col_stride=256)
dq = torch.zeros(M, 768, dtype=bf, device=dev)
st = ops.stream_ptr()


def timeit(fn, n=20):
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


a = ops.make_chain_t_bwd(B=B, SA=SA, segs=[(ops.ptr(wt), 8)], dy_bf16=ops.ptr(dy), qkv=ops.ptr(qkv), dqkv=ops.ptr(dq), attn_scale=0.25)
t_c = timeit(lambda: _lib.call("hma_chain_t_bwd", st, C.byref(a)))
line = f"{os.environ.get('HMA_LIB', 'default'):>28s}  chain T bwd {t_c:7.1f} us ({M * 3584.0 / t_c / 1e6:5.2f} TB/s)"
if not os.environ.get("HMA_LIB"):
    wpt = wp.t().contiguous().to(bf)
    d_o = torch.zeros(M, 256, dtype=bf, device=dev)
    t_g = timeit(lambda: ops.linear(dy, wpt, epi=ops.EPI_BF16, out=d_o))
    o = ops.attn_temporal_fwd(qkv, B, T, SA, 0.25)
    t_a = timeit(lambda: _lib.call("hma_attn_temporal_bwd", st, ops.ptr(qkv), ops.ptr(o), ops.ptr(d_o), ops.ptr(dq), B, T, SA, 0.25))
    line += f"  [2 launches: {t_g:.0f} + {t_a:.0f} = {t_g + t_a:.0f} us]"
print(line)
