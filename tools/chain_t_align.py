"""hma_chain_t_bwd at the bench shape with the three arrays carved out of one pool at chosen byte offsets: does its time depend on the
arrays' relative placement (HBM channel interleave)?  usage: python3 tools/chain_t_align.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hma_amd import _lib, ops  # noqa: E402

B, SA, T = 32, 320, 16
M = B * T * SA
dev = "cuda"
bf = torch.bfloat16
pool = torch.empty(3 << 30, dtype=torch.uint8, device=dev)
base = (-pool.data_ptr()) % (1 << 21)  # 2 MB aligned origin
wp = torch.randn(256, 256, device=dev) * 0.06
wt = ops.chain_pack(wp.contiguous(), kind=0, rows=256, cols=256, row_stride=1, col_stride=256)
st = ops.stream_ptr()


def carve(off, nbytes):
    return pool[base + off: base + off + nbytes]


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


nq, nd = M * 768 * 2, M * 256 * 2
for skew in [0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096, 3 << 19]:
    o_q = 0
    o_dq = (nq + (1 << 21) - 1) // (1 << 21) * (1 << 21) + skew
    o_dy = o_dq + (nq + (1 << 21) - 1) // (1 << 21) * (1 << 21) + 2 * skew
    qkv = carve(o_q, nq).view(bf).view(M, 768)
    dq = carve(o_dq, nq).view(bf).view(M, 768)
    dy = carve(o_dy, nd).view(bf).view(M, 256)
    qkv.copy_(torch.randn(M, 768, device=dev).to(bf))
    dy.copy_((torch.randn(M, 256, device=dev) * 0.05).to(bf))
    a = ops.make_chain_t_bwd(B=B, SA=SA, segs=[(ops.ptr(wt), 8)], dy_bf16=ops.ptr(dy), qkv=ops.ptr(qkv), dqkv=ops.ptr(dq), attn_scale=0.25)
    ts = [timeit(lambda: _lib.call("hma_chain_t_bwd", st, C.byref(a))) for _ in range(3)]
    print(f"skew {skew:8d}: " + "  ".join(f"{t:6.1f}" for t in ts) + " us")
