"""Time hma_adaln_bwd_acc / hma_adaln_fwd at the MAR head's shape (rows x 1024, modulation rows of 6 x 1024 bf16): HMA_LIB selects the library."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hma_amd import _lib

if os.environ.get("HMA_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["HMA_LIB"])
from hma_amd import ops

dev = "cuda"
n, W = int(os.environ.get("ROWS", 76800)), 1024
x = torch.randn(n, W, device=dev)
dout = torch.randn(n, W, device=dev).bfloat16()
mod = torch.randn(n, 6 * W, device=dev).bfloat16()
dmod = torch.empty_like(mod)
dx = torch.empty(n, W, device=dev)
p = lambda t: t.data_ptr()
gam, bet = torch.randn(W, device=dev), torch.randn(W, device=dev)
dgam, dbet = torch.zeros(W, device=dev), torch.zeros(W, device=dev)
for acc, aff in ((0, 0), (1, 0), (0, 1), (1, 1)):
    def run():
        _lib.call("hma_adaln_bwd_acc", ops.stream_ptr(), p(dout), p(x), p(mod), 6 * W, 0, W, p(gam) if aff else None, p(bet) if aff else None,
                  1e-6, p(dx), p(dmod), p(dgam) if aff else None, p(dbet) if aff else None, n, W, acc)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    gb = n * W * (4 + 2 + 2 + 4 + 4 + (4 if acc else 0)) / 1e9
    print(f"adaln_bwd rows {n} accumulate {acc} affine {aff}: {us:.1f} us, {gb / us * 1e3:.2f} TB/s of algorithmic bytes")
