"""Time hma_action_stem_fwd / _bwd at the bench shape (rows = B T = 512, d_a = 7..70): HMA_LIB selects the library."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hma_amd import _lib

if os.environ.get("HMA_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["HMA_LIB"])
from hma_amd import ops

dev = "cuda"
rows = int(os.environ.get("ROWS", 512))
for d_a in (7, 70):
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    a, mean, std = r(rows, d_a), r(7), torch.rand(7, generator=g).to(dev) + 0.5
    W1, b1, lw, lb, W2, b2 = r(256, d_a), r(256), r(256), r(256), r(256, 256), r(256)
    an, xh, rs, h, out = r(rows, d_a), r(rows, 256), r(rows), r(rows, 256), r(rows, 256)
    dout, scratch = r(rows, 256), r(rows, 256)
    G = [torch.zeros_like(t) for t in (W1, b1, lw, lb, W2, b2)]
    p = lambda t: t.data_ptr()

    def fwd():
        _lib.call("hma_action_stem_fwd", ops.stream_ptr(), p(a), p(mean), p(std), 7, p(W1), p(b1), p(lw), p(lb), p(W2), p(b2), p(an), p(xh),
                  p(rs), p(h), p(out), rows, d_a, 0)

    def bwd():
        _lib.call("hma_action_stem_bwd", ops.stream_ptr(), p(dout), p(an), p(xh), p(rs), p(h), p(lw), p(W2), p(G[0]), p(G[1]), p(G[2]),
                  p(G[3]), p(G[4]), p(G[5]), p(scratch), rows, d_a)

    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"stem {name} rows {rows} d_a {d_a}: {e0.elapsed_time(e1) * 50:.1f} us per call")
