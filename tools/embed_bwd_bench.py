"""Time hma_embed_bwd at the bench shape (B 32, T 16, S 256, A 64, V 512): HMA_LIB selects the library."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hma_amd import _lib

if os.environ.get("HMA_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["HMA_LIB"])
from hma_amd import ops

B, T, S, A, V = 32, 16, 256, 64, 512
dev = "cuda"
g = torch.Generator().manual_seed(1)
ids = torch.randint(0, V * V, (B, T, S), generator=g)
ids[torch.rand(B, T, S, generator=g) < 0.5] = V * V
ids = ids.to(dev)
dx = torch.randn(B, T, S + A, 256, device=dev)
G = [torch.zeros(V, 256, device=dev), torch.zeros(V, 256, device=dev), torch.zeros(1, 256, device=dev),
     torch.zeros(T + 1, S + A, 256, device=dev), torch.zeros(B, T, 256, device=dev)]


def run():
    _lib.call("hma_embed_bwd", ops.stream_ptr(), ids.data_ptr(), dx.data_ptr(), *[t.data_ptr() for t in G], B, T, S, A, S + A, V, V * V)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
print("hma_embed_bwd (tok + pos + act kernels): %.1f us per call" % (e0.elapsed_time(e1) * 50))
