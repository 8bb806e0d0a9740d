import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hma_amd import _lib, ops
x = torch.randn(4096, 256, device="cuda")
w = torch.randn(256, 256, device="cuda").bfloat16()
xb = x.bfloat16()
which = sys.argv[1]
torch.cuda.synchronize()
for _ in range(200):
    if which == "ln":
        ops.ln_fwd(x, 1e-5)
    elif which == "gemm":
        ops.linear(xb, w, None, epi=0)
    elif which == "attn":
        ops.attn_temporal_fwd(torch.empty(16 * 64, 768, device="cuda", dtype=torch.bfloat16), 1, 16, 64, 0.25)
torch.cuda.synchronize()
