#!/usr/bin/env python3
"""Diffusion-head timing at the C4 shape (rows = 16 x 16 frames x 256 patches x diffusion_batch_mul, width 1024, depth 4)
and its parity numbers on the golden fixture."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safetensors.torch import load_file
from hma_amd.model.diffloss import DiffLoss

dev = "cuda"
G = load_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g9_diffloss.safetensors"))
m = DiffLoss(16, 256, 2, 256, "10")
m.load_state_dict({k[3:]: v for k, v in G.items() if k.startswith("sd.")})
m = m.to(dev)
z = G["z"].to(dev).requires_grad_(True)
loss = m(G["target"].to(dev), z, G["mask"].to(dev), t=G["t"].to(dev), noise=G["noise"].to(dev))
loss.backward()
rel = lambda a, b: ((a.float().cpu() - b).norm() / (b.norm() + 1e-12)).item()
print("parity: loss", loss.item(), "ref", G["loss"].item(), "net_out rel", rel(m.last_net_out, G["net_out"]), "dz rel", rel(z.grad, G["dz"]),
      "worst param-grad rel", max(rel(p.grad, G["grad." + n]) for n, p in m.named_parameters()))
x = m.sample(G["s.z"].to(dev), temperature=0.9, noise0=G["s.noise0"].to(dev), step_noises=G["s.draws"].to(dev))
print("sample rel", rel(x, G["s.sample"]))

N = int(os.environ.get("ROWS", 16 * 16 * 256 * 4))
W, D, C = 1024, 4, 16
big = DiffLoss(C, 256, D, W, "100").to(dev)
with torch.no_grad():
    for p in big.parameters():
        if p.dim() == 2: p.normal_(0, 0.02)
tgt, zz = torch.randn(N, C, device=dev), torch.randn(N, 256, device=dev, requires_grad=True)
mask = (torch.rand(N, device=dev) < 0.7).float()
def step():
    big.zero_grad(set_to_none=True); zz.grad = None
    big(tgt, zz, mask).backward()
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.time()
reps = 3
for _ in range(reps): step()
torch.cuda.synchronize(); dt = (time.time() - t0) / reps
macs = 256 * W + W * W + 256 * W + C * W + D * (3 * W * W + 2 * W * W) + 2 * W * W + W * 2 * C
print(f"C4 head fwd+bwd: rows {N}, {dt * 1e3:.1f} ms, {3 * 2 * macs * N / dt / 1e12:.1f} TFLOP/s algorithmic (3 x fwd), peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
with torch.no_grad():
    zs = torch.randn(16384, 256, device=dev)
    big.sample(zs); torch.cuda.synchronize(); t0 = time.time(); big.sample(zs); torch.cuda.synchronize()
    print(f"sampling 100 steps x 16384 rows: {(time.time() - t0) * 1e3:.1f} ms")
