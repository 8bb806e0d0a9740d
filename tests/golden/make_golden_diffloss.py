#!/usr/bin/env python3
"""Golden vectors for the diffusion head from the REAL reference: python tests/golden/make_golden_diffloss.py

Writes tests/golden/g9_diffloss.safetensors: a seeded DiffLoss state dict (width 256, depth 2, 16 channels), inputs
(target, z, mask), the draws (t, noise; every randn_like of the sampling loop), and the reference's loss, raw network
output, gradients (parameters and z) and a 10-step sample.  Data only; runs in the build container only.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: F401,E402  (stubs + /root/reference on sys.path)

from hma.model.diffloss import DiffLoss  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

W, DEPTH, C, ZC, N, STEPS = 256, 2, 16, 256, 300, "10"
g = torch.Generator().manual_seed(0)
dl = DiffLoss(target_channels=C, z_channels=ZC, depth=DEPTH, width=W, num_sampling_steps=STEPS)
with torch.no_grad():  # the reference zero-inits the adaLN / output layers: give every tensor signal
    for n, p in dl.named_parameters():
        if p.dim() == 2:
            p.copy_(torch.randn(p.shape, generator=g) * (1.0 / p.shape[1] ** 0.5))
        elif "in_ln.weight" in n:
            p.copy_(1 + 0.2 * torch.randn(p.shape, generator=g))
        else:
            p.copy_(0.1 * torch.randn(p.shape, generator=g))
out = {f"sd.{k}": v.detach().clone() for k, v in dl.state_dict().items()}
target = torch.randn(N, C, generator=g) * 0.8
target[0, 0], target[1, 1] = -1.2, 1.3          # exercise the |x| > 0.999 branches of the discretised likelihood
z = torch.randn(N, ZC, generator=g).requires_grad_(True)
mask = (torch.rand(N, generator=g) < 0.6).float()
t = torch.randint(0, 1000, (N,), generator=g)
t[:6] = torch.tensor([0, 0, 1, 999, 0, 500])    # t == 0 selects the decoder NLL
noise = torch.randn(N, C, generator=g)
terms = dl.train_diffusion.training_losses(dl.net, target, t, dict(c=z), noise=noise)
loss = (terms["loss"] * mask).sum() / (mask.sum() + 1e-8)
loss.backward()
raw = dl.net(dl.train_diffusion.q_sample(target, t, noise=noise), t, z)
out.update({"target": target, "z": z.detach(), "mask": mask, "t": t, "noise": noise, "loss": loss.detach().reshape(1),
            "loss_rows": terms["loss"].detach(), "mse_rows": terms["mse"].detach(), "vb_rows": terms["vb"].detach(),
            "net_out": raw.detach(), "dz": z.grad.clone()})
for k, p in dl.named_parameters():
    out[f"grad.{k}"] = p.grad.clone()

# sampling: capture every randn_like of p_sample
ns = 64
zs = torch.randn(ns, ZC, generator=g)
noise0 = torch.randn(ns, C, generator=g)
draws = []
orig = torch.randn_like
def rec(x, *a, **k):
    v = torch.randn(x.shape, generator=g)
    draws.append(v)
    return v
torch.randn_like = rec
try:
    with torch.no_grad():
        smp = dl.gen_diffusion.p_sample_loop(dl.net.forward, noise0.shape, noise0, clip_denoised=False, model_kwargs=dict(c=zs),
                                             progress=False, temperature=0.9)
finally:
    torch.randn_like = orig
out.update({"s.z": zs, "s.noise0": noise0, "s.draws": torch.stack(draws), "s.sample": smp})

# classifier-free guidance (diffloss.py:39-43 + forward_with_cfg :235-243): DiffLoss.sample's own cfg != 1 branch, its
# torch.randn(...).cuda() start noise replaced by a seeded CPU draw (no GPU in the build container)
CFG = 2.5
zg = torch.randn(ns, ZC, generator=g)
half0 = torch.randn(ns // 2, C, generator=g)
draws_g = []
def rec_g(x, *a, **k):
    v = torch.randn(x.shape, generator=g)
    draws_g.append(v)
    return v
class _Start:  # what `torch.randn(n, c).cuda()` returns inside DiffLoss.sample
    def __init__(self, v): self.v = v
    def cuda(self): return self.v
orig_randn = torch.randn
torch.randn_like = rec_g
torch.randn = lambda *a, **k: _Start(half0) if "generator" not in k else orig_randn(*a, **k)
try:
    with torch.no_grad():
        smp_g = dl.sample(zg, temperature=0.9, cfg=CFG)
finally:
    torch.randn_like, torch.randn = orig, orig_randn
out.update({"g.z": zg, "g.half0": half0, "g.draws": torch.stack(draws_g), "g.sample": smp_g, "g.cfg": torch.tensor([CFG])})
save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "g9_diffloss.safetensors"))
print("wrote g9_diffloss:", len(out), "tensors,", os.path.getsize(os.path.join(HERE, "g9_diffloss.safetensors")) // 1024, "KB; loss", float(loss))
