import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_fulldepth_gpu import _model, FULL, DEV
m = _model(train=False, readout_gain=300.0)
B, T0 = 64, 4
g = torch.Generator().manual_seed(9)
prompt = torch.randint(0, 8192, (B, T0 * 256), generator=g).to(DEV)
act = torch.randn(B, FULL["T"], 7, generator=g).to(DEV)
for Tw in (6, 8, 16):
    win = torch.full((B, Tw, 16, 16), FULL["image_vocab_size"], dtype=torch.long, device=DEV)
    win[:, :T0] = prompt.reshape(B, T0, 16, 16)
    with torch.no_grad():
        eng = m._get_engine(torch.device(DEV, 0))
        eng.fused_mlp_min_rows = 128 * 256
        a, _ = m.compute_logits(win, action_ids=act, domain=["domA"] * B)
        a = a[:, :, T0].permute(0, 2, 3, 1).reshape(B * 256, 1024).float().clone()
        eng.fused_mlp_min_rows = 10 ** 9
        eng._ws_key = None
        b, _ = m.compute_logits(win, action_ids=act, domain=["domA"] * B)
        b = b[:, :, T0].permute(0, 2, 3, 1).reshape(B * 256, 1024).float().clone()
        eng.decode_prefill(prompt.reshape(B, T0, 256).contiguous(), act.float(), "domA", Tw)
        c = eng.decode_frame(win[:, T0].reshape(B, 256).contiguous(), act[:, T0].float(), "domA", T0, Tw).float().clone()
    d_ab = (a - b).abs().amax(1)
    d_bc = (b - c).abs().amax(1)
    print(f"T={Tw}: fused-vs-unfused max {d_ab.max().item():.4f} rows>0.1: {(d_ab > 0.1).sum().item()} | unfused-vs-cached max {d_bc.max().item():.4f} rows>0.1: {(d_bc > 0.1).sum().item()}  scale {b.abs().max().item():.2f}")
    bad = (d_ab > 0.1).nonzero().flatten()
    if len(bad):
        print("   bad rows (b*256+s):", bad[:20].tolist(), "...", bad[-5:].tolist())
