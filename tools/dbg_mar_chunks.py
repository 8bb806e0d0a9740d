import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_fulldepth_stmar_gpu import _model, _inputs, _kw
L = int(os.environ.get("L", 2))
m, _ = _model(train=False, num_layers=L)
B = int(os.environ.get("B", 16))
inp = _inputs(B, seed=6)
n1 = 16 * 256
def chunk(lo, hi):
    c = {k: (v[lo:hi] if k in ("lat", "masked", "act") else v[lo * n1:hi * n1]) for k, v in inp.items()}
    return _kw(c, hi - lo)
with torch.no_grad():
    o16 = m(**_kw(inp, B))
    z16 = o16.logits.clone()
    print("B loss", o16.loss.item())
    for c in range(0, B, 4):
        oc = m(**chunk(c, c + 4))
        print("chunk", c, "loss", oc.loss.item(), "z diff", (oc.logits - z16[c:c+4]).abs().max().item())
    # the head alone on the same z: loss of rows in chunks
    from hma_amd.model.diffloss import DiffLoss
