import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_fulldepth_gpu as F
F.FULL["num_layers"] = int(os.environ.get("L", 2))
m = F._model(train=False)
for B in (8, 16, 32, 64):
    ids, labels, act = F._inputs(B, 5, 7)
    dev = lambda t: t.to("cuda")
    with torch.no_grad():
        out = m(input_ids=dev(ids), labels=dev(labels), action_ids=dev(act), domain=["domA"] * B)
        lg = out.logits.clone()
        ws = m._engine._ws
        ss_all = ws["ss"].clone() if "ss" in ws else None
        ds = []
        for c in range(0, B, 4):
            sl = slice(c, c + 4)
            oc = m(input_ids=dev(ids[sl]), labels=dev(labels[sl]), action_ids=dev(act[sl]), domain=["domA"] * 4)
            ds.append(float((oc.logits - lg[sl]).abs().max()))
            if c == 0 and ss_all is not None:
                ss4 = m._engine._ws["ss"]
                print("   ss (shift/scale) of layer 0, frames of samples 0-3: max diff", float((ss_all[0, :64] - ss4[0, :64]).abs().max()),
                      " a_emb diff", float((ws["a_emb"][:64] - m._engine._ws["a_emb"][:64]).abs().max()) if False else "")
    print("B", B, "max |logits(B) - logits(chunks of 4)| per chunk:", [round(d, 4) for d in ds], "scale", float(lg.abs().max()))
