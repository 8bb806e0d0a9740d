#!/usr/bin/env python3
"""G18: STMaskGIT with qk_norm=True (the GenieConfig dataclass default; hma/model/attention.py:31-35,44-48, st_transformer.py:55,62:
per-head LayerNorm of q and k, norm1 / norm2 become identities) from the REAL reference.  Build container only.

    python tests/golden/make_golden_qknorm.py     # writes tests/golden/g18_qknorm.safetensors"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
import torch  # noqa: E402

from oracle.param_spec import seeded_state_dict, state_dict_spec  # noqa: E402
from oracle.st_maskgit_ref import RefConfig  # noqa: E402
from tests.golden.golden_cfg import TINY, tiny_inputs  # noqa: E402


def main():
    over = dict(TINY["config"], qk_norm=True)
    cfg = MG.GenieConfig(**over)
    model = MG.STMaskGIT(cfg)
    model.init_action_projectors(TINY["domains"], TINY["d_actions"], TINY["action_stats"], cfg.action_network)
    rc = RefConfig(**{k: v for k, v in over.items() if k in RefConfig.__dataclass_fields__})
    spec = state_dict_spec(rc, TINY["domains"], TINY["d_actions"], [len(s[0]) for s in TINY["action_stats"]])
    ref_sd = model.state_dict()
    assert sorted(spec) == sorted(ref_sd), set(spec) ^ set(ref_sd)
    for k in spec:
        assert tuple(ref_sd[k].shape) == tuple(spec[k]), (k, ref_sd[k].shape, spec[k])
    sd = seeded_state_dict(spec, seed=TINY["seed"] + 2, std=0.02, embed_std=0.02)  # init-scale: without norm1 / norm2 the fixture's
    for dom in TINY["domains"]:                                                     # unit-scale embeddings would saturate everything
        sd[f"action_preprocessor.{dom}.mean"] = ref_sd[f"action_preprocessor.{dom}.mean"].clone()
        sd[f"action_preprocessor.{dom}.std"] = ref_sd[f"action_preprocessor.{dom}.std"].clone()
    model.load_state_dict(sd, strict=True)
    model.train()
    inp = tiny_inputs()
    out = {}
    for tag, act in (("domA", inp["actions_domA"]), ("noact", None)):
        model.zero_grad(set_to_none=True)
        kw = dict(input_ids=inp["input_ids"], labels=inp["labels"], h=[16, 16], w=[16, 16])
        kw.update(dict(action_ids=act, domain=[tag] * 2) if act is not None else dict(domain=None))
        o = model(**kw)
        o.loss.backward()
        out[f"{tag}.loss"] = o.loss.detach()
        out[f"{tag}.acc"] = o.acc.detach()
        out[f"{tag}.logits_sub"] = o.logits.detach()[:, :, :, ::4, ::4]
        for k, v in MG.grad_digest(model.named_parameters()).items():
            out[f"{tag}.{k}"] = v
    MG.save("g18_qknorm", out)


if __name__ == "__main__":
    main()
