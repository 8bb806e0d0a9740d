#!/usr/bin/env python3
"""G16: STMaskGIT with jointly_predict_actions=True, from the REAL reference (build container only; same stubs as make_golden.py).

    python tests/golden/make_golden_jpa.py     # writes tests/golden/g16_jpa.safetensors

The reference draws the per-frame action mask inside forward (st_mask_git.py:704-710); the draw is recorded (the model keeps it as
`relevant_action_mask`) and handed to the oracle / the HIP path as an input.  Total loss = loss + 0.5 * action_loss
(train_multi.py:576 with config.action_loss_weight)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (installs the stubs, imports the reference)
import torch  # noqa: E402

from oracle.param_spec import seeded_state_dict, state_dict_spec  # noqa: E402
from oracle.st_maskgit_ref import RefConfig  # noqa: E402
from tests.golden.golden_cfg import TINY, tiny_inputs  # noqa: E402


def main():
    cfg = MG.GenieConfig(**dict(TINY["config"], jointly_predict_actions=True))
    model = MG.STMaskGIT(cfg)
    model.init_action_projectors(TINY["domains"], TINY["d_actions"], TINY["action_stats"], cfg.action_network)
    rc = RefConfig(**{k: v for k, v in TINY["config"].items() if k in RefConfig.__dataclass_fields__})
    spec = state_dict_spec(rc, TINY["domains"], TINY["d_actions"], [len(s[0]) for s in TINY["action_stats"]])
    ref_sd = model.state_dict()
    sd = seeded_state_dict(spec, seed=TINY["seed"])
    for dom in TINY["domains"]:
        sd[f"action_preprocessor.{dom}.mean"] = ref_sd[f"action_preprocessor.{dom}.mean"].clone()
        sd[f"action_preprocessor.{dom}.std"] = ref_sd[f"action_preprocessor.{dom}.std"].clone()
    model.load_state_dict(sd, strict=True)
    model.train()
    inp = tiny_inputs()
    out = {}
    for tag, seed in (("domA", 101), ("domB", 202)):
        model.zero_grad(set_to_none=True)
        torch.manual_seed(seed)
        o = model(input_ids=inp["input_ids"], labels=inp["labels"], action_ids=inp[f"actions_{tag}"], domain=[tag] * 2, h=[16, 16], w=[16, 16])
        mask = model.relevant_action_mask[..., 0, 0].float()  # (B, T)
        assert 0 < mask.sum() < mask.numel(), mask
        (o.loss + 0.5 * o.action_loss).backward()
        out[f"{tag}.action_mask"] = mask
        out[f"{tag}.loss"] = o.loss.detach()
        out[f"{tag}.acc"] = o.acc.detach()
        out[f"{tag}.action_loss"] = o.action_loss.detach()
        out[f"{tag}.actions"] = o.actions.detach()
        out[f"{tag}.logits_sub"] = o.logits.detach()[:, :, :, ::4, ::4]
        for k, v in MG.grad_digest(model.named_parameters()).items():
            out[f"{tag}.{k}"] = v
    # policy mode: no action ids, every action token is a mask token, the decoder runs unconditioned (st_mask_git.py:663-666)
    model.eval()
    with torch.no_grad():
        # (B = 1: the reference concatenates the un-expanded (1, T, A, D) mask tokens, :665-666, so this branch only runs for one sample)
        x_THW = inp["input_ids"].reshape(2, cfg.T, 16, 16)[:1]
        logits, actions = model.compute_logits(x_THW, action_ids=None, domain=["domB"], h=[16], w=[16])
    out["policy.actions"] = actions.detach()
    out["policy.logits_sub"] = logits.detach()[:, :, :, ::4, ::4]
    MG.save("g16_jpa", out)


if __name__ == "__main__":
    main()
