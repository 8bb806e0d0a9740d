"""(f) row 1 on the GPU: the device collator (hma_maskgit_collate through the C ABI) against the reference's outputs for
the same RNG state, and against the CPU restatement on device-side draws."""
import json
import os
import random

import pytest
import torch
from safetensors.torch import load_file

from hma_amd import data as hdata
from tests.test_data_cpu import CFGS, G, J, TAGS

pytestmark = pytest.mark.gpu


def _features(tag, device="cpu"):
    meta = J[tag]
    return [{"input_ids": G[f"{tag}.features"][b].to(device), "h": meta["h"], "w": meta["w"], "domain": "d",
             "action_ids": G[f"{tag}.actions"][b].to(device)} for b in range(meta["B"])]


@pytest.mark.parametrize("tag", TAGS)
def test_device_collator_equals_reference_for_the_same_rng_state(tag):
    """CPU features -> draws from the CPU generators in the reference's order -> identical batch, bit for bit."""
    cfg, meta = CFGS[tag], J[tag]
    torch.manual_seed(meta["seed"])
    random.seed(meta["seed"])
    batch = hdata.get_maskgit_collator(cfg)(_features(tag))
    assert batch["input_ids"].is_cuda and batch["input_ids"].dtype == torch.int64
    assert torch.equal(batch["input_ids"].cpu(), G[f"{tag}.input_ids"])
    assert torch.equal(batch["labels"].cpu(), G[f"{tag}.labels"])
    assert torch.equal(batch["action_ids"].cpu(), G[f"{tag}.actions"])
    assert batch["domain"] == ["d"] * meta["B"] and batch["h"] == [meta["h"]] * meta["B"]


def test_device_collator_on_device_features():
    """GPU-resident features: draws on the device; structure of the result (frame 0 never masked, labels untouched,
    masked fraction following the cosine schedule's mean 2/pi within sampling error)."""
    cfg = CFGS["nocorrupt"]
    B, h, w = 64, 8, 8
    g = torch.Generator().manual_seed(0)
    feats = [{"input_ids": torch.randint(0, 262144, (cfg.T * h * w,), generator=g).cuda(), "h": h, "w": w, "domain": "d"}
             for _ in range(B)]
    torch.manual_seed(1)
    batch = hdata.get_maskgit_collator(cfg)(feats)
    ids = batch["input_ids"].reshape(B, cfg.T, h * w)
    lab = batch["labels"].reshape(B, cfg.T, h * w)
    assert torch.equal(lab, torch.stack([f["input_ids"] for f in feats]).reshape(B, cfg.T, h * w))
    masked = ids == cfg.image_vocab_size
    assert not masked[:, 0].any()
    assert torch.equal(ids[~masked], lab[~masked])
    frac = masked[:, 1:].float().mean().item()
    assert abs(frac - 2 / torch.pi) < 0.06, frac


def test_collator_needs_the_gpu():
    with pytest.raises(RuntimeError):
        hdata.get_maskgit_collator(CFGS["mlm"], device="cpu")(_features("mlm"))
