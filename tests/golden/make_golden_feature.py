#!/usr/bin/env python3
"""G15: the continuous-feature (MAR) data path from the REAL reference (build container only).

    python tests/golden/make_golden_feature.py     # writes tests/golden/g15_feature.safetensors + g15_feature.json

(1) `get_maskgit_collator_feature` (hma/data.py:103-157) for fixed python / torch RNG states: the indicator it draws;
(2) `RawFeatureDataset` (hma/data.py:298-435) reading files written by OUR `write_feature_dataset`: windows and items.
"""
import json
import os
import random
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_golden_data as MGD  # noqa: E402,F401  (stubs, reference on sys.path, `datasets` stub with dom_fast = 6 Hz)

from hma.config import DiffusionGenieConfig  # noqa: E402
from hma import data as rdata  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

sys.path.insert(0, ROOT)
from hma_amd.data import write_feature_dataset  # noqa: E402

out_t, out_j = {}, {}
base = dict(num_layers=2, num_heads=8, d_model=256, T=6, S=64)


def run(tag, cfg, seed, B, h, w):
    g = torch.Generator().manual_seed(seed)
    feats = [{"input_ids": torch.randn(cfg.T * h * w, 4, generator=g), "h": h, "w": w, "domain": "d",
              "action_ids": torch.randn(cfg.T, 3, generator=g)} for _ in range(B)]
    torch.manual_seed(seed)
    random.seed(seed)
    batch = rdata.get_maskgit_collator_feature(cfg)(feats)
    out_t[f"{tag}.features"] = torch.stack([f["input_ids"] for f in feats])
    out_t[f"{tag}.indicator"] = batch["masked_tokens_indicator"].to(torch.uint8)
    assert torch.equal(batch["input_ids"], out_t[f"{tag}.features"]) and torch.equal(batch["labels"], batch["input_ids"])
    out_j[tag] = {"seed": seed, "B": B, "h": h, "w": w, "T": cfg.T}


cfg = DiffusionGenieConfig(**base)
run("mlm", cfg, 3, 3, 8, 8)
for s in range(4, 80):
    random.seed(s)
    if random.random() < cfg.non_mlm_ratio:
        run("nonmlm", cfg, s, 2, 8, 8)
        break
run("nomask", DiffusionGenieConfig(**base, dataloader_apply_mask=False), 9, 2, 8, 8)

with tempfile.TemporaryDirectory() as td:
    rng = np.random.default_rng(1)
    n, c, h, w = 50, 4, 4, 4
    lat = rng.standard_normal((n, c, h, w)).astype(np.float16)
    seg = np.repeat(np.arange(5), 10).astype(np.int32)
    actions = rng.standard_normal((n, 3)).astype(np.float32)
    recs = {}
    for name in ("dom_slow", "dom_fast"):
        write_feature_dataset(os.path.join(td, name), lat, seg, actions, name=name)
        for tag, kw in (("plain", {}), ("overlaps", dict(filter_overlaps=True)), ("cap", dict(max_traj_num=7))):
            ds = rdata.RawFeatureDataset(os.path.join(td, name), window_size=3, use_actions=True, **kw)
            item = ds[len(ds) // 2]
            recs[f"{name}.{tag}"] = {"stride": ds.stride, "n_action": ds.n_action, "starts": list(map(int, ds.valid_start_inds)),
                                     "domain": item["domain"], "h": item["h"], "c": item["c"]}
            out_t[f"ds.{name}.{tag}.input_ids"] = item["input_ids"]
            out_t[f"ds.{name}.{tag}.action_ids"] = item["action_ids"]
    out_t["ds.latents"] = torch.from_numpy(lat.astype(np.float32))
    out_t["ds.seg"] = torch.from_numpy(seg)
    out_t["ds.actions"] = torch.from_numpy(actions)
    out_j["datasets"] = recs

save_file({k: v.contiguous() for k, v in out_t.items()}, os.path.join(HERE, "g15_feature.safetensors"))
json.dump(out_j, open(os.path.join(HERE, "g15_feature.json"), "w"), indent=1)
print("wrote g15_feature:", len(out_t), "tensors", os.path.getsize(os.path.join(HERE, "g15_feature.safetensors")) // 1024, "KB")
