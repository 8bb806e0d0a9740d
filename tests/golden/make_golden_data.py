#!/usr/bin/env python3
"""Golden vectors for the (f) rows -- collator, dataset reader, batch sampler -- from the REAL reference.

    python tests/golden/make_golden_data.py     # writes tests/golden/g10_data.safetensors + g10_data.json

Runs only in the build container (imports /root/reference; the GPU box never sees it).  The collator's random
draws are captured by wrapping torch.rand / torch.randint / torch.rand_like while the reference's collate_fn runs,
so the fixture holds (inputs, every draw in call order, outputs): data only.
"""
import json
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402,F401  (installs the mup / xformers stubs and puts /root/reference on sys.path)

ds_stub = types.ModuleType("datasets.encode_openx_dataset")
ds_stub.DATA_FREQ_TABLE = {"dom_fast": 6}
pkg = types.ModuleType("datasets")
pkg.encode_openx_dataset = ds_stub
sys.modules["datasets"] = pkg
sys.modules["datasets.encode_openx_dataset"] = ds_stub

from hma.config import GenieConfig  # noqa: E402
from hma import data as rdata  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

sys.path.insert(0, "/root/reference/external")
spec_path = "/root/reference/external/data_sampler.py"
if not os.path.exists(spec_path):
    import glob
    spec_path = glob.glob("/root/reference/**/data_sampler.py", recursive=True)[0]
import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location("ref_data_sampler", spec_path)
try:
    ref_sampler = importlib.util.module_from_spec(_spec)
    for _m in ("matplotlib", "matplotlib.pyplot", "seaborn"):  # plotting helpers of the same file: never called here
        if _m not in sys.modules:
            try:
                __import__(_m)
            except ImportError:
                sys.modules[_m] = types.ModuleType(_m)
    if not hasattr(sys.modules["matplotlib"], "pyplot"):
        sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    _spec.loader.exec_module(ref_sampler)
except Exception as e:  # noqa: BLE001
    raise SystemExit(f"cannot import the reference sampler: {e}")

sys.path.insert(0, ROOT)
from hma_amd.data import write_token_dataset  # noqa: E402  (the reference must be able to READ what we write)

out_t, out_j = {}, {}

# ------------------------------------------------------------------ collator (G10)
def run_collator(tag, cfg, seed, B, h, w):
    g = torch.Generator().manual_seed(seed)
    feats = [{"input_ids": torch.randint(0, 262144, (cfg.T * h * w,), generator=g), "h": h, "w": w, "domain": "d",
              "action_ids": torch.randn(cfg.T, 3, generator=g)} for _ in range(B)]
    draws = []
    o_rand, o_randint, o_rand_like = torch.rand, torch.randint, torch.rand_like

    def rec(fn):
        def wrapped(*a, **k):
            v = fn(*a, **k)
            draws.append(v.clone())
            return v
        return wrapped

    torch.manual_seed(seed)
    random.seed(seed)
    state = random.getstate()
    torch.rand, torch.randint, torch.rand_like = rec(o_rand), rec(o_randint), rec(o_rand_like)
    try:
        batch = rdata.get_maskgit_collator(cfg)(feats)
    finally:
        torch.rand, torch.randint, torch.rand_like = o_rand, o_randint, o_rand_like
    out_t[f"{tag}.features"] = torch.stack([f["input_ids"] for f in feats])
    out_t[f"{tag}.actions"] = torch.stack([f["action_ids"] for f in feats])
    out_t[f"{tag}.input_ids"] = batch["input_ids"]
    out_t[f"{tag}.labels"] = batch["labels"]
    for i, d in enumerate(draws):
        out_t[f"{tag}.draw{i}"] = d.reshape(-1) if d.dim() == 0 else d
    # replay python's RNG to record the scalar decisions
    random.setstate(state)
    out_j[tag] = {"seed": seed, "B": B, "h": h, "w": w, "n_draws": len(draws), "T": cfg.T,
                  "first_random": random.random()}


base = dict(num_layers=2, num_heads=8, d_model=256, T=6, S=64, image_vocab_size=262144, num_factored_vocabs=2)  # 2 x 512 as shipped
cfg_a = GenieConfig(**base)                                           # defaults: corruption + masking, non-MLM 20 %
run_collator("mlm", cfg_a, 3, 3, 8, 8)
for s in range(4, 60):                                                # find a seed that takes the non-MLM branch
    random.seed(s)
    if random.random() < cfg_a.non_mlm_ratio:
        run_collator("nonmlm", cfg_a, s, 3, 8, 8)
        break
cfg_b = GenieConfig(**base, dataloader_apply_corruption=False, non_mlm_ratio=0.0)
run_collator("nocorrupt", cfg_b, 7, 2, 8, 8)
cfg_c = GenieConfig(**base, dataloader_apply_mask=False, non_mlm_ratio=0.0)
run_collator("nomask", cfg_c, 9, 2, 8, 8)
cfg_d = GenieConfig(**{**base, "num_factored_vocabs": 1}, non_mlm_ratio=0.0)   # GenieConfig default: one factor of 262144
run_collator("onefactor", cfg_d, 11, 2, 8, 8)

# ------------------------------------------------------------------ dataset reader (reads OUR writer's files)
with tempfile.TemporaryDirectory() as td:
    rng = np.random.default_rng(0)
    n, h, w = 60, 4, 4
    tokens = rng.integers(0, 262144, size=(n, h, w), dtype=np.uint32)
    seg = np.repeat(np.arange(4), 15).astype(np.int32)
    actions = rng.standard_normal((n, 3)).astype(np.float32)
    for name in ("dom_slow", "dom_fast"):
        write_token_dataset(os.path.join(td, name), tokens, seg, actions, name=name)
    cases = {}
    for name, kw in (("dom_slow", dict(window_size=5)), ("dom_fast", dict(window_size=4)),
                     ("dom_slow", dict(window_size=5, filter_overlaps=True)),
                     ("dom_slow", dict(window_size=5, filter_interrupts=False, compute_stride_from_freq_table=False, stride=2)),
                     ("dom_fast", dict(window_size=3, max_traj_num=2))):
        ds = rdata.RawTokenDataset(os.path.join(td, name), use_actions=True, **kw)
        key = f"{name}|{json.dumps(kw, sort_keys=True)}"
        np.random.seed(0)
        item = ds[len(ds) // 2]
        cases[key] = {"valid_start_inds": [int(i) for i in ds.valid_start_inds], "stride": int(ds.stride), "n_action": int(ds.n_action),
                      "num_videos": int(ds.num_videos), "len": len(ds), "action_stat": ds.action_stat}
        out_t[f"ds.{len(cases) - 1}.input_ids"] = item["input_ids"]
        out_t[f"ds.{len(cases) - 1}.action_ids"] = item["action_ids"]
    out_j["dataset"] = {"n": n, "h": h, "w": w, "cases": cases}
    out_t["ds.tokens"] = torch.from_numpy(tokens.astype(np.int64))
    out_t["ds.segment_ids"] = torch.from_numpy(seg)
    out_t["ds.actions"] = torch.from_numpy(actions)

# ------------------------------------------------------------------ batch sampler
samp = {}
for tag, kw in (("plain", dict(dataset_sizes=[50, 7, 200], batch_size=4, temperature=3.0, seed=0)),
                ("rank1of2", dict(dataset_sizes=[50, 7, 200], batch_size=4, temperature=3.0, seed=5, num_replicas=2, rank=1)),
                ("groups", dict(dataset_sizes=[30, 10, 90, 12], batch_size=3, temperature=2.0, seed=1, dataset_groups=[(0, 2), (2, 4)])),
                ("noshuffle", dict(dataset_sizes=[20, 40], batch_size=5, temperature=1.0, seed=2, shuffle=False, shuffle_task=False))):
    s = ref_sampler.MultiTaskBatchSampler(**kw)
    e0 = [list(map(int, b)) for b in s]
    s.set_epoch(3)
    e3 = [list(map(int, b)) for b in s]
    samp[tag] = {"kwargs": {k: (list(map(list, v)) if k == "dataset_groups" else v) for k, v in kw.items()}, "epoch0": e0, "epoch3": e3,
                 "weights": s.generate_tasks_distribution().tolist(), "len": len(s)}
out_j["sampler"] = samp

save_file({k: v.contiguous() for k, v in out_t.items()}, os.path.join(HERE, "g10_data.safetensors"))
with open(os.path.join(HERE, "g10_data.json"), "w") as f:
    json.dump(out_j, f)
print("wrote g10_data:", len(out_t), "tensors;", os.path.getsize(os.path.join(HERE, "g10_data.safetensors")) // 1024, "KB")
