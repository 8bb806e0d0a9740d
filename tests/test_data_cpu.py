"""(f) rows on CPU: the collator restatement, the on-disk token format and the batch sampler against vectors captured
from the real reference (tests/golden/make_golden_data.py -> g10_data.safetensors / g10_data.json)."""
import json
import os
import random

import numpy as np
import pytest
import torch
from safetensors.torch import load_file

from hma_amd import data as hdata
from hma_amd.config import GenieConfig
from hma_amd.data_sampler import MultiTaskBatchSampler
from oracle.collator_ref import collate_with_draws

HERE = os.path.dirname(os.path.abspath(__file__))
G = load_file(os.path.join(HERE, "golden", "g10_data.safetensors"))
J = json.load(open(os.path.join(HERE, "golden", "g10_data.json")))

BASE = dict(num_layers=2, num_heads=8, d_model=256, T=6, S=64, image_vocab_size=262144, num_factored_vocabs=2)
CFGS = {
    "mlm": GenieConfig(**BASE),
    "nonmlm": GenieConfig(**BASE),
    "nocorrupt": GenieConfig(**BASE, dataloader_apply_corruption=False, non_mlm_ratio=0.0),
    "nomask": GenieConfig(**BASE, dataloader_apply_mask=False, non_mlm_ratio=0.0),
    "onefactor": GenieConfig(**{**BASE, "num_factored_vocabs": 1}, non_mlm_ratio=0.0),
}
TAGS = list(CFGS)


def staged_draws(tag):
    """Sort the recorded draws of one reference collate_fn call into the stages of data.py:42-76."""
    cfg, meta = CFGS[tag], J[tag]
    draws = [G[f"{tag}.draw{i}"] for i in range(meta["n_draws"])]
    B, T, h, w = meta["B"], meta["T"], meta["h"], meta["w"]
    st = dict(r_corrupt=None, corrupt_thresh=0.0, random_values=None, r_nonmlm=None, correct_rate=None, first_masked_frame=1,
              mask_prob=None, r_mask=None)
    i = 0
    if cfg.dataloader_apply_corruption:
        st["r_corrupt"], u01, st["random_values"] = draws[0], draws[1], draws[2]
        st["corrupt_thresh"] = float(cfg.max_corrupt_rate * u01.reshape(()))
        i = 3
    random.seed(meta["seed"])
    assert random.random() == meta["first_random"]
    if meta["first_random"] < cfg.non_mlm_ratio:
        fmf = random.randint(cfg.num_prompt_frames, cfg.T - 1)
        rate = random.uniform(cfg.dataloader_mask_ratio_min, 1.0)
        rates = []
        for _ in range(T - fmf):
            rate *= random.uniform(0.9, 1.0)
            rates.append(rate)
        st["r_nonmlm"] = torch.stack(draws[i:i + T - fmf], dim=1)
        st["correct_rate"] = torch.tensor(rates, dtype=torch.float32)
        st["first_masked_frame"] = fmf
        i += T - fmf
    if cfg.dataloader_apply_mask:
        assert (len(draws) - i) % 2 == 0 and len(draws) > i
        st["mask_prob"] = hdata.cosine_schedule(draws[-2]).reshape(B, -1)
        st["r_mask"] = draws[-1]
        assert st["mask_prob"].shape[1] == T - st["first_masked_frame"]
    else:
        assert len(draws) == i
    return st


@pytest.mark.parametrize("tag", TAGS)
def test_collator_restatement_matches_reference(tag):
    cfg, meta = CFGS[tag], J[tag]
    ids = G[f"{tag}.features"].reshape(meta["B"], meta["T"], meta["h"], meta["w"])
    out = collate_with_draws(ids, cfg.factored_vocab_size, cfg.image_vocab_size, num_factored=cfg.num_factored_vocabs,
                             **staged_draws(tag))
    assert torch.equal(out.reshape(meta["B"], -1), G[f"{tag}.input_ids"])
    assert torch.equal(G[f"{tag}.labels"], G[f"{tag}.features"])
    if tag == "nonmlm":
        assert staged_draws(tag)["first_masked_frame"] >= cfg.num_prompt_frames
    if cfg.dataloader_apply_mask:
        assert (G[f"{tag}.input_ids"] == cfg.image_vocab_size).any()


def test_token_dataset_reader_matches_reference(tmp_path):
    d = J["dataset"]
    tokens = G["ds.tokens"].numpy().astype(np.uint32)
    seg, actions = G["ds.segment_ids"].numpy(), G["ds.actions"].numpy()
    hdata.DATA_FREQ_TABLE["dom_fast"] = 6  # as in the generating script's stub of the reference's table
    try:
        for name in ("dom_slow", "dom_fast"):
            hdata.write_token_dataset(tmp_path / name, tokens, seg, actions, name=name)
        meta = json.load(open(tmp_path / "dom_slow" / "metadata.json"))
        assert meta["num_images"] == d["n"] and meta["token_dtype"] == "uint32" and meta["action_dim"] == 3
        assert np.array_equal(np.fromfile(tmp_path / "dom_slow" / "video.bin", dtype=np.uint32).reshape(d["n"], d["h"], d["w"]), tokens)
        for i, (key, want) in enumerate(d["cases"].items()):
            name, kw = key.split("|", 1)
            ds = hdata.RawTokenDataset(tmp_path / name, use_actions=True, **json.loads(kw))
            assert [int(v) for v in ds.valid_start_inds] == want["valid_start_inds"], key
            assert (ds.stride, ds.n_action, ds.num_videos, len(ds)) == (want["stride"], want["n_action"], want["num_videos"], want["len"])
            assert np.allclose(ds.action_stat, want["action_stat"], rtol=1e-6, atol=1e-7)
            np.random.seed(0)
            item = ds[len(ds) // 2]
            assert torch.equal(item["input_ids"], G[f"ds.{i}.input_ids"]) and item["input_ids"].dtype == torch.int64
            assert torch.equal(item["action_ids"], G[f"ds.{i}.action_ids"])
            assert item["domain"] == name and item["h"] == d["h"] and torch.equal(item["labels"], item["input_ids"])
    finally:
        hdata.DATA_FREQ_TABLE.pop("dom_fast", None)


@pytest.mark.parametrize("tag", ["plain", "rank1of2", "groups", "noshuffle"])
def test_batch_sampler_matches_reference(tag):
    want = J["sampler"][tag]
    kw = dict(want["kwargs"])
    if "dataset_groups" in kw:
        kw["dataset_groups"] = [tuple(g) for g in kw["dataset_groups"]]
    s = MultiTaskBatchSampler(**kw)
    assert len(s) == want["len"]
    assert np.allclose(s.generate_tasks_distribution().numpy(), want["weights"], rtol=0, atol=1e-15)
    assert [list(b) for b in s] == want["epoch0"]
    s.set_epoch(3)
    assert [list(b) for b in s] == want["epoch3"]
    # every batch comes from ONE dataset (one action head per micro-batch, train_multi.py / st_mask_git.py:648)
    offs = np.cumsum([0] + kw["dataset_sizes"])
    for b in want["epoch0"]:
        assert len({int(np.searchsorted(offs, i, side="right")) for i in b}) == 1


def test_sampler_rank_validation():
    with pytest.raises(ValueError):
        MultiTaskBatchSampler([10, 10], 2, 1.0, num_replicas=2, rank=2)


# ------------------------------------------------------------------------------------------------ continuous features (MAR)
def test_feature_collator_and_dataset_match_reference(tmp_path):
    """G15 (tests/golden/make_golden_feature.py): `get_maskgit_collator_feature` draws the reference's indicator for the same
    RNG state (hma/data.py:103-157), and `RawFeatureDataset` yields the reference's windows / items from files written by
    `write_feature_dataset` (hma/data.py:298-435)."""
    import json
    import random

    import numpy as np
    from safetensors.torch import load_file

    from hma_amd import data as D
    from hma_amd.config import DiffusionGenieConfig

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = load_file(os.path.join(here, "g15_feature.safetensors"))
    meta = json.load(open(os.path.join(here, "g15_feature.json")))
    base = dict(num_layers=2, num_heads=8, d_model=256, T=6, S=64)
    for tag, extra in (("mlm", {}), ("nonmlm", {}), ("nomask", dict(dataloader_apply_mask=False))):
        m = meta[tag]
        cfg = DiffusionGenieConfig(**base, **extra)
        feats = [{"input_ids": g[f"{tag}.features"][b], "h": m["h"], "w": m["w"], "domain": "d", "action_ids": torch.zeros(cfg.T, 3)}
                 for b in range(m["B"])]
        torch.manual_seed(m["seed"])
        random.seed(m["seed"])
        batch = D.get_maskgit_collator_feature(cfg, device=None)(feats)
        assert torch.equal(batch["masked_tokens_indicator"].to(torch.uint8), g[f"{tag}.indicator"]), tag
        assert torch.equal(batch["input_ids"], g[f"{tag}.features"]) and torch.equal(batch["labels"], batch["input_ids"])
        assert batch["input_ids"].shape == (m["B"], cfg.T * m["h"] * m["w"], 4)
    lat = g["ds.latents"].numpy().astype(np.float16)
    old = dict(D.DATA_FREQ_TABLE)
    D.DATA_FREQ_TABLE["dom_fast"] = 6  # (the fixture's table entry)
    try:
        for name in ("dom_slow", "dom_fast"):
            D.write_feature_dataset(tmp_path / name, lat, g["ds.seg"].numpy(), g["ds.actions"].numpy(), name=name)
            for tag, kw in (("plain", {}), ("overlaps", dict(filter_overlaps=True)), ("cap", dict(max_traj_num=7))):
                ds = D.RawFeatureDataset(tmp_path / name, window_size=3, use_actions=True, **kw)
                rec = meta["datasets"][f"{name}.{tag}"]
                assert ds.stride == rec["stride"] and ds.n_action == rec["n_action"] and ds.valid_start_inds == rec["starts"], (name, tag)
                item = ds[len(ds) // 2]
                assert torch.equal(item["input_ids"], g[f"ds.{name}.{tag}.input_ids"]) and item["domain"] == rec["domain"]
                assert torch.equal(item["action_ids"], g[f"ds.{name}.{tag}.action_ids"]) and item["c"] == rec["c"]
    finally:
        D.DATA_FREQ_TABLE.clear()
        D.DATA_FREQ_TABLE.update(old)


def test_named_openx_domain_uses_the_reference_frequency(tmp_path):
    """the table ships populated -- a real OpenX name gets stride = hz // 2 and n_action = action_dim * stride
    (datasets/encode_openx_dataset.py:51-109, hma/data.py:204-208) without any test-side patching."""
    import numpy as np

    from hma_amd import data as D

    assert D.DATA_FREQ_TABLE["austin_sailor_dataset_converted_externally_to_rlds"] == 20 and len(D.DATA_FREQ_TABLE) == 54
    n = 80
    tokens = np.zeros((n, 4, 4), dtype=np.uint32)
    D.write_token_dataset(tmp_path / "s", tokens, np.zeros(n, dtype=np.int32), np.zeros((n, 7), dtype=np.float32),
                          name="austin_sailor_dataset_converted_externally_to_rlds")
    ds = D.RawTokenDataset(tmp_path / "s", window_size=3, use_actions=True)
    assert ds.stride == 10 and ds.n_action == 70
    assert ds[0]["action_ids"].shape == (3, 70)
