"""Data side of the hot path (SURVEY (f) rows 1 and 2): the on-disk token format and the MaskGIT collator.

* `RawTokenDataset` reads the reference's dataset directories (hma/data.py:159-294): `metadata.json`,
  `video.bin` (uint32 tokens, (num_images, h, w)), `segment_ids.bin` (int32), `actions/*.bin` (float32), memory-mapped;
  `write_token_dataset` writes that format (datasets/encode_openx_dataset.py:340-387, and what hma/generate.py:193-206
  emits), so either side can consume the other's files.
* `get_maskgit_collator` has the reference's signature and outputs (hma/data.py:28-98) but does the per-token work in
  one HIP kernel (`hma_maskgit_collate`): the draws are made with torch's RNG in the reference's order -- on the
  CPU generator when the features are CPU tensors (then the result equals the reference's for the same RNG state,
  tests/test_model_gpu.py), on the device otherwise -- and the ids never leave the GPU afterwards.  No CPU fallback:
  the kernel library is required.
"""
from __future__ import annotations

import json
import math
import os
import random
from pathlib import Path
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch
from torch.utils.data import Dataset as TorchDataset

from . import _lib
from .config import GenieConfig

# frames per second of the source datasets (datasets/encode_openx_dataset.py DATA_FREQ_TABLE); the stride of a
# dataset is hz // natural_hz.  Unknown names use 1 like the reference's `.get(name, 1)`.
_X = "_converted_externally_to_rlds"
_HZ_TO_NAMES = {
    1: ["robo_net", "uiuc_d3field", "ego4d"],
    2: ["ucsd_kitchen_dataset" + _X, "language_table", "kuka"],
    3: ["ucsd_pick_and_place_dataset" + _X, "nyu_door_opening_surprising_effectiveness", "nyu_franka_play_dataset" + _X,
        "fractal20220817_data"],
    5: ["berkeley_mvp" + _X, "stanford_robocook" + _X, "cmu_play_fusion", "bridge", "robo_set",
        "dlr_edan_shared_control" + _X, "berkeley_autolab_ur5"],
    6: ["robomimic", "metaworld", "robomimic_new", "robomimic_multitask_new", "robomimic_new_perturb",
        "robomimic_multitask_new_perturb"],
    10: ["stanford_hydra_dataset" + _X, "imperialcollege_sawyer_wrist_cam", "bc_z", "dlr_sara_pour" + _X,
         "furniture_bench_dataset" + _X, "usc_cloth_sim" + _X, "roboturk", "kaist_nonprehensile" + _X,
         "utokyo_xarm_pick_and_place" + _X, "berkeley_cable_routing", "columbia_cairlab_pusht_real", "berkeley_gnm_sac_son"],
    12: ["asu_table_top" + _X],
    15: ["droid", "mimic_play"],
    20: ["austin_sailor_dataset" + _X, "austin_buds_dataset" + _X, "austin_sirius_dataset" + _X,
         "iamlab_cmu_pickup_insert" + _X, "utaustin_mutex", "stanford_kuka_multimodal_dataset" + _X,
         "maniskill_dataset" + _X],
    30: ["berkeley_rpt" + _X, "toto", "conq_hose_manipulation", "aloha_mobile", "1x_humanoid", "epic_kitchen_originalres",
         "epic_kitchen", "exoego4d", "frodobot"],
}
DATA_FREQ_TABLE: Dict[str, int] = {name: hz for hz, names in _HZ_TO_NAMES.items() for name in names}


def normalize_actions(actions: np.ndarray):
    """Per-dimension mean / std of the action table; the normalisation itself happens in the network (data.py:18-24)."""
    return actions, [np.mean(actions, axis=0).tolist(), np.std(actions, axis=0).tolist()]


def write_token_dataset(data_dir, tokens: np.ndarray, segment_ids: np.ndarray, actions: Optional[np.ndarray] = None,
                        name: str = "synthetic", hz: int = 1, **extra_metadata) -> Path:
    """Writes `video.bin`, `segment_ids.bin`, `actions/actions.bin`, `metadata.json` in the reference layout."""
    data_dir = Path(data_dir)
    data_dir.mkdir(parents=True, exist_ok=True)
    tokens = np.ascontiguousarray(tokens)
    assert tokens.ndim == 3 and len(segment_ids) == len(tokens)
    np.ascontiguousarray(tokens).tofile(data_dir / "video.bin")
    np.asarray(segment_ids, dtype=np.int32).tofile(data_dir / "segment_ids.bin")
    meta = {"token_dtype": str(tokens.dtype), "s": 16, "h": int(tokens.shape[1]), "w": int(tokens.shape[2]),
            "vocab_size": int(2 ** 18), "hz": hz, "num_images": int(len(tokens)), "name": name, "quantized": True}
    if actions is not None:
        assert len(actions) == len(tokens)
        (data_dir / "actions").mkdir(exist_ok=True)
        np.asarray(actions, dtype=np.float32).tofile(data_dir / "actions" / "actions.bin")
        meta["action_dim"] = int(np.asarray(actions).shape[-1])
    meta.update(extra_metadata)
    with open(data_dir / "metadata.json", "w") as f:
        json.dump(meta, f)
    return data_dir


class RawTokenDataset(TorchDataset):
    """Sliding windows of `window_size` frames, `stride` apart, over a memory-mapped token file (data.py:159-294)."""

    def __init__(self, data_dir, window_size, stride=1, filter_interrupts=True, filter_overlaps=False, use_actions=False,
                 name="", max_traj_num=1000000, compute_stride_from_freq_table=True, natural_hz=2, drop_action_ratio=0.0):
        data_dir = Path(data_dir)
        with open(data_dir / "metadata.json") as f:
            self.metadata = json.load(f)
        n, h, w = self.metadata["num_images"], self.metadata["h"], self.metadata["w"]
        self.data = np.memmap(data_dir / "video.bin", dtype=np.dtype(self.metadata.get("token_dtype", "uint32")), mode="r",
                              shape=(n, h, w))
        self.window_size, self.stride = window_size, stride
        self.name = name if name else self.metadata["name"]
        if compute_stride_from_freq_table:
            self.stride = max(DATA_FREQ_TABLE.get(self.name, 1) // natural_hz, 1)
        self.n_action = self.metadata.get("action_dim", 1) * self.stride
        self.drop_action_ratio = drop_action_ratio
        if use_actions:
            parts = [np.memmap(p, dtype=np.float32, mode="r").reshape(n, -1) for p in sorted((data_dir / "actions").iterdir())]
            self.actions, self.action_stat = normalize_actions(np.concatenate(parts, axis=-1))
        seg_path = data_dir / "segment_ids.bin"
        if os.path.isfile(seg_path):
            self.segment_ids = np.memmap(seg_path, dtype=np.int32, mode="r", shape=(n,))
        else:
            self.segment_ids = None
            if filter_interrupts:
                raise NotImplementedError("Cannot filter interrupted sequences without segment ids.")
        # frames between the first and last frame of a window (one endpoint excluded)
        self.video_len = (window_size - 1) * self.stride
        starts = np.arange(max(n - self.video_len - self.stride, 0))
        if self.segment_ids is not None and len(starts):
            seg = np.asarray(self.segment_ids)
            # the reference's scan stops after the first start whose trajectory index reaches max_traj_num
            over = np.nonzero(seg[starts] >= max_traj_num)[0]
            if len(over):
                starts = starts[: over[0] + 1]
            if filter_interrupts:  # a window spanning two trajectories has different ids at its ends
                starts = starts[seg[starts] == seg[starts + self.video_len]]
        self.valid_start_inds = starts.tolist()
        if filter_overlaps:  # each frame at most once
            kept: List[int] = []
            for s in self.valid_start_inds:
                clash = {s - i * self.stride for i in range(1, window_size)}
                if not any(e in clash for e in kept[-window_size * self.stride:]):
                    kept.append(s)
            self.valid_start_inds = kept
        self.num_videos = len(np.unique(self.valid_start_inds))

    def __len__(self):
        return len(self.valid_start_inds)

    def __getitem__(self, idx):
        s = self.valid_start_inds[idx]
        x = torch.from_numpy(self.data[s: s + self.video_len + 1: self.stride].astype(np.int64)).flatten()
        item = {"input_ids": x, "labels": x, "attention_mask": torch.ones_like(x), "h": self.metadata["h"],
                "w": self.metadata["w"]}
        if hasattr(self, "actions") and np.random.uniform() > self.drop_action_ratio:
            # all actions inside a stride, concatenated: (window, d_action * stride)
            a = self.actions[s: s + self.video_len + self.stride].reshape(self.window_size, -1)
            item["action_ids"] = torch.from_numpy(a.astype(np.float32))
        item["domain"] = self.name
        return item


def cosine_schedule(u):
    """u in [0, 1] (st_mask_git.py:116-125)."""
    if isinstance(u, torch.Tensor):
        return torch.cos(u * torch.pi / 2)
    if isinstance(u, float):
        return math.cos(u * math.pi / 2)
    raise NotImplementedError(f"Unexpected {type(u)=} {u=}")


def get_maskgit_collator(config: GenieConfig, device: Optional[str] = "cuda") -> Callable:
    """collate_fn(features) -> {"input_ids", "labels" (B, T*H*W) int64 on `device`, "action_ids", "domain", "h", "w"}.

    Draw order (it defines the result for a given RNG state; hma/data.py:42-76): rand(B,T,H,W,F), rand(()),
    randint(B,T,H,W,F) [corruption; F = num_factored_vocabs]; python random(): non-MLM?  randint(num_prompt_frames, T-1), uniform(min, 1), then
    per later frame uniform(0.9, 1) and rand(B,H,W,F); finally, until something is masked, rand(B,T-fmf,1,1) and
    rand(B,T-fmf,H,W)."""
    mask_token_id, V = config.image_vocab_size, config.factored_vocab_size
    NF = config.num_factored_vocabs
    if NF not in (1, 2):
        raise NotImplementedError("the collate kernel factorises into 1 or 2 sub-vocabularies")

    def collate_fn(features) -> dict:
        h, w = features[0]["h"], features[0]["w"]
        ids = torch.stack([ex["input_ids"] for ex in features])
        B, T, HW = len(features), config.T, h * w
        src = ids.device  # the draws are made where the features live (CPU features: reference-identical stream)
        dev = torch.device(device) if device is not None else src
        if dev.type != "cuda":
            raise RuntimeError("get_maskgit_collator runs its kernel on the GPU: no CPU path is built (device='cuda')")
        to = lambda t: None if t is None else t.to(dev, non_blocking=True).contiguous()
        ids_d = to(ids.reshape(B, T, HW))
        r_corrupt = random_values = None
        thresh = 0.0
        if config.dataloader_apply_corruption:
            r_corrupt = torch.rand((B, T, h, w, NF), device=src)
            thresh = float(config.max_corrupt_rate * torch.rand((), device=src))
            random_values = torch.randint(low=0, high=V, size=(B, T, h, w, NF), dtype=torch.long, device=src)
        r_nonmlm = correct = None
        if random.random() < config.non_mlm_ratio:  # closer to autoregressive inference: leave a prefix unmasked
            if random_values is None:
                raise RuntimeError("non_mlm_ratio > 0 needs dataloader_apply_corruption (the reference reuses its draws)")
            fmf = random.randint(config.num_prompt_frames, config.T - 1)
            rate = random.uniform(config.dataloader_mask_ratio_min, 1.0)
            rates, draws = [], []
            for _ in range(T - fmf):  # later frames are corrupted more
                rate *= random.uniform(0.9, 1.0)
                rates.append(rate)
                draws.append(torch.rand((B, h, w, NF), device=src))
            r_nonmlm, correct = torch.stack(draws, dim=1), torch.tensor(rates, dtype=torch.float32)
        else:
            fmf = 1
        out = torch.empty_like(ids_d)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        rc_d, rv_d, rn_d, cr_d = to(r_corrupt), to(random_values), to(r_nonmlm), to(correct)
        stream = torch.cuda.current_stream(dev).cuda_stream
        if config.dataloader_apply_mask:
            while True:  # "we could get unlucky and mask no tokens" (data.py:72)
                mask_prob = to(cosine_schedule(torch.rand(B, T - fmf, 1, 1, device=src)).reshape(B, T - fmf))
                r_mask = to(torch.rand((B, T - fmf, h, w), device=src))
                _lib.call("hma_maskgit_collate", stream, ids_d.data_ptr(), out.data_ptr(), _p(rc_d), thresh, _p(rv_d), _p(rn_d),
                          _p(cr_d), _p(mask_prob), _p(r_mask), B, T, HW, fmf, V, NF, mask_token_id, flag.data_ptr())
                if int(flag.item()):
                    break
        else:
            _lib.call("hma_maskgit_collate", stream, ids_d.data_ptr(), out.data_ptr(), None, 0.0, None, None, None, None, None,
                      B, T, HW, fmf, V, NF, mask_token_id, None)
        batch = {"input_ids": out.reshape(B, T * HW), "labels": ids_d.reshape(B, T * HW).clone()}
        if "action_ids" in features[0]:
            batch["action_ids"] = torch.stack([ex["action_ids"] for ex in features]).to(dev, non_blocking=True)
        batch["domain"] = [ex["domain"] for ex in features]
        batch["h"] = [ex["h"] for ex in features]
        batch["w"] = [ex["w"] for ex in features]
        return batch

    return collate_fn


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------ continuous features (MAR)
SVD_SCALE = 0.18215  # latents are stored unscaled and multiplied on load (hma/data.py:16, 416)


def write_feature_dataset(data_dir, latents: np.ndarray, segment_ids: np.ndarray, actions: Optional[np.ndarray] = None,
                          name: str = "synthetic", hz: int = 1, **extra_metadata) -> Path:
    """Writes `video.bin` (float16 [n, c, h, w] VAE latents), `segment_ids.bin`, `actions/actions.bin`, `metadata.json` in the
    layout `RawFeatureDataset` (hma/data.py:298-376) reads."""
    data_dir = Path(data_dir)
    data_dir.mkdir(parents=True, exist_ok=True)
    latents = np.ascontiguousarray(latents)
    assert latents.ndim == 4 and len(segment_ids) == len(latents)
    latents.tofile(data_dir / "video.bin")
    np.asarray(segment_ids, dtype=np.int32).tofile(data_dir / "segment_ids.bin")
    meta = {"token_dtype": str(latents.dtype), "h": int(latents.shape[2]), "w": int(latents.shape[3]), "latent_channels": int(latents.shape[1]),
            "hz": hz, "num_images": int(len(latents)), "name": name, "quantized": False}
    if actions is not None:
        assert len(actions) == len(latents)
        (data_dir / "actions").mkdir(exist_ok=True)
        np.asarray(actions, dtype=np.float32).tofile(data_dir / "actions" / "actions.bin")
        meta["action_dim"] = int(np.asarray(actions).shape[-1])
    meta.update(extra_metadata)
    with open(data_dir / "metadata.json", "w") as f:
        json.dump(meta, f)
    return data_dir


class RawFeatureDataset(TorchDataset):
    """Sliding windows over a memory-mapped file of continuous VAE latents [n, c, h, w] (hma/data.py:298-435): an item is the
    window's frames as (T*h*w, c) float32 rows scaled by SVD_SCALE, with the stride's actions concatenated per frame."""

    def __init__(self, data_dir, window_size, stride=1, filter_interrupts=True, filter_overlaps=False, use_actions=False,
                 max_traj_num=1000000, compute_stride_from_freq_table=True, natural_hz=2, datio_noise_ratio=0.0,
                 use_raw_image_as_latent=False, domain=None):
        data_dir = Path(data_dir)
        with open(data_dir / "metadata.json") as f:
            self.metadata = json.load(f)
        n, h, w = self.metadata["num_images"], self.metadata["h"], self.metadata["w"]
        c = self.metadata.get("latent_channels", 4)
        self.data = np.memmap(data_dir / "video.bin", mode="r", shape=(n, c, h, w),
                              dtype=np.dtype(self.metadata.get("token_dtype", "float16")))
        self.window_size, self.datio_noise_ratio = window_size, datio_noise_ratio
        self.name = (domain if domain is not None else self.metadata["name"]).replace("_noquant", "")
        self.stride = max(DATA_FREQ_TABLE.get(self.name, 1) // natural_hz, 1) if compute_stride_from_freq_table else stride
        self.n_action = self.metadata.get("action_dim", 1) * self.stride
        if use_actions:
            parts = [np.memmap(p, dtype=np.float32, mode="r").reshape(n, -1) for p in sorted((data_dir / "actions").iterdir())]
            self.actions, self.action_stat = normalize_actions(np.concatenate(parts, axis=-1))
        seg_path = data_dir / "segment_ids.bin"
        if os.path.isfile(seg_path):
            self.segment_ids = np.memmap(seg_path, dtype=np.int32, mode="r", shape=(n,))
        else:
            self.segment_ids = None
            if filter_interrupts:
                raise NotImplementedError("Cannot filter interrupted sequences without segment ids.")
        self.video_len = (window_size - 1) * self.stride
        starts = np.arange(max(n - self.video_len - self.stride, 0))
        if filter_interrupts and len(starts):
            seg = np.asarray(self.segment_ids)
            starts = starts[seg[starts] == seg[starts + self.video_len]]
        self.valid_start_inds = starts[:max_traj_num].tolist()  # (this class caps the number of WINDOWS, :374-375)
        if filter_overlaps:
            kept: List[int] = []
            for s in self.valid_start_inds:
                clash = {s - i * self.stride for i in range(1, window_size)}
                if not any(e in clash for e in kept[-window_size * self.stride:]):
                    kept.append(s)
            self.valid_start_inds = kept

    def __len__(self):
        return len(self.valid_start_inds)

    def __getitem__(self, idx):
        s = self.valid_start_inds[idx]
        x = torch.from_numpy(np.array(self.data[s: s + self.video_len + 1: self.stride])).float() * SVD_SCALE   # (T, c, h, w)
        x = x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])                                                          # (t h w) c
        item = {"input_ids": x, "labels": x, "attention_mask": torch.ones_like(x), "h": self.metadata["h"], "w": self.metadata["w"],
                "c": self.metadata["latent_channels"]}
        if hasattr(self, "actions"):
            a = self.actions[s: s + self.video_len + self.stride].reshape(self.window_size, -1)
            item["action_ids"] = torch.from_numpy(a.astype(np.float32))
        item["domain"] = self.name
        return item


def get_maskgit_collator_feature(config, device: Optional[str] = "cuda") -> Callable:
    """collate_fn(features) -> {"input_ids", "labels" (B, T*h*w, c) float32, "masked_tokens_indicator" (B, T, h, w),
    "action_ids", "domain", "h", "w"} on `device` (hma/data.py:103-157).  The latents themselves are not changed here (STMAR
    substitutes its mask latent, st_mar.py:245); only the indicator is drawn.  Draw order -- it defines the result for a given
    RNG state: python random() [non-MLM?], randint(num_prompt_frames, T-1), then until something is masked
    rand(B, T-fmf, 1, 1) and rand(B, T-fmf, h, w)."""

    def collate_fn(features) -> dict:
        h, w = features[0]["h"], features[0]["w"]
        x = torch.stack([ex["input_ids"] for ex in features])
        B, T = len(features), config.T
        ind = torch.zeros((B, T, h, w), dtype=torch.long)
        if config.dataloader_apply_mask:
            fmf = random.randint(config.num_prompt_frames, T - 1) if random.random() < config.non_mlm_ratio else 1
            mask = torch.zeros(1, dtype=torch.bool)
            while not bool(mask.any()):  # "we could get unlucky and mask no tokens"
                rate = torch.rand(B, T - fmf, 1, 1) * (1 - config.dataloader_mask_ratio_min) + config.dataloader_mask_ratio_min
                mask = torch.rand((B, T - fmf, h, w), dtype=torch.float) < cosine_schedule(rate)
            ind = torch.cat([torch.zeros((B, fmf, h, w), dtype=mask.dtype), mask], dim=1)
        to = (lambda t: t.to(device, non_blocking=True)) if device is not None else (lambda t: t)
        batch = {"input_ids": to(x), "labels": to(x.clone()), "masked_tokens_indicator": to(ind)}
        if "action_ids" in features[0]:
            batch["action_ids"] = to(torch.stack([ex["action_ids"] for ex in features]))
        batch["domain"] = [ex["domain"] for ex in features]
        batch["h"] = [ex["h"] for ex in features]
        batch["w"] = [ex["w"] for ex in features]
        return batch

    return collate_fn
