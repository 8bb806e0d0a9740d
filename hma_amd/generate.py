#!/usr/bin/env python3
"""Autoregressive MaskGIT rollout driver: `python -m hma_amd.generate ...`.

Mirror of hma/generate.py:25-206: same flags, same frame loop (prompt frames kept, later frames masked, one
`maskgit_generate` per frame, optional teacher forcing in time), and the same output files -- `video.bin` holding
[prompt | generated | ground truth] per example in the dataset's token dtype and a `metadata.json` (`generate.py:193-206`)
that the reference's `visualize.py` reads.  `--use_feature` (generate.py:75,108-117,141,155,187-189) runs the continuous
model: `STMAR` on a `RawFeatureDataset` of VAE latents, frames masked with the model's mask latent, output float32 laid out
(b, t, c, h, w).
"""
from __future__ import annotations

import argparse
import json
import os
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import DataLoader

from .data import RawFeatureDataset, RawTokenDataset
from .model.st_mask_git import STMaskGIT


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Generates samples (as tokens) with the MI355X-native HMA engine.")
    p.add_argument("--val_data_dir", type=str, default="data/1x_humanoid_magvit_traj10_val",
                   help="A directory with `metadata.json` and `video.bin`; generation starts from its first frames.")
    p.add_argument("--checkpoint_dir", type=str, help="Path to a HuggingFace-style checkpoint (config.json + model.safetensors).")
    p.add_argument("--output_dir", type=str, default="data/genie_generated", help="Directory to save generated outputs.")
    p.add_argument("--num_prompt_frames", type=int, default=4, help="The number of context frames.")
    p.add_argument("--window_size", type=int, default=12, help="Will generate `window_size - num_prompt_frames` frames.")
    p.add_argument("--example_ind", type=int, default=0, help="The index in the dataset of the example to generate on.")
    p.add_argument("--teacher_force_time", action="store_true", help="Teacher-forces generation in the time dimension.")
    p.add_argument("--maskgit_steps", type=int, default=2, help="Number of MaskGIT sampling steps.")
    p.add_argument("--temperature", type=float, default=0, help="Sampling temperature (only greedy, 0, is built).")
    p.add_argument("--add_action_input", action="store_true", help="Condition on the dataset's actions.")
    p.add_argument("--batch_size", type=int, default=4, help="Batch size (single GPU).")
    p.add_argument("--max_example", type=int, default=16, help="Maximum number of examples.")
    p.add_argument("--use_feature", action="store_true", help="Continuous VAE-latent features (STMAR) rather than tokens.")
    return p.parse_args(argv)


def get_model_step(checkpoint_dir) -> int:
    """Optimizer steps behind a checkpoint: `scheduler.bin`'s `_step_count` when the Accelerate-layout file is there
    (generate.py:80-84), else `.../step_1234` -> 1234, else 0."""
    sch = os.path.join(str(checkpoint_dir), "scheduler.bin")
    if os.path.exists(sch):
        return int(torch.load(sch, weights_only=True)["_step_count"])
    tail = os.path.basename(os.path.normpath(str(checkpoint_dir)))
    digits = "".join(ch for ch in tail.split("_")[-1] if ch.isdigit())
    return int(digits) if digits else 0


@torch.no_grad()
def main(argv=None):
    args = parse_args(argv)
    if args.use_feature:
        from .model.st_mar import STMAR
        model = STMAR.from_pretrained(args.checkpoint_dir).to("cuda").eval()
        ds = RawFeatureDataset(args.val_data_dir, window_size=args.window_size, compute_stride_from_freq_table=False,
                               filter_interrupts=True, filter_overlaps=False, use_actions=args.add_action_input)
        ds.metadata["token_dtype"] = "float32"  # (generate.py:117)
    else:
        ds = RawTokenDataset(args.val_data_dir, window_size=args.window_size, compute_stride_from_freq_table=False,
                             filter_interrupts=True, filter_overlaps=False, use_actions=args.add_action_input)
        model = STMaskGIT.from_pretrained(args.checkpoint_dir).to("cuda").eval()
    side = ds.metadata["h"]
    if args.window_size > model.config.T:
        raise ValueError(f"window_size {args.window_size} exceeds the model's T = {model.config.T}")
    loader = DataLoader(ds, batch_size=args.batch_size, shuffle=False, drop_last=True)
    outputs_all = []
    for batch_idx, batch in enumerate(loader):
        if args.use_feature:
            example = batch["input_ids"].to("cuda").reshape(-1, args.window_size, side, side, batch["input_ids"].shape[-1])
        else:
            example = batch["input_ids"].to("cuda").reshape(-1, args.window_size, side, side)
        mask_value = model.mask_token.detach().reshape(-1) if args.use_feature else model.mask_token_id
        actions = domain = None
        if model.config.use_actions and "action_ids" in batch and args.add_action_input:
            actions = batch["action_ids"].to("cuda")
            domain = [ds.name.replace("_noquant", "")] * example.shape[0]
        prompt = example.clone()
        prompt[:, args.num_prompt_frames:] = mask_value
        samples = []
        for t in range(args.num_prompt_frames, args.window_size):
            if args.teacher_force_time:  # masked prediction of this frame only; ground truth before it
                prompt = example.clone()
                prompt[:, t:] = mask_value
            kw = {} if args.use_feature else dict(maskgit_steps=args.maskgit_steps)  # (the reference's MAR call leaves its default)
            frame, _, _ = model.maskgit_generate(prompt, out_t=t, temperature=args.temperature, action_ids=actions, domain=domain, **kw)
            samples.append(frame)
            if not args.teacher_force_time:
                prompt[:, t] = frame
        out = torch.cat([example[:, : args.num_prompt_frames], torch.stack(samples, dim=1), example[:, args.num_prompt_frames:]], dim=1)
        outputs_all.append(out)  # [prompt | predicted | ground truth]
        if batch_idx >= args.max_example // args.batch_size:
            break
    outputs = torch.cat(outputs_all, dim=0)
    if args.use_feature:
        outputs = outputs.permute(0, 1, 4, 2, 3).contiguous()  # "b t h w c -> b t c h w" (generate.py:187-189)
    out_dir = Path(args.output_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    outputs.cpu().numpy().astype(np.dtype(ds.metadata.get("token_dtype", "uint32"))).tofile(out_dir / "video.bin")
    meta = dict(vars(args))
    meta.update(ds.metadata)
    meta.update({"num_images": outputs.shape[1], "h": side, "w": side, "t": args.window_size,
                 "model_checkpoint": args.checkpoint_dir, "dataset": ds.name, "trained_steps": get_model_step(args.checkpoint_dir)})
    with open(out_dir / "metadata.json", "w") as f:
        json.dump(meta, f)
    print(f"Saved generated video to {out_dir / 'video.bin'} {tuple(outputs.shape)}")
    return outputs


if __name__ == "__main__":
    main()
