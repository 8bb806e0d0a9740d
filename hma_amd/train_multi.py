#!/usr/bin/env python3
"""Multi-dataset training driver (SURVEY row a16 + (f) rows): `python -m hma_amd.train_multi --output_dir ...`.

The reference's loop (hma/train_multi.py:556-599, 779-1030) without Accelerate: one `RawTokenDataset` per data
directory (one action domain each), `MultiTaskBatchSampler` (one domain per batch, temperature 3), the device-side
MaskGIT collator, and the fused `Trainer` step (forward + backward + sparse-by-domain all-reduce + clip + AdamW).
Launched under `torch.distributed.run` it is one process per GPU: like the reference every process builds the SAME
sampler (num_replicas=1, rank=0, `train_multi.py:928-932`) and rank r takes batches r, r + world, ... of it
(Accelerate's BatchSamplerShard, `:939,990`).  Flags keep the reference's names; the ones that have no meaning here
(`--no_compile`, `--mu_transfer`, ...) are accepted and ignored so existing launch scripts keep working.
"""
from __future__ import annotations

import argparse
import json
import os
import random
import time
from pathlib import Path
from typing import List

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import ConcatDataset

from .config import DiffusionGenieConfig, GenieConfig
from .data import RawFeatureDataset, RawTokenDataset, get_maskgit_collator, get_maskgit_collator_feature
from .data_sampler import MultiTaskBatchSampler
from .model.st_mask_git import STMaskGIT
from .train import MarTrainer, Trainer


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Train the spatial-temporal MaskGIT on tokenised video + action datasets.")
    p.add_argument("--train_data_dir", type=str, nargs="+", required=True, help="One dataset directory per action domain.")
    p.add_argument("--window_size", type=int, default=12)
    p.add_argument("--stride", type=int, default=None)
    p.add_argument("--filter_overlaps", action="store_true")
    p.add_argument("--genie_config", type=str, required=True, help="GenieConfig JSON (e.g. magvit_n32_h8_d256_action.json).")
    p.add_argument("--resume_from_checkpoint", type=str, default=None)
    p.add_argument("--per_device_train_batch_size", type=int, default=4)
    p.add_argument("--gradient_accumulation_steps", type=int, default=1)
    p.add_argument("--learning_rate", type=float, default=1e-4)
    p.add_argument("--weight_decay", type=float, default=0.05)
    p.add_argument("--num_train_epochs", type=int, default=2)
    p.add_argument("--max_train_steps", type=int, default=None)
    p.add_argument("--num_warmup_steps", type=int, default=500)
    p.add_argument("--max_grad_norm", type=float, default=1.0)
    p.add_argument("--adam_beta_1", type=float, default=0.9)
    p.add_argument("--adam_beta_2", type=float, default=0.95)
    p.add_argument("--adam_eps", type=float, default=1e-8)
    p.add_argument("--output_dir", type=str, required=True)
    p.add_argument("--checkpointing_steps", type=int, default=None)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--sampling_temperature", type=float, default=3.0, help="Dataset-mix temperature (train_multi.py:931).")
    p.add_argument("--log_every", type=int, default=10)
    p.add_argument("--model_type", type=str, default="discrete", choices=["discrete", "continuous"],
                   help="discrete: STMaskGIT on VQ tokens; continuous: STMAR on VAE latents with a diffusion head (train_multi.py:756-775)")
    for ignored in ("--no_compile", "--mu_transfer", "--pin_memory", "--overfit_first_batch"):
        p.add_argument(ignored, action="store_true", help="accepted for script compatibility; no effect")
    for ignored in ("--report_to", "--run_name", "--lr_scheduler_type", "--val_data_dir", "--domain", "--num_workers"):
        p.add_argument(ignored, default=None, help="accepted for script compatibility; no effect")
    return p.parse_args(argv)


def build_datasets(args) -> list:
    kw = dict(window_size=args.window_size, filter_overlaps=args.filter_overlaps, use_actions=True)
    if args.stride is not None:
        kw.update(stride=args.stride, compute_stride_from_freq_table=False)
    cls = RawTokenDataset if args.model_type == "discrete" else RawFeatureDataset
    return [cls(d, **kw) for d in args.train_data_dir]


def main(argv=None):
    args = parse_args(argv)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: RCCL between processes needs it on this driver)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl")
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)

    datasets = build_datasets(args)
    domains = [d.name for d in datasets]
    continuous = args.model_type == "continuous"
    config = (DiffusionGenieConfig if continuous else GenieConfig).from_pretrained(args.genie_config)
    config.T = args.window_size
    config.use_mup = True  # train_multi.py forces the muP attention scale (8 / head_dim)
    if continuous:
        from .model.st_mar import STMAR as model_cls
        config.S = datasets[0].metadata["h"] * datasets[0].metadata["w"]
        config.vae_embed_dim = datasets[0].metadata["latent_channels"]
    else:
        model_cls = STMaskGIT
    if args.resume_from_checkpoint:
        model = model_cls.from_pretrained(args.resume_from_checkpoint)
    else:
        model = model_cls(config)
        model.init_action_projectors(domains, [d.n_action for d in datasets], [d.action_stat for d in datasets], config.action_network)
    model = model.to("cuda")
    accum = args.gradient_accumulation_steps
    # the reference scales the base rate with the effective batch (train_multi.py:902-904)
    lr = args.learning_rate * min(max(1, args.per_device_train_batch_size * accum * world / 64), 8)
    trainer = (MarTrainer if continuous else Trainer)(
        model, lr=lr, betas=(args.adam_beta_1, args.adam_beta_2), eps=args.adam_eps, weight_decay=args.weight_decay,
        max_grad_norm=args.max_grad_norm, warmup_steps=args.num_warmup_steps, grad_accum=accum)
    start_step = 0
    if args.resume_from_checkpoint and (os.path.exists(os.path.join(args.resume_from_checkpoint, "optimizer.bin")) or
                                        os.path.exists(os.path.join(args.resume_from_checkpoint, Trainer.STATE_FILE))):
        trainer.load_state(args.resume_from_checkpoint)  # Adam moments + step counts (train_multi.py:484-533)
        start_step = trainer.completed

    concat = ConcatDataset(datasets)
    sizes = [len(d) for d in datasets]
    sampler = MultiTaskBatchSampler(sizes, args.per_device_train_batch_size, args.sampling_temperature,
                                    seed=args.seed)  # defaults num_replicas=1, rank=0: every process draws the same sequence
    bounds = np.cumsum(sizes)
    domain_of = lambda indices: domains[int(np.searchsorted(bounds, indices[0], side="right"))]  # one domain per batch
    collate = get_maskgit_collator_feature(config) if continuous else get_maskgit_collator(config)
    group = world * accum                                   # batches consumed by one optimizer step, all ranks together
    if len(sampler) < group:
        raise ValueError(f"the dataset mix yields {len(sampler)} batches per epoch, fewer than the {group} one optimizer step consumes "
                         f"(world {world} x gradient accumulation {accum}): lower the batch size / accumulation or add data")
    # (the reference stops an epoch 8 iterations early, train_multi.py:554 `len(train_dataloader) - 8`; here an epoch is every
    # whole group of batches -- a resume position computed from a reference-written step count can differ by that margin)
    steps_per_epoch = len(sampler) // group                 # (every rank gets the same number of batches per epoch)
    max_steps = args.max_train_steps or args.num_train_epochs * steps_per_epoch
    out_dir = Path(args.output_dir)
    save = trainer.save_state
    step, t0, tokens = start_step, time.time(), 0
    # resume: skip what the finished steps consumed (train_multi.py:519-528, 547-549) instead of replaying epoch 0
    first_epoch, skip_steps = divmod(start_step, steps_per_epoch)
    for epoch in range(first_epoch, 10 ** 9):
        if step >= max_steps:
            break
        sampler.set_epoch(epoch)
        batches = list(sampler)
        for k in range(skip_steps if epoch == first_epoch else 0, steps_per_epoch):
            window = batches[k * group:(k + 1) * group]
            if len(window) < group:
                break
            # Every process iterates the SAME sampler, so the domains all ranks will touch in this optimizer step are known
            # here without communication; rank r takes every world-th batch (BatchSamplerShard, train_multi.py:939, 990).
            step_domains = [domain_of(ix) for ix in window]
            for j in range(accum):
                indices = window[j * world + rank]
                batch = collate([concat[i] for i in indices])
                if continuous:
                    ws = trainer.micro_step(step_domains=step_domains, **batch)
                    tokens += batch["input_ids"].shape[0] * batch["input_ids"].shape[1]
                else:
                    ws = trainer.micro_step(batch["input_ids"], batch["labels"], batch.get("action_ids"), batch["domain"],
                                            step_domains=step_domains)
                    tokens += batch["input_ids"].numel()
            trainer.optimizer_step()
            step += 1
            if rank == 0 and (step % args.log_every == 0 or step == max_steps):
                acc = 0.0 if continuous else float(trainer.loss_and_acc(ws)[1])
                dt = time.time() - t0
                print(json.dumps({"step": step, "loss": float(trainer.reduced_loss()), "acc": acc, "domain": batch["domain"][0],
                                  "skipped": bool(trainer.last_loss_info[2].item() > 0), "tokens_per_s_per_gpu": tokens / dt}), flush=True)
            if rank == 0 and args.checkpointing_steps and step % args.checkpointing_steps == 0:
                save(out_dir / f"step_{step}")
            if step >= max_steps:
                break
    if rank == 0:
        save(out_dir / f"step_{step}")
    if world > 1:
        dist.barrier()
    return step


if __name__ == "__main__":
    main()
