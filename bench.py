#!/usr/bin/env python3
"""Headline benchmark: video-tokens/s of one HMA-base optimizer step (fwd + bwd + all-reduce + clip + AdamW).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): `magvit_n32_h8_d256_action.json` with T = 16,
use_mup = True, 40 action domains (the "362M" model), per-GPU batch 32 of synthetic VQ tokens
(ids ~ U{0..8191} inside the 2 x 512 factorised vocabulary, frames 1..15 masked at the collator's
cos(u pi/2) rate), fp32 master weights, bf16 MFMA compute.  Inputs are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the NT GEMM): algorithmic
FLOPs / HIP-event time over the timed steps; `cpu_baseline` times the CPU oracle on the host cores.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FREQ = [20, 10, 20, 20, 5, 30, 2, 20, 20, 10, 2, 2, 10, 1, 10, 5, 5, 5, 10, 3, 10, 20, 10, 10, 12, 10, 10, 15, 1, 5, 30,
        3, 3, 15, 20, 10, 30, 5, 10, 5]
FLOP_PER_TOKEN_FWD_BWD = 3.104e8
MFMA_PEAK = 2.5e15                 # dense bf16, MI355X_MICROARCH.md


def build_model(num_domains: int, T: int, layers: int):
    from hma_amd.config import GenieConfig
    from hma_amd.model import STMaskGIT

    cfg = GenieConfig(num_layers=layers, num_heads=8, d_model=256, T=T, S=256, image_vocab_size=262144, use_mup=True,
                      action_network="concat+modulate", num_factored_vocabs=2, qkv_bias=False, proj_bias=True,
                      attn_drop=0.0, qk_norm=False, mlp_ratio=4.0, mlp_drop=0.0, mlp_bias=True, use_actions=True)
    torch.manual_seed(0)
    model = STMaskGIT(cfg)
    domains = [f"dom{i:02d}" for i in range(num_domains)]
    d_actions = [7 * max(1, f // 2) for f in FREQ[:num_domains]]
    stats = [[[0.0] * 7, [1.0] * 7] for _ in domains]
    model.init_action_projectors(domains, d_actions, stats, cfg.action_network)
    # non-degenerate embeddings (the reference leaves pos/mask embeddings at zero at init)
    with torch.no_grad():
        model.pos_embed_TSC.normal_(0, 0.02)
        model.token_embed.mask_token_embed.normal_(0, 0.02)
    return model, domains, d_actions


def synthetic_batch(B, T, seed, d_a, device):
    g = torch.Generator().manual_seed(seed)
    labels = torch.randint(0, 8192, (B, T, 256), generator=g)
    u = torch.rand(B, T - 1, 1, generator=g)
    m = torch.rand(B, T - 1, 256, generator=g) < torch.cos(u * math.pi / 2)
    ids = labels.clone()
    ids[:, 1:][m] = 262144
    act = torch.randn(B, T, d_a, generator=g)
    return ids.reshape(B, -1).to(device), labels.reshape(B, -1).to(device), act.to(device)


def domain_sequence(n_domains, n_draws, seed=0):
    sizes = torch.tensor([1000.0 * (1 + (i * 7) % 13) for i in range(n_domains)], dtype=torch.double)
    w = (sizes / sizes.sum()) ** (1.0 / 3.0)  # temperature-3 multinomial, external/data_sampler.py:244-263
    g = torch.Generator().manual_seed(seed)
    return torch.multinomial(w / w.sum(), n_draws, replacement=True, generator=g).tolist()


def pmc_traffic_per_launch(kernel_substr="gemm_nt"):
    """HBM bytes per launch of the NT GEMM kernels from the committed rocprofv3 PMC passes (profiles/pmc_hbm_r1.json:
    separate FETCH_SIZE / WRITE_SIZE runs of this bench on an 8-layer model; counters are in KB and, on gfx950,
    FETCH_SIZE reports half of a wide coalesced read -- MI355X_MICROARCH.md section HBM -- so it is doubled)."""
    path = os.path.join(ROOT, "profiles", "pmc_hbm_r1.json")
    try:
        d = json.load(open(path))
        tot, launches = 0.0, 0
        for name, v in d["fetch"].items():
            if kernel_substr in name:
                tot += 2.0 * v["counter_sum_kb"] * 1024.0
                launches += v["launches"]
        for name, v in d["write"].items():
            if kernel_substr in name:
                tot += v["counter_sum_kb"] * 1024.0
        return tot / launches if launches else None
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(model, domain, d_a, T, budget_s=25.0):
    """The CPU oracle (a port of the reference path, oracle/st_maskgit_ref.py) on the host cores: B = 1
    fwd + bwd + clip + AdamW, timed for a bounded number of steps."""
    from oracle import st_maskgit_ref as R

    cfg = model.config
    rc = R.RefConfig(num_layers=cfg.num_layers, num_heads=8, d_model=256, T=T, use_mup=True)
    keep = lambda k: (".action_projectors." not in k and not k.startswith("action_")) or f".{domain}." in k
    sd = {k: v.detach().to("cpu", copy=True) for k, v in model.state_dict().items() if keep(k)}
    names = [k for k in sd if not (k.endswith(".mean") or k.endswith(".std"))]
    params = {k: sd[k] for k in names}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v = {k: torch.zeros_like(v) for k, v in params.items()}
    ids, labels, act = synthetic_batch(1, T, 1234, d_a, "cpu")
    cores = torch.get_num_threads()
    times = []
    t_begin = time.perf_counter()
    step = 0
    while True:
        t0 = time.perf_counter()
        leaf = {k: p.clone().requires_grad_(True) for k, p in params.items()}
        full = dict(sd)
        full.update(leaf)
        loss, _, _ = R.forward(full, rc, ids, labels, act, [domain])
        loss.backward()
        step += 1
        R.clip_and_adamw(params, {k: leaf[k].grad for k in names}, m, v, step, 1e-4)
        times.append(time.perf_counter() - t0)
        if len(times) >= 4 or time.perf_counter() - t_begin > budget_s:
            break
    warm = times[1:] if len(times) > 1 else times
    med = sorted(warm)[len(warm) // 2]
    return {"value": T * 256 / med, "unit": "video-tokens/s", "cores": cores, "kind": "port",
            "sample": f"CPU oracle (plain PyTorch fp32), B=1 T={T} L={cfg.num_layers}, fwd+bwd+clip+AdamW, "
                      f"median of {len(warm)} warm steps ({med:.2f} s/step)"}


def mar_bench(args, dev):
    """BASELINE.json configs[3] on ONE GPU (its 8-GPU form shards samples exactly like the headline config): STMAR (continuous
    VAE latents 32x32x4 -> 256 patch tokens + 64 action tokens per frame, T = 16, diffusion head width 1024 / depth 4), batch 16,
    forward + backward + clip + AdamW.  Not the headline metric: a measured line for the C4 row."""
    import time
    from hma_amd.config import DiffusionGenieConfig
    from hma_amd.model.st_mar import STMAR

    B, T = (args.batch if args.batch != 32 else 16), args.frames
    cfg = DiffusionGenieConfig(num_layers=args.layers, num_heads=8, d_model=256, T=T, S=1024, use_mup=True, action_network="concat+modulate",
                               num_factored_vocabs=2, qkv_bias=True, proj_bias=True, qk_norm=False, mlp_drop=0.0, mlp_bias=False,
                               patch_size=2, vae_embed_dim=4, diffloss_w=1024, diffloss_d=4, num_sampling_steps="100", attn_drop=0.0)
    m = STMAR(cfg)
    m.init_action_projectors(["dom0", "dom1"], [14, 7], [[[0.0] * 7, [1.0] * 7]] * 2, cfg.action_network)
    with torch.no_grad():
        for p_ in m.parameters():
            if p_.dim() >= 2 and float(p_.abs().max()) == 0.0:
                p_.normal_(0, 0.02)
    m = m.to(dev).train()
    g = torch.Generator(device=dev).manual_seed(0)
    lat = torch.randn(B, T * 1024, 4, device=dev, generator=g) * 0.7
    masked = torch.rand(B, T, 32, 32, device=dev, generator=g) < 0.6
    act = torch.randn(B, T, 14, device=dev, generator=g)

    def step():
        m.zero_grad()
        out = m(input_ids=lat, labels=lat, action_ids=act, domain=["dom0"] * B, masked_tokens_indicator=masked, h=[32] * B, w=[32] * B)
        out.loss.backward()
        m.optimizer_step(1e-4, "dom0")
        return out.loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({
        "metric": "patch-tokens/sec (STMAR train step: fwd+bwd+clip+AdamW) HMA-MAR T=16 32x32x4 latents", "value": B * T * 256 / dt,
        "unit": "patch-tokens/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"HMA-MAR d256/h8/L{args.layers}, diffusion head 1024x4, synthetic latents T={T} 32x32x4 (+64 action "
                               f"tokens/frame), batch {B}/GPU, eager launches (no hipGraph)", "global_batch": B, "parallelism": "dp1"},
        "final_loss": float(loss)}), flush=True)


def decode_bench(args, dev):
    """BASELINE.json configs[4]: MaskGIT iterative decode, T = 16, 4 prompt + 12 generated frames, 8 iterations,
    batch 64 -> generated frames/s (replicas only: no exchange step).  `--steps` rollouts are timed."""
    from hma_amd.model import STMaskGIT  # noqa: F401

    B, T, P, iters = args.batch, args.frames, 4, 8
    model, domains, d_actions = build_model(args.domains, T, args.layers)
    model = model.to(dev).eval()
    g = torch.Generator().manual_seed(5)
    prompt = torch.randint(0, 8192, (B, P * 256), generator=g).to(dev)
    acts = torch.randn(B, T, d_actions[0], generator=g).to(dev)
    kw = dict(max_new_tokens=(T - P) * 256, maskgit_steps=iters, temperature=0.0, action_ids=acts, domain=[domains[0]] * B,
              unmask_mode="random")
    for _ in range(max(1, args.warmup)):
        out = model.generate(prompt, None, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = model.generate(prompt, None, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    assert int((out == 262144).sum()) == 0
    frames = B * (T - P)
    flops_min = 2.97e12 * B * (args.layers / 32.0)  # minimal algorithmic count with frame-causal reuse, SURVEY.md 8d
    res = {
        "metric": "generated frames/sec (MaskGIT iterative decode, autoregressive rollout) HMA-base T=16 16x16",
        "value": frames / dt, "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic",
        "config": {"workload": f"HMA-base-disc L{args.layers}, prompt {P} + {T - P} generated frames, {iters} MaskGIT iterations, "
                               f"batch {B}, per-layer temporal K/V cache (one 320-row frame per pass)", "global_batch": B,
                   "parallelism": "dp1"},
        "roofline": {"bound": "mfma", "kernel": "whole rollout (minimal algorithmic FLOPs)", "achieved": flops_min / dt / 1e12,
                     "peak": 2500.0, "unit": "TFLOP/s", "frac": flops_min / dt / 2.5e15, "traffic": None},
    }
    if not args.no_cpu_baseline:
        from oracle import st_maskgit_ref as R
        rc = R.RefConfig(num_layers=args.layers, num_heads=8, d_model=256, T=T, use_mup=True)
        dom = domains[0]
        keep = lambda k: (".action_projectors." not in k and not k.startswith("action_")) or f".{dom}." in k
        sd = {k: v.detach().to("cpu", copy=True) for k, v in model.state_dict().items() if keep(k)}
        p1 = torch.full((1, T, 16, 16), 262144, dtype=torch.long)
        p1[:, :P] = prompt[:1].cpu().reshape(1, P, 16, 16)
        t1 = time.perf_counter()
        R.maskgit_generate(sd, rc, p1, P, 2, 0.0, "greedy", acts[:1].cpu(), [dom])
        el = time.perf_counter() - t1
        res["cpu_baseline"] = {"value": 1.0 / (el * iters / 2), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"CPU oracle, B=1, one frame, 2 of {iters} MaskGIT iterations timed ({el:.1f} s) and scaled"}
    print(json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--domains", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--mode", choices=["train", "decode", "mar"], default="train")
    ap.add_argument("--unfused-mlp", action="store_true",
                    help="measurement only: train with the unfused MLP GEMMs (fc1 / fc2 / dfc2 / dfc1 + LayerNorm kernels) instead of "
                         "the fused block (hma_mlp_fwd / hma_mlp_bwd)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # Debug only: HMA_BENCH_ONE_DEVICE=1 puts every rank on GPU 0 with the gloo backend, so the N > 1 trainer path
    # (domain all-gather, bucketed all-reduce on the side stream between per-bucket hipGraphs) can be exercised on a
    # one-GPU box.  The numbers of such a run mean nothing.
    one_device = os.environ.get("HMA_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if args.mode == "decode":
        if rank == 0:
            decode_bench(args, dev)
        return
    if args.mode == "mar":
        if rank == 0:
            mar_bench(args, dev)
        return

    from hma_amd.engine import LaunchTimer
    from hma_amd.train import Trainer

    B, T = args.batch, args.frames
    model, domains, d_actions = build_model(args.domains, T, args.layers)
    model = model.to(dev).train()
    trainer = Trainer(model, lr=1e-4 * min(max(1, B * world / 64), 8), warmup_steps=500, device=dev)
    total = args.warmup + args.steps
    seq = domain_sequence(len(domains), total * world)
    mine = [seq[k * world + rank] for k in range(total)]  # rank r takes every world-th draw
    batches = {}
    for di in sorted(set(mine)):
        batches[di] = synthetic_batch(B, T, 100 + di, d_actions[di], dev)
    eng = trainer.engine
    if args.unfused_mlp:
        eng.fused_mlp_train = False

    def one(k):
        di = mine[k]
        ids, labels, act = batches[di]
        return trainer.step(ids, labels, act, [domains[di]] * B)

    # untimed preparation: every (shape, domain) pair of the schedule is run until its launch plan is captured
    # as a hipGraph (N = 1 path), so the timed steps replay graphs only -- the analogue of a compiler warm-up
    # (every rank runs the same NUMBER of steps per slot so the collectives stay matched)
    prepare_steps = 0
    for rnd in range(3):
        for k in range(total):
            first = mine.index(mine[k]) == k
            ids, labels, act = batches[mine[k]]
            if world == 1 and not first:
                continue
            trainer.step(ids, labels, act, [domains[mine[k]]] * B)
            prepare_steps += 1
    for k in range(args.warmup):
        ws = one(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.warmup, total):
        ws = one(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer = None
    if not args.no_kernel_timing:
        # per-launch HIP events need eager launches: an instrumented pass of the same steps right after the timed
        # region (the timed region itself replays hipGraphs on the N = 1 path)
        eng.timer = LaunchTimer(["hma_gemm_nt"])
        for k in range(args.warmup, min(total, args.warmup + 3)):
            one(k)
        torch.cuda.synchronize()
        timer, eng.timer = eng.timer, None
        timed_ms = sum(e0.elapsed_time(e1) for _, _, e0, e1 in timer.pairs)
        inst_steps = min(total, args.warmup + 3) - args.warmup
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loss, acc = trainer.loss_and_acc(ws)
    loss = float(loss.item())

    if rank == 0:
        tokens = world * B * T * 256 * args.steps
        value = tokens / dt
        out = {
            "metric": "video-tokens/sec (train step: fwd+bwd+all-reduce+clip+AdamW) HMA-base T=16 16x16",
            "value": value, "unit": "video-tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"HMA-base-disc d256/h8/L{args.layers}, {args.domains} action domains "
                                   f"({sum(p.numel() for p in model.parameters()) / 1e6:.1f}M params), synthetic VQ tokens "
                                   f"T={T} H=W=16 ids<8192 (+64 action tokens/frame), batch {B}/GPU",
                       "global_batch": B * world, "seq_len": T * 256, "parallelism": f"dp{world}"},
            "per_gpu": value / world,
            "mfma_roofline_frac_step": value / world * FLOP_PER_TOKEN_FWD_BWD / MFMA_PEAK,
            "final_loss": loss,
        }
        if timer is not None:
            s = timer.summary().get("hma_gemm_nt")
            if s:
                ach = s["flops"] / (s["ms"] * 1e-3) / 1e12
                traffic = pmc_traffic_per_launch()
                avg_s = 1e-3 * s["ms"] / s["launches"]
                out["roofline"] = {"bound": "mfma", "kernel": "hma_gemm_nt (gemm_nt_sw_kernel at K = 256, gemm_nt_ring_kernel at K = 768 / 1024)",
                                   "achieved": ach, "peak": 2500.0,
                                   "unit": "TFLOP/s", "frac": ach / 2500.0, "traffic": traffic,
                                   # the same launches against the HBM roofline (they are output-dominated streams at K = 256)
                                   "hbm": None if not traffic else {"achieved": traffic / avg_s / 1e9, "peak": 8000.0, "unit": "GB/s",
                                                                    "frac": traffic / avg_s / 8e12},
                                   "traffic_note": "HBM bytes per launch, FETCH_SIZE x2 + WRITE_SIZE from profiles/pmc_hbm_r1.json",
                                   "launches": s["launches"], "avg_launch_us": 1e3 * s["ms"] / s["launches"],
                                   "flops_per_launch": s["flops"] / s["launches"],
                                   "share_of_step_time": (s["ms"] / inst_steps) / (1e3 * dt / args.steps),
                                   "measured": f"HIP events around every hma_gemm_nt launch of {inst_steps} eager steps run "
                                               "right after the timed region"}
            out["config"]["prepare_steps"] = prepare_steps
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, domains[mine[0]], d_actions[mine[0]], T)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
