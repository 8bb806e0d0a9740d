#!/usr/bin/env python3
"""Headline benchmark: video-tokens/s of one HMA-base optimizer step (fwd + bwd + all-reduce + clip + AdamW).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N ...            (no launcher around it: starts the N ranks itself, `launch_ranks`)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): `magvit_n32_h8_d256_action.json` with T = 16,
use_mup = True, 40 action domains (the "362M" model), per-GPU batch 32 of synthetic VQ tokens
(ids ~ U{0..8191} inside the 2 x 512 factorised vocabulary, frames 1..15 masked at the collator's
cos(u pi/2) rate), fp32 master weights, bf16 MFMA compute.  Inputs are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel family: algorithmic
FLOPs / HIP-event time over the timed steps; `cpu_baseline` times the CPU oracle on the host cores following
BASELINE.md section 3 (fp32, all host cores, >= 2 warm-up + median of >= 5 steps at B = 1; B = 4 beside it).

At --gpus 1 the same line carries two more measured sub-objects (each with value / unit / ms_per_step / roofline /
cpu_baseline), so that the driver's one command sees every BASELINE config that fits one GPU:
  "decode": configs[4]  MaskGIT iterative decode, B = 64, 4 prompt + 12 generated frames, 8 iterations -> frames/s
  "mar":    configs[3]  STMAR (continuous latents + diffusion head) train step at its per-GPU batch 16 -> patch-tokens/s
`--mode decode` / `--mode mar` run one of them alone (and print it as the line).
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
# every MFMA kernel family of the step (C-ABI entry points); the per-launch HIP-event pass times each of them
FAMILIES = ["hma_gemm_nt", "hma_mlp_fwd", "hma_mlp_bwd", "hma_gemm_tn", "hma_gemm_tn_pair", "hma_gemm_tn_multi", "hma_attn_spatial_fwd", "hma_attn_spatial_bwd_blocked",
            "hma_attn_spatial_bwd", "hma_attn_temporal_fwd", "hma_attn_temporal_bwd", "hma_chain_a_fwd", "hma_chain_a_bwd", "hma_chain_b_fwd", "hma_chain_ab_fwd",
            "hma_chain_s_bwd", "hma_chain_t_bwd", "hma_readout_ce"]
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


def pmc_traffic_per_call(kernels, count_kernel):
    """HBM bytes per C-ABI call of a kernel family from the newest committed PMC passes: the FETCH_SIZE (x 2, see above) and WRITE_SIZE
    sums of EVERY kernel the call launches (`kernels`: name substrings, e.g. the weight-gradient ring kernel AND its reduction) over the
    number of calls (= launches of `count_kernel`).  Comparable with `bytes_per_launch` (algorithmic bytes per call).  FETCH_SIZE is
    doubled: on gfx950 it reports half of a wide coalesced read (MI355X_MICROARCH.md, section HBM)."""
    path = next((q for q in (os.path.join(ROOT, "profiles", f"pmc_hbm_r{r}.json") for r in (6, 5, 4, 3, 2, 1)) if os.path.exists(q)), "")
    try:
        d = json.load(open(path))
        tot, calls = 0.0, 0
        for name, v in d["fetch"].items():
            if any(k in name for k in kernels):
                tot += 2.0 * v["counter_sum_kb"] * 1024.0
            if count_kernel in name:
                calls += v["launches"]
        for name, v in d["write"].items():
            if any(k in name for k in kernels):
                tot += v["counter_sum_kb"] * 1024.0
        return (tot / calls if calls else None), os.path.basename(path)
    except (OSError, KeyError, ValueError):
        return None, None


def pmc_traffic_total(mode):
    """Whole-workload HBM bytes of the `decode` / `mar` legs from their committed PMC passes (profiles/pmc_hbm_<mode>_r4.json: FETCH_SIZE
    and WRITE_SIZE summed over every kernel of the measured unit, written by tools/prof_bench.sh MODE=<mode>)."""
    path = next((q for q in (os.path.join(ROOT, "profiles", f"pmc_hbm_{mode}_r{r}.json") for r in (6, 5, 4)) if os.path.exists(q)), "")
    try:
        return json.load(open(path))["summary"]
    except (OSError, KeyError, ValueError):
        return None


def _cpu_train_steps(sd, rc, names, domain, d_a, T, B, warm, timed):
    """`warm` untimed + `timed` timed optimizer steps of the CPU oracle at batch B; returns the step times."""
    from oracle import st_maskgit_ref as R

    params = {k: sd[k].clone() for k in names}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v = {k: torch.zeros_like(v) for k, v in params.items()}
    ids, labels, act = synthetic_batch(B, T, 1234, d_a, "cpu")
    times = []
    for step in range(warm + timed):
        t0 = time.perf_counter()
        leaf = {k: p.clone().requires_grad_(True) for k, p in params.items()}
        full = dict(sd)
        full.update(leaf)
        loss, _, _ = R.forward(full, rc, ids, labels, act, [domain])
        loss.backward()
        R.clip_and_adamw(params, {k: leaf[k].grad for k in names}, m, v, step + 1, 1e-4)
        if step >= warm:
            times.append(time.perf_counter() - t0)
    return times


def cpu_baseline(model, domain, d_a, T, quick=False):
    """The CPU oracle (a port of the reference path, oracle/st_maskgit_ref.py) on the host cores, BASELINE.md section 3:
    fp32, every host core, one optimizer step = fwd + bwd + clip + AdamW; B = 1: 2 warm-up + median of 5.  Beside that protocol
    figure: `best_threads` -- the same step at 8 / 16 / 32 / 64 threads (1 warm-up + median of 3 each; K = 256 matmuls do not
    scale to 128 threads, the protocol figure understates what the host can do) -- and B = 4 at the best thread count
    (1 warm-up + median of 5).  `quick`: one warm-up + one timed step at B = 1 only."""
    from oracle import st_maskgit_ref as R

    cfg = model.config
    rc = R.RefConfig(num_layers=cfg.num_layers, num_heads=8, d_model=256, T=T, use_mup=True)
    keep = lambda k: (".action_projectors." not in k and not k.startswith("action_")) or f".{domain}." in k
    sd = {k: v.detach().to("cpu", copy=True) for k, v in model.state_dict().items() if keep(k)}
    names = [k for k in sd if not (k.endswith(".mean") or k.endswith(".std"))]
    cores = torch.get_num_threads()
    med = lambda ts: sorted(ts)[len(ts) // 2]
    w1, n1 = (1, 1) if quick else (2, 5)
    t1 = _cpu_train_steps(sd, rc, names, domain, d_a, T, 1, w1, n1)
    out = {"value": T * 256 / med(t1), "unit": "video-tokens/s", "cores": cores, "kind": "port",
           "sample": f"CPU oracle (plain PyTorch fp32, {cores} threads), B=1 T={T} L={cfg.num_layers}, fwd+bwd+clip+AdamW, "
                     f"{w1} warm-up + median of {n1} steps ({med(t1):.2f} s/step)"}
    if not quick:
        sweep = {}
        try:
            for nthr in (8, 16, 32, 64):
                if nthr >= cores:
                    break
                torch.set_num_threads(nthr)
                sweep[nthr] = T * 256 / med(_cpu_train_steps(sd, rc, names, domain, d_a, T, 1, 1, 3))
            best = max(sweep, key=sweep.get) if sweep else cores
            if sweep and sweep[best] < out["value"]:
                best = cores
            torch.set_num_threads(best)
            t4 = _cpu_train_steps(sd, rc, names, domain, d_a, T, 4, 1, 5)
        finally:
            torch.set_num_threads(cores)
        out["best_threads"] = {"threads": best, "value": max([out["value"]] + list(sweep.values())), "unit": "video-tokens/s",
                               "sweep": {str(k): v for k, v in sweep.items()},
                               "sample": "same step, B=1, 1 warm-up + median of 3 per thread count"}
        out["b4"] = {"value": 4 * T * 256 / med(t4), "unit": "video-tokens/s", "cores": best,
                     "sample": f"same, B=4 at {best} threads, 1 warm-up + median of 5 steps ({med(t4):.2f} s/step)"}
    return out


MAR_FLOP_PER_TOKEN = 3.0 * (1.04e8 + 4.94e7)  # trunk 104 MFLOP + diffusion head 49.4 MFLOP per patch token fwd, x3


def mar_bench(args, dev, steps=None, warmup=None):
    """BASELINE.json configs[3] on ONE GPU (its 8-GPU form shards samples exactly like the headline config): STMAR (continuous
    VAE latents 32x32x4 -> 256 patch tokens + 64 action tokens per frame, T = 16, diffusion head width 1024 / depth 4), batch 16,
    forward + backward + clip + AdamW through `MarTrainer` (the data-parallel step driver; world 1 here)."""
    from hma_amd.config import DiffusionGenieConfig
    from hma_amd.model.st_mar import STMAR
    from hma_amd.train import MarTrainer

    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    B, T = (args.batch if args.batch != 32 else 16), args.frames
    # hma/configs/mar_n32_h8_d256_action.json as shipped (mlp_drop 0.05: the MLP's two Dropout sites are live in training),
    # use_mup as train_multi.py forces it, 30 action domains (the "1B" model, run_30datasets_mar_waction.sh)
    cfgd = dict(num_layers=args.layers, num_heads=8, d_model=256, T=T, S=1024, use_mup=True, action_network="concat+modulate",
                num_factored_vocabs=2, qkv_bias=True, proj_bias=True, qk_norm=False, mlp_drop=0.05, mlp_bias=False,
                patch_size=2, vae_embed_dim=4, diffloss_w=1024, diffloss_d=4, num_sampling_steps="100", attn_drop=0.0)
    m = STMAR(DiffusionGenieConfig(**cfgd))
    ndom = 30
    doms = [f"dom{i}" for i in range(ndom)]
    # (action strides capped at 9 steps: the per-domain action-diffusion heads the config also builds -- st_mar.py:89-100, never
    # exercised without jointly_predict_actions -- take at most 64 target channels here)
    m.init_action_projectors(doms, [14] + [7 * min(max(1, f // 2), 9) for f in FREQ[1:ndom]], [[[0.0] * 7, [1.0] * 7]] * ndom,
                             cfgd["action_network"])
    with torch.no_grad():
        for p_ in m.parameters():
            if p_.dim() >= 2 and float(p_.abs().max()) == 0.0:
                p_.normal_(0, 0.02)
    m = m.to(dev).train()
    tr = MarTrainer(m, lr=1e-4, warmup_steps=0, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    lat = torch.randn(B, T * 1024, 4, device=dev, generator=g) * 0.7
    masked = torch.rand(B, T, 32, 32, device=dev, generator=g) < 0.6
    act = torch.randn(B, T, 14, device=dev, generator=g)
    kw = dict(input_ids=lat, labels=lat, action_ids=act, domain=["dom0"] * B, masked_tokens_indicator=masked, h=[32] * B, w=[32] * B)

    for _ in range(max(1, warmup)):
        out = tr.step(step_domains=["dom0"], **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = tr.step(step_domains=["dom0"], **kw)
    t_issue = time.perf_counter() - t0  # the host is done enqueueing: below the step time means the GPU is the limit
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tokens = B * T * 256
    res = {
        "metric": "patch-tokens/sec (STMAR train step: fwd+bwd+clip+AdamW) HMA-MAR T=16 32x32x4 latents", "value": tokens / dt,
        "unit": "patch-tokens/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"HMA-MAR d256/h8/L{args.layers} (mar_n32_h8_d256_action.json: mlp_drop 0.05), {ndom} action domains "
                               f"({sum(p_.numel() for p_ in m.parameters()) / 1e6:.0f}M params), diffusion head 1024x4, synthetic latents "
                               f"T={T} 32x32x4 (+64 action tokens/frame), batch {B}/GPU, eager launches (no hipGraph)",
                   "global_batch": B, "parallelism": "dp1"},
        "roofline": {"bound": "mfma", "kernel": "whole step (algorithmic FLOPs: trunk 3 x 1.04e8 + diffusion head 3 x 4.94e7 per patch token)",
                     "achieved": tokens / dt * MAR_FLOP_PER_TOKEN / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                     "frac": tokens / dt * MAR_FLOP_PER_TOKEN / MFMA_PEAK, "traffic": None},
        "final_loss": float(out.loss.detach()), "host_issue_ms_per_step": 1e3 * t_issue / steps}
    traffic = pmc_traffic_total("mar")
    if traffic:
        res["roofline"]["traffic"] = traffic["bytes_per_step"]
        res["roofline"]["traffic_note"] = traffic["note"]
    if not args.no_cpu_baseline:
        from oracle import st_maskgit_ref as R
        from oracle import st_mar_ref as MR
        rc = R.RefConfig(num_layers=args.layers, num_heads=8, d_model=256, T=T, use_mup=True, qkv_bias=True, mlp_bias=False)
        keep = lambda k: (".action_projectors." not in k and not k.startswith("action_")) or ".dom0." in k
        sd = {k: v.detach().to("cpu", copy=True) for k, v in m.state_dict().items() if keep(k)}
        names = [k for k in sd if not (k.endswith(".mean") or k.endswith(".std")) and not k.startswith("action_diff_losses")]
        n = T * 256
        gc = torch.Generator().manual_seed(1)
        lat1, mk1, act1 = lat[:1].cpu(), masked[:1].cpu(), act[:1].cpu()
        tt, nz = torch.randint(0, 1000, (n,), generator=gc), torch.randn(n, 16, generator=gc)
        times = []
        for it in range(3):
            t1 = time.perf_counter()
            leaf = {k: sd[k].clone().requires_grad_(True) for k in names}
            full = dict(sd)
            full.update(leaf)
            loss, _ = MR.forward(full, rc, lat1, lat1, act1, ["dom0"], mk1, tt, nz, 2, 32, 32, 4)
            loss.backward()
            with torch.no_grad():  # clip + a plain AdamW-shaped pass over every tensor (the update itself is < 1 % of the step)
                tot = torch.sqrt(sum((leaf[k].grad.double() ** 2).sum() for k in names if leaf[k].grad is not None))
                for k in names:
                    if leaf[k].grad is not None:
                        sd[k].add_(leaf[k].grad * min(1.0, 1.0 / (float(tot) + 1e-6)), alpha=-1e-4)
            if it >= 1:
                times.append(time.perf_counter() - t1)
        med = sorted(times)[len(times) // 2]
        res["cpu_baseline"] = {"value": n / med, "unit": "patch-tokens/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"CPU oracle (oracle/st_mar_ref.py, fp32), B=1 T={T} L={args.layers}, fwd+bwd+clip+update, "
                                         f"1 warm-up + median of 2 steps ({med:.2f} s/step)"}
    del tr, m
    torch.cuda.empty_cache()
    return res


def decode_bench(args, dev, steps=None, warmup=None, batch=None):
    """BASELINE.json configs[4]: MaskGIT iterative decode, T = 16, 4 prompt + 12 generated frames, 8 iterations,
    batch 64 -> generated frames/s (replicas only: no exchange step).  `steps` rollouts are timed."""
    from hma_amd.model import STMaskGIT  # noqa: F401

    class _A:  # (the rollout count of this leg is its own: a rollout is ~1 s)
        pass
    a2 = _A()
    a2.__dict__.update(vars(args))
    a2.steps = args.steps if steps is None else steps
    a2.warmup = args.warmup if warmup is None else warmup
    a2.batch = args.batch if batch is None else batch
    args = a2
    B, T, P, iters = args.batch, args.frames, 4, 8
    model, domains, d_actions = build_model(args.domains, T, args.layers)
    model = model.to(dev).eval()
    if os.environ.get("HMA_FUSED_MLP_MIN_ROWS"):  # measurement only: policy threshold of the fused MLP block
        model._get_engine(dev).fused_mlp_min_rows = int(os.environ["HMA_FUSED_MLP_MIN_ROWS"])
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
    flops_ref = 4.07e13 * B * (args.layers / 32.0)  # what the reference computes: a full-window forward per iteration
    res = {
        "metric": "generated frames/sec (MaskGIT iterative decode, autoregressive rollout) HMA-base T=16 16x16",
        "value": frames / dt, "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic",
        "config": {"workload": f"HMA-base-disc L{args.layers}, prompt {P} + {T - P} generated frames, {iters} MaskGIT iterations, "
                               f"batch {B}, per-layer temporal K/V cache (one 320-row frame per pass)", "global_batch": B,
                   "parallelism": "dp1"},
        "roofline": {"bound": "mfma", "kernel": "whole rollout (minimal algorithmic FLOPs, frame-causal reuse)",
                     "achieved": flops_min / dt / 1e12, "peak": 2500.0, "unit": "TFLOP/s", "frac": flops_min / dt / 2.5e15,
                     "traffic": None, "reference_equivalent_tflops": flops_ref / dt / 1e12},
    }
    # the interactive caller (sim/simulator.py:286-293: one environment, one frame at a time): B = 1 latency of the same rollout
    if not getattr(args, "no_latency", False):
        p1, a1 = prompt[:1].contiguous(), acts[:1].contiguous()
        kw1 = dict(kw, action_ids=a1, domain=[domains[0]])
        for _ in range(2):
            model.generate(p1, None, **kw1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            model.generate(p1, None, **kw1)
        t_issue1 = (time.perf_counter() - t0) / 3
        torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t0) / 3
        res["latency_b1"] = {"ms_per_frame": 1e3 * dt1 / (T - P), "frames_per_s": (T - P) / dt1,
                             "host_ms_per_frame": 1e3 * t_issue1 / (T - P),
                             "sample": f"batch 1, {T - P} generated frames x {iters} MaskGIT iterations, 2 warm-up + mean of 3 rollouts"}
    traffic = pmc_traffic_total("decode")
    if traffic:
        res["roofline"]["traffic"] = traffic["bytes_per_rollout"]
        res["roofline"]["traffic_note"] = traffic["note"]
    if not args.no_cpu_baseline:
        from oracle import st_maskgit_ref as R
        rc = R.RefConfig(num_layers=args.layers, num_heads=8, d_model=256, T=T, use_mup=True)
        dom = domains[0]
        keep = lambda k: (".action_projectors." not in k and not k.startswith("action_")) or f".{dom}." in k
        sd = {k: v.detach().to("cpu", copy=True) for k, v in model.state_dict().items() if keep(k)}
        p1 = torch.full((1, T, 16, 16), 262144, dtype=torch.long)
        p1[:, :P] = prompt[:1].cpu().reshape(1, P, 16, 16)
        t1 = time.perf_counter()
        R.maskgit_generate(sd, rc, p1, P, iters, 0.0, "greedy", acts[:1].cpu(), [dom])
        el = time.perf_counter() - t1
        res["cpu_baseline"] = {"value": 1.0 / el, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"CPU oracle (full-window recompute like the reference), B=1, one frame, {iters} MaskGIT "
                                         f"iterations ({el:.1f} s)"}
    del model
    torch.cuda.empty_cache()
    return res


def sample_clock_power(run):
    """Shader clock (MHz) and package power (W) of the device WHILE `run()` replays the timed steps once more (behind the timed region:
    nothing of this is inside it): one `rocm-smi --showclocks --showpower` sample taken from a helper thread ~0.4 s into the loop.
    None when rocm-smi is missing or says nothing -- the line then simply has no `power` object."""
    import re
    import subprocess
    import threading
    out = {}

    def probe():
        time.sleep(0.4)
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10)
            m = re.search(r"sclk clock level[^(]*\((\d+)Mhz\)", r.stdout)
            w = re.search(r"Power \(W\):\s*([0-9.]+)", r.stdout)
            if m:
                out["sclk_mhz"] = int(m.group(1))
            if w:
                out["watts"] = float(w.group(1))
        except Exception:
            pass

    th = threading.Thread(target=probe, daemon=True)
    th.start()
    run()
    torch.cuda.synchronize()
    th.join(timeout=12)
    if not out:
        return None
    out["note"] = ("one rocm-smi sample while the timed steps are replayed once more after the timed region; the part's clock maximum is "
                   "2400 MHz and its package power limit 1400 W (DESIGN.md section 6.0 item 5)")
    return out


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run --nproc-per-node N bench.py <same
    arguments>` as a child process, relay rank 0's JSON line, return the child's exit code.  Runs BEFORE anything initialises the
    GPU in this process (counting devices does not), and the ranks are fresh processes, never an exec of this one."""
    import socket
    import subprocess

    one_device = os.environ.get("HMA_BENCH_ONE_DEVICE") == "1"
    have = torch.cuda.device_count()
    if have < (1 if one_device else n):
        print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        if out.startswith("{") and '"metric"' in out:
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc != 0:
        print(f"bench.py: a rank failed (torch.distributed.run exit code {rc})", file=sys.stderr, flush=True)
        return rc
    if line is None or json.loads(line).get("n_gpus") != n:
        print(f"bench.py: the ranks did not produce an n_gpus = {n} line", file=sys.stderr, flush=True)
        return 3
    print(line, flush=True)
    return 0


def child_line(extra_args, env_extra, steps=8, warmup=3):
    """One more train-leg line of this script in a CHILD process (its own model, plans and graphs; the parent's memory is released
    first), reduced to the numbers the parent's line carries."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HMA_FORCE_COLLECTIVES", "HMA_BENCH_STEP_DOMAINS")}
    env.update(env_extra)
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--mode", "train", "--steps", str(steps), "--warmup", str(warmup),
           "--no-cpu-baseline", "--no-kernel-timing"] + list(extra_args)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or len(lines) != 1:
        return {"error": (r.stderr or r.stdout)[-400:]}
    d = json.loads(lines[0])
    keep = {k: d[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "final_loss", "backend", "allreduce_bytes_per_step") if k in d}
    keep["workload"] = d["config"]["workload"]
    return keep


def forced_collectives_check():
    """What ONE GPU can measure of the 8-rank step: the same step with and without the reducer's collectives in a one-rank RCCL group
    (HMA_FORCE_COLLECTIVES=1), both announcing the 8 domains an 8-rank step sees (HMA_BENCH_STEP_DOMAINS=8: 4 x (1 + 8) + 1 + 8 + 1
    all-reduce calls, ~380-414 MB, on the side stream between the per-bucket hipGraphs).  What the delta contains: the step split into
    one hipGraph per gradient bucket, the event waits between the two streams, RCCL's host-side enqueue -- NOT device-side contention:
    an in-place all-reduce over one rank launches no kernel (profiles/prof_forced_r6.txt: no RCCL kernel in the trace), so the CUs
    RCCL's ring kernels take from the persistent 256-workgroup launches at 8 ranks stay unmeasured on one GPU.
    forced_collectives_delta_ms = step with - step without."""
    plain = child_line([], {"HMA_BENCH_STEP_DOMAINS": "8"})
    forced = child_line([], {"HMA_BENCH_STEP_DOMAINS": "8", "HMA_FORCE_COLLECTIVES": "1", "MASTER_ADDR": "127.0.0.1",
                             "MASTER_PORT": str(29600 + os.getpid() % 300)})
    res = {"announced_domains_per_step": 8, "plain": plain, "forced": forced,
           "note": "one-rank RCCL group: the reducer's calls, streams and per-bucket graphs of an 8-rank step without device-side "
                   "all-reduce work (RCCL launches no kernel for one rank)"}
    if "ms_per_step" in plain and "ms_per_step" in forced:
        res["forced_collectives_delta_ms"] = forced["ms_per_step"] - plain["ms_per_step"]
        by = forced.get("allreduce_bytes_per_step") or []
        res["allreduce_mb_per_step"] = (sum(by) / len(by) / 1e6) if by else None
    return res


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
    ap.add_argument("--mode", choices=["all", "train", "decode", "mar"], default="all",
                    help="all (default): the train line, plus the decode and mar sub-objects when --gpus 1")
    ap.add_argument("--no-latency", action="store_true", help="decode: skip the batch-1 latency leg (profiling passes)")
    ap.add_argument("--lib", type=str, default=None, help="measurement only: another build of libhma_hip.so (same-box A / B of a kernel change)")
    ap.add_argument("--quick-cpu", action="store_true", help="one warm-up + one timed CPU-oracle step instead of the BASELINE.md protocol")
    ap.add_argument("--unfused-mlp", action="store_true",
                    help="measurement only: train with the unfused MLP GEMMs (fc1 / fc2 / dfc2 / dfc1 + LayerNorm kernels) instead of "
                         "the fused block (hma_mlp_fwd / hma_mlp_bwd)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Launched bare with --gpus N: this process becomes the launcher -- it never touches a GPU -- and starts N fresh ranks
        # (the reference: `torchrun --nproc_per_node=8`, experiments/scripts/run_30datasets_waction.sh:17-19).
        raise SystemExit(launch_ranks(args.gpus))
    if args.lib:
        from hma_amd import _lib
        _lib.LIB_PATH = os.path.abspath(args.lib)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: RCCL between processes needs it on this driver)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
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
    # HMA_FORCE_COLLECTIVES=1 (debug): a process group -- and with it the reducer's RCCL all-reduces on the side stream between the
    # per-bucket graphs -- also at one rank, so a one-GPU box executes the N > 1 code path with the real backend
    force = os.environ.get("HMA_FORCE_COLLECTIVES") == "1"
    if world > 1 or force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
    backend = (dist.get_backend() + (" (RCCL over xGMI)" if dist.get_backend() == "nccl" else "")) if (world > 1 or force) else None

    if args.mode in ("decode", "mar") and world > 1:
        raise SystemExit(f"--mode {args.mode} is a one-GPU leg (BASELINE configs[3] / configs[4]); run it with --gpus 1")
    if args.mode == "decode":
        if rank == 0:
            print(json.dumps(decode_bench(args, dev, batch=64 if args.batch == 32 else args.batch)), flush=True)
        if force:
            dist.destroy_process_group()
        return
    if args.mode == "mar":
        if rank == 0:
            print(json.dumps(mar_bench(args, dev)), flush=True)
        if force:
            dist.destroy_process_group()
        return

    from hma_amd.engine import LaunchTimer
    from hma_amd.train import Trainer

    B, T = args.batch, args.frames
    model, domains, d_actions = build_model(args.domains, T, args.layers)
    model = model.to(dev).train()
    trainer = Trainer(model, lr=1e-4 * min(max(1, B * world / 64), 8), warmup_steps=500, device=dev)
    if os.environ.get("HMA_BENCH_WGRAD_LAYERS"):  # measurement only (same-box A / B): weight gradients of 1 | 2 blocks per launch
        trainer.engine.wgrad_layers = int(os.environ["HMA_BENCH_WGRAD_LAYERS"])
    total = args.warmup + args.steps
    # HMA_BENCH_STEP_DOMAINS=n (one-rank runs, `forced_collectives_check` below): every step ANNOUNCES the n domains an n-rank job would
    # see in it (this rank trains the first; the others' blocks are zeroed, all-reduced and stepped with zero gradients), so that the
    # reducer issues the all-reduce calls and bytes of an n-rank step
    span = world
    if world == 1 and int(os.environ.get("HMA_BENCH_STEP_DOMAINS", "0")) > 1:
        span = int(os.environ["HMA_BENCH_STEP_DOMAINS"])
    seq = domain_sequence(len(domains), total * span)
    mine = [seq[k * span + rank] for k in range(total)]  # rank r takes every world-th draw
    batches = {}
    for di in sorted(set(mine)):
        batches[di] = synthetic_batch(B, T, 100 + di, d_actions[di], dev)
    eng = trainer.engine
    if args.unfused_mlp:
        eng.fused_mlp_train = False

    def one(k):
        di = mine[k]
        ids, labels, act = batches[di]
        # every rank knows the whole draw sequence (one shared sampler): the step's domain set needs no collective
        return trainer.step(ids, labels, act, [domains[di]] * B, step_domains=[domains[j] for j in seq[k * span:(k + 1) * span]])

    # untimed preparation: every (shape, domain) pair of the schedule is run until its launch plan is captured
    # as a hipGraph (N = 1 path), so the timed steps replay graphs only -- the analogue of a compiler warm-up
    # (every rank runs the same NUMBER of steps per slot so the collectives stay matched)
    prepare_steps = 0
    for rnd in range(2):  # (a (shape, domain) is captured on its first use once the trainer has captured any: two rounds leave every slot graphed)
        for k in range(total):
            first = mine.index(mine[k]) == k
            ids, labels, act = batches[mine[k]]
            if world == 1 and not first:
                continue
            trainer.step(ids, labels, act, [domains[mine[k]]] * B, step_domains=[domains[j] for j in seq[k * span:(k + 1) * span]])
            prepare_steps += 1
    for k in range(args.warmup):
        ws = one(k)
    torch.cuda.synchronize()
    if world > 1 or force:
        dist.barrier()
    torch.cuda.synchronize()
    # HMA_BENCH_DP_CHECK=1 (debug, tests/test_dp_gpu.py): after every timed step the ranks compare a digest of their weights, and the
    # line reports how many gradient buckets were all-reduced from inside the backward and how many bytes went through all-reduce
    dp_check = {"steps": 0, "weights_equal": True, "early_buckets": [], "bytes_per_step": [], "buckets": len(trainer.reducer.dense_buckets)} \
        if (os.environ.get("HMA_BENCH_DP_CHECK") == "1" and (world > 1 or force)) else None
    if force and world == 1:
        bytes_seen = []
    t0 = time.perf_counter()
    for k in range(args.warmup, total):
        ws = one(k)
        if force and world == 1:
            bytes_seen.append(int(trainer.reducer.bytes_step))
        if dp_check is not None:
            P = trainer.engine.P
            idx = torch.arange(P.numel(), device=P.device, dtype=torch.float64)
            dig = torch.stack([P.double().sum(), P.double().abs().sum(), (P.double() * (1.0 + (idx % 977.0))).sum()])
            alld = [torch.zeros_like(dig) for _ in range(world)]
            dist.all_gather(alld, dig)
            dp_check["weights_equal"] = dp_check["weights_equal"] and all(torch.equal(a, alld[0]) for a in alld)
            dp_check["steps"] += 1
            dp_check["early_buckets"].append(int(trainer.reducer.early_buckets))
            dp_check["bytes_per_step"].append(int(trainer.reducer.bytes_step))
    t_issue = time.perf_counter() - t0  # the host is done enqueueing the timed steps (far below dt: the GPU is the limit)
    torch.cuda.synchronize()
    if world > 1 or force:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loss, acc = trainer.loss_and_acc(ws)  # (of the LAST TIMED step: read before the power / instrumented passes below run more steps)
    loss = float(loss.item())
    timer = None
    power = None
    if not args.no_kernel_timing and world == 1:
        power = sample_clock_power(lambda: [one(args.warmup + (k % max(1, args.steps))) for k in range(14)])
    if not args.no_kernel_timing:
        # per-launch HIP events need eager launches: an instrumented pass of the same steps right after the timed
        # region (the timed region itself replays hipGraphs on the N = 1 path)
        eng.timer = LaunchTimer(FAMILIES)
        for k in range(args.warmup, min(total, args.warmup + 3)):
            one(k)
        torch.cuda.synchronize()
        timer, eng.timer = eng.timer, None
        timed_ms = sum(p[2].elapsed_time(p[3]) for p in timer.pairs)
        inst_steps = min(total, args.warmup + 3) - args.warmup
    if world > 1 or force:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        tokens = world * B * T * 256 * args.steps
        value = tokens / dt
        out = {
            "metric": "video-tokens/sec (train step: fwd+bwd+all-reduce+clip+AdamW) HMA-base T=16 16x16",
            "value": value, "unit": "video-tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic", "host_issue_ms_per_step": 1e3 * t_issue / args.steps,
            "config": {"workload": f"HMA-base-disc d256/h8/L{args.layers}, {args.domains} action domains "
                                   f"({sum(p.numel() for p in model.parameters()) / 1e6:.1f}M params), synthetic VQ tokens "
                                   f"T={T} H=W=16 ids<8192 (+64 action tokens/frame), batch {B}/GPU",
                       "global_batch": B * world, "seq_len": T * 256, "parallelism": f"dp{world}"},
            "per_gpu": value / world, "backend": backend,
            "mfma_roofline_frac_step": value / world * FLOP_PER_TOKEN_FWD_BWD / MFMA_PEAK,
            "final_loss": loss,
        }
        if power is not None:
            out["power"] = power
        if dp_check is not None:
            out["dp_check"] = dp_check
        if force and world == 1:
            out["allreduce_bytes_per_step"] = bytes_seen
        if timer is not None:
            summ = timer.summary()
            step_ms = 1e3 * dt / args.steps
            fams = {}
            for name, sm in summ.items():
                if sm["ms"] <= 0:
                    continue
                sec = sm["ms"] * 1e-3
                ach = sm["flops"] / sec / 1e12
                gbs = sm["bytes"] / sec / 1e9
                # the roofline that bounds the family: arithmetic intensity against the machine balance 2500 TFLOP/s / 8 TB/s
                ai = sm["flops"] / sm["bytes"] if sm["bytes"] > 0 else float("inf")
                fams[name] = {"bound": "hbm" if ai < MFMA_PEAK / 8e12 else "mfma", "achieved": ach, "frac": ach / 2500.0,
                              "hbm_achieved_gbs": gbs, "hbm_frac": gbs / 8000.0, "flop_per_byte": ai,
                              "launches": sm["launches"], "avg_launch_us": 1e3 * sm["ms"] / sm["launches"],
                              "flops_per_launch": sm["flops"] / sm["launches"], "bytes_per_launch": sm["bytes"] / sm["launches"],
                              "share_of_step_time": (sm["ms"] / inst_steps) / step_ms}
            # The weight-gradient ring kernel (gemm_tn_dma_kernel + its reduction tn_reduce_native_kernel) serves the three
            # hma_gemm_tn_pair calls of a layer AND linear_out's single hma_gemm_tn call: one entry for every C-ABI call that runs
            # it, so that its average is the number a rocprofv3 kernel summary gives (sum of the two kernels' time / ring launches)
            # (the launches are selected by the ENTRY POINT that runs the ring kernel, not by matching FLOP counts; the entry points'
            # own family entries are dropped in favour of this one, so that the dominant-family selection below sees the launch once)
            ring = [p_ for p_ in timer.pairs if p_[0] in ("hma_gemm_tn_pair", "hma_gemm_tn_multi")]
            if ring:
                fams.pop("hma_gemm_tn_pair", None)
                fams.pop("hma_gemm_tn_multi", None)
                ms = sum(p_[2].elapsed_time(p_[3]) for p_ in ring)
                fl, by = sum(p_[1] for p_ in ring), sum(p_[4] for p_ in ring)
                ach, gbs, ai = fl / (ms * 1e-3) / 1e12, by / (ms * 1e-3) / 1e9, fl / by
                fams["wgrad_ring"] = {"bound": "hbm" if ai < MFMA_PEAK / 8e12 else "mfma", "achieved": ach, "frac": ach / 2500.0,
                                      "hbm_achieved_gbs": gbs, "hbm_frac": gbs / 8000.0, "flop_per_byte": ai, "launches": len(ring),
                                      "avg_launch_us": 1e3 * ms / len(ring), "flops_per_launch": fl / len(ring),
                                      "bytes_per_launch": by / len(ring), "share_of_step_time": (ms / inst_steps) / step_ms,
                                      "note": "the C-ABI calls that run gemm_tn_dma_kernel + tn_reduce_native_kernel: per TWO layers ONE "
                                              "hma_gemm_tn_multi call with the blocks' fourteen weight gradients (engine.wgrad_layers)"}
            # `roofline` = the family with the largest share of the step, against the roof its arithmetic intensity puts it under
            dom_name = max(fams, key=lambda k: fams[k]["share_of_step_time"]) if fams else None
            if dom_name:
                d0 = fams[dom_name]
                # (every kernel the C-ABI call launches, per CALL: the weight-gradient entry points launch the ring kernel and its reduction)
                fam_kernels = {"hma_gemm_nt": (("gemm_nt",), "gemm_nt"), "hma_mlp_bwd": (("mlp_bwd",), "mlp_bwd"), "hma_mlp_fwd": (("mlp_fwd",), "mlp_fwd"),
                               "hma_gemm_tn_pair": (("gemm_tn_dma", "tn_reduce_native"), "gemm_tn_dma"), "hma_chain_a_fwd": (("chain_a_fwd",), "chain_a_fwd"),
                               "wgrad_ring": (("gemm_tn_dma", "tn_reduce_native"), "gemm_tn_dma"), "hma_chain_s_bwd": (("chain_s_bwd",), "chain_s_bwd"),
                               "hma_chain_t_bwd": (("chain_t_bwd",), "chain_t_bwd"),
                               "hma_gemm_tn_multi": (("gemm_tn_dma", "tn_reduce_native"), "gemm_tn_dma"),
                               "hma_chain_a_bwd": (("chain_a_bwd",), "chain_a_bwd"), "hma_chain_b_fwd": (("chain_b_fwd",), "chain_b_fwd"),
                               "hma_attn_spatial_bwd": (("attn_bwd_fused",), "attn_bwd_fused")}.get(dom_name, ((dom_name,), dom_name))
                traffic, pmc_file = pmc_traffic_per_call(*fam_kernels)
                same_scope = d0["bytes_per_launch"]
                mfma = {"achieved": d0["achieved"], "peak": 2500.0, "unit": "TFLOP/s", "frac": d0["frac"]}
                if power is not None and power.get("sclk_mhz"):
                    # the dense-MFMA peak at the clock the step actually runs at (the part's 2.5 PF are quoted at 2 400 MHz; the step sits
                    # at the package power limit): the same launches against that roof
                    mfma["peak_in_situ"] = 2500.0 * power["sclk_mhz"] / 2400.0
                    mfma["frac_in_situ"] = d0["achieved"] / mfma["peak_in_situ"]
                    mfma["sclk_mhz"] = power["sclk_mhz"]
                hbm = {"achieved": d0["hbm_achieved_gbs"], "peak": 8000.0, "unit": "GB/s", "frac": d0["hbm_frac"]}
                # headline = the MFMA roof (SURVEY.md 8d / north_star: dense contraction => MFMA); the HBM view of the same
                # launches (algorithmic bytes / time) rides beside it as `hbm`
                out["roofline"] = {"bound": "mfma", "kernel": dom_name, "achieved": mfma["achieved"], "peak": mfma["peak"],
                                   "unit": mfma["unit"], "frac": mfma["frac"], "traffic": traffic,
                                   "flop_per_byte": d0["flop_per_byte"], "mfma": mfma, "hbm": hbm,
                                   "bound_by_intensity": d0["bound"],
                                   "bytes_per_launch": d0["bytes_per_launch"], "algorithmic_bytes_in_traffic_scope": same_scope,
                                   "traffic_note": f"traffic = measured HBM bytes per C-ABI call (every kernel the call launches: for the weight gradients the "
                                                   f"ring kernel + its reduction; the ring kernel also serves linear_out's single hma_gemm_tn call, so it is the "
                                                   f"average over the layer's four weight-gradient calls), FETCH_SIZE x2 + WRITE_SIZE from the committed PMC passes "
                                                   f"(profiles/{pmc_file}); achieved = ALGORITHMIC bytes (every operand once, every result once) / "
                                                   "launch time; bound = hbm when FLOP per algorithmic byte < 2500e12 / 8e12",
                                   "launches": d0["launches"], "avg_launch_us": d0["avg_launch_us"], "flops_per_launch": d0["flops_per_launch"],
                                   "share_of_step_time": d0["share_of_step_time"],
                                   "rocprof_kernels": list(fam_kernels[0]),
                                   "measured": f"HIP events around every launch of the MFMA kernel families in {inst_steps} eager steps run "
                                               "right after the timed region, on the stream the kernels run on.  The timed region replays the "
                                               "SAME launches in the SAME order on one stream as a hipGraph (round 5: no forked launches), so "
                                               "avg_launch_us is comparable with the per-kernel averages of a rocprofv3 --kernel-trace --stats "
                                               "run of this command (profiles/kernel_stats_r6.csv; tools/roofline_check.py adds the kernels of a "
                                               "C-ABI call -- for the weight gradients the ring kernel AND its reduction -- and compares).  "
                                               "Algorithmic FLOPs: recomputation is not counted",
                                   "families": fams}
            out["config"]["prepare_steps"] = prepare_steps
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, domains[mine[0]], d_actions[mine[0]], T, quick=args.quick_cpu)
        if world == 1 and args.mode == "all" and args.layers == 32:
            # the other single-GPU BASELINE configs, measured by the same command (sub-objects of the one line)
            del trainer, eng, batches
            model = None
            torch.cuda.empty_cache()
            out["decode"] = decode_bench(args, dev, steps=3, warmup=2, batch=64)  # (two warm-up rollouts: a frame pass is captured on its second use)
            out["mar"] = mar_bench(args, dev, steps=5, warmup=4)  # (the step is launched eagerly: the allocator settles over the first steps)
            torch.cuda.empty_cache()
            # the reference's own default window (train_multi.py:78-83, both shipped d256 JSONs: T = 12) and what one GPU can measure of
            # the 8-rank risk (RCCL's kernels beside the persistent 256-workgroup launches), each in a child process of this command
            out["train_T12"] = child_line(["--frames", "12"], {})
            out["dp_check"] = forced_collectives_check()
        print(json.dumps(out), flush=True)
    if world > 1 or force:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
