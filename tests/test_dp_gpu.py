"""Data-parallel equivalence through the REAL engine on the GPU box: two fresh processes (one per rank, both on GPU 0, gloo),
each running `Trainer.step` on its own domain and batch with the per-bucket hipGraph path and the bucketed all-reduce on the
side stream, against ONE process that accumulates the same two micro-batches with 1/2 scaling.

What must hold (reference semantics of DDP + zero_grad(set_to_none), train_multi.py:556-599, 779, 990):
  * grad = sum over ranks / world; ranks without a domain contribute 0 to its block;
  * clip over the REDUCED gradients; AdamW over the dense range and the domains active on some rank;
  * the idle domain (domC) is never touched: no decay, no moments;
  * a non-finite loss on one rank skips the update on EVERY rank, and the skipped step does not count for Adam;
  * the reduced loss is the mean over ranks (accelerator.reduce(loss_info), :599).
"""
import os
import subprocess
import sys

import pytest
import torch
from safetensors.torch import load_file

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _launch_pair(out_path, nan_step=-1, mode=None):
    port = 29600 + os.getpid() % 1000 + (7 if nan_step >= 0 else 0) + ({None: 0, "mar": 13, "mar_mixed": 29}[mode])
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
        args = [sys.executable, os.path.join(HERE, "dp_child.py"), str(out_path), str(nan_step)] + ([mode] if mode else [])
        procs.append(subprocess.Popen(args, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return load_file(str(out_path))


def _single_process(nan_step=-1):
    import dp_child as C
    from hma_amd.train import Trainer

    model = C.build_model()
    tr = Trainer(model, lr=1e-3, warmup_steps=0, layers_per_bucket=2, grad_accum=2)
    init = {n: p.detach().float().cpu().clone() for n, p in model.named_parameters()}
    losses = []
    for step in range(C.STEPS):
        for which in range(2):
            ids, labels, act, dom = C.batch(which, step)
            if step == nan_step and which == 1:
                act = act.clone()
                act[0, 0, 0] = float("nan")
            tr.micro_step(ids, labels, act, dom)  # (domains are NOT announced: mixed-domain accumulation, domB appears late)
        tr.optimizer_step()
        losses.append(tr.reduced_loss().detach().clone())
    torch.cuda.synchronize()
    return C.digest(model, tr, losses), init


@pytest.mark.timeout(900)
@pytest.mark.parametrize("nan_step", [-1, 2])
def test_two_ranks_equal_one_rank_accumulating(tmp_path, nan_step):
    import dp_child as C
    two = _launch_pair(tmp_path / "rank0.safetensors", nan_step)
    one, init = _single_process(nan_step)
    applied = C.STEPS - (1 if nan_step >= 0 else 0)
    assert int(two["_opt_step"]) == int(one["_opt_step"]) == applied
    assert two["_dom_steps"].tolist() == one["_dom_steps"].tolist() == [applied, applied, 0]
    # the reduced loss of every step (NaN-contributing micro-batches are left out of the mean, train_multi.py:572-577)
    assert torch.allclose(two["_losses"], one["_losses"], rtol=2e-6, atol=0), (two["_losses"], one["_losses"])
    worst = 0.0
    for name, w2 in two.items():
        if name.startswith("_"):
            continue
        w1 = one[name]
        if ".domC." in name:
            assert torch.equal(w2, init[name]) and torch.equal(w1, init[name]), f"idle domain touched: {name}"
            continue
        moved = (w1 - init[name]).double().pow(2).mean().sqrt().item()
        assert moved > 0, name
        err = (w2 - w1).double().pow(2).mean().sqrt().item()
        worst = max(worst, err / moved)
        # Same arithmetic up to the order of fp32 additions (all-reduce vs in-place accumulation, the atomics of the norm and of
        # the embedding / stem gradients).  Adam divides by sqrt(v): where a gradient element is itself rounding noise its update
        # is +-lr either way, so the comparison is rms over the tensor, with a wider band for the small
        # vectors whose gradients are mostly such noise (measured worst: see gpurun_out/dp_equivalence_*.txt)
        assert err <= (0.15 if w2.numel() <= 4096 else 3e-2) * moved, (name, err, moved)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/dp_equivalence_{'nan' if nan_step >= 0 else 'plain'}.txt", "w") as f:
        f.write(f"worst |w_2ranks - w_1rank| / |w - w_init| over checked tensors: {worst:.3e}\n")


@pytest.mark.timeout(900)
def test_stmar_two_ranks_equal_one_rank_accumulating(tmp_path):
    """configs[3] (STMAR, an 8-GPU config): the trunk's flat ranges AND the model's own flat range (input / output stages +
    diffusion head) are all-reduced and stepped; two ranks with different domains == one rank accumulating both."""
    import dp_child as C
    from hma_amd.train import MarTrainer

    two = _launch_pair(tmp_path / "mar0.safetensors", mode="mar")
    model = C.build_mar()
    init = {n: p.detach().float().cpu().clone() for n, p in model.named_parameters()}
    tr = MarTrainer(model, lr=1e-3, warmup_steps=0, grad_accum=2)
    losses = []
    for step in range(3):
        for which in range(2):
            tr.micro_step(**C.mar_batch(which, step))
        tr.optimizer_step()
        losses.append(tr.reduced_loss().detach().clone())
    torch.cuda.synchronize()
    one = C.mar_digest(model, losses)
    assert abs(float(two["_losses"][0]) - float(one["_losses"][0])) <= 1e-5 * float(one["_losses"][0])   # same weights, same batches
    assert torch.allclose(two["_losses"], one["_losses"], rtol=1e-3, atol=0), (two["_losses"], one["_losses"])  # after noise-level weight differences
    worst, per = 0.0, {}
    for name, w2 in two.items():
        if name.startswith("_"):
            continue
        moved = (one[name] - init[name]).double().pow(2).mean().sqrt().item()
        assert moved > 0, name
        err = (w2 - one[name]).double().pow(2).mean().sqrt().item()
        worst = max(worst, err / moved)
        per[name] = err / moved
        # (three Adam steps of +-lr: the small bias / gain vectors, whose gradients are partly bf16 noise that the reduction
        # order perturbs, sit further apart than the matrices -- measured values are written to gpurun_out/)
        assert err <= (0.15 if w2.numel() <= 1024 else 3e-2) * moved, (name, err, moved)
    with open("gpurun_out/dp_equivalence_mar.txt", "w") as f:
        f.write(f"worst rms |w_2ranks - w_1rank| / rms |w - w_init| over checked tensors: {worst:.3e}\n")
        for k, v in sorted(per.items()):
            f.write(f"  {k}: {v:.3e}\n")


@pytest.mark.timeout(900)
def test_stmar_mixed_domains_under_accumulation_without_step_domains(tmp_path):
    """with gradient accumulation, world > 1 and no `step_domains`, every rank has to enter the domain gather on
    every micro-batch (a rank whose own domain was already known used to skip it: the other rank's all_gather then paired with its
    all_reduce).  Rank 0 sees domA twice, rank 1 domA then domB; the result must equal one process accumulating the four batches."""
    import dp_child as C
    from hma_amd.train import MarTrainer

    two = _launch_pair(tmp_path / "marmix0.safetensors", mode="mar_mixed")
    model = C.build_mar()
    init = {n: p.detach().float().cpu().clone() for n, p in model.named_parameters()}
    tr = MarTrainer(model, lr=1e-3, warmup_steps=0, grad_accum=4)
    losses = []
    for step in range(2):
        for k in range(2):
            for rank in range(2):
                tr.micro_step(**C.mar_batch(C.MIXED[rank][k], 4 * step + 2 * k + rank))
        tr.optimizer_step()
        losses.append(tr.reduced_loss().detach().clone())
    torch.cuda.synchronize()
    one = C.mar_digest(model, losses)
    assert abs(float(two["_losses"][0]) - float(one["_losses"][0])) <= 1e-5 * float(one["_losses"][0])
    for name, w2 in two.items():
        if name.startswith("_"):
            continue
        moved = (one[name] - init[name]).double().pow(2).mean().sqrt().item()
        assert moved > 0, name
        err = (w2 - one[name]).double().pow(2).mean().sqrt().item()
        assert err <= (0.15 if w2.numel() <= 1024 else 3e-2) * moved, (name, err, moved)


@pytest.mark.timeout(900)
def test_bench_gpus_2_spawns_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two fresh ranks itself (torch.distributed.run as a child
    process) and relays ONE line with n_gpus = 2; HMA_BENCH_ONE_DEVICE=1 puts both ranks on GPU 0 over gloo (a one-GPU box: the
    numbers mean nothing, the path -- rendezvous, the 40-domain layout of configs[2] with each rank drawing its own domain per step
    from the shared sequence, per-bucket graphs + side-stream all-reduces of the dense slices and the active domains' slices -- is
    the N-rank one).  The reference launches the same way: experiments/scripts/run_30datasets_waction.sh:17-19."""
    import json

    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HMA_BENCH_ONE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--layers", "2", "--steps", "3", "--warmup", "1",
                        "--domains", "40", "--batch", "4", "--mode", "train", "--no-cpu-baseline", "--no-kernel-timing"],
                       env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 8
    assert line["backend"] == "gloo" and line["value"] > 0
    # ---- the same launch over SIX timed steps with the ranks checking each other (HMA_BENCH_DP_CHECK=1): after EVERY optimizer step the
    # two ranks hold the same weights (each drew its own domain: dense slices + two domains' slices summed, one clip, one AdamW), and
    # every gradient bucket's all-reduce was issued from inside the backward (between the per-bucket graphs), not after it
    env["HMA_BENCH_DP_CHECK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--layers", "4", "--steps", "6", "--warmup", "1",
                        "--domains", "40", "--batch", "4", "--mode", "train", "--no-cpu-baseline", "--no-kernel-timing"],
                       env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    chk = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["dp_check"]
    assert chk["steps"] == 6 and chk["weights_equal"], chk
    assert chk["buckets"] >= 1 and all(e == chk["buckets"] for e in chk["early_buckets"]), chk
    assert all(b > 0 for b in chk["bytes_per_step"]), chk
    env.pop("HMA_BENCH_DP_CHECK")
    # more ranks than visible devices (no one-device override): a loud failure, not a silent single-rank line
    env.pop("HMA_BENCH_ONE_DEVICE")
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--layers", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.timeout(900)
def test_rccl_all_reduces_run_in_a_one_rank_group():
    """HMA_FORCE_COLLECTIVES=1: a ONE-rank `nccl` process group, so that a one-GPU box executes what an N-GPU job executes -- RCCL
    initialised by this code, the bucketed all-reduces issued on the side stream between the per-bucket hipGraphs, the timing
    barriers and the max-over-ranks of bench.py -- except the transport.  An all-reduce over one rank is the identity: the step's
    loss must equal the plain one-process run's on the same seeds."""
    import json

    root = os.path.dirname(HERE)
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                              "HMA_BENCH_ONE_DEVICE", "HMA_FORCE_COLLECTIVES")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--layers", "4", "--steps", "3", "--warmup", "2", "--domains", "4",
           "--batch", "4", "--mode", "train", "--no-cpu-baseline", "--no-kernel-timing"]

    def run(env):
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        return json.loads(lines[0])

    plain = run(base)
    forced = run(dict(base, HMA_FORCE_COLLECTIVES="1", MASTER_PORT=str(29500 + os.getpid() % 400)))
    assert plain["backend"] is None and forced["backend"].startswith("nccl")
    assert forced["n_gpus"] == 1 and forced["value"] > 0
    assert abs(forced["final_loss"] - plain["final_loss"]) <= 1e-4 * max(1.0, abs(plain["final_loss"])), (forced["final_loss"], plain["final_loss"])
