"""N > 1 path on CPU: the flat layout, bucket plan and the sparse-by-domain gradient all-reduce
(world_size 2, gloo).  The same GradReducer runs over RCCL on the GPU box."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hma_amd.config import GenieConfig
from hma_amd.params import ALIGN, ParamLayout
from hma_amd.train import GradReducer, lr_at
from oracle.param_spec import state_dict_spec
from tests.golden.golden_cfg import TINY
from tests.helpers import tiny_ref_config


def make_layout(num_layers=4, domains=("domA", "domB", "domC"), d_actions=(7, 14, 21)):
    cfg = GenieConfig(**{**TINY["config"], "num_layers": num_layers})
    return cfg, ParamLayout(cfg, list(domains), list(d_actions), [7] * len(domains))


def test_layout_matches_reference_state_dict_and_is_aligned():
    cfg, lay = make_layout(2, TINY["domains"], TINY["d_actions"])
    spec = state_dict_spec(tiny_ref_config(), TINY["domains"], TINY["d_actions"], [7, 7])
    params = {k: v for k, v in spec.items() if not (k.endswith(".mean") or k.endswith(".std"))}
    assert set(lay.entries) == set(params)
    spans = []
    for name, e in lay.entries.items():
        assert tuple(params[name]) == e.shape, name
        assert e.offset % ALIGN == 0
        spans.append((e.offset, e.offset + e.numel))
    spans.sort()
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0, "views overlap"
    assert spans[-1][1] <= lay.total
    # constant per-layer stride, layers stored L-1 .. 0 (backward-completion order)
    assert lay.off("decoder.layers.0.mlp.fc1.weight") - lay.off("decoder.layers.1.mlp.fc1.weight") == lay.layer_stride > 0
    # a domain's block is layer-major with a constant layer stride (the batch stride of the adaLN GEMMs), so the modulation tensors of
    # a run of layers -- a gradient bucket -- are one contiguous slice
    for key in ("adaLN_modulation.0.weight", "adaLN_modulation.2.weight", "linear_out.weight", "adaLN_modulation.2.bias", "linear_out.bias"):
        a = lay.off(f"decoder.layers.0.action_projectors.domA.{key}")
        b = lay.off(f"decoder.layers.1.action_projectors.domA.{key}")
        assert b - a == lay.dom_layer_stride == 256 * 256 * 4 + 256 * 4, key


def test_domain_slices_of_the_buckets_tile_the_domain_block():
    _, lay = make_layout(4)
    for lpb in (1, 3, 8):
        dense = lay.buckets(lpb)
        for dom in lay.domains:
            sl = lay.dom_buckets(dom, lpb)
            assert len(sl) == len(dense) + 1
            a0, a1 = lay.regions[f"dom:{dom}"]
            cover = sorted(sl)
            assert cover[0][0] == a0 and cover[-1][1] == a1
            for (x0, x1), (y0, y1) in zip(cover, cover[1:]):
                assert x1 == y0 and x0 < x1
        # bucket 0 holds the LAST layers (backward order): the slice contains layer L - 1's tensors and not layer 0's
        s0 = lay.dom_buckets("domA", 1)[0]
        assert s0[0] <= lay.off("decoder.layers.3.action_projectors.domA.linear_out.bias") < s0[1]
        assert not (s0[0] <= lay.off("decoder.layers.0.action_projectors.domA.linear_out.weight") < s0[1])
        assert s0[0] <= lay.off("decoder.layers.3.action_projectors.domA.adaLN_modulation.0.weight") < s0[1]


def test_decay_flags_follow_the_reference_grouping():
    _, lay = make_layout(2, TINY["domains"], TINY["d_actions"])
    flags = lay.decay_flags()
    for name, e in lay.entries.items():
        f = int(flags[e.offset // ALIGN])
        if e.region == "frozen":
            assert f == 0, name
        else:
            assert f == (1 if "bias" in name else 2), name  # train_multi.py:907-918: only "bias" names are un-decayed
    assert int(flags[lay.off("decoder.layers.0.norm1.weight") // ALIGN]) == 2  # LN weights ARE decayed


@pytest.mark.parametrize("lpb", [1, 3, 8])
def test_buckets_tile_the_dense_region_in_backward_order(lpb):
    _, lay = make_layout(4)
    buckets = lay.buckets(lpb)
    assert buckets[0][0] == lay.regions["head"][0]
    assert buckets[-1][1] == lay.regions["layer0"][1] == lay.regions["tail"][0]
    for (a0, a1), (b0, b1) in zip(buckets, buckets[1:]):
        assert a1 == b0 and a0 < a1
    ranges = lay.trainable_ranges(["domB"])
    assert ranges[0] == (lay.regions["head"][0], lay.regions["tail"][1])
    assert ranges[1:] == [lay.regions["dom:domB"]]


def test_lr_schedule_constant_with_warmup():
    # LambdaLR(constant_with_warmup): the k-th update (0-based) runs at base * k / warmup (train_multi.py:194-199, 979-986)
    assert lr_at(0, 1e-4, 500) == 0.0
    assert lr_at(1, 1e-4, 500) == pytest.approx(1e-4 / 500)
    assert lr_at(500, 1e-4, 500) == pytest.approx(1e-4)
    assert lr_at(10_000, 1e-4, 500) == 1e-4
    assert lr_at(3, 2e-4, 0) == 2e-4


def _worker(rank, world, port, lpb, early, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, lay = make_layout(4)
        G = torch.zeros(lay.total)
        red = GradReducer(lay, G, layers_per_bucket=lpb)
        local = ["domC", "domA"][rank]  # ranks hold different domains; domB is idle everywhere
        active = red.active_domains(local)
        assert active == ["domA", "domC"], active
        # the collective-free form: the driver knows every rank's domain (all processes iterate one shared sampler)
        assert red.order(["domC", "domA", "domC", None]) == active
        loss_info = torch.tensor([2.5 * (rank + 1), 2.0, float(rank), 0.0])  # [sum loss * B, count, non-finite, -]
        # what a rank's backward leaves in G: dense range + its own domain block, zeros elsewhere
        dense = lay.trainable_ranges([])[0]
        G[dense[0]:dense[1]] = torch.arange(dense[1] - dense[0], dtype=torch.float32) * (rank + 1) * 1e-3
        a, b = lay.regions[f"dom:{local}"]
        G[a:b] = float(rank + 1)
        red.begin(active if early else ())   # (early: the domains' per-bucket slices ride with the dense buckets)
        L = cfg.num_layers
        for l in reversed(range(L)):           # the labels STEngine.backward emits with segment_layers = lpb
            if (L - l) % lpb == 0 or l == 0:
                red.on_segment(f"layer{l}")
        red.on_segment("end")
        n_early = len(red._pending)
        red.finish(active, extra=[loss_info])
        # dense buckets (+ two domain slices each when early) were launched from inside the "backward"
        assert n_early == len(red.dense_buckets) * (3 if early else 1), n_early
        assert loss_info.tolist() == [7.5, 4.0, 1.0, 0.0]  # rides along with the gradients (train_multi.py:599)
        exp = torch.zeros(lay.total)
        exp[dense[0]:dense[1]] = torch.arange(dense[1] - dense[0], dtype=torch.float32) * 3e-3  # ranks 1x + 2x
        a, b = lay.regions["dom:domC"]
        exp[a:b] = 1.0   # only rank 0 had domC
        a, b = lay.regions["dom:domA"]
        exp[a:b] = 2.0   # only rank 1 had domA
        ok = torch.allclose(G, exp, rtol=1e-6, atol=0)
        idle = lay.regions["dom:domB"]
        ok = ok and float(G[idle[0]:idle[1]].abs().sum()) == 0.0
        fr = lay.regions["frozen"]
        ok = ok and float(G[fr[0]:fr[1]].abs().sum()) == 0.0
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lpb,early", [(1, True), (3, True), (3, False)])
def test_sparse_by_domain_allreduce_world2_gloo(lpb, early):
    world = 2
    port = 29500 + (os.getpid() % 2000) + lpb + (7 if early else 0)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, lpb, early, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_accelerate_layout_optimizer_state_loads_into_the_reference_optimizer():
    """SURVEY (f) row 4: `optimizer.bin` / `scheduler.bin` in the layout accelerator.save_state writes (train_multi.py:310-321).
    The exported dict must be what torch's AdamW -- built the way the reference builds it (train_multi.py:907-922) -- produces
    itself: same groups, same indices, state only for stepped parameters; and it must load into that optimizer."""
    from hma_amd.model import STMaskGIT
    from hma_amd.train import build_optimizer_state_dict, build_scheduler_state_dict, reference_param_groups

    cfg = GenieConfig(**{**TINY["config"], "num_layers": 1})
    m = STMaskGIT(cfg)  # (construction is plain nn.Module code: no GPU needed)
    m.init_action_projectors(TINY["domains"], TINY["d_actions"], TINY["action_stats"], cfg.action_network)
    named = list(m.named_parameters())
    no_decay = ["bias", "layer_norm.weight"]
    groups = [{"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": 0.05},
              {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
    opt = torch.optim.AdamW(groups, lr=1e-3, betas=(0.9, 0.95), eps=1e-8)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: min(1.0, s / 500))
    g = torch.Generator().manual_seed(0)
    stepped = {n for n, _ in named if ".domB." not in n and "action_out" not in n and n != "action_mask_tokens"}
    for it in range(2):
        for n, p in named:
            p.grad = torch.randn(p.shape, generator=g) if n in stepped else None
        opt.step()
        sched.step()
    ref = opt.state_dict()
    names = [n for n, _ in named]
    g0, g1 = reference_param_groups(names)
    order = g0 + g1
    by_name = {order[i]: st for i, st in ref["state"].items()}
    ours = build_optimizer_state_dict(names, lambda n: (by_name[n]["exp_avg"], by_name[n]["exp_avg_sq"]),
                                      lambda n: 2 if n in stepped else 0, sched.get_last_lr()[0], 1e-3, (0.9, 0.95), 1e-8, 0.05)
    assert [gr["params"] for gr in ours["param_groups"]] == [gr["params"] for gr in ref["param_groups"]]
    assert set(ours["state"]) == set(ref["state"])
    for i in ref["state"]:
        assert float(ours["state"][i]["step"]) == float(ref["state"][i]["step"]) == 2.0
        assert torch.equal(ours["state"][i]["exp_avg"], ref["state"][i]["exp_avg"])
    for a, b in zip(ours["param_groups"], ref["param_groups"]):
        assert set(b) <= set(a), set(b) - set(a)
        assert all(a[k] == b[k] for k in b if k != "params"), {k: (a[k], b[k]) for k in b if a[k] != b[k]}
    opt2 = torch.optim.AdamW(groups, lr=1e-3, betas=(0.9, 0.95), eps=1e-8)
    opt2.load_state_dict(ours)  # what accelerator.load_state does with optimizer.bin
    sch = build_scheduler_state_dict(2, 1, sched.get_last_lr()[0], 1e-3)
    want = sched.state_dict()
    assert {k: sch[k] for k in ("last_epoch", "_step_count", "base_lrs", "_last_lr")} == {k: want[k] for k in ("last_epoch", "_step_count", "base_lrs", "_last_lr")}
    torch.optim.lr_scheduler.LambdaLR(opt2, lambda s: min(1.0, s / 500)).load_state_dict(sch)
