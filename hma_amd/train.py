"""Training-step driver for the MI355X engine: the hot slice of hma/train_multi.py:556-599.

One process per GPU.  Per optimizer step:
  1. the set of action domains that receive a gradient on SOME rank in SOME micro-batch of the step is fixed: passed in by
     the driver (every process builds the same sampler, train_multi.py:928-932, so it is local knowledge -- no collective,
     no host sync), or, for callers that cannot know it, found by a tiny all-gather per micro-batch;
  2. forward + fused CE, hand-written backward into the flat gradient buffer (micro-batches of ANY domain accumulate);
  3. gradients are all-reduced (RCCL, `nccl` backend) in contiguous buckets of the flat buffer that
     become final as backward walks layers L-1..0 -- launched on a side HIP stream behind an event so
     they overlap the remaining backward; only the dense trunk and the domains that are active on
     SOME rank are reduced (the reference's DDP reduces all 362 M parameters, ~90 % zeros); the step's
     loss bookkeeping [sum loss * batch, batch count, non-finite count] rides along (train_multi.py:599);
  4. global-norm clip + AdamW over exactly those ranges.  Parameters unused on every rank are left
     untouched (no moment update, no weight decay) -- DDP's globally-unused rule.
     A non-finite loss on ANY rank makes the reduced gradient norm non-finite on EVERY rank; the update
     kernel then leaves weights, moments and update counts alone (the all-rank-consistent form of
     train_multi.py:572-583), decided on the device.
The per-micro-step barrier of the reference (train_multi.py:568) is dropped on purpose.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .params import ParamLayout


class GradReducer:
    """Bucketed sum-all-reduce of a flat gradient buffer, sparse by action domain.

    Device-agnostic on purpose (pure torch.distributed on 1-D slices) so the N > 1 logic is covered
    by world_size-2 gloo tests on CPU; on the GPU box the backend is RCCL over xGMI."""

    def __init__(self, layout: ParamLayout, G: torch.Tensor, layers_per_bucket: int = 8, group=None):
        self.layout, self.G, self.group = layout, G, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # whether the collectives run: always with more than one rank; HMA_FORCE_COLLECTIVES=1 also runs them in a ONE-rank process
        # group (an all-reduce over one rank is the identity, but it is RCCL that executes it, on the side stream, between the
        # per-bucket graphs: the N > 1 code path end to end on a one-GPU box -- tests/test_dp_gpu.py)
        self.collective = self.world > 1 or (dist.is_initialized() and os.environ.get("HMA_FORCE_COLLECTIVES") == "1")
        self.dense_buckets = layout.buckets(layers_per_bucket)
        L = layout.cfg.num_layers
        order = list(reversed(range(L)))
        # label of the backward segment after which bucket i is final
        self.bucket_label = [f"layer{order[min(i + layers_per_bucket, L) - 1]}" for i in range(0, L, layers_per_bucket)]
        self.tail = layout.regions["tail"]
        self._pending: List = []
        self._next = 0
        self.side = torch.cuda.Stream() if G.is_cuda else None
        self._dom_index = {d: i for i, d in enumerate(layout.domains)}
        # a domain's block is layer-major (params.py): its slice for the layers of dense bucket i is contiguous and final with the bucket
        self._dom_buckets = {d: layout.dom_buckets(d, layers_per_bucket) for d in layout.domains}
        self._active: List[str] = []
        self.early_buckets = 0
        self.bytes_step = 0

    def active_domains(self, local_domain: Optional[str]) -> List[str]:
        """Union over ranks of this micro-batch's domains, in layout order (same list on every rank).  FALLBACK for callers
        that do not pass `step_domains` to `Trainer.micro_step`: a blocking all-gather + host read per call."""
        if not self.collective:
            return [local_domain] if local_domain is not None else []
        idx = -1 if local_domain is None else self._dom_index[local_domain]
        mine = torch.tensor([idx], dtype=torch.int64, device=self.G.device)
        allv = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(allv, mine, group=self.group)
        seen = sorted({int(v) for v in torch.cat(allv).tolist() if int(v) >= 0})
        return [self.layout.domains[i] for i in seen]

    def order(self, domains) -> List[str]:
        """`domains` de-duplicated, in layout order."""
        want = {d for d in domains if d is not None}
        unknown = want - set(self._dom_index)
        if unknown:
            raise KeyError(f"unknown action domains {sorted(unknown)}")
        return [d for d in self.layout.domains if d in want]

    def _launch(self, a: int, b: int) -> None:
        if not self.collective or b <= a:
            return
        self.bytes_step += 4 * (b - a)
        sl = self.G[a:b]
        if self.side is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                self._pending.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._pending.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def begin(self, active: Sequence[str] = ()) -> None:
        """`active`: the step's domains (known before its last backward starts): their slices ride with the dense buckets."""
        self._pending, self._next = [], 0
        self._active = list(active)
        self.early_buckets = 0   # dense buckets (each with the active domains' slices) reduced from INSIDE the backward this step
        self.bytes_step = 0      # bytes handed to all-reduce this step (gradient slices; `extra` tensors not counted)

    def _launch_bucket(self, i: int) -> None:
        self._launch(*self.dense_buckets[i])
        for dom in self._active:
            self._launch(*self._dom_buckets[dom][i])

    def on_segment(self, label: str) -> None:
        """Called by STEngine.backward after each enqueued segment; reduces every bucket that is final: the dense range of its
        layers and, for every active domain, the modulation tensors of the same layers."""
        while self._next < len(self.dense_buckets) and label == self.bucket_label[self._next]:
            self._launch_bucket(self._next)
            self._next += 1
            self.early_buckets += 1

    def _launch_tensor(self, t: torch.Tensor) -> None:
        if not self.collective or t.numel() == 0:
            return
        if self.side is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                self._pending.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._pending.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self, active: Sequence[str], extra: Sequence[torch.Tensor] = ()) -> None:
        """Reduce what is left (unlaunched dense buckets, embeddings, active domain blocks, `extra` flat tensors such as the
        loss bookkeeping or STMAR's own gradient range) and wait."""
        early = set(self._active)
        while self._next < len(self.dense_buckets):
            self._launch_bucket(self._next)
            self._next += 1
        self._launch(*self.tail)
        for dom in active:
            if dom in early:   # (its per-bucket slices went with the buckets: the action stem is what is left)
                self._launch(*self._dom_buckets[dom][-1])
            else:
                self._launch(*self.layout.regions[f"dom:{dom}"])
        for t in extra:
            self._launch_tensor(t)
        for w in self._pending:
            w.wait()
        if self.side is not None and self.collective:
            torch.cuda.current_stream().wait_stream(self.side)
        self._pending = []


def lr_at(step: int, base_lr: float, warmup_steps: int) -> float:
    """`constant_with_warmup` (train_multi.py:194-199, 979-986); `step` counts completed optimizer steps."""
    if warmup_steps <= 0:
        return base_lr
    return base_lr * min(1.0, float(step) / float(warmup_steps))  # LambdaLR: lr of update k + 1 = base * k / warmup


# ---------------------------------------------------------------------------------------------------------------------
# Accelerate-layout optimizer / scheduler files (SURVEY (f) row 4; hma/train_multi.py:310-321 `accelerator.save_state`,
# :484-533 `accelerator.load_state`, hma/generate.py:80-84 reads scheduler.bin).  `optimizer.bin` is
# torch.save(optimizer.state_dict()) of the AdamW the reference builds (train_multi.py:907-922): two parameter groups --
# [names without "bias" / "layer_norm.weight"], [the rest] -- each in named_parameters() order, state keyed by the running
# index over both groups, only for parameters that have been stepped.  `scheduler.bin` is the LambdaLR state dict; Accelerate
# steps the scheduler once per process per optimizer step, so its counters run at `world` times the optimizer-step count.
NO_DECAY_SUBSTRINGS = ("bias", "layer_norm.weight")


def reference_param_groups(names: Sequence[str]) -> Tuple[List[str], List[str]]:
    """(decayed, un-decayed) parameter names in the order the reference hands them to AdamW (train_multi.py:907-918)."""
    nd = lambda n: any(k in n for k in NO_DECAY_SUBSTRINGS)
    return [n for n in names if not nd(n)], [n for n in names if nd(n)]


def build_optimizer_state_dict(names: Sequence[str], moments, steps, lr: float, base_lr: float, betas, eps: float,
                               weight_decay: float) -> dict:
    """`optimizer.state_dict()` of the reference's AdamW.  `moments(name) -> (exp_avg, exp_avg_sq)` CPU tensors in the
    parameter's shape; `steps(name) -> int` updates applied to that parameter (0: never stepped -> no state entry)."""
    g0, g1 = reference_param_groups(names)
    template = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=base_lr, betas=tuple(betas), eps=eps,
                                 weight_decay=weight_decay).state_dict()["param_groups"][0]
    groups, state, idx = [], {}, 0
    for members, wd in ((g0, weight_decay), (g1, 0.0)):
        ids = []
        for n in members:
            k = int(steps(n))
            if k > 0:
                m, v = moments(n)
                state[idx] = {"step": torch.tensor(float(k)), "exp_avg": m, "exp_avg_sq": v}
            ids.append(idx)
            idx += 1
        groups.append(dict(template, lr=lr, initial_lr=base_lr, weight_decay=wd, params=ids))
    return {"state": state, "param_groups": groups}


def build_scheduler_state_dict(completed: int, world: int, lr: float, base_lr: float) -> dict:
    """LambdaLR.state_dict() after `completed` optimizer steps under Accelerate (`world` scheduler steps per optimizer step)."""
    n = completed * world
    return {"base_lrs": [base_lr, base_lr], "last_epoch": n, "_step_count": n + 1, "_get_lr_called_within_step": False,
            "_last_lr": [lr, lr], "lr_lambdas": [None, None], "verbose": False}


class Trainer:
    """Fused train step on one GPU of a data-parallel job (no autograd, no per-tensor optimizer)."""

    def __init__(self, model, lr: float = 1e-4, betas=(0.9, 0.95), eps: float = 1e-8, weight_decay: float = 0.05,
                 max_grad_norm: Optional[float] = 1.0, warmup_steps: int = 0, layers_per_bucket: int = 8,
                 grad_accum: int = 1, device=None):
        self.model = model
        dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.engine = model._get_engine(dev)
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.max_grad_norm, self.warmup = max_grad_norm, warmup_steps
        self.accum = grad_accum
        self.reducer = GradReducer(self.engine.layout, self.engine.G, layers_per_bucket)
        self.layers_per_bucket = layers_per_bucket
        self.engine.ada_group = layers_per_bucket  # (the adaLN stacks' backward per bucket: a domain's slice of a bucket is final with it)
        self.completed = 0
        self._micro = 0
        self._active: List[str] = []
        self._known = False  # the step's domain set was given up front (no collective needed)
        # [sum over finite micro-batches of loss * batch, their batch count, non-finite micro-batches, unused]; all-reduced with
        # the gradients of each optimizer step (train_multi.py:560-577, 599) and snapshotted into `last_loss_info`
        self.loss_info = torch.zeros(4, dtype=torch.float32, device=dev)
        self.last_loss_info = torch.zeros(4, dtype=torch.float32, device=dev)
        # hipGraph replay of (zero-grad, forward, loss, backward) per (shape, domain): ~1500 launches per step
        # become one graph launch.  Single-GPU fast path only; the N > 1 path interleaves all-reduces eagerly.
        self.use_graphs = True
        self._graphs: Dict[tuple, "torch.cuda.CUDAGraph"] = {}
        self._seen: Dict[tuple, int] = {}

    def _announce(self, dom: Optional[str], step_domains) -> None:
        """Fix (first micro-batch) or extend (later ones) the set of domains whose blocks are zeroed, reduced and stepped."""
        eng, red = self.engine, self.reducer
        first = self._micro == 0
        if first:
            self._known = step_domains is not None
            self._active = red.order(step_domains) if self._known else red.active_domains(dom)
            eng.zero_grad(self._active)
            self.loss_info.zero_()
            return
        if self._known:
            if dom is not None and dom not in self._active:
                raise RuntimeError(f"domain {dom!r} is not in the step_domains announced for this optimizer step")
            return
        # not announced up front: a later micro-batch may bring a new domain (mixed-domain accumulation, train_multi.py:563-579);
        # its block still holds an earlier step's gradient -> zero it on first appearance
        now = red.active_domains(dom)
        fresh = [d for d in now if d not in self._active]
        if fresh:
            eng.zero_grad_domains(fresh)
            self._active = red.order(list(self._active) + fresh)

    def _book(self, ws, B: int) -> None:
        """loss bookkeeping of one micro-batch, on the device (train_multi.py:572-577: non-finite losses are not summed)."""
        st = ws["stats"]
        loss = st[0] / st[2]
        ok = torch.isfinite(loss)
        okf = ok.to(torch.float32)
        self.loss_info += torch.stack([torch.where(ok, loss, torch.zeros_like(loss)) * B, okf * B, 1.0 - okf, okf * 0.0])

    def micro_step(self, input_ids, labels, action_ids=None, domain=None, step_domains=None, action_mask=None) -> Dict[str, torch.Tensor]:
        """forward + backward of one micro-batch; gradients accumulate in the flat buffer.

        `step_domains`: every domain that any rank sees in any micro-batch of THIS optimizer step (only read on the first
        micro-batch).  The driver knows it without communication -- all processes iterate the same sampler
        (train_multi.py:928-932) -- and passing it removes the per-step all-gather and its host sync."""
        eng, red = self.engine, self.reducer
        dom = None if action_ids is None else (domain if isinstance(domain, str) else domain[0])
        last = self._micro == self.accum - 1
        self._announce(dom, step_domains)
        B = input_ids.shape[0]
        T = self.model.config.T
        eng.grad_scale.value = 1.0 / (self.accum * red.world)
        eng.gscale.fill_(1.0)
        # the adaLN stacks' backward per gradient bucket (a domain's slice of a bucket is final with it) -- or, with nothing to reduce,
        # once for all layers: four batched launches instead of four per bucket (round 6: -0.3 ms of a 76 ms step)
        segmented = red.collective or getattr(self, "force_segments", False)
        eng.ada_group = self.layers_per_bucket if segmented else eng.cfg.num_layers
        if self.use_graphs and self.accum == 1 and eng.timer is None and eng.device.type == "cuda" and not eng.jpa:
            ws = self._graphed_micro_step(input_ids.reshape(B, T, -1), labels, action_ids, dom)
            self._micro += 1
            return ws
        if eng.jpa and action_ids is not None and action_mask is None:  # the reference draws it inside forward (st_mask_git.py:704-710)
            drop_ratio = torch.rand(B, 1)
            action_mask = (torch.rand(B, T) < drop_ratio).to(eng.device)
        ws = eng.forward(input_ids.reshape(B, T, -1), labels, action_ids, dom, train=True, loss_grad=True, need_logits=False,
                         action_mask=action_mask)
        self._book(ws, B)
        if eng.jpa:  # loss += config.action_loss_weight * action_loss (train_multi.py:574-576)
            eng.act_scale = float(getattr(self.model.config, "action_loss_weight", 0.5)) * eng.grad_scale.value
        if last and red.collective:
            red.begin(self._active)
            eng.backward(eng.grad_scale.value, on_segment=red.on_segment, segment_layers=self.layers_per_bucket)
            red.finish(self._active, extra=[self.loss_info])
        else:
            eng.backward(eng.grad_scale.value)
        self._micro += 1
        return ws

    def _segments(self, pl) -> List[Tuple[int, Optional[int], Optional[str]]]:
        """(start, stop, label) slices of the backward plan: one per gradient bucket when data-parallel."""
        L = self.model.config.num_layers
        if not self.reducer.collective and not getattr(self, "force_segments", False):
            return [(0, None, None)]
        out, start = [], 0
        for l in reversed(range(L)):
            if (L - l) % self.layers_per_bucket == 0 or l == 0:
                stop = pl.marks[f"layer{l}"]
                out.append((start, stop, f"layer{l}"))
                start = stop
        out.append((start, None, "end"))
        return out

    def _graphed_micro_step(self, ids_BTS, labels, action_ids, dom):
        """forward + loss + backward replayed from captured hipGraphs (built on the second use of a (shape, domain)):
        ~1500 launches per step become one graph launch per gradient bucket; the all-reduce of a bucket is issued
        between two graph launches exactly where the eager path issues it."""
        eng, red = self.engine, self.reducer
        eng.bump_dropout()
        B, T, S = ids_BTS.shape
        A_now = eng.cfg.action_token_size if action_ids is not None else 0
        eng._workspace(B, T, S, A_now, True)  # (re)allocates only when the shape changed
        if getattr(self, "_graph_gen", None) != eng.ws_generation:  # captured graphs point into the old buffers
            self._graphs, self._seen = {}, {}
            self._graph_gen = eng.ws_generation
        key = (B, T, S, dom, action_ids is not None)
        graphs = self._graphs.get(key)
        if graphs is None:
            n = self._seen.get(key, 0) + 1
            self._seen[key] = n
            ws = eng.forward(ids_BTS, labels, action_ids, dom, train=True, loss_grad=True, need_logits=False)
            self._book(ws, B)
            if red.collective:
                red.begin(self._active)
                eng.backward(eng.grad_scale.value, on_segment=red.on_segment, segment_layers=self.layers_per_bucket)
                red.finish(self._active, extra=[self.loss_info])
            else:
                eng.backward(eng.grad_scale.value)
            # the first (shape, domain) of a trainer is captured on its SECOND use (buffers, plans and lazily-initialised kernel
            # attributes exist then); every further domain on its first: the kernels are the same, only pointers into the flat
            # buffers differ (with 40 domains: ~40 eager steps until a rank replays graphs only, instead of ~80)
            if n >= (1 if self._graphs else 2):
                torch.cuda.synchronize()
                Bq, Tq, Sq, A, domq = eng._last
                bwd = eng._backward_plan(Bq, Tq, Sq, A, domq)
                graphs = []
                for i, (start, stop, label) in enumerate(self._segments(bwd)):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        stream = torch.cuda.current_stream().cuda_stream
                        if i == 0:
                            fce = eng._use_fused_ce(Bq, Tq, Sq)
                            eng._forward_plan(Bq, Tq, Sq, A, True, domq, readout=not fce).run(stream)
                            eng._ws["stats"].zero_()
                            eng._loss_plan(Bq, Tq, Sq, True, fused=fce, A=A).run(stream)
                            eng.zero_dx_action_rows(eng._ws, Bq * Tq, Sq, A)
                            if A > 0:
                                eng._ws["da_emb"].zero_()
                        bwd.run(stream, start, stop)
                    graphs.append((g, label))
                # capture only records: the gradients of this step are already in G
                self._graphs[key] = graphs
            return ws
        # replay: stage the inputs exactly as STEngine.forward does, then one graph launch per bucket
        A = eng.cfg.action_token_size if action_ids is not None else 0
        ws = eng._workspace(B, T, S, A, True)
        stream = torch.cuda.current_stream().cuda_stream
        eng.refresh_weights(dom if action_ids is not None else None, stream)
        ws["ids"].copy_(ids_BTS, non_blocking=True)
        ws["labels"].copy_(labels.reshape(B, T * S), non_blocking=True)
        if action_ids is not None:
            d_a = eng.d_actions[dom]
            ws["actions"][: B * T * d_a].copy_(action_ids[:, :T].reshape(-1), non_blocking=True)
        eng._last = (B, T, S, A, dom if A > 0 else None)
        if red.collective:
            red.begin(self._active)
        for i, (g, label) in enumerate(graphs):
            g.replay()
            if i == 0:
                self._book(ws, B)  # (the first graph holds forward + loss: `stats` is final behind it)
            if label is not None:
                red.on_segment(label)
        if red.collective:
            red.finish(self._active, extra=[self.loss_info])
        return ws

    def optimizer_step(self) -> None:
        lr = lr_at(self.completed, self.lr, self.warmup)
        self.engine.optimizer_step(lr, self._active, self.betas, self.eps, self.wd, self.max_grad_norm)
        self.completed += 1
        self._micro = 0
        self.last_loss_info.copy_(self.loss_info)

    def reduced_loss(self) -> torch.Tensor:
        """Mean training loss of the last optimizer step over all ranks and micro-batches with a finite loss
        (`accelerator.reduce(loss_info)`, train_multi.py:599-601); a device scalar, read it only when logging."""
        return self.last_loss_info[0] / self.last_loss_info[1]

    def skipped_last_step(self) -> bool:
        """Whether the last optimizer step was skipped on every rank because some rank's loss was not finite (host read)."""
        return bool(self.last_loss_info[2].item() > 0) or not bool(torch.isfinite(self.engine.sqnorm).item())

    def step(self, input_ids, labels, action_ids=None, domain=None, step_domains=None, action_mask=None) -> Dict[str, torch.Tensor]:
        """One full optimizer step on one micro-batch (grad_accum must be 1)."""
        ws = self.micro_step(input_ids, labels, action_ids, domain, step_domains=step_domains, action_mask=action_mask)
        self.optimizer_step()
        return ws

    # ------------------------------------------------------------------ resume (SURVEY (f) row 4)
    STATE_FILE = "trainer_state.safetensors"

    def _range_steps(self) -> Dict[str, int]:
        """updates applied per parameter name (its flat range's device counter)."""
        eng = self.engine
        dense, doms = eng.opt_step, eng.dom_steps
        out = {}
        for name, e in eng.layout.entries.items():
            if e.region == "frozen":
                out[name] = 0
            elif e.region.startswith("dom:"):
                out[name] = doms.get(e.region[4:], 0)
            else:
                out[name] = dense
        return out

    def save_state(self, directory) -> None:
        """Everything needed to resume, in the layout `accelerator.save_state` leaves (hma/train_multi.py:310-321): the model
        (`config.json` + `model.safetensors`), `optimizer.bin` (= torch AdamW's state dict for the reference's two parameter
        groups: per-parameter step / exp_avg / exp_avg_sq for every parameter that has been stepped) and `scheduler.bin`
        (LambdaLR counters) -- the reference's `accelerator.load_state` and `generate.py:80-84` read them as their own.
        `trainer_state.safetensors` keeps the same moments in this trainer's flat order (kept for older checkpoints' readers)."""
        import json
        import os
        from safetensors.torch import save_file
        eng = self.engine
        self.model.save_pretrained(directory)
        tensors = {}
        if eng.M is not None:
            for name, e in eng.layout.entries.items():
                tensors[f"m.{name}"] = eng.M[e.offset:e.offset + e.numel].detach().cpu().clone()
                tensors[f"v.{name}"] = eng.V[e.offset:e.offset + e.numel].detach().cpu().clone()
        meta = {"completed": str(self.completed), "opt_step": str(eng.opt_step), "dom_steps": json.dumps(eng.dom_steps)}
        tensors["_"] = torch.zeros(1)
        save_file(tensors, os.path.join(str(directory), self.STATE_FILE), metadata=meta)
        names = [n for n, _ in self.model.named_parameters()]
        steps = self._range_steps() if eng.M is not None else {n: 0 for n in names}
        shape = lambda n: eng.layout.entries[n].shape
        moments = lambda n: (tensors[f"m.{n}"].view(shape(n)), tensors[f"v.{n}"].view(shape(n)))
        lr_now = lr_at(self.completed, self.lr, self.warmup)
        torch.save(build_optimizer_state_dict(names, moments, lambda n: steps[n], lr_now, self.lr, self.betas, self.eps, self.wd),
                   os.path.join(str(directory), "optimizer.bin"))
        torch.save(build_scheduler_state_dict(self.completed, self.reducer.world, lr_now, self.lr),
                   os.path.join(str(directory), "scheduler.bin"))

    def load_state(self, directory) -> None:
        """Inverse of `save_state` for a Trainer built on a model loaded from the same directory.  Reads the Accelerate-layout
        `optimizer.bin` / `scheduler.bin` when present (also a checkpoint the REFERENCE wrote), else `trainer_state.safetensors`."""
        import json
        import os
        from safetensors import safe_open
        eng = self.engine
        opt_path = os.path.join(str(directory), "optimizer.bin")
        if os.path.exists(opt_path):
            sd = torch.load(opt_path, map_location="cpu", weights_only=True)
            names = [n for n, _ in self.model.named_parameters()]
            g0, g1 = reference_param_groups(names)
            order = g0 + g1
            if [len(g["params"]) for g in sd["param_groups"]] != [len(g0), len(g1)]:
                raise RuntimeError("optimizer.bin does not have the reference's two parameter groups for this model")
            if eng.M is None:
                eng.M, eng.V = torch.zeros_like(eng.P), torch.zeros_like(eng.P)
            dense, doms = 0, {}
            for idx, st in sd["state"].items():
                name = order[int(idx)]
                e = eng.layout.entries[name]
                eng.M[e.offset:e.offset + e.numel].copy_(st["exp_avg"].reshape(-1))
                eng.V[e.offset:e.offset + e.numel].copy_(st["exp_avg_sq"].reshape(-1))
                k = int(float(st["step"]))
                if e.region.startswith("dom:"):
                    doms[e.region[4:]] = max(doms.get(e.region[4:], 0), k)
                elif e.region != "frozen":
                    dense = max(dense, k)
            eng.set_steps(dense, doms)
            sch_path = os.path.join(str(directory), "scheduler.bin")
            if os.path.exists(sch_path):
                sch = torch.load(sch_path, map_location="cpu", weights_only=True)
                self.completed = int(sch["last_epoch"]) // max(self.reducer.world, 1)
            else:
                self.completed = dense
            return
        with safe_open(os.path.join(str(directory), self.STATE_FILE), framework="pt") as f:
            meta = f.metadata()
            names = set(f.keys())
            if any(k.startswith("m.") for k in names):
                if eng.M is None:
                    eng.M, eng.V = torch.zeros_like(eng.P), torch.zeros_like(eng.P)
                for name, e in eng.layout.entries.items():
                    if f"m.{name}" in names:
                        eng.M[e.offset:e.offset + e.numel].copy_(f.get_tensor(f"m.{name}"))
                        eng.V[e.offset:e.offset + e.numel].copy_(f.get_tensor(f"v.{name}"))
        self.completed = int(meta["completed"])
        eng.set_steps(int(meta["opt_step"]), {k: int(v) for k, v in json.loads(meta["dom_steps"]).items()})

    def loss_and_acc(self, ws) -> Tuple[torch.Tensor, torch.Tensor]:
        st = ws["stats"]
        return st[0] / st[2], st[1] / st[2]


class MarTrainer:
    """Data-parallel train step for STMAR (BASELINE configs[3] is an 8-GPU config; the reference wraps the whole model in DDP,
    train_multi.py:779, 990).  Same semantics as `Trainer`: gradients pre-scaled by 1 / (accum * world) and SUM-reduced -- the
    engine's dense range and active domain blocks in buckets, the model's own flat range (input / output stages + diffusion
    head, `STMAR._own_flat`) as one more collective, the loss bookkeeping behind them -- then ONE global-norm clip and fused
    AdamW over both flat ranges, skipped on every rank when any rank's loss was not finite."""

    def __init__(self, model, lr: float = 1e-4, betas=(0.9, 0.95), eps: float = 1e-8, weight_decay: float = 0.05,
                 max_grad_norm: Optional[float] = 1.0, warmup_steps: int = 0, layers_per_bucket: int = 8, grad_accum: int = 1,
                 device=None):
        self.model = model
        dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.engine = model._engine(dev)
        self.own = model._own_flat(dev)
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.max_grad_norm, self.warmup, self.accum = max_grad_norm, warmup_steps, grad_accum
        self.reducer = GradReducer(self.engine.layout, self.engine.G, layers_per_bucket)
        self.layers_per_bucket = layers_per_bucket
        self.engine.ada_group = layers_per_bucket
        self.completed, self._micro = 0, 0
        self._active: List[str] = []
        self._known = False
        self.loss_info = torch.zeros(4, dtype=torch.float32, device=dev)
        self.last_loss_info = torch.zeros(4, dtype=torch.float32, device=dev)
        self.force_overlap = False   # (tests: take the segmented / hooked backward on one rank too)
        self.early_launches = 0      # reductions issued before the backward had finished (head range + trunk buckets)

    def micro_step(self, step_domains=None, **batch):
        """forward + backward of one micro-batch (`batch` = STMAR.forward's keyword arguments)."""
        red = self.reducer
        domain = batch["domain"]
        dom = domain if isinstance(domain, str) else domain[0]
        # Same rule as Trainer._announce: the set of domains that are reduced and stepped is either announced up front (every rank
        # passes the same step_domains; a micro-batch of another domain is an error) or gathered with a collective that EVERY rank
        # enters on EVERY micro-batch -- a rank that skipped it because its own domain was already known would pair the other
        # ranks' all_gather with its own later all_reduce.
        if self._micro == 0:
            self._known = step_domains is not None
            self._active = red.order(step_domains) if self._known else red.active_domains(dom)
            self.model.zero_grad(active_domains=self._active)  # (the dense range + these blocks: a 30-domain G is 4 GB)
            self.loss_info.zero_()
        elif self._known:
            if dom not in self._active:
                raise RuntimeError(f"domain {dom!r} is not in the step_domains announced for this optimizer step")
        else:
            fresh = [d for d in red.active_domains(dom) if d not in self._active]
            if fresh:
                self.engine.zero_grad_domains(fresh)  # (a block that was not zeroed with the first micro-batch's)
                self._active = red.order(list(self._active) + fresh)
        # (the adaLN stacks' backward: per gradient bucket, or once for all layers when nothing is reduced -- see Trainer.micro_step)
        self.engine.ada_group = self.layers_per_bucket if (red.collective or self.force_overlap) else self.engine.cfg.num_layers
        out = self.model(**batch)
        loss = out.loss.detach()
        ok = torch.isfinite(loss)
        okf = ok.to(torch.float32)
        B = batch["input_ids"].shape[0]
        self.loss_info += torch.stack([torch.where(ok, loss, torch.zeros_like(loss)) * B, okf * B, 1.0 - okf, okf * 0.0])
        scaled = out.loss * (1.0 / (self.accum * red.world))
        jpa = getattr(out, "action_loss", None) is not None
        if jpa:  # loss += config.action_loss_weight * action_loss (train_multi.py:574-576)
            scaled = scaled + out.action_loss * (float(self.model.config.action_loss_weight) / (self.accum * red.world))
        self._micro += 1
        if self._micro == self.accum and (red.collective or self.force_overlap):
            # Last micro-batch of the step: gradients become final in backward order -- the diffusion head first, then the trunk
            # bucket by bucket, then the input / output stages -- and every finished range is all-reduced on the side stream
            # while the backward goes on (the reference: DDP's bucketed reduction, train_multi.py:779, 990).
            own, m = self.own, self.model
            ha, hb = m._own_head_range(own)
            red.begin(self._active)

            def after_head():
                m._own_gather_grads(own, prefix="diffloss.")
                red._launch_tensor(own["G"][ha:hb])
                self.early_launches += 1

            def on_segment(label):
                n0 = red._next
                red.on_segment(label)
                self.early_launches += red._next - n0

            m.__dict__["_bwd_hooks"] = dict(after_head=after_head, on_segment=on_segment, segment_layers=self.layers_per_bucket)
            try:
                scaled.backward()
            finally:
                m.__dict__["_bwd_hooks"] = None
            m._own_gather_grads(own, skip="diffloss.")
            extra = [own["G"][:ha], own["G"][hb:]]
            if m.config.jointly_predict_actions:  # the active domains' action heads (their own flat ranges)
                for d_ in self._active:
                    af = m._act_flat(d_, self.engine.device)
                    m._gather(m._act_named(d_), af)
                    extra.append(af["G"])
            red.finish(self._active, extra=extra + [self.loss_info])
        else:
            scaled.backward()
        return out

    def optimizer_step(self) -> None:
        lr = lr_at(self.completed, self.lr, self.warmup)
        self.model.optimizer_step(lr, self._active, self.betas, self.eps, self.wd, self.max_grad_norm)
        self.completed += 1
        self._micro = 0
        self.last_loss_info.copy_(self.loss_info)

    def step(self, step_domains=None, **batch):
        out = self.micro_step(step_domains=step_domains, **batch)
        self.optimizer_step()
        return out

    def reduced_loss(self) -> torch.Tensor:
        return self.last_loss_info[0] / self.last_loss_info[1]

    # ------------------------------------------------------------------ resume: the same Accelerate-layout files as `Trainer`
    def _locate(self, name: str):
        """(moment buffers, offset, numel, shape, updates applied) of a named parameter, or None if it is never stepped."""
        eng, own = self.engine, self.own
        flats = [own]
        if name.startswith("action_diff_losses.") and self.model.config.jointly_predict_actions:
            flats = [self.model._act_flat(name.split(".")[1], eng.device)]
        for fl in flats:
            if name in fl["names"]:
                i = fl["names"].index(name)
                pv = fl["pviews"][i]
                off = (pv.data_ptr() - fl["P"].data_ptr()) // 4
                return fl["M"], fl["V"], off, pv.numel(), tuple(pv.shape), int(fl["steps"][fl["calls"] & 1].item())
        e = eng.layout.entries.get(name)
        if e is None or e.region == "frozen" or eng.M is None:
            return None
        k = eng.dom_steps.get(e.region[4:], 0) if e.region.startswith("dom:") else eng.opt_step
        return eng.M, eng.V, e.offset, e.numel, tuple(e.shape), k

    def save_state(self, directory) -> None:
        """`config.json` + `model.safetensors` + `optimizer.bin` + `scheduler.bin` as `accelerator.save_state` leaves them
        (hma/train_multi.py:310-321): the optimizer file is torch AdamW's state dict for the reference's two parameter groups."""
        import os
        self.model.save_pretrained(directory)
        names = [n for n, _ in self.model.named_parameters()]
        loc = {n: self._locate(n) for n in names}
        steps = lambda n: 0 if loc[n] is None else loc[n][5]

        def moments(n):
            M, V, off, numel, shape, _ = loc[n]
            return M[off:off + numel].detach().cpu().clone().view(shape), V[off:off + numel].detach().cpu().clone().view(shape)

        lr_now = lr_at(self.completed, self.lr, self.warmup)
        torch.save(build_optimizer_state_dict(names, moments, steps, lr_now, self.lr, self.betas, self.eps, self.wd),
                   os.path.join(str(directory), "optimizer.bin"))
        torch.save(build_scheduler_state_dict(self.completed, self.reducer.world, lr_now, self.lr),
                   os.path.join(str(directory), "scheduler.bin"))

    def load_state(self, directory) -> None:
        import os
        eng, own = self.engine, self.own
        sd = torch.load(os.path.join(str(directory), "optimizer.bin"), map_location="cpu", weights_only=True)
        names = [n for n, _ in self.model.named_parameters()]
        order = sum(reference_param_groups(names), [])
        if eng.M is None:
            eng.M, eng.V = torch.zeros_like(eng.P), torch.zeros_like(eng.P)
        dense, doms, own_k = 0, {}, 0
        for idx, st in sd["state"].items():
            name = order[int(idx)]
            k = int(float(st["step"]))
            if name in own["names"]:
                M, V, off, numel, _, _ = self._locate(name)
                own_k = max(own_k, k)
            elif name.startswith("action_diff_losses."):  # jointly_predict_actions: the domain's action head (its own flat range)
                M, V, off, numel, _, _ = self._locate(name)
                af = self.model._act_flat(name.split(".")[1], eng.device)
                af["steps"][af["calls"] & 1] = max(int(af["steps"][af["calls"] & 1].item()), k)
            else:
                e = eng.layout.entries[name]
                M, V, off, numel = eng.M, eng.V, e.offset, e.numel
                if e.region.startswith("dom:"):
                    doms[e.region[4:]] = max(doms.get(e.region[4:], 0), k)
                else:
                    dense = max(dense, k)
            M[off:off + numel].copy_(st["exp_avg"].reshape(-1))
            V[off:off + numel].copy_(st["exp_avg_sq"].reshape(-1))
        eng.set_steps(dense, doms)
        own["steps"][own["calls"] & 1] = own_k
        sch_path = os.path.join(str(directory), "scheduler.bin")
        self.completed = (int(torch.load(sch_path, map_location="cpu", weights_only=True)["last_epoch"]) // max(self.reducer.world, 1)
                          if os.path.exists(sch_path) else dense)


class FusedAdamW:
    """torch.optim-shaped facade over the engine's fused clip + AdamW for the autograd (drop-in) path:
    `loss.backward(); opt.step(); opt.zero_grad()`."""

    def __init__(self, model, lr: float = 1e-4, betas=(0.9, 0.95), eps: float = 1e-8, weight_decay: float = 0.05,
                 max_grad_norm: Optional[float] = 1.0):
        self.model, self.lr, self.betas, self.eps, self.wd, self.max_grad_norm = model, lr, betas, eps, weight_decay, max_grad_norm

    def step(self) -> None:
        eng = self.model._engine
        if eng is None:
            raise RuntimeError("no backward has run yet")
        active = [d for d in eng.domains if d in self.model.touched_domains]
        eng.optimizer_step(self.lr, active, self.betas, self.eps, self.wd, self.max_grad_norm)

    def zero_grad(self, set_to_none: bool = True) -> None:
        self.model.zero_grad()
