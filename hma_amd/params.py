"""Flat parameter store: every STMaskGIT tensor is a view into ONE fp32 buffer.

Why: the reference keeps ~1500 separate tensors (362 M parameters, ~90 % of them per-domain heads
that are idle in any given step).  A single flat buffer laid out in
backward-completion order gives (i) one fused clip+AdamW launch per active range instead of one
per tensor, (ii) contiguous gradient buckets for the RCCL all-reduce that become final in address
order as backward walks layers L-1 .. 0, (iii) per-layer weight stacks with a constant stride so
the adaLN GEMMs of all layers run as one batched launch.  Names, shapes and dtypes of the views are
exactly the reference state-dict (hma/model/st_mask_git.py:152-251).

Flat order:  head (out_x_proj) | trunk layers L-1 .. 0 | tail (embeddings, pos) | one block per
action domain (layer-major: a layer's six modulation tensors together, constant layer stride; then the action stem) |
never-trained tensors.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

ALIGN = 64  # elements; also the granularity of the AdamW decay flags


def _align(n: int) -> int:
    return (n + ALIGN - 1) // ALIGN * ALIGN


@dataclass
class Entry:
    name: str
    shape: Tuple[int, ...]
    offset: int
    decay: bool
    region: str

    @property
    def numel(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n


class ParamLayout:
    """Pure-Python description of the flat buffer (no device memory): usable on CPU for tests."""

    def __init__(self, cfg, domains: Sequence[str] = (), d_actions: Sequence[int] = (), action_dims: Sequence[int] = ()):
        self.cfg = cfg
        self.domains = list(domains)
        self.d_actions = list(d_actions)
        d, L = cfg.d_model, cfg.num_layers
        hid = int(d * cfg.mlp_ratio)
        C = cfg.factored_vocab_size * cfg.num_factored_vocabs
        self.entries: "OrderedDict[str, Entry]" = OrderedDict()
        self.regions: "OrderedDict[str, Tuple[int, int]]" = OrderedDict()
        self._cur = 0

        def add(name, shape, region):
            e = Entry(name, tuple(shape), self._cur, "bias" not in name, region)  # train_multi.py:907-918
            self.entries[name] = e
            self._cur += _align(e.numel)
            return e

        def begin(region):
            self._cur = _align(self._cur)
            self._rstart = self._cur
            self._rname = region

        def end():
            self.regions[self._rname] = (self._rstart, self._cur)

        begin("head")
        add("out_x_proj.weight", (C, d), "head")
        add("out_x_proj.bias", (C,), "head")
        end()

        qkn = bool(getattr(cfg, "qk_norm", False))  # norm1 / norm2 are nn.Identity; each attention owns a LayerNorm(head_dim) for q, k

        def layer_tensors():
            t = [] if qkn else [("norm1.weight", (d,))]
            for a in ("spatial_attn", "temporal_attn"):
                t.append((f"{a}.qkv.weight", (3 * d, d)))
                t.append((f"{a}.proj.weight", (d, d)))
                if qkn:
                    t.append((f"{a}.norm.weight", (d // cfg.num_heads,)))
            t += ([] if qkn else [("norm2.weight", (d,))]) + [("mlp.fc1.weight", (hid, d)), ("mlp.fc2.weight", (d, hid))]
            if not qkn:
                t.append(("norm1.bias", (d,)))
            for a in ("spatial_attn", "temporal_attn"):
                if cfg.qkv_bias:
                    t.append((f"{a}.qkv.bias", (3 * d,)))
                if cfg.proj_bias:
                    t.append((f"{a}.proj.bias", (d,)))
                if qkn:
                    t.append((f"{a}.norm.bias", (d // cfg.num_heads,)))
            if not qkn:
                t.append(("norm2.bias", (d,)))
            if cfg.mlp_bias:
                t += [("mlp.fc1.bias", (hid,)), ("mlp.fc2.bias", (d,))]
            return t

        self.layer_suffixes = [s for s, _ in layer_tensors()]
        self.layer_start: Dict[int, int] = {}
        for l in reversed(range(L)):  # backward finishes layer L-1 first
            begin(f"layer{l}")
            self.layer_start[l] = self._cur
            for suffix, shape in layer_tensors():
                add(f"decoder.layers.{l}.{suffix}", shape, f"layer{l}")
            end()
        self.layer_stride = (self.layer_start[0] - self.layer_start[1]) if L > 1 else 0

        begin("tail")
        add("token_embed.mask_token_embed", (1, d), "tail")
        for i in range(cfg.num_factored_vocabs):
            add(f"token_embed.factored_embeds.{i}.weight", (cfg.factored_vocab_size, d), "tail")
        add("pos_embed_TSC", (1, cfg.T, cfg.S + cfg.action_token_size, d), "tail")
        jpa = bool(getattr(cfg, "jointly_predict_actions", False))
        if jpa:  # trainable with action prediction (st_mask_git.py:656-660): lives with the embeddings
            add("action_mask_tokens", (1, cfg.T, 1, d), "tail")
        end()

        # A domain's block is LAYER-major: the six modulation tensors of layer 0, of layer 1, ... at a constant stride (the 32 layers'
        # adaLN GEMMs still run as batched launches -- `dom_layer_stride` is their batch stride), then the action stem.  So the part
        # of a domain's block that belongs to a gradient bucket (a run of consecutive layers) is ONE contiguous slice, and it is final
        # when that bucket's backward is: it rides with the bucket's all-reduce instead of waiting for the end of the step.
        modulate = "modulate" in cfg.action_network
        self.dom_layer_stride = 0
        self.dom_layer_start: Dict[str, int] = {}
        for dom, da in zip(self.domains, self.d_actions):
            begin(f"dom:{dom}")
            self.dom_layer_start[dom] = self._cur
            if modulate:
                for l in range(L):
                    l0 = self._cur
                    for key, shape in (("adaLN_modulation.0.weight", (d, d)), ("adaLN_modulation.2.weight", (2 * d, d)),
                                       ("linear_out.weight", (d, d)), ("adaLN_modulation.0.bias", (d,)),
                                       ("adaLN_modulation.2.bias", (2 * d,)), ("linear_out.bias", (d,))):
                        add(f"decoder.layers.{l}.action_projectors.{dom}.{key}", shape, f"dom:{dom}")
                    self.dom_layer_stride = self._cur - l0
            add(f"action_mlp.{dom}.model.0.weight", (d, da), f"dom:{dom}")
            add(f"action_mlp.{dom}.model.1.weight", (d,), f"dom:{dom}")
            add(f"action_mlp.{dom}.model.3.weight", (d, d), f"dom:{dom}")
            add(f"action_mlp.{dom}.model.0.bias", (d,), f"dom:{dom}")
            add(f"action_mlp.{dom}.model.1.bias", (d,), f"dom:{dom}")
            add(f"action_mlp.{dom}.model.3.bias", (d,), f"dom:{dom}")
            if jpa:  # the domain's action read-out (st_mask_git.py:676-678)
                add(f"action_out_projectors.{dom}.weight", (da, d), f"dom:{dom}")
                add(f"action_out_projectors.{dom}.bias", (da,), f"dom:{dom}")
            end()

        begin("frozen")  # never receive a gradient with jointly_predict_actions=False
        if not jpa:
            add("action_mask_tokens", (1, cfg.T, 1, d), "frozen")
            for dom, da in zip(self.domains, self.d_actions):
                add(f"action_out_projectors.{dom}.weight", (da, d), "frozen")
                add(f"action_out_projectors.{dom}.bias", (da,), "frozen")
        end()
        self.total = _align(self._cur)

    # ------------------------------------------------------------------ queries
    def off(self, name: str) -> int:
        return self.entries[name].offset

    def trainable_ranges(self, active_domains: Sequence[str]) -> List[Tuple[int, int]]:
        """Contiguous [start, end) ranges that receive gradients this step, in flat order."""
        out = []
        dense_start = self.regions["head"][0]
        dense_end = self.regions["tail"][1]
        out.append((dense_start, dense_end))
        for dom in self.domains:
            if dom in active_domains:
                out.append(self.regions[f"dom:{dom}"])
        return out

    def buckets(self, layers_per_bucket: int) -> List[Tuple[int, int]]:
        """Dense gradient buckets in backward-completion order: head+first layers, ..., last layers."""
        L = self.cfg.num_layers
        out = []
        start = self.regions["head"][0]
        order = list(reversed(range(L)))
        for i in range(0, L, layers_per_bucket):
            last = order[min(i + layers_per_bucket, L) - 1]
            end = self.regions[f"layer{last}"][1]
            out.append((start, end))
            start = end
        return out

    def dom_buckets(self, dom: str, layers_per_bucket: int) -> List[Tuple[int, int]]:
        """The slices of a domain's block that become final with each dense bucket (same order and count as `buckets`), then the
        rest of the block (the action stem, final at the end of the backward) as one more slice."""
        L = self.cfg.num_layers
        a0, a1 = self.regions[f"dom:{dom}"]
        if self.dom_layer_stride == 0:
            return [(a0, a0)] * ((L + layers_per_bucket - 1) // layers_per_bucket) + [(a0, a1)]
        base, st = self.dom_layer_start[dom], self.dom_layer_stride
        out = []
        order = list(reversed(range(L)))
        for i in range(0, L, layers_per_bucket):
            hi, lo = order[i], order[min(i + layers_per_bucket, L) - 1]
            out.append((base + lo * st, base + (hi + 1) * st))
        out.append((base + L * st, a1))
        return out

    def decay_flags(self) -> torch.Tensor:
        """uint8 per 64-element block: 2 = decay, 1 = no decay, 0 = frozen/padding."""
        flags = torch.zeros(self.total // ALIGN, dtype=torch.uint8)
        for e in self.entries.values():
            if e.region == "frozen":
                continue
            b0, b1 = e.offset // ALIGN, (e.offset + _align(e.numel)) // ALIGN
            flags[b0:b1] = 2 if e.decay else 1
        return flags
