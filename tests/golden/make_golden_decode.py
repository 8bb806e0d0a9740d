#!/usr/bin/env python3
"""G13: per-step MaskGIT decode records from the REAL reference (runs only in the build container).

    python tests/golden/make_golden_decode.py        # writes tests/golden/g13_decode_steps.safetensors

G7 stores final ids only, which lets a test assert an agreement percentage but not SURVEY.md section 7's rule ("an id may
differ only where the reference's top-2 margin is below the logits tolerance").  This fixture records, for every MaskGIT step
of `STMaskGIT.maskgit_generate` (hma/model/st_mask_git.py:338-467) on the tiny model, what is needed to check each step in
isolation (teacher forcing: the test feeds OUR model the reference's state before the step):

  <run>.frame_in[k]   ids of frame out_t before step k                 (B, 256) int32
  <run>.frame_out[k]  ids of frame out_t after step k                  (B, 256) int32
  <run>.top1[k]       per-factor arg-max of the reference logits       (B, 2, 256) int16   (factor v of channel v*512 + k)
  <run>.margin[k]     per-factor top-1 minus top-2 logit               (B, 2, 256) f32
  <run>.conf[k]       the tensor handed to torch.argsort (:442-443)    (B, 256) f32       (+inf on unmasked; absent on last step)
  <run>.logits_sub[k] reference logits, every 32nd channel             (B, 32, 256) f32

runs: greedy1 / greedy2 / greedy8 (temperature 0, unmask_mode="greedy"), random4 (unmask_mode="random"; `draws` = the
torch.rand_like tensors), sampled3 (temperature 1.0 -- the Categorical branch, :411-416 -- with unmask_mode="greedy").

Categorical branch: `Categorical(probs).sample()` is `torch.multinomial(probs_2d, 1, True)`, whose single-sample path draws
q = empty_like(p).exponential_(1) and returns argmax(p / q) (ATen native/Sampling.cpp).  The generator wraps
torch.multinomial, re-derives the sample from a q drawn at the same RNG state, ASSERTS it equals what torch returned and
that the RNG ends in the same state, and records for every (step, factor) call
  sampled3.sample[k]  the drawn factor ids                             (B, 2, 256) int16
  sampled3.smargin[k] log-domain margin of the winner of p / q         (B, 2, 256) f32   ((x - log q) top-1 minus top-2)
q itself (1 MB per step) is not stored: the test re-draws it with `sampled3.seed` in the recorded call order, and
`sampled3.qsum[k]` (float64 sums) pins the stream.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden as MG  # noqa: E402  (installs the stubs, imports the reference)
from tests.golden.golden_cfg import tiny_inputs  # noqa: E402

SAMPLED_SEED = 2024


def record_run(model, cfg, prompt0, out_t, steps, temperature, unmask_mode, inp, seed=None):
    """Runs the reference once with compute_logits / argsort / rand_like / multinomial wrapped to record per-step state."""
    rec = {"frame_in": [], "logits": [], "conf": [], "draws": [], "sample": [], "q": []}
    orig_cl = model.compute_logits
    orig_argsort, orig_rand_like, orig_multinomial = torch.argsort, torch.rand_like, torch.multinomial

    def cl(prompt_THW, *a, **k):
        rec["frame_in"].append(prompt_THW[:, out_t].clone())
        out = orig_cl(prompt_THW, *a, **k)
        rec["logits"].append(out[0][:, :, out_t].detach().clone())  # (B, 1024, H, W)
        return out

    def argsort(t, *a, **k):
        rec["conf"].append(t.detach().clone())
        return orig_argsort(t, *a, **k)

    def rand_like(t, *a, **k):
        r = orig_rand_like(t, *a, **k)
        rec["draws"].append(r.clone())
        return r

    def multinomial(p, n, replacement=False, **k):
        assert n == 1 and not k
        state = torch.get_rng_state()
        real = orig_multinomial(p, n, replacement)
        after = torch.get_rng_state()
        torch.set_rng_state(state)
        q = torch.empty_like(p).exponential_(1)
        assert torch.equal(torch.get_rng_state(), after), "torch.multinomial consumed a different random stream"
        mine = (p / q).argmax(dim=-1, keepdim=True)
        assert torch.equal(mine, real), "torch.multinomial is not argmax(p / q) here"
        rec["sample"].append(real.clone())
        rec["q"].append(q)
        return real

    model.compute_logits = cl
    torch.argsort, torch.rand_like, torch.multinomial = argsort, rand_like, multinomial
    try:
        if seed is not None:
            torch.manual_seed(seed)
        p = prompt0.clone()
        s, _, _ = model.maskgit_generate(p, out_t=out_t, maskgit_steps=steps, temperature=temperature, unmask_mode=unmask_mode,
                                         action_ids=inp["actions_domA"], domain=["domA"] * prompt0.shape[0])
    finally:
        torch.argsort, torch.rand_like, torch.multinomial = orig_argsort, orig_rand_like, orig_multinomial
        del model.compute_logits
    assert len(rec["logits"]) == steps and len(rec["conf"]) == steps - 1
    rec["final"] = s.clone()
    return rec


def pack(tag, rec, out, B):
    steps = len(rec["logits"])
    frames = rec["frame_in"] + [rec["final"]]
    out[f"{tag}.frame_in"] = torch.stack([f.reshape(B, 256) for f in frames[:-1]]).to(torch.int32)
    out[f"{tag}.frame_out"] = torch.stack([f.reshape(B, 256) for f in frames[1:]]).to(torch.int32)
    lg = torch.stack([l.reshape(B, 2, 512, 256) for l in rec["logits"]])          # (steps, B, factor, vocab, token)
    top2 = lg.topk(2, dim=3)
    out[f"{tag}.top1"] = top2.indices[:, :, :, 0].to(torch.int16)
    out[f"{tag}.margin"] = (top2.values[:, :, :, 0] - top2.values[:, :, :, 1]).float()
    out[f"{tag}.logits_sub"] = lg.reshape(steps, B, 1024, 256)[:, :, ::32].contiguous()
    if rec["conf"]:
        out[f"{tag}.conf"] = torch.stack([c.reshape(B, 256) for c in rec["conf"]]).float()
    if rec["draws"]:
        out[f"{tag}.draws"] = torch.stack(rec["draws"])
    if rec["sample"]:
        # calls come in (step, factor 1, factor 0) order (:408 flips the factor axis); rows are (b, h, w)
        smp = torch.stack(rec["sample"]).reshape(steps, 2, B, 256).flip(1).permute(0, 2, 1, 3)     # (steps, B, factor, token)
        q = torch.stack(rec["q"]).reshape(steps, 2, B, 256, 512).flip(1).permute(0, 2, 1, 3, 4)      # (steps, B, factor, token, vocab)
        z = lg.permute(0, 1, 2, 4, 3) - q.log()                                                       # argmax_k p_k / q_k = argmax_k x_k - log q_k
        t2 = z.topk(2, dim=-1)
        assert torch.equal(t2.indices[..., 0], smp), "log-domain restatement of the draw disagrees"
        out[f"{tag}.sample"] = smp.to(torch.int16)
        out[f"{tag}.smargin"] = (t2.values[..., 0] - t2.values[..., 1]).float()
        out[f"{tag}.qsum"] = torch.stack([x.double().sum() for x in rec["q"]])
        out[f"{tag}.seed"] = torch.tensor(SAMPLED_SEED)


def main():
    cfg, model, _ = MG.build_tiny()
    model.eval()
    inp = tiny_inputs()
    B = 2
    out_t = cfg.T - 1
    prompt0 = inp["labels"].reshape(B, cfg.T, 16, 16).clone()
    prompt0[:, out_t:] = cfg.image_vocab_size
    out = {"prompt0": prompt0}
    with torch.no_grad():
        for steps in (1, 2, 8):
            pack(f"greedy{steps}", record_run(model, cfg, prompt0, out_t, steps, 0.0, "greedy", inp), out, B)
        pack("random4", record_run(model, cfg, prompt0, out_t, 4, 0.0, "random", inp, seed=123), out, B)
        pack("sampled3", record_run(model, cfg, prompt0, out_t, 3, 1.0, "greedy", inp, seed=SAMPLED_SEED), out, B)
    MG.save("g13_decode_steps", out)


if __name__ == "__main__":
    main()
