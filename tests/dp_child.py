"""One rank of the 2-process data-parallel equivalence test (tests/test_dp_gpu.py): a fresh interpreter per rank, both on GPU 0,
gloo between them (the GPU box has ONE device; the same GradReducer code runs over RCCL on a multi-GPU node).

    RANK=r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/dp_child.py OUT.safetensors [nan_step]

Runs STEPS optimizer steps of `Trainer.step` with rank-specific domains and batches (rank 0: domA, rank 1: domB; domC idle
everywhere) and rank 0 writes the resulting weights, Adam update counts and the reduced per-step losses."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STEPS = 4
L = 4
DOMAINS, D_ACTIONS = ["domA", "domB", "domC"], [7, 14, 7]
CFG = dict(num_layers=L, num_heads=8, d_model=256, T=3, S=256, image_vocab_size=262144, use_mup=True,
           action_network="concat+modulate", num_factored_vocabs=2, qkv_bias=False, proj_bias=True, attn_drop=0.0, qk_norm=False,
           mlp_ratio=4.0, mlp_drop=0.0, mlp_bias=True)
STATS = [[[0.0] * 7, [1.0] * 7]] * 3


def build_model():
    from hma_amd.config import GenieConfig
    from hma_amd.model import STMaskGIT

    torch.manual_seed(1234)
    m = STMaskGIT(GenieConfig(**CFG))
    m.init_action_projectors(DOMAINS, D_ACTIONS, STATS, CFG["action_network"])
    with torch.no_grad():
        for p in m.parameters():  # (no exact zeros: an indexing error in a zero tensor would hide)
            if float(p.abs().max()) == 0.0:
                p.normal_(0, 0.02)
    return m.to("cuda").train()


def batch(which: int, step: int):
    """micro-batch `which` (0: domA, 1: domB) of optimizer step `step`."""
    import math
    g = torch.Generator().manual_seed(100 + 10 * step + which)
    B, T = 2, CFG["T"]
    labels = torch.randint(0, 8192, (B, T, 256), generator=g)
    m = torch.rand(B, T - 1, 256, generator=g) < math.cos(0.3 * math.pi / 2)
    ids = labels.clone()
    ids[:, 1:][m] = CFG["image_vocab_size"]
    act = torch.randn(B, T, D_ACTIONS[which], generator=g)
    return ids.reshape(B, -1).cuda(), labels.reshape(B, -1).cuda(), act.cuda(), [DOMAINS[which]] * B


def digest(model, trainer, losses):
    out = {n: p.detach().float().cpu().clone() for n, p in model.named_parameters()
           if n.startswith(("decoder.layers.0.", f"decoder.layers.{L - 1}.", "out_x_proj", "token_embed", "action_mlp"))}
    eng = trainer.engine
    out["_opt_step"] = torch.tensor(eng.opt_step)
    out["_dom_steps"] = torch.tensor([eng.dom_steps.get(d, 0) for d in DOMAINS])
    out["_losses"] = torch.stack(losses).cpu()
    return out


# ---------------------------------------------------------------------------------------------------- STMAR (configs[3])
def build_mar():
    from hma_amd.config import DiffusionGenieConfig
    from hma_amd.model.st_mar import STMAR
    from tests.golden.stmar_cfg import CFG as MCFG, DOMAINS as MD, D_ACTIONS as MDA, STATS as MST, seeded_state

    m = STMAR(DiffusionGenieConfig(**MCFG))
    m.init_action_projectors(MD, MDA, MST, MCFG["action_network"])
    m.load_state_dict(seeded_state(m.state_dict()))
    return m.to("cuda").train()


def mar_batch(which: int, step: int):
    from tests.golden.stmar_cfg import inputs, D_ACTIONS as MDA, DOMAINS as MD
    inp = inputs(seed=50 + 10 * step + which)
    g = torch.Generator().manual_seed(7 + which)
    act = torch.randn(2, 3, MDA[which], generator=g)
    return dict(input_ids=inp["latents"].cuda(), labels=inp["latents"].cuda(), action_ids=act.cuda(), domain=[MD[which]] * 2,
                masked_tokens_indicator=inp["masked"].cuda(), h=[32, 32], w=[32, 32], diffusion_t=inp["t"].cuda(),
                diffusion_noise=inp["noise"].cuda())


def mar_digest(model, losses):
    keep = ("token_embed.weight", "mask_token", "decoder_norm.weight", "out_x_proj.weight", "out_x_proj.bias", "pos_embed_TSC",
            "diffloss.net.cond_embed.weight", "diffloss.net.res_blocks.0.mlp.0.weight", "diffloss.net.final_layer.linear.bias",
            "decoder.layers.0.mlp.fc1.weight", "decoder.layers.1.spatial_attn.qkv.bias", "action_mlp.domA.model.0.weight",
            "action_mlp.domB.model.0.weight", "decoder.layers.0.action_projectors.domB.linear_out.weight")
    out = {n: p.detach().float().cpu().clone() for n, p in model.named_parameters() if n in keep}
    out["_losses"] = torch.stack(losses).cpu()
    return out


# mixed domains under gradient accumulation WITHOUT step_domains: micro-batch k of rank r
MIXED = {0: [0, 0], 1: [0, 1]}  # rank 0: domA, domA; rank 1: domA, domB (a domain that is new on ONE rank in the second micro-batch)


def main_mar_mixed(out_path):
    import torch.distributed as dist
    from safetensors.torch import save_file
    from hma_amd.train import MarTrainer

    rank = int(os.environ["RANK"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    model = build_mar()
    tr = MarTrainer(model, lr=1e-3, warmup_steps=0, grad_accum=2)
    losses = []
    for step in range(2):
        for k in range(2):
            tr.micro_step(**mar_batch(MIXED[rank][k], 4 * step + 2 * k + rank))  # (every rank enters the domain gather every micro-batch)
        assert tr._active == ["domA", "domB"], tr._active
        tr.optimizer_step()
        losses.append(tr.reduced_loss().detach().clone())
    torch.cuda.synchronize()
    if rank == 0:
        save_file(mar_digest(model, losses), out_path)
    dist.barrier()
    dist.destroy_process_group()


def main_mar(out_path):
    import torch.distributed as dist
    from safetensors.torch import save_file
    from hma_amd.train import MarTrainer

    rank = int(os.environ["RANK"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    model = build_mar()
    tr = MarTrainer(model, lr=1e-3, warmup_steps=0, layers_per_bucket=1)
    losses = []
    for step in range(3):
        tr.step(step_domains=["domA", "domB"], **mar_batch(rank, step))
        losses.append(tr.reduced_loss().detach().clone())
    torch.cuda.synchronize()
    # the bucketed path ran: per step the diffusion head's range and the two one-layer trunk buckets were all-reduced from inside
    # the backward (before the input stage's gradients existed), the rest in finish()
    assert tr.early_launches == 3 * 3, tr.early_launches
    if rank == 0:
        save_file(mar_digest(model, losses), out_path)
    dist.barrier()
    dist.destroy_process_group()


def main():
    if len(sys.argv) > 3 and sys.argv[3] == "mar":
        return main_mar(sys.argv[1])
    if len(sys.argv) > 3 and sys.argv[3] == "mar_mixed":
        return main_mar_mixed(sys.argv[1])
    import torch.distributed as dist
    from safetensors.torch import save_file
    from hma_amd.train import Trainer

    out_path = sys.argv[1]
    nan_step = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    rank = int(os.environ["RANK"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    model = build_model()
    tr = Trainer(model, lr=1e-3, warmup_steps=0, layers_per_bucket=2)
    assert tr.reducer.world == 2
    losses = []
    for step in range(STEPS):
        ids, labels, act, dom = batch(rank, step)
        if step == nan_step and rank == 1:
            act = act.clone()
            act[0, 0, 0] = float("nan")  # a non-finite loss on ONE rank: every rank must skip this update
        tr.step(ids, labels, act, dom, step_domains=["domA", "domB"])
        losses.append(tr.reduced_loss().detach().clone())
    torch.cuda.synchronize()
    assert len(tr._graphs) == 1 and len(next(iter(tr._graphs.values()))) == L // 2 + 1  # per-bucket graph segments were used
    if rank == 0:
        save_file(digest(model, tr, losses), out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
