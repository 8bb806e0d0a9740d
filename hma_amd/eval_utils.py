"""Evaluator loss (SURVEY (f) row 4): `compute_loss` of hma/eval_utils.py:60-93 on the device.

Plain cross-entropy (no label smoothing) of the factorised logits of frames 1..T-1, summed over the two factors and
averaged over every token -- an independent parity metric on generated logits.  Runs `hma_ce_fwd_bwd` (label smoothing
0, every token of frames >= 1 counted) instead of materialising log-softmax tensors.
"""
from __future__ import annotations

import torch

from . import _lib


def compute_loss(labels_flat: torch.LongTensor, factored_logits: torch.Tensor, num_factored_vocabs: int = 2,
                 factored_vocab_size: int = 512) -> float:
    """labels_flat (B, T*H*W); factored_logits (B, 512, 2, T-1, H, W) as returned by the model (any strides)."""
    assert factored_logits.dim() == 6 and tuple(factored_logits.shape[:3]) == (labels_flat.size(0), factored_vocab_size,
                                                                             num_factored_vocabs), \
        f"Shape of `logits` should be (B, {factored_vocab_size}, {num_factored_vocabs}, T-1, H, W)"
    if (num_factored_vocabs, factored_vocab_size) != (2, 512):
        raise NotImplementedError("the loss kernel is built for the 2 x 512 factorisation")
    B = labels_flat.size(0)
    t = factored_logits.size(3) + 1
    h, w = factored_logits.shape[-2:]
    assert t * h * w == labels_flat.size(1), "Shape of `factored_logits` does not match flattened latent image size."
    dev = factored_logits.device
    if dev.type != "cuda":
        raise RuntimeError("compute_loss runs its kernel on the GPU")
    S = h * w
    # kernel layout: rows (b, t, s) over ALL T frames, channel = factor * 512 + k; frame 0 is never read
    logits = torch.zeros(B, t, S, 2 * 512, dtype=torch.float32, device=dev)
    logits[:, 1:] = factored_logits.permute(0, 3, 4, 5, 2, 1).reshape(B, t - 1, S, 1024).float()
    labels = labels_flat.to(dev).reshape(B, t, S).contiguous()
    mask_id = 1 << 40  # an id no label has: every token of frames >= 1 is counted
    ids = torch.full_like(labels, mask_id)
    stats = torch.zeros(8, dtype=torch.float32, device=dev)  # HMA_CE_STATS_FLOATS
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.call("hma_count_masked", stream, ids.data_ptr(), stats.data_ptr(), B, t, S, mask_id)
    _lib.call("hma_ce_fwd_bwd", stream, logits.data_ptr(), ids.data_ptr(), labels.data_ptr(), stats.data_ptr(), None, None, 1.0,
              B, t, S, mask_id, 0.0)
    return (stats[0] / stats[2]).item()
