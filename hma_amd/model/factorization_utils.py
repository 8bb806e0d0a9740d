"""Factorised-vocabulary helpers: same names / semantics as hma/model/factorization_utils.py.

The integer helpers are host-side glue for collators and tests (bit-exact by construction); the hot
path never calls them -- the embedding, CE and MaskGIT kernels factorise ids in registers."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _lib
from ..ops import ptr, stream_ptr


def nth_root(x: int, n: int) -> int:
    root = round(x ** (1 / n))
    assert root ** n == x, (x, n, root)
    return root


def factorize_token_ids(token_ids: torch.Tensor, num_factored_vocabs: int = 2, factored_vocab_size: int = 512) -> torch.Tensor:
    """(...,) ids in [0, V**n) -> (..., n) with factor i = (id // V**i) % V  (factorization_utils.py:57-68)."""
    parts = [(token_ids // (factored_vocab_size ** i)) % factored_vocab_size for i in range(num_factored_vocabs)]
    return torch.stack(parts, dim=-1)


def unfactorize_token_ids(factored: torch.Tensor, num_factored_vocabs: int = 2, factored_vocab_size: int = 512) -> torch.Tensor:
    """Inverse of `factorize_token_ids` (factorization_utils.py:71-82)."""
    out = torch.zeros_like(factored[..., 0])
    for i in range(num_factored_vocabs):
        out = out + factored[..., i] * (factored_vocab_size ** i)
    return out


def factorize_labels(labels_THW: torch.Tensor, num_factored_vocabs: int = 2, factored_vocab_size: int = 512) -> torch.Tensor:
    """(B,T,H,W) -> (B, n, T, H, W)  (factorization_utils.py:85-96)."""
    return factorize_token_ids(labels_THW, num_factored_vocabs, factored_vocab_size).permute(0, 4, 1, 2, 3).contiguous()


class FactorizedEmbedding(nn.Module):
    """Sum of per-factor embeddings with a learned mask-token row (factorization_utils.py:6-54)."""

    def __init__(self, factored_vocab_size: int, num_factored_vocabs: int, d_model: int, mask_token_id: int):
        super().__init__()
        self.factored_vocab_size = factored_vocab_size
        self.num_factored_vocabs = num_factored_vocabs
        self.d_model = d_model
        self.mask_token_id = mask_token_id
        self.factored_embeds = nn.ModuleList([nn.Embedding(factored_vocab_size, d_model) for _ in range(num_factored_vocabs)])
        self.mask_token_embed = nn.Parameter(torch.zeros(1, d_model))

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor) -> torch.Tensor:
        """ids (B, T, S) -> (B, T, S, d) through the fused embedding kernel (zero positional table, no actions)."""
        assert self.num_factored_vocabs == 2 and self.d_model == 256
        B, T, S = input_ids.shape
        ids = input_ids.contiguous()
        x = torch.empty(B, T, S, self.d_model, dtype=torch.float32, device=ids.device)
        pos = torch.zeros(T, S, self.d_model, dtype=torch.float32, device=ids.device)
        _lib.call("hma_embed_fwd", stream_ptr(), ptr(ids), ptr(self.factored_embeds[0].weight), ptr(self.factored_embeds[1].weight),
                  ptr(self.mask_token_embed), ptr(pos), None, ptr(x), B, T, S, 0, S, self.factored_vocab_size, self.mask_token_id)
        return x
