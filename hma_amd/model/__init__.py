"""Drop-in mirror of the reference's `hma.model` module API."""
from .st_mask_git import STMaskGIT, FixedMuReadout, ModulateLayer, BasicMLP, ActionStat, cosine_schedule  # noqa: F401
from .st_transformer import STTransformerDecoder, STBlock, Mlp  # noqa: F401
from .attention import SelfAttention, BasicSelfAttention  # noqa: F401
from .factorization_utils import (FactorizedEmbedding, factorize_token_ids, unfactorize_token_ids,  # noqa: F401
                                  factorize_labels, nth_root)
