"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the MaskGIT collator's per-token arithmetic.

Follows get_maskgit_collator.collate_fn, /root/reference/hma/data.py:28-98, with every random draw passed in (the
shapes and order documented in hma_amd/data.py).  Pinned by tests/golden/g10_data.safetensors: draws, inputs and
outputs captured from the real reference collator (tests/golden/make_golden_data.py).  Only tests may import this.
"""
import torch


def collate_with_draws(ids_BTHW, V, mask_id, r_corrupt=None, corrupt_thresh=0.0, random_values=None, r_nonmlm=None,
                       correct_rate=None, first_masked_frame=1, mask_prob=None, r_mask=None, num_factored=0):
    """ids (B, T, H, W) int64 -> input ids after corruption / non-MLM corruption / masking."""
    x = ids_BTHW.clone()
    nf = 1 if (r_corrupt is None and r_nonmlm is None and random_values is None) else (
        random_values if random_values is not None else r_nonmlm).shape[-1]
    nf = num_factored or nf
    c = torch.stack([(x // V ** k) % V for k in range(nf)], dim=-1)      # data.py:39 (factorize_token_ids)
    if r_corrupt is not None:                                            # :42-49
        m = r_corrupt < corrupt_thresh
        c[m] = random_values[m]
    fmf = first_masked_frame
    if r_nonmlm is not None:                                             # :51-64
        for i in range(x.shape[1] - fmf):
            m = r_nonmlm[:, i] > correct_rate[i]
            c[:, fmf + i][m] = random_values[:, fmf + i][m]
    if mask_prob is None:                                                # no masking: the original ids come back
        return x
    out = sum(c[..., k] * V ** k for k in range(nf))                     # :79 (unfactorize_token_ids)
    m = r_mask < mask_prob[:, :, None, None]                             # :74-76
    out[:, fmf:][m] = mask_id                                            # :80
    return out
