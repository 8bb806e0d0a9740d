// MaskGIT training collator on the device (SURVEY (f) row 1): the per-batch corruption / non-MLM corruption /
// cosine masking of hma/data.py:28-98 applied to a (B, T, H*W) grid of token ids in one elementwise pass.
// All randomness arrives as tensors (the host draws them, in the reference's order when parity is wanted), so the
// kernel is a pure function: bit-exact against the reference given the same draws.
//
// Per token (b, t, s), following get_maskgit_collator.collate_fn:
//   c = (id % V, id / V % V)  (or just id % V with one factor)        data.py:39 (factorize_token_ids)
//   corruption  (data.py:42-49): c[k] = random_values[k]  where r_corrupt[k] < corrupt_thresh
//   non-MLM     (data.py:51-64): for t >= first_masked_frame, c[k] = random_values[k] where r_nonmlm[k] > correct_rate[t - fmf]
//   masking     (data.py:68-83): out = c0 + V * c1, and for t >= first_masked_frame: out = mask_id where r_mask < mask_prob[b, t - fmf]
//   (without masking the reference returns the ORIGINAL ids: its unfactorize sits inside `if dataloader_apply_mask`)
#include "hma_common.h"
#include "../../include/hma_hip.h"

namespace {

__global__ __launch_bounds__(256) void collate_kernel(const int64_t* __restrict__ ids, int64_t* __restrict__ out,
                                                      const float* __restrict__ r_corrupt, float corrupt_thresh,
                                                      const int64_t* __restrict__ random_values,
                                                      const float* __restrict__ r_nonmlm, const float* __restrict__ correct_rate,
                                                      const float* __restrict__ mask_prob, const float* __restrict__ r_mask,
                                                      int64_t total, int T, int HW, int fmf, int V, int NF, int64_t mask_id,
                                                      int* __restrict__ any_masked) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  bool masked = false;
  if (i < total) {
    const int64_t id = ids[i];
    int64_t result = id;
    if (mask_prob) {
      const int s = (int)(i % HW);
      const int t = (int)((i / HW) % T);
      const int64_t b = i / ((int64_t)HW * T);
      int64_t c0 = id % V, c1 = NF == 2 ? (id / V) % V : 0;
      if (r_corrupt) {
        if (r_corrupt[NF * i] < corrupt_thresh) c0 = random_values[NF * i];
        if (NF == 2 && r_corrupt[2 * i + 1] < corrupt_thresh) c1 = random_values[2 * i + 1];
      }
      if (t >= fmf) {
        const int64_t j = (b * (T - fmf) + (t - fmf)) * HW + s;  // index in the (B, T - fmf, HW) draws
        if (r_nonmlm) {
          const float cr = correct_rate[t - fmf];
          if (r_nonmlm[NF * j] > cr) c0 = random_values[NF * i];
          if (NF == 2 && r_nonmlm[2 * j + 1] > cr) c1 = random_values[2 * i + 1];
        }
        masked = r_mask[j] < mask_prob[b * (T - fmf) + (t - fmf)];
      }
      result = masked ? mask_id : c0 + (int64_t)V * c1;
    }
    out[i] = result;
  }
  if (any_masked && __any(masked) && (threadIdx.x & 63) == 0) atomicOr(any_masked, 1);
}

}  // namespace

extern "C" int hma_maskgit_collate(void* stream, const int64_t* ids, int64_t* out_ids, const float* r_corrupt,
                                   float corrupt_thresh, const int64_t* random_values, const float* r_nonmlm,
                                   const float* correct_rate, const float* mask_prob, const float* r_mask, int64_t B, int32_t T,
                                   int32_t HW, int32_t first_masked_frame, int32_t V, int32_t num_factored, int64_t mask_id,
                                   int32_t* any_masked) {
  if (!ids || !out_ids) return HMA_EINVAL;
  if (T < 1 || HW < 1 || V < 1 || first_masked_frame < 0 || first_masked_frame > T) return HMA_EINVAL;
  if (num_factored != 1 && num_factored != 2) return HMA_EINVAL;
  if (r_corrupt && !random_values) return HMA_EINVAL;
  if (r_nonmlm && (!random_values || !correct_rate || !mask_prob)) return HMA_EINVAL;
  if (mask_prob && !r_mask) return HMA_EINVAL;
  if (B <= 0) return 0;
  const int64_t total = B * T * HW;
  hipLaunchKernelGGL(collate_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ids, out_ids,
                     r_corrupt, corrupt_thresh, random_values, r_nonmlm, correct_rate, mask_prob, r_mask, total, (int)T, (int)HW,
                     (int)first_masked_frame, (int)V, (int)num_factored, mask_id, (int*)any_masked);
  HMA_CHECK_LAUNCH();
  return 0;
}
