// Factorised-vocabulary cross-entropy (2 x 512-way), accuracy and logits gradient; MaskGIT step.
//
// Reference: STMaskGIT.compute_video_loss_and_acc hma/model/st_mask_git.py:603-630 with the mask
// rule of forward (:714-716), F.cross_entropy(label_smoothing=0.01, reduction="none").sum(factors),
// factorize_labels defaults (2, 512) (:617, factorization_utils.py:85-96); readout channel
// c = v * 512 + k (v = factor, :397-402, 610-615).  MaskGIT step: :397-453.
// One wave64 per token row: 1024 fp32 logits = 16 per lane (8 per factor), wave reductions only.
#include "hma_common.h"
#include "../../include/hma_hip.h"

using namespace hma;

namespace {

constexpr int V = 512;
constexpr int C = 1024;

__global__ __launch_bounds__(256) void count_masked_kernel(const int64_t* __restrict__ ids, float* __restrict__ stats,
                                                           int64_t B, int T, int S, int64_t mask_id) {
  const int64_t total = B * T * S;
  float c = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int t = (int)((i / S) % T);
    if (t >= 1 && ids[i] == mask_id) c += 1.f;
  }
  // one atomic per workgroup (thousands of single-lane atomics on one address took 20 us of this kernel's 26)
  __shared__ float part[4];
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    c = (part[0] + part[1]) + (part[2] + part[3]);
    if (c != 0.f) atomicAdd(stats + 2, c);
  }
}

struct FactorStats {
  float m, sumexp, sumx, xt;
  int arg;
};

// lane holds x[0..8) = logits[f*512 + lane*8 .. +8]
__device__ __forceinline__ FactorStats factor_stats(const float (&x)[8], int lane, int target) {
  FactorStats r;
  float m = x[0];
  int a = 0;
#pragma unroll
  for (int j = 1; j < 8; ++j)
    if (x[j] > m) { m = x[j]; a = j; }
  a += lane * 8;
  float sx = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) sx += x[j];
  float xt = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (lane * 8 + j == target) xt = x[j];
  // wave arg-max, lowest index wins ties (torch.argmax on the reference path)
  float bm = m;
  int ba = a;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(bm, o, 64);
    const int oa = __shfl_xor(ba, o, 64);
    if (om > bm || (om == bm && oa < ba)) { bm = om; ba = oa; }
  }
  float se = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) se += __expf(x[j] - bm);
  r.m = bm;
  r.arg = ba;
  r.sumexp = wave_sum(se);
  r.sumx = wave_sum(sx);
  r.xt = wave_sum(xt);
  return r;
}

__device__ __forceinline__ void load8(const float* p, float (&x)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
  x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
}

__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ input_ids,
                                                 const int64_t* __restrict__ labels, float* __restrict__ stats,
                                                 uint16_t* __restrict__ dlogits, const float* __restrict__ grad_scale_dev,
                                                 float grad_scale, int64_t rows, int T, int S, int64_t mask_id, float eps_ls) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  const float nmask = stats[2];
  const float wgt = grad_scale * (grad_scale_dev ? *grad_scale_dev : 1.0f) / nmask;
  float loss_acc = 0.f, acc_acc = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += nw) {
    const int t = (int)((row / S) % T);
    const bool live = t >= 1 && input_ids[row] == mask_id;
    if (!live) {
      if (dlogits) {
        const uint4 z = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(dlogits + row * C + lane * 8) = z;
        *reinterpret_cast<uint4*>(dlogits + row * C + V + lane * 8) = z;
      }
      continue;
    }
    const int64_t lab = labels[row];
    float row_loss = 0.f;
    bool ok = true;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int target = (int)(f == 0 ? lab % V : (lab / V) % V);
      float x[8];
      load8(logits + row * C + f * V + lane * 8, x);
      const FactorStats st = factor_stats(x, lane, target);
      const float lse = st.m + __logf(st.sumexp);
      row_loss += (1.f - eps_ls) * (lse - st.xt) + eps_ls * (lse - st.sumx * (1.0f / V));
      ok = ok && (st.arg == target);
      if (dlogits) {
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float p = __expf(x[j] - lse);
          const float tgt = (lane * 8 + j == target ? (1.f - eps_ls) : 0.f) + eps_ls * (1.0f / V);
          g[j] = wgt * (p - tgt);
        }
        *reinterpret_cast<uint4*>(dlogits + row * C + f * V + lane * 8) = pack8(g);
      }
    }
    loss_acc += row_loss;
    acc_acc += ok ? 1.f : 0.f;
  }
  if (lane == 0 && acc_acc != 0.f) atomicAdd(stats + 1, acc_acc);
  det_loss_add(stats, loss_acc, lane, gridDim.x * 4u);
}

// --------------------------------------------------------------------------------- MaskGIT step
// one token's 2 x 512 logits (lane holds 8 per factor) -> (sampled id, confidence); every lane returns the same values
__device__ __forceinline__ void maskgit_token(const float (&xc)[2][8], int lane, const float* __restrict__ noise_tok, int& sample,
                                              float& c) {
  sample = 0;
  c = 1.f;
#pragma unroll
  for (int f = 1; f >= 0; --f) {  // flip(2): highest factor first (:408)
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = xc[f][j];
    const FactorStats st = factor_stats(x, lane, -1);
    if (!noise_tok) {  // greedy (temperature <= 1e-8, :409-410)
      sample = sample * V + st.arg;
      c *= 1.0f / st.sumexp;  // softmax prob of the arg-max = exp(0) / sum exp(x - max)
    } else {
      // Categorical(probs).sample() (:411-416; the temperature cancels in the normalisation): torch.multinomial draws ONE
      // sample as argmax_k p_k / q_k with q ~ Exp(1) -- q is the injected draw, so a run is replayable bit for bit.
      float q[8];
      load8(noise_tok + f * V + lane * 8, q);
      const float inv = 1.0f / st.sumexp;
      float best = -1.f, pbest = 0.f;
      int a = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float pj = __expf(x[j] - st.m) * inv;
        const float r = pj / q[j];
        if (r > best) { best = r; a = j; pbest = pj; }
      }
      a += lane * 8;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64), op = __shfl_xor(pbest, o, 64);
        const int oa = __shfl_xor(a, o, 64);
        if (ob > best || (ob == best && oa < a)) { best = ob; a = oa; pbest = op; }
      }
      sample = sample * V + a;
      c *= pbest;  // torch.gather(probs, 1, sample) (:420)
    }
  }
}

// S <= 256: thread s owns token s.  Rank = position in a STABLE ascending sort of the confidences (previously unmasked tokens
// pinned to +inf, :442-443); the n_mask lowest are re-masked (:446), the rest become unmasked (:445); previously unmasked tokens
// keep their prompt value (:449).  conf / samp: LDS, filled for every token; conf is overwritten with the ranking keys.
__device__ __forceinline__ void maskgit_rank_update(float* conf, const int* samp, int64_t* __restrict__ prompt,
                                                    uint8_t* __restrict__ unmasked, const float* __restrict__ conf_override,
                                                    float* __restrict__ conf_out, int64_t b, int T, int S, int out_t, int n_mask,
                                                    int last, int64_t mask_id) {
  const int s = threadIdx.x;
  const bool active = s < S;
  bool prev_unm = false;
  int64_t out = 0;
  if (active) {
    prev_unm = unmasked[b * S + s] != 0;
    out = samp[s];
    if (conf_out) conf_out[b * S + s] = conf[s];
  }
  // the confidences the ranking uses (override, previously unmasked tokens pinned to +inf), once per token, through LDS
  float c_eff = INFINITY;
  if (active && !last) {
    c_eff = conf_override ? conf_override[b * S + s] : conf[s];
    if (prev_unm) c_eff = INFINITY;
  }
  __syncthreads();  // (everyone has read conf[s] / samp[s])
  if (active) conf[s] = c_eff;
  __syncthreads();
  if (active) {
    if (!last) {
      int rank = 0;
      for (int j = 0; j < S; ++j) {
        const float cj = conf[j];
        rank += (cj < c_eff || (cj == c_eff && j < s)) ? 1 : 0;
      }
      if (rank < n_mask) out = mask_id;
    }
    if (prev_unm) out = prompt[(b * T + out_t) * (int64_t)S + s];
  }
  __syncthreads();  // every thread has read `unmasked` before anyone updates it
  if (active) {
    prompt[(b * T + out_t) * (int64_t)S + s] = out;
    if (!last && out != mask_id) unmasked[b * S + s] = 1;
  }
}

// One launch, one workgroup per sample (no scratch needed): a wave walks its tokens with the NEXT token's 4 KB of logits in flight.
__global__ __launch_bounds__(256) void maskgit_kernel(const float* __restrict__ logits, int64_t* __restrict__ prompt,
                                                      uint8_t* __restrict__ unmasked, const float* __restrict__ conf_override,
                                                      float* __restrict__ conf_out, int T, int S, int out_t, int n_mask, int last,
                                                      int64_t mask_id, int logits_T, int logits_t,
                                                      const float* __restrict__ sample_noise) {
  extern __shared__ float sm[];              // conf[S] | sample[S] (as int)
  float* conf = sm;
  int* samp = reinterpret_cast<int*>(sm + S);
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* lg = logits + ((b * logits_T + logits_t) * (int64_t)S) * C;
  float xn[2][8];
  if (w < S) {
    load8(lg + (int64_t)w * C + V + lane * 8, xn[1]);
    load8(lg + (int64_t)w * C + lane * 8, xn[0]);
  }
  for (int s = w; s < S; s += 4) {
    float xc[2][8];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int j = 0; j < 8; ++j) xc[f][j] = xn[f][j];
    {
      const int sn = s + 4 < S ? s + 4 : s;
      load8(lg + (int64_t)sn * C + V + lane * 8, xn[1]);
      load8(lg + (int64_t)sn * C + lane * 8, xn[0]);
    }
    int sample;
    float c;
    maskgit_token(xc, lane, sample_noise ? sample_noise + (b * (int64_t)S + s) * 2 * V : nullptr, sample, c);
    if (lane == 0) {
      conf[s] = c;
      samp[s] = sample;
    }
  }
  __syncthreads();
  maskgit_rank_update(conf, samp, prompt, unmasked, conf_override, conf_out, b, T, S, out_t, n_mask, last, mask_id);
}

// Two launches when the caller provides scratch (conf_out [B, S] f32 + samp_scratch [B, S] i32): one wave per TOKEN over the whole
// chip (a decode step has 64 samples -- 64 workgroups walking 256 tokens each left three quarters of the CUs idle: 126 us), then
// the per-sample ranking.
__global__ __launch_bounds__(256) void maskgit_sample_kernel(const float* __restrict__ logits, float* __restrict__ conf_out,
                                                             int* __restrict__ samp_out, int64_t tokens, int S, int logits_T,
                                                             int logits_t, const float* __restrict__ sample_noise) {
  const int lane = threadIdx.x & 63;
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= tokens) return;
  const int64_t b = tok / S, s = tok - b * S;
  const float* lg = logits + (((b * logits_T + logits_t) * (int64_t)S) + s) * C;
  float xc[2][8];
  load8(lg + V + lane * 8, xc[1]);
  load8(lg + lane * 8, xc[0]);
  int sample;
  float c;
  maskgit_token(xc, lane, sample_noise ? sample_noise + tok * 2 * V : nullptr, sample, c);
  if (lane == 0) {
    conf_out[tok] = c;
    samp_out[tok] = sample;
  }
}
__global__ __launch_bounds__(256) void maskgit_rank_kernel(int64_t* __restrict__ prompt, uint8_t* __restrict__ unmasked,
                                                           const float* __restrict__ conf_override, const float* __restrict__ conf_in,
                                                           const int* __restrict__ samp_in, int T, int S, int out_t, int n_mask,
                                                           int last, int64_t mask_id) {
  extern __shared__ float sm[];
  float* conf = sm;
  int* samp = reinterpret_cast<int*>(sm + S);
  const int64_t b = blockIdx.x;
  if ((int)threadIdx.x < S) {
    conf[threadIdx.x] = conf_in[b * S + threadIdx.x];
    samp[threadIdx.x] = samp_in[b * S + threadIdx.x];
  }
  __syncthreads();
  maskgit_rank_update(conf, samp, prompt, unmasked, conf_override, nullptr, b, T, S, out_t, n_mask, last, mask_id);
}

}  // namespace

extern "C" int hma_count_masked(void* stream, const int64_t* input_ids, float* stats, int64_t B, int32_t T, int32_t S,
                                int64_t mask_id) {
  if (!input_ids || !stats) return HMA_EINVAL;
  const int64_t total = B * T * S;
  if (total <= 0) return 0;
  int64_t blocks = (total + 2047) / 2048;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(count_masked_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, input_ids, stats, B, (int)T,
                     (int)S, mask_id);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_ce_fwd_bwd(void* stream, const float* logits, const int64_t* input_ids, const int64_t* labels, float* stats,
                              void* dlogits, const float* grad_scale_dev, float grad_scale, int64_t B, int32_t T, int32_t S,
                              int64_t mask_id, float label_smoothing) {
  if (!logits || !input_ids || !labels || !stats) return HMA_EINVAL;
  const int64_t rows = B * T * S;
  if (rows <= 0) return 0;
  int64_t blocks = (rows + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(ce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, input_ids, labels, stats,
                     (uint16_t*)dlogits, grad_scale_dev, grad_scale, rows, (int)T, (int)S, mask_id, label_smoothing);
  HMA_CHECK_LAUNCH();
  return 0;
}

static int maskgit_launch(void* stream, const float* logits, int64_t* prompt, uint8_t* unmasked, const float* conf_override,
                          float* conf_out, int64_t B, int32_t T, int32_t S, int32_t out_t, int32_t n_mask, int32_t last,
                          int64_t mask_id, int32_t logits_T, int32_t logits_t, const float* sample_noise, int32_t* samp_scratch = nullptr) {
  if (!logits || !prompt || !unmasked) return HMA_EINVAL;
  if (S > 256 || out_t < 0 || out_t >= T) return HMA_EINVAL;  // one pass of 256 threads covers the frame
  if (logits_T <= 0) { logits_T = T; logits_t = out_t; }
  if (logits_t < 0 || logits_t >= logits_T) return HMA_EINVAL;
  if (B <= 0) return 0;
  const size_t smem = (size_t)S * 8;
  if (samp_scratch && conf_out) {
    const int64_t tokens = B * S;
    hipLaunchKernelGGL(maskgit_sample_kernel, dim3((unsigned)((tokens + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, conf_out,
                       samp_scratch, tokens, (int)S, (int)logits_T, (int)logits_t, sample_noise);
    HMA_CHECK_LAUNCH();
    hipLaunchKernelGGL(maskgit_rank_kernel, dim3((unsigned)B), dim3(256), smem, (hipStream_t)stream, prompt, unmasked, conf_override,
                       (const float*)conf_out, (const int*)samp_scratch, (int)T, (int)S, (int)out_t, (int)n_mask, (int)last, mask_id);
    HMA_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(maskgit_kernel, dim3((unsigned)B), dim3(256), smem, (hipStream_t)stream, logits, prompt, unmasked,
                     conf_override, conf_out, (int)T, (int)S, (int)out_t, (int)n_mask, (int)last, mask_id, (int)logits_T,
                     (int)logits_t, sample_noise);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_maskgit_step(void* stream, const float* logits, int64_t* prompt, uint8_t* unmasked,
                                const float* conf_override, float* conf_out, int64_t B, int32_t T, int32_t S,
                                int32_t out_t, int32_t n_mask, int32_t last, int64_t mask_id, int32_t logits_T,
                                int32_t logits_t) {
  return maskgit_launch(stream, logits, prompt, unmasked, conf_override, conf_out, B, T, S, out_t, n_mask, last, mask_id, logits_T,
                        logits_t, nullptr);
}

extern "C" int hma_maskgit_step_sampled(void* stream, const float* logits, int64_t* prompt, uint8_t* unmasked,
                                        const float* conf_override, float* conf_out, const float* sample_noise, int64_t B,
                                        int32_t T, int32_t S, int32_t out_t, int32_t n_mask, int32_t last, int64_t mask_id,
                                        int32_t logits_T, int32_t logits_t) {
  if (!sample_noise) return HMA_EINVAL;
  return maskgit_launch(stream, logits, prompt, unmasked, conf_override, conf_out, B, T, S, out_t, n_mask, last, mask_id, logits_T,
                        logits_t, sample_noise);
}

extern "C" int hma_maskgit_step_wide(void* stream, const float* logits, int64_t* prompt, uint8_t* unmasked,
                                     const float* conf_override, float* conf_out, int32_t* samp_scratch, const float* sample_noise,
                                     int64_t B, int32_t T, int32_t S, int32_t out_t, int32_t n_mask, int32_t last, int64_t mask_id,
                                     int32_t logits_T, int32_t logits_t) {
  if (!conf_out || !samp_scratch) return HMA_EINVAL;
  return maskgit_launch(stream, logits, prompt, unmasked, conf_override, conf_out, B, T, S, out_t, n_mask, last, mask_id, logits_T,
                        logits_t, sample_noise, samp_scratch);
}
