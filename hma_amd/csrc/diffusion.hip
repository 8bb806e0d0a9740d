// Row kernels of the diffusion head (DiffLoss / SimpleMLPAdaLN): everything around its GEMMs.
//
// Reference: hma/model/diffloss.py (TimestepEmbedder :66-96, ResBlock :99-124, FinalLayer :127-149,
// SimpleMLPAdaLN.forward :212-233) and hma/diffusion/gaussian_diffusion.py (q_sample :203-219, p_mean_variance
// :250-325 with ModelVarType.LEARNED_RANGE, _vb_terms_bpd :650-673, training_losses :675-745 with LossType.MSE,
// p_sample :358-394), diffusion_utils.py (normal_kl, discretized_gaussian_log_likelihood).
// One wave per row; the hidden width W (256 .. 2048, multiple of 256) is walked in float4 pieces per lane.  The
// schedule tables are fp32 casts of the fp64 host tables (as the reference's _extract_into_tensor does).
#include "hma_common.h"
#include "../../include/hma_hip.h"

using namespace hma;

namespace {

constexpr int MAXP = 8;  // W / 256 float4 pieces per lane

__device__ __forceinline__ float sigmoid_f(float u) { return 1.0f / (1.0f + __expf(-u)); }

// ---------------------------------------------------------------- q_sample + timestep embedding
__global__ __launch_bounds__(256) void diff_prepare_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                           const int64_t* __restrict__ t, const float* __restrict__ sqrt_ac,
                                                           const float* __restrict__ sqrt_1mac, const int32_t* __restrict__ tmap,
                                                           float* __restrict__ xt, uint16_t* __restrict__ xt_pad,
                                                           uint16_t* __restrict__ tfreq, int64_t n, int C, int pad) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int64_t ti = t[row];
  const float a = sqrt_ac ? sqrt_ac[ti] : 1.f, b = sqrt_1mac ? sqrt_1mac[ti] : 0.f;
  for (int c = lane; c < pad; c += 64) {
    float v = 0.f;
    if (c < C) {
      v = noise ? a * x0[row * C + c] + b * noise[row * C + c] : x0[row * C + c];
      if (xt) xt[row * C + c] = v;
    }
    xt_pad[row * pad + c] = to_bf16(v);
  }
  // timestep_embedding(t, 256): [cos(t f_k) | sin(t f_k)], f_k = exp(-ln(10000) k / 128)   diffloss.py:79-90
  const float tm = (float)(tmap ? (int64_t)tmap[ti] : ti);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = lane + 64 * j;
    const float f = __expf(-9.210340371976184f * (float)k / 128.0f);
    const float arg = tm * f;
    tfreq[row * 256 + k] = to_bf16(cosf(arg));
    tfreq[row * 256 + 128 + k] = to_bf16(sinf(arg));
  }
}

// ---------------------------------------------------------------- SiLU of the conditioning vector
__global__ __launch_bounds__(256) void silu_cast_kernel(const float* __restrict__ y, uint16_t* __restrict__ sy, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  const float4 v = *reinterpret_cast<const float4*>(y + i);
  *reinterpret_cast<uint2*>(sy + i) = make_uint2(pack_bf16(silu_f(v.x), silu_f(v.y)), pack_bf16(silu_f(v.z), silu_f(v.w)));
}
__global__ __launch_bounds__(256) void silu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dsy,
                                                       float* __restrict__ dy, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  const float4 v = *reinterpret_cast<const float4*>(y + i), g = *reinterpret_cast<const float4*>(dsy + i);
  *reinterpret_cast<float4*>(dy + i) = make_float4(g.x * dsilu_f(v.x), g.y * dsilu_f(v.y), g.z * dsilu_f(v.z), g.w * dsilu_f(v.w));
}

// ---------------------------------------------------------------- adaLN: LN(x) (1 + scale) + shift
// (P = W / 256 is a template parameter: with a run-time trip count the per-lane arrays are indexed dynamically and live in
// scratch memory -- the backward kernel took 1.1 ms for 1.3 GB of traffic)
template <int P>
__device__ __forceinline__ void row_stats(const float4 (&v)[P], int W, float eps, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int p = 0; p < P; ++p) s += v[p].x + v[p].y + v[p].z + v[p].w;
  mean = wave_sum(s) / W;
  float q = 0.f;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const float a = v[p].x - mean, b = v[p].y - mean, c = v[p].z - mean, d = v[p].w - mean;
    q += a * a + b * b + c * c + d * d;
  }
  rstd = rsqrtf(wave_sum(q) / W + eps);
}
__device__ __forceinline__ float4 ld4_bf16(const uint16_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y));
}

template <int P>
__global__ __launch_bounds__(256) void adaln_fwd_kernel(const float* __restrict__ x, const uint16_t* __restrict__ mod, int64_t ldm,
                                                        int off_shift, int off_scale, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, uint16_t* __restrict__ out,
                                                        int64_t n, int W) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  float4 v[P];
#pragma unroll
  for (int p = 0; p < P; ++p) v[p] = *reinterpret_cast<const float4*>(x + row * W + p * 256 + lane * 4);
  float mean, rstd;
  row_stats<P>(v, W, eps, mean, rstd);
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int c = p * 256 + lane * 4;
    float h[4] = {(v[p].x - mean) * rstd, (v[p].y - mean) * rstd, (v[p].z - mean) * rstd, (v[p].w - mean) * rstd};
    if (gamma) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
      h[0] = h[0] * g.x + b.x; h[1] = h[1] * g.y + b.y; h[2] = h[2] * g.z + b.z; h[3] = h[3] * g.w + b.w;
    }
    const float4 sh = ld4_bf16(mod + row * ldm + off_shift + c), sc = ld4_bf16(mod + row * ldm + off_scale + c);
    *reinterpret_cast<uint2*>(out + row * W + c) =
        make_uint2(pack_bf16(h[0] * (1.f + sc.x) + sh.x, h[1] * (1.f + sc.y) + sh.y),
                   pack_bf16(h[2] * (1.f + sc.z) + sh.z, h[3] * (1.f + sc.w) + sh.w));
  }
}

// dout (bf16) = grad of the modulated output.  dx += LN-backward; dmod[shift] = dout, dmod[scale] = dout * ln;
// dgamma += sum dout (1 + scale) xhat, dbeta += sum dout (1 + scale)   (per-workgroup partial sums, then atomics)
template <int P>
__global__ __launch_bounds__(512) void adaln_bwd_kernel(const uint16_t* __restrict__ dout, const float* __restrict__ x,
                                                        const uint16_t* __restrict__ mod, int64_t ldm, int off_shift, int off_scale,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                        float* __restrict__ dx, uint16_t* __restrict__ dmod,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t n, int W,
                                                        int rows_per_wave, int accumulate) {
  extern __shared__ float red[];  // (affine only) [waves][2][W]: the workgroup's dgamma / dbeta partials
  const int lane = threadIdx.x & 63, nwv = blockDim.x >> 6, wv = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * nwv + wv;
  float4 dg[P], db[P];
#pragma unroll
  for (int p = 0; p < P; ++p) dg[p] = db[p] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int rr = 0; rr < rows_per_wave; ++rr) {
    const int64_t row = wave * rows_per_wave + rr;
    if (row >= n) break;
    float4 v[P];
#pragma unroll
    for (int p = 0; p < P; ++p) v[p] = *reinterpret_cast<const float4*>(x + row * W + p * 256 + lane * 4);
    float mean, rstd;
    row_stats<P>(v, W, eps, mean, rstd);
    float4 gl[P];  // gradient wrt xhat
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int c = p * 256 + lane * 4;
      const float4 d = ld4_bf16(dout + row * W + c), sc = ld4_bf16(mod + row * ldm + off_scale + c);
      const float xh[4] = {(v[p].x - mean) * rstd, (v[p].y - mean) * rstd, (v[p].z - mean) * rstd, (v[p].w - mean) * rstd};
      float4 g = make_float4(1.f, 1.f, 1.f, 1.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gamma) { g = *reinterpret_cast<const float4*>(gamma + c); b = *reinterpret_cast<const float4*>(beta + c); }
      const float ln[4] = {xh[0] * g.x + b.x, xh[1] * g.y + b.y, xh[2] * g.z + b.z, xh[3] * g.w + b.w};
      const float dl[4] = {d.x * (1.f + sc.x), d.y * (1.f + sc.y), d.z * (1.f + sc.z), d.w * (1.f + sc.w)};  // grad wrt LN output
      *reinterpret_cast<uint2*>(dmod + row * ldm + off_shift + c) = make_uint2(pack_bf16(d.x, d.y), pack_bf16(d.z, d.w));
      *reinterpret_cast<uint2*>(dmod + row * ldm + off_scale + c) =
          make_uint2(pack_bf16(d.x * ln[0], d.y * ln[1]), pack_bf16(d.z * ln[2], d.w * ln[3]));
      dg[p].x += dl[0] * xh[0]; dg[p].y += dl[1] * xh[1]; dg[p].z += dl[2] * xh[2]; dg[p].w += dl[3] * xh[3];
      db[p].x += dl[0]; db[p].y += dl[1]; db[p].z += dl[2]; db[p].w += dl[3];
      gl[p] = make_float4(dl[0] * g.x, dl[1] * g.y, dl[2] * g.z, dl[3] * g.w);
      s1 += gl[p].x + gl[p].y + gl[p].z + gl[p].w;
      s2 += gl[p].x * xh[0] + gl[p].y * xh[1] + gl[p].z * xh[2] + gl[p].w * xh[3];
      v[p] = make_float4(xh[0], xh[1], xh[2], xh[3]);
    }
    s1 = wave_sum(s1) / W;
    s2 = wave_sum(s2) / W;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      float4* d = reinterpret_cast<float4*>(dx + row * W + p * 256 + lane * 4);
      float4 o = accumulate ? *d : make_float4(0.f, 0.f, 0.f, 0.f);  // (accumulate = 0: the first writer of dx -- no zero-fill, no read)
      o.x += rstd * (gl[p].x - s1 - v[p].x * s2); o.y += rstd * (gl[p].y - s1 - v[p].y * s2);
      o.z += rstd * (gl[p].z - s1 - v[p].z * s2); o.w += rstd * (gl[p].w - s1 - v[p].w * s2);
      *d = o;
    }
  }
  if (gamma) {
    // One atomic per workgroup and column: every wave adding its own 2 W sums put 4 096 atomics on each of the 2 048 words at the
    // MAR head's shape (65 536 x 1024), ~88 ns apiece on one address: 590 us against 224 us without the affine (round 6).
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int c = p * 256 + lane * 4;
      *reinterpret_cast<float4*>(red + (wv * 2 + 0) * W + c) = dg[p];
      *reinterpret_cast<float4*>(red + (wv * 2 + 1) * W + c) = db[p];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * W; i += blockDim.x) {
      float sum = 0.f;
      for (int w = 0; w < nwv; ++w) sum += red[w * 2 * W + i];
      if (sum != 0.f) atomicAdd(i < W ? dgamma + i : dbeta + (i - W), sum);
    }
  }
}

// ---------------------------------------------------------------- gated residual x += gate * h
__global__ __launch_bounds__(256) void gate_fwd_kernel(float* __restrict__ x, const uint16_t* __restrict__ mod, int64_t ldm, int off_gate,
                                                       const uint16_t* __restrict__ h, int64_t n, int W) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n * W) return;
  const int64_t row = i / W;
  const int c = (int)(i % W);
  const float4 g = ld4_bf16(mod + row * ldm + off_gate + c), hv = ld4_bf16(h + i);
  float4 o = *reinterpret_cast<float4*>(x + i);
  o.x += g.x * hv.x; o.y += g.y * hv.y; o.z += g.z * hv.z; o.w += g.w * hv.w;
  *reinterpret_cast<float4*>(x + i) = o;
}
// dh = dx * gate (bf16), dmod[gate] = dx * h (bf16); dx itself is also the gradient of the identity branch
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dx, const uint16_t* __restrict__ mod, int64_t ldm,
                                                       int off_gate, const uint16_t* __restrict__ h, uint16_t* __restrict__ dh,
                                                       uint16_t* __restrict__ dmod, int64_t n, int W) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n * W) return;
  const int64_t row = i / W;
  const int c = (int)(i % W);
  const float4 g = ld4_bf16(mod + row * ldm + off_gate + c), hv = ld4_bf16(h + i), d = *reinterpret_cast<const float4*>(dx + i);
  *reinterpret_cast<uint2*>(dh + i) = make_uint2(pack_bf16(d.x * g.x, d.y * g.y), pack_bf16(d.z * g.z, d.w * g.w));
  *reinterpret_cast<uint2*>(dmod + row * ldm + off_gate + c) = make_uint2(pack_bf16(d.x * hv.x, d.y * hv.y), pack_bf16(d.z * hv.z, d.w * hv.w));
}

// ---------------------------------------------------------------- loss: mse(eps) + vb(learned range), and its gradient
__device__ __forceinline__ float approx_cdf(float x, float& dcdf) {
  const float k = 0.7978845608028654f;  // sqrt(2 / pi)
  const float u = k * (x + 0.044715f * x * x * x);
  const float th = tanhf(u);
  dcdf = 0.5f * (1.f - th * th) * k * (1.f + 3.f * 0.044715f * x * x);
  return 0.5f * (1.f + th);
}

struct Sched { const float *sqrt_recip_ac, *sqrt_recipm1_ac, *coef1, *coef2, *post_logvar, *log_betas; };

// one thread per (row, channel); rows reduced by a wave (C <= 64)
constexpr int DL_RPW = 16;  // rows per wave
__global__ __launch_bounds__(256) void diff_loss_kernel(const float* __restrict__ out, int64_t ldo, const float* __restrict__ x0,
                                                        const float* __restrict__ xt, const float* __restrict__ noise,
                                                        const int64_t* __restrict__ t, Sched sc, const float* __restrict__ mask,
                                                        const float* __restrict__ denom, float gscale, float* __restrict__ stats,
                                                        float* __restrict__ rows_out, float* __restrict__ dout, int64_t n, int C) {
  const int lane = threadIdx.x & 63;
  float acc = 0.f;  // this wave's share of stats[0]: one atomic per DL_RPW rows (one per row was 65536 serialised atomics, 0.8 ms)
  for (int rr = 0; rr < DL_RPW; ++rr) {
  const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * DL_RPW + rr;
  if (row >= n) break;
  const int64_t ti = t[row];
  float contrib = 0.f, d_eps = 0.f, d_v = 0.f;
  if (lane < C) {
    const float eps = out[row * ldo + lane], v = out[row * ldo + C + lane];
    const float x_0 = x0[row * C + lane], x_t = xt[row * C + lane], nz = noise[row * C + lane];
    const float min_log = sc.post_logvar[ti], max_log = sc.log_betas[ti];
    const float frac = (v + 1.f) * 0.5f;
    const float L = frac * max_log + (1.f - frac) * min_log;  // model log-variance
    const float dL_dv = 0.5f * (max_log - min_log);
    const float pred_x0 = sc.sqrt_recip_ac[ti] * x_t - sc.sqrt_recipm1_ac[ti] * eps;
    const float mean = sc.coef1[ti] * pred_x0 + sc.coef2[ti] * x_t;        // model mean (frozen for the vb term)
    const float true_mean = sc.coef1[ti] * x_0 + sc.coef2[ti] * x_t;
    const float inv_ln2 = 1.4426950408889634f;
    float vb, dvb_dL;
    if (ti != 0) {  // KL(q(x_{t-1} | x_t, x_0) || p)
      const float dm = true_mean - mean, e1 = __expf(min_log - L), e2 = __expf(-L);
      vb = 0.5f * (-1.f + L - min_log + e1 + dm * dm * e2);
      dvb_dL = 0.5f * (1.f - e1 - dm * dm * e2);
    } else {        // decoder NLL: -log of the discretised Gaussian likelihood of x_0
      const float cx = x_0 - mean, inv = __expf(-0.5f * L);
      const float pin = inv * (cx + 1.f / 255.f), min_ = inv * (cx - 1.f / 255.f);
      float dcp, dcm;
      const float cp = approx_cdf(pin, dcp), cm = approx_cdf(min_, dcm);
      float lp, dlp_dL;  // d pin / dL = -0.5 pin, d min / dL = -0.5 min
      if (x_0 < -0.999f) {
        lp = __logf(fmaxf(cp, 1e-12f));
        dlp_dL = cp > 1e-12f ? dcp * (-0.5f * pin) / cp : 0.f;
      } else if (x_0 > 0.999f) {
        lp = __logf(fmaxf(1.f - cm, 1e-12f));
        dlp_dL = (1.f - cm) > 1e-12f ? -dcm * (-0.5f * min_) / (1.f - cm) : 0.f;
      } else {
        const float dlt = cp - cm;
        lp = __logf(fmaxf(dlt, 1e-12f));
        dlp_dL = dlt > 1e-12f ? (dcp * (-0.5f * pin) - dcm * (-0.5f * min_)) / dlt : 0.f;
      }
      vb = -lp;
      dvb_dL = -dlp_dL;
    }
    const float de = nz - eps;
    contrib = (de * de + vb * inv_ln2) / C;   // mean over channels of both terms
    d_eps = -2.f * de / C;
    d_v = dvb_dL * dL_dv * inv_ln2 / C;
  }
  const float row_loss = wave_sum(contrib);
  const float m = mask ? mask[row] : 1.f;
  if (lane == 0 && rows_out) rows_out[row] = row_loss;
  acc += row_loss * m;
  if (dout && lane < C) {
    const float w = gscale * m / (*denom);
    dout[row * ldo + lane] = d_eps * w;
    dout[row * ldo + C + lane] = d_v * w;
  }
  }
  det_loss_add(stats, acc, lane, gridDim.x * 4u);  // (order-independent: hma_common.h)
}

// ---------------------------------------------------------------- one reverse step of the sampler
__global__ __launch_bounds__(256) void p_sample_kernel(const float* __restrict__ out, int64_t ldo, float* __restrict__ x,
                                                       const float* __restrict__ noise, Sched sc, int ti, float temperature,
                                                       int clip, int64_t n, int C, int64_t half, float cfg) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * C) return;
  const int64_t row = i / C;
  const int c = (int)(i % C);
  float eps = out[row * ldo + c];
  if (half) {  // forward_with_cfg (diffloss.py:235-243): rows [0, half) conditional, [half, 2 half) unconditional
    const int64_t r = row >= half ? row - half : row;
    const float cond = out[r * ldo + c], uncond = out[(r + half) * ldo + c];
    eps = uncond + cfg * (cond - uncond);
  }
  const float v = out[row * ldo + C + c], x_t = x[i];
  const float frac = (v + 1.f) * 0.5f;
  const float L = frac * sc.log_betas[ti] + (1.f - frac) * sc.post_logvar[ti];
  float px0 = sc.sqrt_recip_ac[ti] * x_t - sc.sqrt_recipm1_ac[ti] * eps;
  if (clip) px0 = fminf(fmaxf(px0, -10.f), 10.f);
  const float mean = sc.coef1[ti] * px0 + sc.coef2[ti] * x_t;
  x[i] = mean + (ti != 0 ? __expf(0.5f * L) * noise[i] * temperature : 0.f);
}

inline unsigned rows4(int64_t n) { return (unsigned)((n + 3) / 4); }
inline unsigned flat4(int64_t n) { return (unsigned)((n / 4 + 255) / 256); }

}  // namespace

extern "C" int hma_diff_prepare(void* stream, const float* x0, const float* noise, const int64_t* t, const float* sqrt_ac,
                                const float* sqrt_1mac, const int32_t* tmap, float* xt, void* xt_pad, void* tfreq, int64_t n,
                                int32_t C, int32_t pad) {
  if (!x0 || !t || !xt_pad || !tfreq || C < 1 || pad < C) return HMA_EINVAL;
  if (noise && (!sqrt_ac || !sqrt_1mac)) return HMA_EINVAL;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(diff_prepare_kernel, dim3(rows4(n)), dim3(256), 0, (hipStream_t)stream, x0, noise, t, sqrt_ac, sqrt_1mac, tmap, xt,
                     (uint16_t*)xt_pad, (uint16_t*)tfreq, n, (int)C, (int)pad);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_silu_cast(void* stream, const float* y, void* sy, int64_t n) {
  if (!y || !sy || (n & 3)) return HMA_EINVAL;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(silu_cast_kernel, dim3(flat4(n)), dim3(256), 0, (hipStream_t)stream, y, (uint16_t*)sy, n);
  HMA_CHECK_LAUNCH();
  return 0;
}
extern "C" int hma_silu_bwd(void* stream, const float* y, const float* dsy, float* dy, int64_t n) {
  if (!y || !dsy || !dy || (n & 3)) return HMA_EINVAL;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(flat4(n)), dim3(256), 0, (hipStream_t)stream, y, dsy, dy, n);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_adaln_fwd(void* stream, const float* x, const void* mod, int64_t ldm, int32_t off_shift, int32_t off_scale,
                             const float* gamma, const float* beta, float eps, void* out, int64_t n, int32_t W) {
  if (!x || !mod || !out || W < 256 || W > 256 * MAXP || (W & 255) || (gamma && !beta)) return HMA_EINVAL;
  if (n <= 0) return 0;
#define HMA_ADALN_FWD(PP)                                                                                                      \
  case PP:                                                                                                                      \
    hipLaunchKernelGGL(adaln_fwd_kernel<PP>, dim3(rows4(n)), dim3(256), 0, (hipStream_t)stream, x, (const uint16_t*)mod, ldm, \
                       (int)off_shift, (int)off_scale, gamma, beta, eps, (uint16_t*)out, n, (int)W);                            \
    break;
  switch (W / 256) {
    HMA_ADALN_FWD(1) HMA_ADALN_FWD(2) HMA_ADALN_FWD(3) HMA_ADALN_FWD(4) HMA_ADALN_FWD(5) HMA_ADALN_FWD(6) HMA_ADALN_FWD(7) HMA_ADALN_FWD(8)
    default: return HMA_EINVAL;
  }
#undef HMA_ADALN_FWD
  HMA_CHECK_LAUNCH();
  return 0;
}
extern "C" int hma_adaln_bwd(void* stream, const void* dout, const float* x, const void* mod, int64_t ldm, int32_t off_shift,
                             int32_t off_scale, const float* gamma, const float* beta, float eps, float* dx, void* dmod,
                             float* dgamma, float* dbeta, int64_t n, int32_t W) {
  return hma_adaln_bwd_acc(stream, dout, x, mod, ldm, off_shift, off_scale, gamma, beta, eps, dx, dmod, dgamma, dbeta, n, W, 1);
}
extern "C" int hma_adaln_bwd_acc(void* stream, const void* dout, const float* x, const void* mod, int64_t ldm, int32_t off_shift,
                                 int32_t off_scale, const float* gamma, const float* beta, float eps, float* dx, void* dmod,
                                 float* dgamma, float* dbeta, int64_t n, int32_t W, int32_t accumulate) {
  if (!dout || !x || !mod || !dx || !dmod || W < 256 || W > 256 * MAXP || (W & 255)) return HMA_EINVAL;
  if (gamma && (!beta || !dgamma || !dbeta)) return HMA_EINVAL;
  if (n <= 0) return 0;
  const int rpw = n > 16384 ? 16 : 1;  // bounds the dgamma / dbeta atomics
  const int64_t waves = (n + rpw - 1) / rpw;
  // with an affine: as many waves per workgroup as 64 KB of LDS hold partials for (8 at W <= 1024, two workgroups per CU)
  const int nwv = gamma ? (W <= 1024 ? 8 : 4) : 4;
  const size_t smem = gamma ? (size_t)nwv * 2 * W * sizeof(float) : 0;
  const unsigned grid = (unsigned)((waves + nwv - 1) / nwv);
#define HMA_ADALN_BWD(PP)                                                                                                      \
  case PP:                                                                                                                      \
    hipLaunchKernelGGL(adaln_bwd_kernel<PP>, dim3(grid), dim3(64 * nwv), smem, (hipStream_t)stream, (const uint16_t*)dout, x,   \
                       (const uint16_t*)mod, ldm, (int)off_shift, (int)off_scale, gamma, beta, eps, dx, (uint16_t*)dmod, dgamma, \
                       dbeta, n, (int)W, rpw, (int)accumulate);                                                                 \
    break;
  switch (W / 256) {
    HMA_ADALN_BWD(1) HMA_ADALN_BWD(2) HMA_ADALN_BWD(3) HMA_ADALN_BWD(4) HMA_ADALN_BWD(5) HMA_ADALN_BWD(6) HMA_ADALN_BWD(7) HMA_ADALN_BWD(8)
    default: return HMA_EINVAL;
  }
#undef HMA_ADALN_BWD
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_gate_fwd(void* stream, float* x, const void* mod, int64_t ldm, int32_t off_gate, const void* h, int64_t n, int32_t W) {
  if (!x || !mod || !h || (W & 3)) return HMA_EINVAL;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(gate_fwd_kernel, dim3(flat4(n * W)), dim3(256), 0, (hipStream_t)stream, x, (const uint16_t*)mod, ldm, (int)off_gate,
                     (const uint16_t*)h, n, (int)W);
  HMA_CHECK_LAUNCH();
  return 0;
}
extern "C" int hma_gate_bwd(void* stream, const float* dx, const void* mod, int64_t ldm, int32_t off_gate, const void* h, void* dh,
                            void* dmod, int64_t n, int32_t W) {
  if (!dx || !mod || !h || !dh || !dmod || (W & 3)) return HMA_EINVAL;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(flat4(n * W)), dim3(256), 0, (hipStream_t)stream, dx, (const uint16_t*)mod, ldm, (int)off_gate,
                     (const uint16_t*)h, (uint16_t*)dh, (uint16_t*)dmod, n, (int)W);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_diff_loss(void* stream, const float* out, int64_t ldo, const float* x0, const float* xt, const float* noise,
                             const int64_t* t, const float* tables6 /* 6 x n_steps */, int32_t n_steps, const float* mask,
                             const float* denom, float grad_scale, float* stats, float* rows_out, float* dout, int64_t n, int32_t C) {
  if (!out || !x0 || !xt || !noise || !t || !tables6 || !stats || C < 1 || C > 64 || ldo < 2 * C) return HMA_EINVAL;
  if (dout && !denom) return HMA_EINVAL;
  if (n <= 0) return 0;
  const Sched sc{tables6, tables6 + n_steps, tables6 + 2 * n_steps, tables6 + 3 * n_steps, tables6 + 4 * n_steps, tables6 + 5 * n_steps};
  hipLaunchKernelGGL(diff_loss_kernel, dim3(rows4((n + DL_RPW - 1) / DL_RPW)), dim3(256), 0, (hipStream_t)stream, out, ldo, x0, xt, noise, t, sc, mask, denom,
                     grad_scale, stats, rows_out, dout, n, (int)C);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_diff_p_sample(void* stream, const float* out, int64_t ldo, float* x, const float* noise, const float* tables6,
                                 int32_t n_steps, int32_t step, float temperature, int32_t clip_denoised, int64_t n, int32_t C) {
  if (!out || !x || !tables6 || step < 0 || step >= n_steps || C < 1 || ldo < 2 * C) return HMA_EINVAL;
  if (step != 0 && !noise) return HMA_EINVAL;
  if (n <= 0) return 0;
  const Sched sc{tables6, tables6 + n_steps, tables6 + 2 * n_steps, tables6 + 3 * n_steps, tables6 + 4 * n_steps, tables6 + 5 * n_steps};
  hipLaunchKernelGGL(p_sample_kernel, dim3((unsigned)((n * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, ldo, x, noise, sc,
                     (int)step, temperature, (int)clip_denoised, n, (int)C, (int64_t)0, 1.0f);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_diff_p_sample_cfg(void* stream, const float* out, int64_t ldo, float* x, const float* noise, const float* tables6,
                                     int32_t n_steps, int32_t step, float temperature, int32_t clip_denoised, int64_t n, int32_t C,
                                     float cfg_scale) {
  if (!out || !x || !tables6 || step < 0 || step >= n_steps || C < 1 || ldo < 2 * C || (n & 1)) return HMA_EINVAL;
  if (step != 0 && !noise) return HMA_EINVAL;
  if (n <= 0) return 0;
  const Sched sc{tables6, tables6 + n_steps, tables6 + 2 * n_steps, tables6 + 3 * n_steps, tables6 + 4 * n_steps, tables6 + 5 * n_steps};
  hipLaunchKernelGGL(p_sample_kernel, dim3((unsigned)((n * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, ldo, x, noise, sc,
                     (int)step, temperature, (int)clip_denoised, n, (int)C, n / 2, cfg_scale);
  HMA_CHECK_LAUNCH();
  return 0;
}
