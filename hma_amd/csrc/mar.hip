// STMAR (continuous-latent model) input / output stages around the ST-transformer trunk.
//
// Reference: hma/model/st_mar.py -- forward :219-275 (mask-latent fill :245, patchify :199-207, patch mask :260),
// compute_latents :146-197: x = z_proj_ln(concat(token_embed(patches), action tokens) + pos_embed_TSC) -> trunk ->
// z = decoder_norm(out_x_proj(x[image tokens])) + diffusion_pos_embed_learned.  The two Linears run through
// hma_gemm_nt / hma_gemm_tn; these kernels are everything else.  d_model = 256: one wave per token row.
#include "hma_common.h"
#include "../../include/hma_hip.h"

using namespace hma;

namespace {

constexpr int D = 256;

// ---------------------------------------------------------------- patchify (+ mask-latent fill, patch mask)
// lat [B*T, H, W, C] -> patches [B*T*(H/p)*(W/p), P = p*p*C] in (p, q, c) channel order; one thread per output element.
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ lat, const uint8_t* __restrict__ masked,
                                                       const float* __restrict__ mask_token, uint16_t* __restrict__ out_bf16,
                                                       int pad, float* __restrict__ out_f32, float* __restrict__ pmask,
                                                       int64_t frames, int H, int W, int C, int p) {
  const int h = H / p, w = W / p, P = p * p * C;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = frames * h * w * P;
  if (i >= total) return;
  const int ch = (int)(i % P);
  const int64_t row = i / P;
  const int c = ch % C, q = (ch / C) % p, pp = ch / (C * p);
  const int wi = (int)(row % w), hi = (int)((row / w) % h);
  const int64_t f = row / ((int64_t)h * w);
  const int64_t pix = (f * H + hi * p + pp) * W + wi * p + q;
  const bool m = masked && masked[pix];
  const float v = m && mask_token ? mask_token[c] : lat[pix * C + c];
  if (out_bf16) out_bf16[row * pad + ch] = to_bf16(v);
  if (out_f32) out_f32[row * P + ch] = v;
  if (pmask && ch == 0) {  // "as long as it's not no mask": any masked pixel in the patch
    bool any = false;
    for (int a = 0; a < p; ++a)
      for (int b = 0; b < p; ++b) any |= masked && masked[(f * H + hi * p + a) * W + wi * p + b];
    pmask[row] = any ? 1.f : 0.f;
  }
}
__global__ __launch_bounds__(256) void zero_pad_kernel(uint16_t* __restrict__ out, int64_t rows, int P, int pad) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int w = pad - P;
  if (i >= rows * w) return;
  out[(i / w) * pad + P + (i % w)] = 0;
}
// dmask_token[c] += sum over masked pixels of d patches (fp32 [rows, ld])
__global__ __launch_bounds__(256) void mask_token_bwd_kernel(const float* __restrict__ dpatch, int64_t ld, const uint8_t* __restrict__ masked,
                                                             float* __restrict__ dmask_token, int64_t frames, int H, int W, int C, int p) {
  __shared__ float red[8];
  if (threadIdx.x < 8) red[threadIdx.x] = 0.f;
  __syncthreads();
  const int h = H / p, w = W / p, P = p * p * C;
  const int64_t total = frames * h * w * P;
  const int64_t stride = (int64_t)gridDim.x * 256;
  float acc = 0.f;
  int myc = -1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int ch = (int)(i % P);
    const int64_t row = i / P;
    const int c = ch % C, q = (ch / C) % p, pp = ch / (C * p);
    const int wi = (int)(row % w), hi = (int)((row / w) % h);
    const int64_t f = row / ((int64_t)h * w);
    if (masked[(f * H + hi * p + pp) * W + wi * p + q]) atomicAdd(&red[c & 7], dpatch[row * ld + ch]);
    (void)acc; (void)myc;
  }
  __syncthreads();
  if (threadIdx.x < C && threadIdx.x < 8) atomicAdd(dmask_token + threadIdx.x, red[threadIdx.x]);
}

// ---------------------------------------------------------------- trunk input: LN_affine(tokens + pos)
__device__ __forceinline__ void ln_row4(const float4 v, float eps, float (&xh)[4], float& rstd) {
  const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / D);
  const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
  rstd = rsqrtf(wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / D) + eps);
  xh[0] = d0 * rstd; xh[1] = d1 * rstd; xh[2] = d2 * rstd; xh[3] = d3 * rstd;
}

__global__ __launch_bounds__(256) void mar_embed_fwd_kernel(const float* __restrict__ xtok, const float* __restrict__ a_emb,
                                                            const float* __restrict__ pos, int64_t pos_frame_stride,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            float* __restrict__ x, uint16_t* __restrict__ xhat,
                                                            float* __restrict__ rstd_out, int64_t frames, int T, int S, int A) {
  const int lane = threadIdx.x & 63, SA = S + A;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= frames * SA) return;
  const int64_t f = row / SA;
  const int s = (int)(row % SA), t = (int)(f % T);
  float4 v = s < S ? *reinterpret_cast<const float4*>(xtok + (f * S + s) * D + lane * 4)
                   : *reinterpret_cast<const float4*>(a_emb + f * D + lane * 4);
  const float4 pe = *reinterpret_cast<const float4*>(pos + t * pos_frame_stride + (int64_t)s * D + lane * 4);
  v.x += pe.x; v.y += pe.y; v.z += pe.z; v.w += pe.w;
  float xh[4], rstd;
  ln_row4(v, eps, xh, rstd);
  const float4 g = *reinterpret_cast<const float4*>(gamma + lane * 4), b = *reinterpret_cast<const float4*>(beta + lane * 4);
  *reinterpret_cast<float4*>(x + row * D + lane * 4) = make_float4(xh[0] * g.x + b.x, xh[1] * g.y + b.y, xh[2] * g.z + b.z, xh[3] * g.w + b.w);
  *reinterpret_cast<uint2*>(xhat + row * D + lane * 4) = make_uint2(pack_bf16(xh[0], xh[1]), pack_bf16(xh[2], xh[3]));
  if (lane == 0) rstd_out[row] = rstd;
}

// dv = LN-backward(dx * gamma); dxtok (image rows) = dv, da_emb[f] += sum over the action rows of frame f, dpos[t][s] += sum over the
// batch, dgamma / dbeta +=.  A wave owns one POSITION (t, s) and walks the batch: the positional gradient is summed in registers and
// added once, without atomics (a workgroup per frame added every row's 256 values with atomics: 21 M lane-atomics, 278 us at
// 16 x 16 frames of 320 rows against 31 us forward, round 6; now 87 us); the action rows' sums go to da_emb with one atomic per (frame, column)
// and action position; dgamma / dbeta meet in LDS first (one atomic per column and workgroup of 16 waves).
constexpr int MW = 16;  // waves per workgroup of the two backward kernels below
__global__ __launch_bounds__(64 * MW) void mar_embed_bwd_kernel(const float* __restrict__ dx, const uint16_t* __restrict__ xhat,
                                                               const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                               float* __restrict__ dxtok, float* __restrict__ da_emb,
                                                               float* __restrict__ dpos, int64_t pos_frame_stride,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t frames, int T,
                                                               int S, int A) {
  __shared__ float red[2][MW][D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, SA = S + A;
  const int64_t pos = (int64_t)blockIdx.x * MW + w;  // (t, s)
  const int64_t B = frames / T;
  const float4 g4 = *reinterpret_cast<const float4*>(gamma + lane * 4);
  const float gm[4] = {g4.x, g4.y, g4.z, g4.w};
  float dg[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0}, dp[4] = {0, 0, 0, 0};
  if (pos < (int64_t)T * SA) {
    const int t = (int)(pos / SA), s = (int)(pos % SA);
    for (int64_t bb = 0; bb < B; ++bb) {
      const int64_t f = bb * T + t, row = f * SA + s;
      const float4 d = *reinterpret_cast<const float4*>(dx + row * D + lane * 4);
      const uint2 hb = *reinterpret_cast<const uint2*>(xhat + row * D + lane * 4);
      const float dy[4] = {d.x, d.y, d.z, d.w};
      const float xh[4] = {bf16_lo(hb.x), bf16_hi(hb.x), bf16_lo(hb.y), bf16_hi(hb.y)};
      float gl[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        gl[j] = dy[j] * gm[j];
        s1 += gl[j];
        s2 += gl[j] * xh[j];
        dg[j] += dy[j] * xh[j];
        db[j] += dy[j];
      }
      s1 = wave_sum(s1) * (1.0f / D);
      s2 = wave_sum(s2) * (1.0f / D);
      const float rs = rstd[row];
      float dv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dv[j] = rs * (gl[j] - s1 - xh[j] * s2);
        dp[j] += dv[j];
      }
      if (s < S) {
        *reinterpret_cast<float4*>(dxtok + (f * S + s) * D + lane * 4) = make_float4(dv[0], dv[1], dv[2], dv[3]);
      } else {
        float* pa = da_emb + f * D + lane * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(pa + j, dv[j]);
      }
    }
    float4* pp = reinterpret_cast<float4*>(dpos + t * pos_frame_stride + (int64_t)s * D + lane * 4);  // this wave's alone
    float4 o = *pp;
    o.x += dp[0]; o.y += dp[1]; o.z += dp[2]; o.w += dp[3];
    *pp = o;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[0][w][lane * 4 + j] = dg[j];
    red[1][w][lane * 4 + j] = db[j];
  }
  __syncthreads();
  if (threadIdx.x < 2 * D) {
    const int k = threadIdx.x >> 8, c = threadIdx.x & (D - 1);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MW; ++i) sum += red[k][i][c];
    if (sum != 0.f) atomicAdd((k ? dbeta : dgamma) + c, sum);
  }
}

// ---------------------------------------------------------------- readout: z = LN_affine(y) + pos2[t, s]
__global__ __launch_bounds__(256) void mar_readout_fwd_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, const float* __restrict__ pos2,
                                                              float* __restrict__ z, uint16_t* __restrict__ yhat,
                                                              float* __restrict__ rstd_out, int64_t rows, int T, int S) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t pr = row % ((int64_t)T * S);  // (t, s) index into the (T*S, d) table
  const float4 v = *reinterpret_cast<const float4*>(y + row * D + lane * 4);
  float xh[4], rstd;
  ln_row4(v, eps, xh, rstd);
  const float4 g = *reinterpret_cast<const float4*>(gamma + lane * 4), b = *reinterpret_cast<const float4*>(beta + lane * 4);
  const float4 pe = *reinterpret_cast<const float4*>(pos2 + pr * D + lane * 4);
  *reinterpret_cast<float4*>(z + row * D + lane * 4) =
      make_float4(xh[0] * g.x + b.x + pe.x, xh[1] * g.y + b.y + pe.y, xh[2] * g.z + b.z + pe.z, xh[3] * g.w + b.w + pe.w);
  *reinterpret_cast<uint2*>(yhat + row * D + lane * 4) = make_uint2(pack_bf16(xh[0], xh[1]), pack_bf16(xh[2], xh[3]));
  if (lane == 0) rstd_out[row] = rstd;
}
// dy = LN-backward(dz * gamma) (fp32), dpos2 += dz, dgamma / dbeta +=.  As above: a wave per position (t, s), the batch walked in
// registers, no atomics on dpos2 (239 -> 35 us at 65 536 rows).
__global__ __launch_bounds__(64 * MW) void mar_readout_bwd_kernel(const float* __restrict__ dz, const uint16_t* __restrict__ yhat,
                                                                 const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                 float* __restrict__ dy, float* __restrict__ dpos2,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows, int T,
                                                                 int S) {
  __shared__ float red[2][MW][D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t P = (int64_t)T * S, pos = (int64_t)blockIdx.x * MW + w;
  const float4 g4 = *reinterpret_cast<const float4*>(gamma + lane * 4);
  const float gm[4] = {g4.x, g4.y, g4.z, g4.w};
  float dg[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0};
  if (pos < P) {
    for (int64_t row = pos; row < rows; row += P) {
      const float4 d = *reinterpret_cast<const float4*>(dz + row * D + lane * 4);
      const uint2 hb = *reinterpret_cast<const uint2*>(yhat + row * D + lane * 4);
      const float dv[4] = {d.x, d.y, d.z, d.w};
      const float xh[4] = {bf16_lo(hb.x), bf16_hi(hb.x), bf16_lo(hb.y), bf16_hi(hb.y)};
      float gl[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        gl[j] = dv[j] * gm[j];
        s1 += gl[j];
        s2 += gl[j] * xh[j];
        dg[j] += dv[j] * xh[j];
        db[j] += dv[j];
      }
      s1 = wave_sum(s1) * (1.0f / D);
      s2 = wave_sum(s2) * (1.0f / D);
      const float rs = rstd[row];
      *reinterpret_cast<float4*>(dy + row * D + lane * 4) =
          make_float4(rs * (gl[0] - s1 - xh[0] * s2), rs * (gl[1] - s1 - xh[1] * s2), rs * (gl[2] - s1 - xh[2] * s2), rs * (gl[3] - s1 - xh[3] * s2));
    }
    float4* pp = reinterpret_cast<float4*>(dpos2 + pos * D + lane * 4);  // (d pos2 = sum of dz over the batch = db: this wave's alone)
    float4 o = *pp;
    o.x += db[0]; o.y += db[1]; o.z += db[2]; o.w += db[3];
    *pp = o;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[0][w][lane * 4 + j] = dg[j];
    red[1][w][lane * 4 + j] = db[j];
  }
  __syncthreads();
  if (threadIdx.x < 2 * D) {
    const int k = threadIdx.x >> 8, c = threadIdx.x & (D - 1);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MW; ++i) sum += red[k][i][c];
    if (sum != 0.f) atomicAdd((k ? dbeta : dgamma) + c, sum);
  }
}

}  // namespace

extern "C" int hma_mar_patchify(void* stream, const float* latents, const uint8_t* masked, const float* mask_token, void* out_bf16,
                                int32_t pad, float* out_f32, float* patch_mask, int64_t frames, int32_t H, int32_t W, int32_t C,
                                int32_t patch) {
  if (!latents || (!out_bf16 && !out_f32) || patch < 1 || H % patch || W % patch || C < 1) return HMA_EINVAL;
  const int P = patch * patch * C;
  if (out_bf16 && pad < P) return HMA_EINVAL;
  if (patch_mask && !masked) return HMA_EINVAL;
  if (frames <= 0) return 0;
  const int64_t rows = frames * (H / patch) * (W / patch);
  hipStream_t s = (hipStream_t)stream;
  if (out_bf16 && pad > P) {
    hipLaunchKernelGGL(zero_pad_kernel, dim3((unsigned)((rows * (pad - P) + 255) / 256)), dim3(256), 0, s, (uint16_t*)out_bf16, rows, P, (int)pad);
    HMA_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((rows * P + 255) / 256)), dim3(256), 0, s, latents, masked, mask_token,
                     (uint16_t*)out_bf16, (int)pad, out_f32, patch_mask, frames, (int)H, (int)W, (int)C, (int)patch);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_mar_mask_token_bwd(void* stream, const float* dpatches, int64_t ld, const uint8_t* masked, float* dmask_token,
                                      int64_t frames, int32_t H, int32_t W, int32_t C, int32_t patch) {
  if (!dpatches || !masked || !dmask_token || C > 8 || patch < 1 || H % patch || W % patch) return HMA_EINVAL;
  if (frames <= 0) return 0;
  const int64_t total = frames * H * W * C;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(mask_token_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dpatches, ld, masked, dmask_token,
                     frames, (int)H, (int)W, (int)C, (int)patch);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_mar_embed_fwd(void* stream, const float* xtok, const float* a_emb, const float* pos, int64_t pos_frame_stride,
                                 const float* gamma, const float* beta, float eps, float* x, void* xhat, float* rstd, int64_t frames,
                                 int32_t T, int32_t S, int32_t A) {
  if (!xtok || !pos || !gamma || !beta || !x || !xhat || !rstd || T < 1 || S < 1 || A < 0 || (A > 0 && !a_emb)) return HMA_EINVAL;
  if (frames <= 0) return 0;
  const int64_t rows = frames * (S + A);
  hipLaunchKernelGGL(mar_embed_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, xtok, a_emb, pos,
                     pos_frame_stride, gamma, beta, eps, x, (uint16_t*)xhat, rstd, frames, (int)T, (int)S, (int)A);
  HMA_CHECK_LAUNCH();
  return 0;
}
extern "C" int hma_mar_embed_bwd(void* stream, const float* dx, const void* xhat, const float* rstd, const float* gamma, float* dxtok,
                                 float* da_emb, float* dpos, int64_t pos_frame_stride, float* dgamma, float* dbeta, int64_t frames,
                                 int32_t T, int32_t S, int32_t A) {
  if (!dx || !xhat || !rstd || !gamma || !dxtok || !dpos || !dgamma || !dbeta || (A > 0 && !da_emb)) return HMA_EINVAL;
  if (frames <= 0) return 0;
  if (T < 1 || frames % T) return HMA_EINVAL;
  const int64_t positions = (int64_t)T * (S + A);
  hipLaunchKernelGGL(mar_embed_bwd_kernel, dim3((unsigned)((positions + MW - 1) / MW)), dim3(64 * MW), 0, (hipStream_t)stream, dx,
                     (const uint16_t*)xhat, rstd, gamma, dxtok, da_emb, dpos, pos_frame_stride, dgamma, dbeta, frames, (int)T, (int)S, (int)A);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_mar_readout_fwd(void* stream, const float* y, const float* gamma, const float* beta, float eps, const float* pos2,
                                   float* z, void* yhat, float* rstd, int64_t rows, int32_t T, int32_t S) {
  if (!y || !gamma || !beta || !pos2 || !z || !yhat || !rstd || T < 1 || S < 1) return HMA_EINVAL;
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(mar_readout_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, y, gamma, beta, eps, pos2,
                     z, (uint16_t*)yhat, rstd, rows, (int)T, (int)S);
  HMA_CHECK_LAUNCH();
  return 0;
}
extern "C" int hma_mar_readout_bwd(void* stream, const float* dz, const void* yhat, const float* rstd, const float* gamma, float* dy,
                                   float* dpos2, float* dgamma, float* dbeta, int64_t rows, int32_t T, int32_t S) {
  if (!dz || !yhat || !rstd || !gamma || !dy || !dpos2 || !dgamma || !dbeta) return HMA_EINVAL;
  if (rows <= 0) return 0;
  if (T < 1 || S < 1 || rows % ((int64_t)T * S)) return HMA_EINVAL;
  const int64_t positions = (int64_t)T * S;
  hipLaunchKernelGGL(mar_readout_bwd_kernel, dim3((unsigned)((positions + MW - 1) / MW)), dim3(64 * MW), 0, (hipStream_t)stream, dz,
                     (const uint16_t*)yhat, rstd, gamma, dy, dpos2, dgamma, dbeta, rows, (int)T, (int)S);
  HMA_CHECK_LAUNCH();
  return 0;
}
