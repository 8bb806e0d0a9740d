// LayerNorm kernels over d_model = 256 for gfx950: one wave64 per token row, 4 columns per lane
// (one 16-B fp32 load / one 8-B bf16 store per lane), HBM-bound streaming passes.
//
// Reference: nn.LayerNorm at hma/model/st_transformer.py:50,75 (eps 1e-5, affine applied by the
// consumer GEMM prologue) and ModulateLayer.norm_final + modulate at hma/model/st_mask_git.py:58,71-74
// (eps 1e-6, no affine).
#include "hma_common.h"
#include "../../include/hma_hip.h"

using namespace hma;

namespace {

constexpr int D = 256;

__device__ __forceinline__ void ln_row(const float4 v, float eps, float (&xh)[4], float& rstd) {
  const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / D);
  const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
  const float var = wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / D);
  rstd = rsqrtf(var + eps);
  xh[0] = d0 * rstd; xh[1] = d1 * rstd; xh[2] = d2 * rstd; xh[3] = d3 * rstd;
}

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, uint16_t* __restrict__ xhat,
                                                     float* __restrict__ rstd_out, int64_t rows, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t row = wave; row < rows; row += nw) {
    const float4 v = *reinterpret_cast<const float4*>(x + row * D + lane * 4);
    float xh[4], rstd;
    ln_row(v, eps, xh, rstd);
    *reinterpret_cast<uint2*>(xhat + row * D + lane * 4) = make_uint2(pack_bf16(xh[0], xh[1]), pack_bf16(xh[2], xh[3]));
    if (lane == 0) rstd_out[row] = rstd;
  }
}

// dx += rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dxn * gamma
__global__ __launch_bounds__(256) void ln_bwd_kernel(const uint16_t* __restrict__ dxn, const uint16_t* __restrict__ xhat,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     float* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, uint16_t* __restrict__ dxb, int64_t rows) {
  __shared__ float red[2][4][D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * 4 + w;
  const int64_t nw = (int64_t)gridDim.x * 4;
  float gm[4] = {1.f, 1.f, 1.f, 1.f};
  if (gamma) {
    const float4 g4 = *reinterpret_cast<const float4*>(gamma + lane * 4);
    gm[0] = g4.x; gm[1] = g4.y; gm[2] = g4.z; gm[3] = g4.w;
  }
  float dg[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0};
  for (int64_t row = wave; row < rows; row += nw) {
    const uint2 a = *reinterpret_cast<const uint2*>(dxn + row * D + lane * 4);
    const uint2 b = *reinterpret_cast<const uint2*>(xhat + row * D + lane * 4);
    const float dy[4] = {bf16_lo(a.x), bf16_hi(a.x), bf16_lo(a.y), bf16_hi(a.y)};
    const float xh[4] = {bf16_lo(b.x), bf16_hi(b.x), bf16_lo(b.y), bf16_hi(b.y)};
    float g[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      g[j] = dy[j] * gm[j];
      s1 += g[j];
      s2 += g[j] * xh[j];
      dg[j] += dy[j] * xh[j];
      db[j] += dy[j];
    }
    s1 = wave_sum(s1) * (1.0f / D);
    s2 = wave_sum(s2) * (1.0f / D);
    const float rs = rstd[row];
    float4 o = *reinterpret_cast<float4*>(dx + row * D + lane * 4);
    o.x += rs * (g[0] - s1 - xh[0] * s2);
    o.y += rs * (g[1] - s1 - xh[1] * s2);
    o.z += rs * (g[2] - s1 - xh[2] * s2);
    o.w += rs * (g[3] - s1 - xh[3] * s2);
    *reinterpret_cast<float4*>(dx + row * D + lane * 4) = o;
    if (dxb) *reinterpret_cast<uint2*>(dxb + row * D + lane * 4) = make_uint2(pack_bf16(o.x, o.y), pack_bf16(o.z, o.w));
  }
  if (gamma) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      red[0][w][lane * 4 + j] = dg[j];
      red[1][w][lane * 4 + j] = db[j];
    }
    __syncthreads();
    const int c = threadIdx.x;
    atomicAdd(dgamma + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    atomicAdd(dbeta + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
  }
}

__global__ __launch_bounds__(256) void modln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ ss,
                                                        uint16_t* __restrict__ xhat, uint16_t* __restrict__ xm,
                                                        float* __restrict__ rstd_out, int64_t frames,
                                                        int64_t rows_per_frame, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t rows = frames * rows_per_frame;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t row = wave; row < rows; row += nw) {
    const int64_t f = row / rows_per_frame;
    const float4 v = *reinterpret_cast<const float4*>(x + row * D + lane * 4);
    float xh[4], rstd;
    ln_row(v, eps, xh, rstd);
    const float4 sh = *reinterpret_cast<const float4*>(ss + f * 2 * D + lane * 4);
    const float4 sc = *reinterpret_cast<const float4*>(ss + f * 2 * D + D + lane * 4);
    const float m0 = xh[0] * (1.f + sc.x) + sh.x, m1 = xh[1] * (1.f + sc.y) + sh.y;
    const float m2 = xh[2] * (1.f + sc.z) + sh.z, m3 = xh[3] * (1.f + sc.w) + sh.w;
    *reinterpret_cast<uint2*>(xhat + row * D + lane * 4) = make_uint2(pack_bf16(xh[0], xh[1]), pack_bf16(xh[2], xh[3]));
    *reinterpret_cast<uint2*>(xm + row * D + lane * 4) = make_uint2(pack_bf16(m0, m1), pack_bf16(m2, m3));
    if (lane == 0) rstd_out[row] = rstd;
  }
}

// one workgroup per (b, t) frame: the shift/scale gradients are column sums over the frame's rows
__global__ __launch_bounds__(256) void modln_bwd_kernel(const uint16_t* __restrict__ dxm, const uint16_t* __restrict__ xhat,
                                                        const float* __restrict__ rstd, const float* __restrict__ ss,
                                                        float* __restrict__ dx, float* __restrict__ dss,
                                                        uint16_t* __restrict__ dxb, int64_t rows_per_frame) {
  __shared__ float red[2][4][D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t f = blockIdx.x;
  const float4 sc4 = *reinterpret_cast<const float4*>(ss + f * 2 * D + D + lane * 4);
  const float sc[4] = {1.f + sc4.x, 1.f + sc4.y, 1.f + sc4.z, 1.f + sc4.w};
  float dsh[4] = {0, 0, 0, 0}, dsc[4] = {0, 0, 0, 0};
  for (int64_t i = w; i < rows_per_frame; i += 4) {
    const int64_t row = f * rows_per_frame + i;
    const uint2 a = *reinterpret_cast<const uint2*>(dxm + row * D + lane * 4);
    const uint2 b = *reinterpret_cast<const uint2*>(xhat + row * D + lane * 4);
    const float dy[4] = {bf16_lo(a.x), bf16_hi(a.x), bf16_lo(a.y), bf16_hi(a.y)};
    const float xh[4] = {bf16_lo(b.x), bf16_hi(b.x), bf16_lo(b.y), bf16_hi(b.y)};
    float g[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      g[j] = dy[j] * sc[j];
      s1 += g[j];
      s2 += g[j] * xh[j];
      dsh[j] += dy[j];
      dsc[j] += dy[j] * xh[j];
    }
    s1 = wave_sum(s1) * (1.0f / D);
    s2 = wave_sum(s2) * (1.0f / D);
    const float rs = rstd[row];
    float4 o = *reinterpret_cast<float4*>(dx + row * D + lane * 4);
    o.x += rs * (g[0] - s1 - xh[0] * s2);
    o.y += rs * (g[1] - s1 - xh[1] * s2);
    o.z += rs * (g[2] - s1 - xh[2] * s2);
    o.w += rs * (g[3] - s1 - xh[3] * s2);
    *reinterpret_cast<float4*>(dx + row * D + lane * 4) = o;
    if (dxb) *reinterpret_cast<uint2*>(dxb + row * D + lane * 4) = make_uint2(pack_bf16(o.x, o.y), pack_bf16(o.z, o.w));
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[0][w][lane * 4 + j] = dsh[j];
    red[1][w][lane * 4 + j] = dsc[j];
  }
  __syncthreads();
  const int c = threadIdx.x;
  dss[f * 2 * D + c] = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
  dss[f * 2 * D + D + c] = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
}

inline unsigned stream_grid(int64_t rows) {
  int64_t b = (rows + 3) / 4;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ---------------------------------------------------------------------------------------------- q / k LayerNorm (qk_norm=True)
// attention.py:31-35,44-48: q and k are normalised per head (32 values, eps 1e-5) with ONE affine shared by q, k and all heads, before
// the softmax scale.  One wave per token row of the packed qkv buffer: lane l holds 8 consecutive values of the 512 q | k values
// (4 lanes = one head), statistics by two shuffles.  In place; the pre-norm values are saved for the backward.
__device__ __forceinline__ int64_t qkn_row(int64_t r, int64_t g_rows, int64_t g_stride) {
  return g_rows > 0 ? (r / g_rows) * g_stride + r % g_rows : r;
}
__global__ __launch_bounds__(256) void qknorm_fwd_kernel(uint16_t* __restrict__ qkv, int64_t ld, uint16_t* __restrict__ raw,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                         int64_t rows, int64_t g_rows, int64_t g_stride) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  const int j0 = 8 * (lane & 3);
  float gm[8], bt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) gm[e] = gamma[j0 + e], bt[e] = beta[j0 + e];
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += nw) {
    uint16_t* p = qkv + qkn_row(row, g_rows, g_stride) * ld + lane * 8;
    const uint4 pk = *reinterpret_cast<const uint4*>(p);
    if (raw) *reinterpret_cast<uint4*>(raw + row * 512 + lane * 8) = pk;
    float v[8];
    unpack8(pk, v);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    const float mean = s * (1.0f / 32.0f);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) q = __builtin_fmaf(v[e] - mean, v[e] - mean, q);
    q += __shfl_xor(q, 1, 64);
    q += __shfl_xor(q, 2, 64);
    const float rstd = rsqrtf(q * (1.0f / 32.0f) + eps);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (v[e] - mean) * rstd * gm[e] + bt[e];
    *reinterpret_cast<uint4*>(p) = pack8(v);
  }
}
// dqkv (bf16, in place on its q | k parts): gradient wrt the normalised q / k -> gradient wrt the raw ones; dgamma / dbeta += (fp32 atomics)
__global__ __launch_bounds__(256) void qknorm_bwd_kernel(uint16_t* __restrict__ dqkv, int64_t ld, const uint16_t* __restrict__ raw,
                                                         const float* __restrict__ gamma, float eps, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, int64_t rows) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  const int j0 = 8 * (lane & 3);
  float gm[8], dg[8], db[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) gm[e] = gamma[j0 + e], dg[e] = 0.f, db[e] = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += nw) {
    uint16_t* p = dqkv + row * ld + lane * 8;
    float g[8], v[8];
    unpack8(*reinterpret_cast<const uint4*>(p), g);
    unpack8(*reinterpret_cast<const uint4*>(raw + row * 512 + lane * 8), v);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    const float mean = s * (1.0f / 32.0f);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) q = __builtin_fmaf(v[e] - mean, v[e] - mean, q);
    q += __shfl_xor(q, 1, 64);
    q += __shfl_xor(q, 2, 64);
    const float rstd = rsqrtf(q * (1.0f / 32.0f) + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = (v[e] - mean) * rstd;        // xhat
      dg[e] = __builtin_fmaf(g[e], v[e], dg[e]);
      db[e] += g[e];
      g[e] *= gm[e];                      // d xhat
      s1 += g[e];
      s2 = __builtin_fmaf(g[e], v[e], s2);
    }
    s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64);
    s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64);
    s1 *= 1.0f / 32.0f;
    s2 *= 1.0f / 32.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = rstd * (g[e] - s1 - v[e] * s2);
    *reinterpret_cast<uint4*>(p) = pack8(g);
  }
  // the affine's gradients: lanes with the same (lane & 3) hold the same 8 columns
#pragma unroll
  for (int e = 0; e < 8; ++e) {
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) {
      dg[e] += __shfl_xor(dg[e], o, 64);
      db[e] += __shfl_xor(db[e], o, 64);
    }
  }
  if (lane < 4) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      atomicAdd(dgamma + j0 + e, dg[e]);
      atomicAdd(dbeta + j0 + e, db[e]);
    }
  }
}

}  // namespace

extern "C" int hma_ln_fwd(void* stream, const float* x, void* xhat, float* rstd, int64_t rows, float eps) {
  if (!x || !xhat || !rstd) return HMA_EINVAL;
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(stream_grid(rows)), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)xhat,
                     rstd, rows, eps);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_ln_bwd(void* stream, const void* dxn, const void* xhat, const float* rstd, const float* gamma,
                          float* dx, float* dgamma, float* dbeta, int64_t rows, void* dx_bf16) {
  if (!dxn || !xhat || !rstd || !dx) return HMA_EINVAL;
  if (gamma && (!dgamma || !dbeta)) return HMA_EINVAL;
  if (rows <= 0) return 0;
  int64_t b = (rows + 3) / 4;
  if (b > 1024) b = 1024;  // bounds the dgamma/dbeta atomics to 1024 x 512
  hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dxn,
                     (const uint16_t*)xhat, rstd, gamma, dx, dgamma, dbeta, (uint16_t*)dx_bf16, rows);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_modln_fwd(void* stream, const float* x, const float* ss, void* xhat, void* xm, float* rstd,
                             int64_t frames, int64_t rows_per_frame, float eps) {
  if (!x || !ss || !xhat || !xm || !rstd) return HMA_EINVAL;
  if (frames <= 0 || rows_per_frame <= 0) return 0;
  hipLaunchKernelGGL(modln_fwd_kernel, dim3(stream_grid(frames * rows_per_frame)), dim3(256), 0, (hipStream_t)stream,
                     x, ss, (uint16_t*)xhat, (uint16_t*)xm, rstd, frames, rows_per_frame, eps);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_modln_bwd(void* stream, const void* dxm, const void* xhat, const float* rstd, const float* ss,
                             float* dx, float* dss, int64_t frames, int64_t rows_per_frame, void* dx_bf16) {
  if (!dxm || !xhat || !rstd || !ss || !dx || !dss) return HMA_EINVAL;
  if (frames <= 0 || rows_per_frame <= 0) return 0;
  hipLaunchKernelGGL(modln_bwd_kernel, dim3((unsigned)frames), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)dxm, (const uint16_t*)xhat, rstd, ss, dx, dss, (uint16_t*)dx_bf16, rows_per_frame);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_qknorm_fwd(void* stream, void* qkv, int64_t ld, void* raw, const float* gamma, const float* beta, float eps, int64_t rows,
                              int64_t g_rows, int64_t g_stride) {
  if (!qkv || !gamma || !beta || ld < 512 || (ld & 7)) return HMA_EINVAL;
  if (rows <= 0) return 0;
  int64_t blocks = (rows + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(qknorm_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (uint16_t*)qkv, ld, (uint16_t*)raw, gamma,
                     beta, eps, rows, g_rows, g_stride);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_qknorm_bwd(void* stream, void* dqkv, int64_t ld, const void* raw, const float* gamma, float eps, float* dgamma,
                              float* dbeta, int64_t rows) {
  if (!dqkv || !raw || !gamma || !dgamma || !dbeta || ld < 512 || (ld & 7)) return HMA_EINVAL;
  if (rows <= 0) return 0;
  int64_t blocks = (rows + 3) / 4;
  if (blocks > 1024) blocks = 1024;  // bounds the dgamma / dbeta atomics
  hipLaunchKernelGGL(qknorm_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (uint16_t*)dqkv, ld, (const uint16_t*)raw,
                     gamma, eps, dgamma, dbeta, rows);
  HMA_CHECK_LAUNCH();
  return 0;
}
