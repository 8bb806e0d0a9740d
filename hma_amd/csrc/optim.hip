// Flat-buffer optimizer kernels: gradient square-norm, clip + AdamW (+ bf16 weight copy), casts.
//
// Reference: accelerator.clip_grad_norm_(model.parameters(), 1.0) + MuAdamW.step() at
// hma/train_multi.py:593-598, 900-922 (mup width_mult == 1 at d_model 256 => torch.optim.AdamW).
// All parameters live in ONE flat fp32 buffer (ranges per weight-decay group / domain), so the
// update is a handful of HBM-streaming launches instead of one tiny kernel per tensor.
#include "hma_common.h"
#include "../../include/hma_hip.h"

#include <cmath>

using namespace hma;

namespace {

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  const int64_t n4 = n / 4;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) acc += g[i] * g[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, uint16_t* __restrict__ pb, int64_t n, float lr,
                                                    float b1, float b2, float eps, float wd, float inv_bc1,
                                                    float inv_sqrt_bc2, const float* __restrict__ sqnorm, float max_norm,
                                                    const uint8_t* __restrict__ flags, int32_t* __restrict__ step_pair, int parity) {
  float coef = 1.f;
  bool finite = true;
  if (sqnorm) {
    const float sq = *sqnorm;
    finite = sq == sq && sq < INFINITY;  // a NaN / Inf anywhere in the (already all-reduced) gradients lands here on every rank
    if (max_norm > 0.f) {
      const float c = max_norm / (sqrtf(sq) + 1e-6f);
      coef = c < 1.f ? c : 1.f;
    }
  }
  if (step_pair) {
    // Device-side update count of this range: read slot `parity`, publish to the other slot (no launch reads the slot
    // another block of the same launch writes).  A skipped step does not count, so Adam's bias correction stays that
    // of the updates actually applied.
    const int step = step_pair[parity] + 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) step_pair[parity ^ 1] = finite ? step : step - 1;
    inv_bc1 = 1.0f / (1.0f - powf(b1, (float)step));
    inv_sqrt_bc2 = 1.0f / sqrtf(1.0f - powf(b2, (float)step));
  }
  if (!finite) return;  // non-finite loss / gradients: leave weights and moments untouched (train_multi.py:572-583)
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float wdi = wd;
    if (flags) {  // one flag per 64 elements: 0 = frozen, 1 = no weight decay, 2 = weight decay
      const uint8_t f = flags[i >> 6];
      if (f == 0) continue;
      wdi = f == 2 ? wd : 0.f;
    }
    const float gi = g[i] * coef;
    float pi = p[i] * (1.f - lr * wdi);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    pi -= (lr * inv_bc1) * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
    if (pb) pb[i] = to_bf16(pi);
  }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = to_bf16(src[i]);
}

// dst[b][c][r] = bf16(src[b][r][c]); 32x32 tiles through LDS
__device__ __forceinline__ void transpose_cast_body(const float* __restrict__ src, uint16_t* __restrict__ dst, int rows, int cols,
                                                    int64_t src_stride, int64_t dst_stride, int bx, int by, int64_t bz) {
  __shared__ float tile[32][33];
  const float* s = src + bz * src_stride;
  uint16_t* d = dst + bz * dst_stride;
  const int r0 = by * 32, c0 = bx * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? s[(int64_t)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) d[(int64_t)c * rows + r] = to_bf16(tile[tx][i]);
  }
}
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int rows,
                                                             int cols, int64_t src_stride, int64_t dst_stride) {
  transpose_cast_body(src, dst, rows, cols, src_stride, dst_stride, blockIdx.x, blockIdx.y, blockIdx.z);
}
// hma_transpose_cast_bf16_multi: the jobs' (tiles across, tiles down, batch) grids laid end to end in blockIdx.x (b0 = prefix sums)
constexpr int PACK_JOBS = 24;
struct pack_jobs {
  hma_pack_job_t j[PACK_JOBS];
  int b0[PACK_JOBS + 1];
  int n;
};
__global__ __launch_bounds__(256) void transpose_cast_multi_kernel(pack_jobs jobs) {
  int k = 0;
  while (k + 1 < jobs.n && (int)blockIdx.x >= jobs.b0[k + 1]) ++k;
  const hma_pack_job_t& j = jobs.j[k];
  const int nx = (j.cols + 31) / 32, ny = (j.rows + 31) / 32, rem = (int)blockIdx.x - jobs.b0[k];
  transpose_cast_body(j.src, reinterpret_cast<uint16_t*>(j.dst), j.rows, j.cols, j.src_batch_stride, j.dst_batch_stride, rem % nx,
                      rem / nx % ny, rem / (nx * ny));
}

__global__ __launch_bounds__(256) void dropout_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t n,
                                                           float p, const uint32_t* __restrict__ seed_dev, int salt) {
  const uint32_t seed = *seed_dev, th = drop_thresh(p);
  const float sc = drop_scale(p);
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    dst[i] = to_bf16(drop_keep(seed, salt, i, th) ? src[i] * sc : 0.f);
}

inline unsigned flat_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int hma_sqnorm(void* stream, const float* g, int64_t n, float* out) {
  if (!g || !out) return HMA_EINVAL;
  if (n <= 0) return 0;
  if ((reinterpret_cast<uintptr_t>(g) & 15) != 0) return HMA_EINVAL;
  int64_t b = (n / 4 + 255) / 256;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, g, n, out);
  HMA_CHECK_LAUNCH();
  return 0;
}

static int adamw_launch(void* stream, float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                        float beta1, float beta2, float eps, float weight_decay, int32_t step, const float* sqnorm,
                        float max_norm, const uint8_t* flags, int32_t* step_pair, int32_t parity) {
  if (!p || !g || !m || !v || (step < 1 && !step_pair) || (step_pair && (parity < 0 || parity > 1))) return HMA_EINVAL;
  if (n <= 0) return 0;
  const double st = step_pair ? 1.0 : (double)step;
  const double bc1 = 1.0 - std::pow((double)beta1, st);
  const double bc2 = 1.0 - std::pow((double)beta2, st);
  hipLaunchKernelGGL(adamw_kernel, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (uint16_t*)p_bf16, n, lr,
                     beta1, beta2, eps, weight_decay, (float)(1.0 / bc1), (float)(1.0 / std::sqrt(bc2)), sqnorm, max_norm, flags,
                     step_pair, (int)parity);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_adamw(void* stream, float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int32_t step, const float* sqnorm,
                         float max_norm, const uint8_t* flags) {
  return adamw_launch(stream, p, g, m, v, p_bf16, n, lr, beta1, beta2, eps, weight_decay, step, sqnorm, max_norm, flags, nullptr, 0);
}

extern "C" int hma_adamw_counted(void* stream, float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, int32_t* step_pair, int32_t parity,
                                 const float* sqnorm, float max_norm, const uint8_t* flags) {
  if (!step_pair) return HMA_EINVAL;
  return adamw_launch(stream, p, g, m, v, p_bf16, n, lr, beta1, beta2, eps, weight_decay, 0, sqnorm, max_norm, flags, step_pair,
                      parity);
}

extern "C" int hma_cast_bf16(void* stream, const float* src, void* dst, int64_t n) {
  if (!src || !dst) return HMA_EINVAL;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(cast_kernel, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, src, (uint16_t*)dst, n);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_transpose_cast_bf16(void* stream, const float* src, void* dst, int32_t rows, int32_t cols, int32_t batch,
                                       int64_t src_stride, int64_t dst_stride) {
  if (!src || !dst || rows <= 0 || cols <= 0) return HMA_EINVAL;
  if (batch <= 0) return 0;
  const dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32), (unsigned)batch);
  hipLaunchKernelGGL(transpose_cast_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, (uint16_t*)dst, (int)rows, (int)cols,
                     src_stride, dst_stride);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_transpose_cast_bf16_multi(void* stream, const hma_pack_job_t* jobs, int32_t njobs) {
  if (njobs < 0 || (njobs > 0 && !jobs)) return HMA_EINVAL;
  for (int i = 0; i < njobs; ++i)
    if (!jobs[i].src || !jobs[i].dst || jobs[i].rows <= 0 || jobs[i].cols <= 0 || jobs[i].batch < 0) return HMA_EINVAL;
  for (int i0 = 0; i0 < njobs; i0 += PACK_JOBS) {
    pack_jobs pj;
    pj.n = njobs - i0 < PACK_JOBS ? njobs - i0 : PACK_JOBS;
    int64_t blocks = 0;
    for (int i = 0; i < pj.n; ++i) {
      pj.j[i] = jobs[i0 + i];
      pj.b0[i] = (int)blocks;
      blocks += (int64_t)((pj.j[i].cols + 31) / 32) * ((pj.j[i].rows + 31) / 32) * pj.j[i].batch;
      if (blocks > INT32_MAX) return HMA_EINVAL;
    }
    pj.b0[pj.n] = (int)blocks;
    if (blocks == 0) continue;
    hipLaunchKernelGGL(transpose_cast_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pj);
    HMA_CHECK_LAUNCH();
  }
  return 0;
}

// One wave per weight row: Wf[n][:] = bf16(W[n][:] * gamma), bf[n] = bias[n] + W[n][:] . beta  (cols a multiple of 4)
__global__ __launch_bounds__(256) void fold_ln_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ bias,
                                                      uint16_t* __restrict__ Wf, float* __restrict__ bf, int rows, int cols,
                                                      int64_t in_stride, int64_t wf_stride, int64_t bf_stride) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= rows) return;
  const int64_t b = blockIdx.y;
  const float* w = W + b * in_stride + (int64_t)n * cols;
  const float* g = gamma + b * in_stride;
  const float* be = beta + b * in_stride;
  uint16_t* o = Wf + b * wf_stride + (int64_t)n * cols;
  float dot = 0.f;
  for (int k = lane * 4; k < cols; k += 256) {
    const float4 x = *reinterpret_cast<const float4*>(w + k);
    const float4 gg = *reinterpret_cast<const float4*>(g + k);
    const float4 bb = *reinterpret_cast<const float4*>(be + k);
    dot += x.x * bb.x + x.y * bb.y + x.z * bb.z + x.w * bb.w;
    *reinterpret_cast<uint2*>(o + k) = make_uint2(pack_bf16(x.x * gg.x, x.y * gg.y), pack_bf16(x.z * gg.z, x.w * gg.w));
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) dot += __shfl_xor(dot, s);
  if (lane == 0) bf[b * bf_stride + n] = dot + (bias ? bias[b * in_stride + n] : 0.f);
}

extern "C" int hma_fold_ln_bf16(void* stream, const float* W, const float* gamma, const float* beta, const float* bias, void* Wf,
                                float* bf, int32_t rows, int32_t cols, int32_t batch, int64_t in_stride, int64_t wf_stride,
                                int64_t bf_stride) {
  if (!W || !gamma || !beta || !Wf || !bf || rows <= 0 || cols <= 0 || (cols & 3)) return HMA_EINVAL;
  if (batch <= 0) return 0;
  hipLaunchKernelGGL(fold_ln_kernel, dim3((unsigned)((rows + 3) / 4), (unsigned)batch), dim3(256), 0, (hipStream_t)stream, W, gamma,
                     beta, bias, (uint16_t*)Wf, bf, (int)rows, (int)cols, in_stride, wf_stride, bf_stride);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_abi_version(void) { return 0x484d4104; }

extern "C" int hma_dropout_bf16(void* stream, const float* src, void* dst, int64_t rows, int32_t cols, float p, const uint32_t* seed_dev,
                                int32_t salt) {
  if (!src || !dst || !seed_dev || !(p > 0.f && p < 1.f) || cols < 1) return HMA_EINVAL;
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(dropout_bf16_kernel, dim3(flat_grid(rows * cols)), dim3(256), 0, (hipStream_t)stream, src, (uint16_t*)dst, rows * cols, p,
                     seed_dev, (int)salt);
  HMA_CHECK_LAUNCH();
  return 0;
}
