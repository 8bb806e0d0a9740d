// Causal temporal attention over the T (<= 16) frames of one (b, s) token column, 8 heads of 32.
//
// The whole problem per column is 16 x 16 scores per head: far below an MFMA tile's worth of
// reuse, and the kernel is bound by streaming qkv (1536 B per token row, rows T apart by n_s rows)
// so it is written on the VALU with packed-bf16 dot products (v_dot2c_f32_bf16): one workgroup =
// one column, one thread = one (t, head) row; the column's 16 x 1536 B of qkv are staged in LDS
// with fully coalesced 1536-B row reads and K/V rows are broadcast-read by the 16 lanes of a head.
//
// Reference: BasicSelfAttention.forward, hma/model/attention.py:37-61 with causal=True (mask fill
// -finfo.max, :52-56) as called at hma/model/st_transformer.py:111 on the "(B S) T C" view of the
// token grid -- here the rearrange (:89, :113) is never materialised: rows are gathered by stride.
#include "hma_common.h"
#include "../../include/hma_hip.h"

using namespace hma;

namespace {

constexpr int TM = 16;    // max frames
constexpr int LD = 768;   // packed qkv row (elements)
constexpr int DM = 256;

__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}
__device__ __forceinline__ void ld16(const uint16_t* p, uint32_t (&w)[16]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint4 v = *reinterpret_cast<const uint4*>(p + i * 8);
    w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
  }
}
__device__ __forceinline__ float dot32(const uint32_t (&a)[16], const uint32_t (&b)[16]) {
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc = dot2(a[i], b[i], acc);
  return acc;
}
__device__ __forceinline__ void axpy32(float (&y)[32], float a, const uint32_t (&x)[16]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    y[2 * i] += a * bf16_lo(x[i]);
    y[2 * i + 1] += a * bf16_hi(x[i]);
  }
}
__device__ __forceinline__ void st32(uint16_t* p, const float (&y)[32], float mul) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint4 v;
    v.x = pack_bf16(y[8 * i] * mul, y[8 * i + 1] * mul);
    v.y = pack_bf16(y[8 * i + 2] * mul, y[8 * i + 3] * mul);
    v.z = pack_bf16(y[8 * i + 4] * mul, y[8 * i + 5] * mul);
    v.w = pack_bf16(y[8 * i + 6] * mul, y[8 * i + 7] * mul);
    *reinterpret_cast<uint4*>(p + i * 8) = v;
  }
}

// rows of the column: global row (b*T + t) * n_s + s
__device__ __forceinline__ void load_column(uint16_t* dst, int dst_ld, const uint16_t* src, int64_t src_ld, int chunks_per_row,
                                            int64_t row0, int64_t row_stride, int T, int tid) {
  for (int c = tid; c < T * chunks_per_row; c += 128) {
    const int t = c / chunks_per_row, ch = c % chunks_per_row;
    *reinterpret_cast<uint4*>(dst + t * dst_ld + ch * 8) =
        *reinterpret_cast<const uint4*>(src + (row0 + t * row_stride) * src_ld + ch * 8);
  }
}
__device__ __forceinline__ void store_column(uint16_t* dst, int64_t dst_ld, const uint16_t* src, int src_ld, int src_col0,
                                             int chunks_per_row, int64_t row0, int64_t row_stride, int T, int tid) {
  for (int c = tid; c < T * chunks_per_row; c += 128) {
    const int t = c / chunks_per_row, ch = c % chunks_per_row;
    *reinterpret_cast<uint4*>(dst + (row0 + t * row_stride) * dst_ld + ch * 8) =
        *reinterpret_cast<const uint4*>(src + t * src_ld + src_col0 + ch * 8);
  }
}

// scaled, masked scores (log2 domain) -> normalised probabilities for query t
__device__ __forceinline__ void causal_softmax(float (&s)[TM], int t, int T) {
  float m = -INFINITY;
#pragma unroll
  for (int tp = 0; tp < TM; ++tp) {
    if (tp > t || tp >= T) s[tp] = -INFINITY;
    m = fmaxf(m, s[tp]);
  }
  float l = 0.f;
#pragma unroll
  for (int tp = 0; tp < TM; ++tp) {
    s[tp] = __builtin_amdgcn_exp2f(s[tp] - m);
    l += s[tp];
  }
  const float inv = 1.0f / l;
#pragma unroll
  for (int tp = 0; tp < TM; ++tp) s[tp] *= inv;
}

__global__ __launch_bounds__(128, 3) void attn_t_fwd_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ o, int T,
                                                         int n_s, float c_log2, int64_t qkv_batch_rows) {
  __shared__ __attribute__((aligned(16))) uint16_t sm[TM * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = lane & 15, h = wave * 4 + (lane >> 4);
  const int64_t col = blockIdx.x;  // b * n_s + s
  const int64_t b = col / n_s, s_idx = col % n_s;
  const int64_t row0 = b * T * n_s + s_idx;
  load_column(sm, LD, qkv, LD, LD / 8, b * qkv_batch_rows + s_idx, n_s, T, tid);
  __syncthreads();
  float out[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) out[i] = 0.f;
  if (t < T) {
    uint32_t q[16];
    ld16(sm + t * LD + h * 32, q);
    float s[TM];
#pragma unroll
    for (int tp = 0; tp < TM; ++tp) {
      s[tp] = 0.f;
      if (tp < T) {
        uint32_t k[16];
        ld16(sm + tp * LD + DM + h * 32, k);
        s[tp] = dot32(q, k) * c_log2;
      }
    }
    causal_softmax(s, t, T);
#pragma unroll
    for (int tp = 0; tp < TM; ++tp) {
      if (tp < T) {
        uint32_t v[16];
        ld16(sm + tp * LD + 2 * DM + h * 32, v);
        axpy32(out, s[tp], v);
      }
    }
  }
  // each thread overwrites only its own q slot, which no other thread reads
  if (t < T) st32(sm + t * LD + h * 32, out, 1.0f);
  __syncthreads();
  store_column(o, DM, sm, LD, 0, DM / 8, row0, n_s, T, tid);
}

__global__ __launch_bounds__(128) void attn_t_bwd_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ o,
                                                         const uint16_t* __restrict__ d_o, uint16_t* __restrict__ dqkv,
                                                         int T, int n_s, float c_log2, float scale) {
  __shared__ __attribute__((aligned(16))) uint16_t sm[TM * LD];     // qkv, later dq|dk|dv
  __shared__ __attribute__((aligned(16))) uint16_t sg[TM * DM];     // dO
  __shared__ __attribute__((aligned(16))) float sp[8 * TM * TM];    // P[h][t][tp]
  __shared__ __attribute__((aligned(16))) float sd[8 * TM * TM];    // dS[h][t][tp]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = lane & 15, h = wave * 4 + (lane >> 4);
  const int64_t col = blockIdx.x;
  const int64_t b = col / n_s, s_idx = col % n_s;
  const int64_t row0 = b * T * n_s + s_idx;
  load_column(sm, LD, qkv, LD, LD / 8, row0, n_s, T, tid);
  load_column(sg, DM, d_o, DM, DM / 8, row0, n_s, T, tid);
  __syncthreads();

  float dq[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) dq[i] = 0.f;
  float p[TM], ds[TM];
#pragma unroll
  for (int tp = 0; tp < TM; ++tp) { p[tp] = 0.f; ds[tp] = 0.f; }
  if (t < T) {
    uint32_t q[16], g[16], oo[16];
    ld16(sm + t * LD + h * 32, q);
    ld16(sg + t * DM + h * 32, g);
    ld16(o + (row0 + (int64_t)t * n_s) * DM + h * 32, oo);
    const float delta = dot32(g, oo);
#pragma unroll
    for (int tp = 0; tp < TM; ++tp) {
      if (tp < T) {
        uint32_t k[16];
        ld16(sm + tp * LD + DM + h * 32, k);
        p[tp] = dot32(q, k) * c_log2;
      }
    }
    causal_softmax(p, t, T);
#pragma unroll
    for (int tp = 0; tp < TM; ++tp) {
      if (tp < T) {
        uint32_t v[16], k[16];
        ld16(sm + tp * LD + 2 * DM + h * 32, v);
        ds[tp] = p[tp] * (dot32(g, v) - delta);
        ld16(sm + tp * LD + DM + h * 32, k);
        axpy32(dq, ds[tp], k);
      }
    }
  }
  {
    float* pp = sp + (h * TM + t) * TM;
    float* dd = sd + (h * TM + t) * TM;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(pp + 4 * i) = make_float4(p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]);
      *reinterpret_cast<float4*>(dd + 4 * i) = make_float4(ds[4 * i], ds[4 * i + 1], ds[4 * i + 2], ds[4 * i + 3]);
    }
  }
  __syncthreads();
  // pass 2: this thread now owns key/value row tp = t
  float dk[32], dv[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) { dk[i] = 0.f; dv[i] = 0.f; }
  if (t < T) {
#pragma unroll
    for (int tq = 0; tq < TM; ++tq) {
      if (tq < T) {
        const float pv = sp[(h * TM + tq) * TM + t];
        const float dsv = sd[(h * TM + tq) * TM + t];
        uint32_t q[16], g[16];
        ld16(sm + tq * LD + h * 32, q);
        ld16(sg + tq * DM + h * 32, g);
        axpy32(dk, dsv, q);
        axpy32(dv, pv, g);
      }
    }
  }
  __syncthreads();  // every read of q/k/v is done: reuse the slab for the gradients
  if (t < T) {
    st32(sm + t * LD + h * 32, dq, scale);
    st32(sm + t * LD + DM + h * 32, dk, scale);
    st32(sm + t * LD + 2 * DM + h * 32, dv, 1.0f);
  }
  __syncthreads();
  store_column(dqkv, LD, sm, LD, 0, LD / 8, row0, n_s, T, tid);
}

// Incremental decode: only frame t_query is new.  One lane per (column, head): q from the cache row of frame
// t_query, K/V rows of frames 0..t_query streamed straight from the per-layer cache (64 B per head and frame).
// Rows of the cache are (b, t, s) with T_cache frames per sample; o holds frame t_query only, rows (b, s).
__global__ __launch_bounds__(64) void attn_t_decode_kernel(const uint16_t* __restrict__ cache, uint16_t* __restrict__ o,
                                                           int64_t cols, int t_query, int T_cache, int n_s, float c_log2) {
  const int lane = threadIdx.x;
  const int64_t col = (int64_t)blockIdx.x * 8 + (lane >> 3);
  const int h = lane & 7;
  if (col >= cols) return;
  const int64_t b = col / n_s, s_idx = col % n_s;
  const uint16_t* base = cache + ((b * T_cache) * n_s + s_idx) * LD + h * 32;
  uint32_t q[16];
  ld16(base + (int64_t)t_query * n_s * LD, q);
  float s[TM];
#pragma unroll
  for (int tp = 0; tp < TM; ++tp) {
    s[tp] = 0.f;
    if (tp <= t_query) {
      uint32_t k[16];
      ld16(base + (int64_t)tp * n_s * LD + DM, k);
      s[tp] = dot32(q, k) * c_log2;
    }
  }
  causal_softmax(s, t_query, t_query + 1);
  float out[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) out[i] = 0.f;
#pragma unroll
  for (int tp = 0; tp < TM; ++tp) {
    if (tp <= t_query) {
      uint32_t v[16];
      ld16(base + (int64_t)tp * n_s * LD + 2 * DM, v);
      axpy32(out, s[tp], v);
    }
  }
  st32(o + col * DM + h * 32, out, 1.0f);
}

constexpr float LOG2E = 1.4426950408889634f;

}  // namespace

extern "C" int hma_attn_temporal_fwd(void* stream, const void* qkv, void* o, int64_t batch, int32_t T, int32_t n_s,
                                     float scale) {
  if (!qkv || !o) return HMA_EINVAL;
  if (T < 1 || T > TM || n_s < 1) return HMA_EINVAL;
  if (batch <= 0) return 0;
  hipLaunchKernelGGL(attn_t_fwd_kernel, dim3((unsigned)(batch * n_s)), dim3(128), 0, (hipStream_t)stream,
                     (const uint16_t*)qkv, (uint16_t*)o, (int)T, (int)n_s, scale * LOG2E, (int64_t)T * n_s);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_attn_temporal_cached(void* stream, const void* qkv_cache, void* o, int64_t batch, int32_t T, int32_t t_query,
                                        int32_t T_cache, int32_t n_s, float scale) {
  if (!qkv_cache || !o) return HMA_EINVAL;
  if (T_cache < 1 || T_cache > TM || n_s < 1 || T < 1 || T > T_cache) return HMA_EINVAL;
  if (batch <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (t_query < 0) {  // prefill: all T frames attend causally, keys read from the cache layout
    hipLaunchKernelGGL(attn_t_fwd_kernel, dim3((unsigned)(batch * n_s)), dim3(128), 0, s, (const uint16_t*)qkv_cache,
                       (uint16_t*)o, (int)T, (int)n_s, scale * LOG2E, (int64_t)T_cache * n_s);
  } else {
    if (t_query >= T_cache) return HMA_EINVAL;
    hipLaunchKernelGGL(attn_t_decode_kernel, dim3((unsigned)((batch * n_s + 7) / 8)), dim3(64), 0, s,
                       (const uint16_t*)qkv_cache, (uint16_t*)o, batch * n_s, (int)t_query, (int)T_cache, (int)n_s,
                       scale * LOG2E);
  }
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_attn_temporal_bwd(void* stream, const void* qkv, const void* o, const void* d_o, void* dqkv,
                                     int64_t batch, int32_t T, int32_t n_s, float scale) {
  if (!qkv || !o || !d_o || !dqkv) return HMA_EINVAL;
  if (T < 1 || T > TM || n_s < 1) return HMA_EINVAL;
  if (batch <= 0) return 0;
  hipLaunchKernelGGL(attn_t_bwd_kernel, dim3((unsigned)(batch * n_s)), dim3(128), 0, (hipStream_t)stream,
                     (const uint16_t*)qkv, (const uint16_t*)o, (const uint16_t*)d_o, (uint16_t*)dqkv, (int)T, (int)n_s,
                     scale * LOG2E, scale);
  HMA_CHECK_LAUNCH();
  return 0;
}
