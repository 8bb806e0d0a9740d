// Causal temporal attention over the T (<= 16) frames of one (b, s) token column, 8 heads of 32.
//
// The whole problem per column is 16 x 16 scores per head, and the kernel should be bound by
// streaming qkv (1536 B per token row, rows of a column n_s rows apart): one workgroup = one
// column, whose 16 x 1536 B of qkv are staged in LDS with fully coalesced 1536-B row reads; each of
// its two waves then walks four heads with 16x16 MFMAs (see the note above the kernels).  Only the
// single-query decode kernel stays on the VALU (v_dot2c_f32_bf16), where there is no tile to form.
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

// ---------------------------------------------------------------------------------------------------------
// MFMA formulation.  One column and head is exactly one v_mfma_f32_16x16x32_bf16: S^T = K Q^T contracts the
// 32 head channels and yields the 16 x 16 score tile with lane&15 = query t, registers = keys tp = 4*(lane>>4)+r,
// so the causal softmax is 4 registers + two cross-group shuffles per lane.  Products that contract over time
// (P V, dS K, dS^T Q, P^T dO) use v_mfma_f32_16x16x16_bf16 with the probability tile as the B operand straight
// from those registers and the [time][channel] matrix gathered from LDS as A (4 x ds_read_u16 per 16 channels);
// they are computed transposed (channels x time) so each lane ends up with 4 consecutive channels of one token
// row -> 8-byte LDS writes into the slab, which leaves the CU through the same coalesced 1536-B row stores.
// The VALU version of this kernel was instruction-bound (~3600 VALU ops per lane and column in backward).
// ---------------------------------------------------------------------------------------------------------
constexpr int QLD = LD + 8;   // padded LDS rows (16-B aligned, breaks the 1536-B bank period)
constexpr int GLD = DM + 8;

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__device__ __forceinline__ f32x4_t mfma16k32(const bf16x8_t& a, const bf16x8_t& b) {
  const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, z, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t mfma16k16(const s16x4_t& a, const s16x4_t& b) {
  const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, z, 0, 0, 0);
}
__device__ __forceinline__ s16x4_t pack4(float a, float b, float c, float d) {
  const uint2 v = make_uint2(pack_bf16(a, b), pack_bf16(c, d));
  return __builtin_bit_cast(s16x4_t, v);
}
// A operand of a time-contracting product: element (m = channel c, k = time 4g + j) of a [time][channel] slab
__device__ __forceinline__ s16x4_t gather_time(const uint16_t* base, int ld) {
  s16x4_t r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = (short)base[j * ld];
  return r;
}
__device__ __forceinline__ float group_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// stage T rows of CHUNKS 16-B pieces into a padded slab, zero-filling rows T..15 (they feed MFMAs);
// all of a thread's loads are issued before the first LDS write so that the column is in flight at once
template <int CHUNKS>
__device__ __forceinline__ void stage_rows(uint16_t* dst, int dst_ld, const uint16_t* src, int64_t src_ld, int64_t row0,
                                           int64_t row_stride, int T, int tid) {
  constexpr int N = TM * CHUNKS / 128;
  uint4 v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int c = tid + i * 128;
    const int t = c / CHUNKS, ch = c % CHUNKS;
    v[i] = make_uint4(0u, 0u, 0u, 0u);
    if (t < T) v[i] = *reinterpret_cast<const uint4*>(src + (row0 + t * row_stride) * src_ld + ch * 8);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int c = tid + i * 128;
    const int t = c / CHUNKS, ch = c % CHUNKS;
    *reinterpret_cast<uint4*>(dst + t * dst_ld + ch * 8) = v[i];
  }
}

__global__ __launch_bounds__(128, 3) void attn_t_fwd_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ o, int T,
                                                         int n_s, float c_log2, int64_t qkv_batch_rows) {
  __shared__ __attribute__((aligned(16))) uint16_t sm[TM * QLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int64_t col = blockIdx.x;  // b * n_s + s
  const int64_t b = col / n_s, s_idx = col % n_s;
  const int64_t row0 = b * T * n_s + s_idx;
  stage_rows<LD / 8>(sm, QLD, qkv, LD, b * qkv_batch_rows + s_idx, n_s, T, tid);
  __syncthreads();
#pragma unroll 1
  for (int hh = 0; hh < 4; ++hh) {
    const int h = wave * 4 + hh;
    const uint16_t* qh = sm + c * QLD + h * 32 + 8 * g;
    const bf16x8_t aQ = *reinterpret_cast<const bf16x8_t*>(qh);
    const bf16x8_t aK = *reinterpret_cast<const bf16x8_t*>(qh + DM);
    f32x4_t st = mfma16k32(aK, aQ);  // [tp = 4g + r][t = c]
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      st[r] = (4 * g + r <= c) ? st[r] * c_log2 : -INFINITY;
      m = fmaxf(m, st[r]);
    }
    m = group_max(m);
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      st[r] = __builtin_amdgcn_exp2f(st[r] - m);
      l += st[r];
    }
    const float inv = 1.0f / group_sum(l);
    const s16x4_t pT = pack4(st[0] * inv, st[1] * inv, st[2] * inv, st[3] * inv);  // B: k = tp, n = t
    f32x4_t ot[2];
#pragma unroll
    for (int dd = 0; dd < 2; ++dd) {
      const s16x4_t vT = gather_time(sm + (4 * g) * QLD + 2 * DM + h * 32 + 16 * dd + c, QLD);
      ot[dd] = mfma16k16(vT, pT);  // [d = 16dd + 4g + r][t = c]
    }
    __builtin_amdgcn_wave_barrier();  // this wave's reads of head h are done; its q slot becomes the output
#pragma unroll
    for (int dd = 0; dd < 2; ++dd)
      *reinterpret_cast<uint2*>(sm + c * QLD + h * 32 + 16 * dd + 4 * g) =
          make_uint2(pack_bf16(ot[dd][0], ot[dd][1]), pack_bf16(ot[dd][2], ot[dd][3]));
  }
  __syncthreads();
  store_column(o, DM, sm, QLD, 0, DM / 8, row0, n_s, T, tid);
}

__global__ __launch_bounds__(128, 2) void attn_t_bwd_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ d_o,
                                                         uint16_t* __restrict__ dqkv, int T, int n_s, float c_log2,
                                                         float scale) {
  __shared__ __attribute__((aligned(16))) uint16_t sm[TM * QLD];   // qkv, later dq|dk|dv
  __shared__ __attribute__((aligned(16))) uint16_t sg[TM * GLD];   // dO
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int64_t col = blockIdx.x;
  const int64_t b = col / n_s, s_idx = col % n_s;
  const int64_t row0 = b * T * n_s + s_idx;
  stage_rows<LD / 8>(sm, QLD, qkv, LD, row0, n_s, T, tid);
  stage_rows<DM / 8>(sg, GLD, d_o, DM, row0, n_s, T, tid);
  __syncthreads();
#pragma unroll 1
  for (int hh = 0; hh < 4; ++hh) {
    const int h = wave * 4 + hh;
    const uint16_t* qh = sm + c * QLD + h * 32 + 8 * g;
    const bf16x8_t aQ = *reinterpret_cast<const bf16x8_t*>(qh);
    const bf16x8_t aK = *reinterpret_cast<const bf16x8_t*>(qh + DM);
    const bf16x8_t aV = *reinterpret_cast<const bf16x8_t*>(qh + 2 * DM);
    const bf16x8_t aG = *reinterpret_cast<const bf16x8_t*>(sg + c * GLD + h * 32 + 8 * g);
    // orientation 1: lane = query t (c), registers = keys tp = 4g + r
    f32x4_t st = mfma16k32(aK, aQ);
    const f32x4_t dpt = mfma16k32(aV, aG);
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      st[r] = (4 * g + r <= c) ? st[r] * c_log2 : -INFINITY;
      m = fmaxf(m, st[r]);
    }
    m = group_max(m);
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      st[r] = __builtin_amdgcn_exp2f(st[r] - m);
      l += st[r];
    }
    const float inv = 1.0f / group_sum(l);
    float delta = 0.f;  // sum_tp P dP  ( = dO . O, the softmax-backward row term)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      st[r] *= inv;
      delta += st[r] * dpt[r];
    }
    delta = group_sum(delta);
    const s16x4_t dsT = pack4(st[0] * (dpt[0] - delta), st[1] * (dpt[1] - delta), st[2] * (dpt[2] - delta),
                              st[3] * (dpt[3] - delta));  // B: k = tp, n = t
    // orientation 2: lane = key tp (c), registers = queries t = 4g + r
    const f32x4_t s2 = mfma16k32(aQ, aK);
    const f32x4_t dp2 = mfma16k32(aG, aV);
    float p2[4], ds2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tq = 4 * g + r;
      const float mq = __shfl(m, tq, 64), iq = __shfl(inv, tq, 64), dq_ = __shfl(delta, tq, 64);
      p2[r] = (c <= tq) ? __builtin_amdgcn_exp2f(s2[r] * c_log2 - mq) * iq : 0.f;
      ds2[r] = p2[r] * (dp2[r] - dq_);
    }
    const s16x4_t pB = pack4(p2[0], p2[1], p2[2], p2[3]);       // B: k = t, n = tp
    const s16x4_t dsB = pack4(ds2[0], ds2[1], ds2[2], ds2[3]);
    f32x4_t dq[2], dk[2], dv[2];
#pragma unroll
    for (int dd = 0; dd < 2; ++dd) {
      const int off = (4 * g) * QLD + h * 32 + 16 * dd + c;
      const s16x4_t qT = gather_time(sm + off, QLD);
      const s16x4_t kT = gather_time(sm + off + DM, QLD);
      const s16x4_t gT = gather_time(sg + (4 * g) * GLD + h * 32 + 16 * dd + c, GLD);
      dq[dd] = mfma16k16(kT, dsT);  // dQ^T [d][t = c]
      dk[dd] = mfma16k16(qT, dsB);  // dK^T [d][tp = c]
      dv[dd] = mfma16k16(gT, pB);   // dV^T [d][tp = c]
    }
    __builtin_amdgcn_wave_barrier();  // every read of head h by this wave is done: reuse its slots for the gradients
#pragma unroll
    for (int dd = 0; dd < 2; ++dd) {
      uint16_t* dst = sm + c * QLD + h * 32 + 16 * dd + 4 * g;
      *reinterpret_cast<uint2*>(dst) =
          make_uint2(pack_bf16(dq[dd][0] * scale, dq[dd][1] * scale), pack_bf16(dq[dd][2] * scale, dq[dd][3] * scale));
      *reinterpret_cast<uint2*>(dst + DM) =
          make_uint2(pack_bf16(dk[dd][0] * scale, dk[dd][1] * scale), pack_bf16(dk[dd][2] * scale, dk[dd][3] * scale));
      *reinterpret_cast<uint2*>(dst + 2 * DM) = make_uint2(pack_bf16(dv[dd][0], dv[dd][1]), pack_bf16(dv[dd][2], dv[dd][3]));
    }
  }
  __syncthreads();
  store_column(dqkv, LD, sm, QLD, 0, LD / 8, row0, n_s, T, tid);
}

// Incremental decode: only frame t_query is new.  One WAVE per column: q, and the K and V rows of frames 0..t_query, are read
// as whole 512-byte rows (8 bytes per lane: lane 8 h + i holds channels 4 i .. 4 i + 3 of head h), every load of a column in
// flight at once; a head's score is a 4-channel partial dot reduced over its 8 lanes, the softmax runs redundantly in them, and
// each lane accumulates its 4 output channels.  (One lane per (column, head) with 64-byte pieces -- the first version -- had every
// load instruction touch 64 cache lines for 1 KB.)  Rows of the cache are (b, t, s) with T_cache frames per sample; o holds frame
// t_query only, rows (b, s).
__global__ __launch_bounds__(256) void attn_t_decode_kernel(const uint16_t* __restrict__ cache, uint16_t* __restrict__ o,
                                                            int64_t cols, int t_query, int T_cache, int n_s, float c_log2) {
  const int lane = threadIdx.x & 63;
  const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (col >= cols) return;
  const int64_t b = col / n_s, s_idx = col % n_s;
  const int64_t fs = (int64_t)n_s * LD;
  const uint16_t* base = cache + ((b * T_cache) * n_s + s_idx) * LD + lane * 4;
  const uint2 qw = *reinterpret_cast<const uint2*>(base + (int64_t)t_query * fs);
  uint2 kw[TM], vw[TM];
#pragma unroll
  for (int tp = 0; tp < TM; ++tp) {
    kw[tp] = make_uint2(0u, 0u), vw[tp] = make_uint2(0u, 0u);
    if (tp <= t_query) {
      kw[tp] = *reinterpret_cast<const uint2*>(base + (int64_t)tp * fs + DM);
      vw[tp] = *reinterpret_cast<const uint2*>(base + (int64_t)tp * fs + 2 * DM);
    }
  }
  const float q0 = bf16_lo(qw.x), q1 = bf16_hi(qw.x), q2 = bf16_lo(qw.y), q3 = bf16_hi(qw.y);
  float s[TM];
#pragma unroll
  for (int tp = 0; tp < TM; ++tp) {
    float d = q0 * bf16_lo(kw[tp].x) + q1 * bf16_hi(kw[tp].x) + q2 * bf16_lo(kw[tp].y) + q3 * bf16_hi(kw[tp].y);
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 4, 64);
    s[tp] = d * c_log2;
  }
  causal_softmax(s, t_query, t_query + 1);
  float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
  for (int tp = 0; tp < TM; ++tp) {
    const float pr = from_bf16(to_bf16(s[tp]));  // P enters the PV product as bf16, as in the MFMA kernels
    o0 += pr * bf16_lo(vw[tp].x), o1 += pr * bf16_hi(vw[tp].x), o2 += pr * bf16_lo(vw[tp].y), o3 += pr * bf16_hi(vw[tp].y);
  }
  *reinterpret_cast<uint2*>(o + col * DM + lane * 4) = make_uint2(pack_bf16(o0, o1), pack_bf16(o2, o3));
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
    hipLaunchKernelGGL(attn_t_decode_kernel, dim3((unsigned)((batch * n_s + 3) / 4)), dim3(256), 0, s,
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
                     (const uint16_t*)qkv, (const uint16_t*)d_o, (uint16_t*)dqkv, (int)T, (int)n_s, scale * LOG2E, scale);
  HMA_CHECK_LAUNCH();
  return 0;
}
