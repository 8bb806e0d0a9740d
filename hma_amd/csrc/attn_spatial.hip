// Spatial (per-frame, bidirectional) multi-head attention for gfx950, head_dim 32, n = 256/320
// tokens per frame: the whole K/V of one (frame, head) lives in LDS (20 KB each) and a 32-query
// tile walks the keys in two register-resident chunks (one online-softmax merge).
//
// All three kernels use the "swapped" product S^T = K Q^T (v_mfma_f32_32x32x16_bf16): the MFMA
// result puts one query (fwd, dq) or one key (dkv) per lane and 16 of the other index per lane in
// registers, so the softmax reduction is in-lane + one lane^32 exchange, and the packed bf16
// probabilities ARE the B operand of the second product (no LDS round trip): the contraction index
// of the second MFMA is permuted (kappa below) to match how the first MFMA left it in registers.
//
// Reference: BasicSelfAttention.forward, hma/model/attention.py:37-61 (causal=False), called from
// STBlock.forward hma/model/st_transformer.py:85-86; backward = its autograd mirror.
#include <type_traits>

#include "hma_common.h"
#include "../../include/hma_hip.h"

using namespace hma;

namespace {

constexpr int HD = 32;       // head dim
constexpr int NH = 8;        // heads
constexpr int DM = 256;      // d_model
constexpr int QKV_LD = 768;  // packed qkv row
constexpr int LDR = 40;      // row-major [token][32] tile, padded to 80 B rows

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// (frame, head) of a workgroup: the 8 heads of a frame are dispatched back-to-back on ONE XCD
// (block b runs on XCD b % 8) so the 128-B lines shared by neighbouring heads hit the same L2.
__device__ __forceinline__ void decode_block(int64_t b, int64_t frames, int64_t& frame, int& head) {
  const int64_t full = (frames / 8) * 64;  // blocks covered by whole groups of 8 frames
  if (b < full) {
    const int xcd = (int)(b & 7);
    const int64_t r = b >> 3;
    head = (int)(r & 7);
    frame = (r >> 3) * 8 + xcd;
  } else {
    const int64_t r = b - full;
    head = (int)(r & 7);
    frame = (frames / 8) * 8 + (r >> 3);
  }
}

// Staging is split into "issue every global load" and "write LDS" so that the whole (frame, head) slab is in
// flight at once: with one load per loop trip the kernels spent most of their time in ten-odd serialized
// HBM round trips per workgroup (dkv: ~30 us of staging latency around ~7 us of MFMA work).
//
// RowStage: ROWS rows of 64 B -> row-major [row][LDR].
// Items past the end are clamped to the last one (its owner writes the same bytes), which keeps both phases
// free of branches: a guarded load makes the compiler wait for it inside the guard.
template <int ROWS>
struct RowStage {
  static constexpr int ITEMS = ROWS * 4, R = (ITEMS + 255) / 256;
  uint4 v[R];
  static __device__ __forceinline__ int item(int tid, int i) { return min(tid + i * 256, ITEMS - 1); }
  __device__ __forceinline__ void load(const uint16_t* src, int64_t ld, int tid) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int c = item(tid, i);
      v[i] = *reinterpret_cast<const uint4*>(src + (int64_t)(c >> 2) * ld + (c & 3) * 8);
    }
  }
  __device__ __forceinline__ void store(uint16_t* dst, int tid) const {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int c = item(tid, i);
      *reinterpret_cast<uint4*>(&dst[(c >> 2) * LDR + (c & 3) * 8]) = v[i];
    }
  }
};
// PairStage: ROWS rows handled as row pairs, written row-major ([row][LDR]) and/or transposed
// ([d][ldv], token contiguous, two tokens per 4-byte write).
template <int ROWS>
struct PairStage {
  static constexpr int ITEMS = ROWS * 2, R = (ITEMS + 255) / 256;
  uint4 a[R], b[R];
  static __device__ __forceinline__ int item(int tid, int i) { return min(tid + i * 256, ITEMS - 1); }
  __device__ __forceinline__ void load(const uint16_t* src, int64_t ld, int tid) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int c = item(tid, i);
      const uint16_t* p = src + (int64_t)(2 * (c >> 2)) * ld + (c & 3) * 8;
      a[i] = *reinterpret_cast<const uint4*>(p);
      b[i] = *reinterpret_cast<const uint4*>(p + ld);
    }
  }
  __device__ __forceinline__ void store_rows(uint16_t* dst, int tid) const {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int c = item(tid, i);
      uint16_t* q = &dst[(2 * (c >> 2)) * LDR + (c & 3) * 8];
      *reinterpret_cast<uint4*>(q) = a[i];
      *reinterpret_cast<uint4*>(q + LDR) = b[i];
    }
  }
  __device__ __forceinline__ void store_cols(uint16_t* dst, int ldv, int tid) const {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int c = item(tid, i);
      const int kp = c >> 2, ch = c & 3;
      const uint32_t aw[4] = {a[i].x, a[i].y, a[i].z, a[i].w}, bw[4] = {b[i].x, b[i].y, b[i].z, b[i].w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t x = (j & 1) ? ((aw[j >> 1] >> 16) | (bw[j >> 1] & 0xffff0000u))
                                   : ((aw[j >> 1] & 0xffffu) | (bw[j >> 1] << 16));
        *reinterpret_cast<uint32_t*>(&dst[(ch * 8 + j) * ldv + 2 * kp]) = x;
      }
    }
  }
};
// The n/32 tiles of a (frame, head) do not divide over 4 waves (10 -> 3,3,2,2): rotate which waves get the
// extra tile from workgroup to workgroup so that no SIMD is always the loaded one.
__device__ __forceinline__ int wave_rot(int wave) {
  const unsigned r = blockIdx.x >> 3;
  return (int)((wave + r + (r >> 5)) & 3);
}

// A operand from a row-major tile: lane (row = l & 31, hi) reads dims 16*s + 8*hi .. +8
__device__ __forceinline__ bf16x8_t frag_rows(const uint16_t* tile, int row0, int s, int lane) {
  return *reinterpret_cast<const bf16x8_t*>(&tile[(row0 + (lane & 31)) * LDR + s * 16 + (lane >> 5) * 8]);
}
// A operand from a transposed tile for the second product: lane (d = l & 31, hi), k-step s2 of the
// 32-token tile at tok0: tokens tok0 + 16*s2 + 4*hi + {0..3} and + 8 (the kappa permutation that
// matches rows 8*s2 .. 8*s2+7 of a 32x32 accumulator).
__device__ __forceinline__ bf16x8_t frag_cols(const uint16_t* tile, int ldv, int tok0, int s2, int lane) {
  const uint16_t* p = &tile[(lane & 31) * ldv + tok0 + 16 * s2 + 4 * (lane >> 5)];
  const uint2 lo = *reinterpret_cast<const uint2*>(p);
  const uint2 hi = *reinterpret_cast<const uint2*>(p + 8);
  const uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);
  return __builtin_bit_cast(bf16x8_t, v);
}
// B operand straight from global: lane (token = l & 31, hi) reads dims 16*s + 8*hi .. +8 of its row
__device__ __forceinline__ bf16x8_t frag_global(const uint16_t* base, int64_t ld, int s, int lane) {
  const uint4 v = *reinterpret_cast<const uint4*>(base + (int64_t)(lane & 31) * ld + s * 16 + (lane >> 5) * 8);
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ bf16x8_t pack_acc_half(const f32x16_t& a, int s2) {
  uint4 v;
  v.x = pack_bf16(a[8 * s2 + 0], a[8 * s2 + 1]);
  v.y = pack_bf16(a[8 * s2 + 2], a[8 * s2 + 3]);
  v.z = pack_bf16(a[8 * s2 + 4], a[8 * s2 + 5]);
  v.w = pack_bf16(a[8 * s2 + 6], a[8 * s2 + 7]);
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ f32x16_t zero16() {
  f32x16_t z;
#pragma unroll
  for (int e = 0; e < 16; ++e) z[e] = 0.f;
  return z;
}
// store a [d x token] accumulator (lane = token, rows = d) as bf16 into row-major [token][ld] at column col0.
// A lane holds d = 8 g + 4 hi + {0..3} for g = 0..3: the two half-waves trade quads (v_permlane32_swap) so that lane
// (r, hi = 0) owns d = 16 q .. 16 q + 7 and lane (r, hi = 1) d = 16 q + 8 .. + 15 -> 16-byte stores (8-byte scattered
// stores reach about half the HBM write rate of 16-byte ones, tools/probes/store_pattern.hip).
__device__ __forceinline__ void store_dt(uint16_t* base, int64_t ld, const f32x16_t& a, float mul, int lane) {
  const int hi = lane >> 5;
  uint16_t* row = base + (int64_t)(lane & 31) * ld;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int g = 2 * q;
    uint32_t ax = pack_bf16(a[4 * g] * mul, a[4 * g + 1] * mul), ay = pack_bf16(a[4 * g + 2] * mul, a[4 * g + 3] * mul);
    uint32_t bx = pack_bf16(a[4 * g + 4] * mul, a[4 * g + 5] * mul), by = pack_bf16(a[4 * g + 6] * mul, a[4 * g + 7] * mul);
    // before: hi = 0 lanes hold d = 8g .. 8g+3 (a*) and 8g+8 .. 8g+11 (b*); hi = 1 lanes hold 8g+4 .. +7 and 8g+12 .. +15
    auto r0 = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
    // v_permlane32_swap exchanges the first operand's upper half-wave with the second operand's lower half-wave
    *reinterpret_cast<uint4*>(row + 16 * q + 8 * hi) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
  }
}

// The same tile into the HEAD-BLOCKED layout (HMA_A_BF16_HEADBLK, include/hma_hip.h): the [32 token][32 d] tile is 2 KB of contiguous
// memory at `blk`.  Same lane mapping as above -- lane (r, hi) holds chunks 2 q + hi (16 bytes each) of token r -- so the two store
// instructions each write 32 bytes of every 64-byte row: 16 cache lines per instruction, each completed by the same wave's next
// instruction, against 32 lines with 32 bytes each whose other pieces come from another CU (row-major [token][768]).  (A form that
// traded chunks between lanes r and r ^ 16 for two fully contiguous 1 KB instructions was slower: its four cross-lane moves sit in
// the dQ wave's path of every step.)
__device__ __forceinline__ void store_dt_blk(uint16_t* blk, const f32x16_t& a, float mul, int lane) {
  const int hi = lane >> 5;
  uint16_t* row = blk + (lane & 31) * 32;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int g = 2 * q;
    uint32_t ax = pack_bf16(a[4 * g] * mul, a[4 * g + 1] * mul), ay = pack_bf16(a[4 * g + 2] * mul, a[4 * g + 3] * mul);
    uint32_t bx = pack_bf16(a[4 * g + 4] * mul, a[4 * g + 5] * mul), by = pack_bf16(a[4 * g + 6] * mul, a[4 * g + 7] * mul);
    auto r0 = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
    *reinterpret_cast<uint4*>(row + 16 * q + 8 * hi) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
  }
}
// a [32 token][32 d] gradient tile of part `which` (0 dq, 1 dk, 2 dv) of (frame, head), tokens tok0 .. tok0 + 31: row-major
// [token][768] or head-blocked
__device__ __forceinline__ void store_grad_tile(uint16_t* dqkv, int hb, int64_t frame, int head, int which, int n, int tok0,
                                                const f32x16_t& a, float mul, int lane) {
  if (hb)
    store_dt_blk(dqkv + ((((frame * NH + head) * 3 + which) * n + tok0) << 5), a, mul, lane);
  else
    store_dt(dqkv + (frame * n + tok0) * QKV_LD + which * DM + head * HD, QKV_LD, a, mul, lane);
}

// ------------------------------------------------------------------------------------- forward
template <int NT>
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ o,
                                                          float* __restrict__ lse, int64_t frames, float c_log2, int qsplit) {
  constexpr int N = NT * 32, LDV = N + 4;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* Ks = smem;             // [N][LDR]
  uint16_t* Vt = smem + N * LDR;   // [32][LDV]
  int64_t frame; int head;
  // qsplit > 1 (a handful of frames: the interactive decode's single frame is 8 workgroups on 256 CUs): the query tiles of a (frame,
  // head) are dealt over qsplit workgroups, four tiles each -- every one stages K / V itself (L2 serves the repeats) and runs ONE round
  // of the tile loop instead of NT / 4; the arithmetic of a tile is the same wherever it runs
  const int64_t item = qsplit > 1 ? (int64_t)blockIdx.x / qsplit : (int64_t)blockIdx.x;
  const int part = qsplit > 1 ? (int)(blockIdx.x - item * qsplit) : 0;
  decode_block(item, frames, frame, head);
  const int qt_lo = qsplit > 1 ? 4 * part : 0, qt_hi = qsplit > 1 ? (4 * part + 4 < NT ? 4 * part + 4 : NT) : NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint16_t* base = qkv + frame * N * QKV_LD + head * HD;
  {
    RowStage<N> ks;
    PairStage<N> vs;
    ks.load(base + DM, QKV_LD, tid);
    vs.load(base + 2 * DM, QKV_LD, tid);
    ks.store(Ks, tid);
    vs.store_cols(Vt, LDV, tid);
  }
  __syncthreads();

  // the query tile of the NEXT trip is fetched while this one is computed (clamped re-load on the last trip)
  const int qt0 = qt_lo + wave_rot(wave);
  bf16x8_t nq0 = frag_global(base + (int64_t)min(qt0, NT - 1) * 32 * QKV_LD, QKV_LD, 0, lane);
  bf16x8_t nq1 = frag_global(base + (int64_t)min(qt0, NT - 1) * 32 * QKV_LD, QKV_LD, 1, lane);
  for (int qt = qt0; qt < qt_hi; qt += 4) {
    const bf16x8_t q0 = nq0, q1 = nq1;
    {
      const uint16_t* qn = base + (int64_t)min(qt + 4, NT - 1) * 32 * QKV_LD;
      nq0 = frag_global(qn, QKV_LD, 0, lane);
      nq1 = frag_global(qn, QKV_LD, 1, lane);
    }
    // two key chunks of NT/2 tiles with one online-softmax merge: keeps the live scores at 80 VGPRs
    constexpr int NC = NT / 2;
    float m = -INFINITY, l = 0.f;
    f32x16_t acc = zero16();
#pragma unroll 1
    for (int c = 0; c < 2; ++c) {
      f32x16_t s[NC];
#pragma unroll
      for (int kt = 0; kt < NC; ++kt) {
        s[kt] = mfma32(frag_rows(Ks, (c * NC + kt) * 32, 0, lane), q0, zero16());
        s[kt] = mfma32(frag_rows(Ks, (c * NC + kt) * 32, 1, lane), q1, s[kt]);
      }
      // max of the RAW scores (the scale, > 0, is folded into the exp2 argument below).  A wave issues ~one instruction per 5
      // cycles whatever its kind, so the soft-max is written to be short: three-input maxima in two chains, the scale-and-shift
      // and the row sums as packed pairs (two sum chains) -- 240 instead of 360 VALU instructions per chunk of 80 scores
      float mc0 = -INFINITY, mc1 = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NC; ++kt)
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
          mc0 = __builtin_fmaxf(__builtin_fmaxf(mc0, s[kt][e]), s[kt][e + 1]);      // -> v_max3_f32
          mc1 = __builtin_fmaxf(__builtin_fmaxf(mc1, s[kt][e + 2]), s[kt][e + 3]);
        }
      float mc = fmaxf(mc0, mc1);
      mc = fmaxf(mc, __shfl_xor(mc, 32, 64)) * c_log2;
      const float m_new = fmaxf(m, mc);
      const float alpha = fast_exp2(m - m_new);  // 0 on the first chunk
      l *= alpha;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] *= alpha;
      const f32x2_t c2 = {c_log2, c_log2}, nm2 = {-m_new, -m_new};
      f32x2_t la = {0.f, 0.f}, lb = {0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < NC; ++kt)
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
          const f32x2_t a0 = __builtin_elementwise_fma(f32x2_t{s[kt][e], s[kt][e + 1]}, c2, nm2);
          const f32x2_t a1 = __builtin_elementwise_fma(f32x2_t{s[kt][e + 2], s[kt][e + 3]}, c2, nm2);
          const f32x2_t p0 = {fast_exp2(a0[0]), fast_exp2(a0[1])}, p1 = {fast_exp2(a1[0]), fast_exp2(a1[1])};
          s[kt][e] = p0[0]; s[kt][e + 1] = p0[1]; s[kt][e + 2] = p1[0]; s[kt][e + 3] = p1[1];
          la += p0;
          lb += p1;
        }
      la += lb;
      l += la[0] + la[1];
#pragma unroll
      for (int kt = 0; kt < NC; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          acc = mfma32(frag_cols(Vt, LDV, (c * NC + kt) * 32, s2, lane), pack_acc_half(s[kt], s2), acc);
      m = m_new;
    }
    l += __shfl_xor(l, 32, 64);
    const int64_t row0 = frame * N + qt * 32;
    store_dt(o + row0 * DM + head * HD, DM, acc, 1.0f / l, lane);
    if (lane < 32) lse[(row0 + lane) * NH + head] = m + __log2f(l);
  }
}

typedef short v4s16_t __attribute__((ext_vector_type(4)));
#define HMA_LDS(T) __attribute__((address_space(3))) T
// Second-product A operand straight from a ROW-major tile (no transposed copy in LDS): 8 bf16 of column (lane & 31), rows
// row0 + 4 hi + {0..3} and the same + 8 -- the order in which pack_acc_half lays the rows of a 32 x 32 accumulator into a
// B operand (the kappa permutation of frag_cols).  ds_read_b64_tr_b16: a 16-lane group reads a [4 row][16 column] block,
// lane i supplies the address of row i >> 2, columns 4 (i & 3) .. + 3 and receives column i of the 4 rows
// (tools/probes/tr_read_dma.hip).  Rows must be 8-byte aligned (LDR = 40 elements: 80 B).
__device__ __forceinline__ bf16x8_t frag_tr(const uint16_t* tile, int ld, int row0, int lane) {
  const int i16 = lane & 15, g16 = lane >> 4;
  const uint16_t* p = tile + (row0 + 4 * (g16 >> 1) + (i16 >> 2)) * ld + 16 * (g16 & 1) + 4 * (i16 & 3);
  const v4s16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((HMA_LDS(v4s16_t)*)p);
  const v4s16_t hv = __builtin_amdgcn_ds_read_tr16_b64_v4i16((HMA_LDS(v4s16_t)*)(p + 8 * ld));
  typedef short v8s16_t __attribute__((ext_vector_type(8)));
  const v8s16_t v = __builtin_shufflevector(lo, hv, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, v);
}

// ------------------------------------------------------------------------------------- dQ
template <int NT>
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ o,
                                                             const uint16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                             float* __restrict__ delta, uint16_t* __restrict__ dqkv,
                                                             int64_t frames, float c_log2, float scale, int hb) {
  constexpr int N = NT * 32;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* Ks = smem;                 // [N][LDR]
  uint16_t* Vs = Ks + N * LDR;         // [N][LDR]
  // (no transposed K: the dQ product reads its K^T operand from Ks with transposing LDS reads -- 51 KB instead of 72 KB
  // of LDS, three workgroups per CU instead of two)
  int64_t frame; int head;
  decode_block(blockIdx.x, frames, frame, head);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint16_t* base = qkv + frame * N * QKV_LD + head * HD;
  {
    RowStage<N> ks, vs;
    ks.load(base + DM, QKV_LD, tid);
    vs.load(base + 2 * DM, QKV_LD, tid);
    ks.store(Ks, tid);
    vs.store(Vs, tid);
  }
  __syncthreads();

  // operands of the NEXT query tile are fetched while this one is computed (clamped re-load on the last trip)
  struct QTile { bf16x8_t q0, q1, g0, g1, o0, o1; float L2; };
  auto fetch = [&](int qt) {
    const int64_t r0 = frame * N + qt * 32;
    const uint16_t* qb = base + (int64_t)qt * 32 * QKV_LD;
    const uint16_t* dob = d_o + r0 * DM + head * HD;
    const uint16_t* ob = o + r0 * DM + head * HD;
    QTile t;
    t.q0 = frag_global(qb, QKV_LD, 0, lane); t.q1 = frag_global(qb, QKV_LD, 1, lane);
    t.g0 = frag_global(dob, DM, 0, lane);    t.g1 = frag_global(dob, DM, 1, lane);
    t.o0 = frag_global(ob, DM, 0, lane);     t.o1 = frag_global(ob, DM, 1, lane);
    t.L2 = lse[(r0 + (lane & 31)) * NH + head];
    return t;
  };
  const int qt0 = wave_rot(wave);
  QTile nxt = fetch(min(qt0, NT - 1));
  for (int qt = qt0; qt < NT; qt += 4) {
    const int64_t row0 = frame * N + qt * 32;
    const QTile cur = nxt;
    nxt = fetch(min(qt + 4, NT - 1));
    const bf16x8_t q0 = cur.q0, q1 = cur.q1, g0 = cur.g0, g1 = cur.g1;
    float dl = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) dl += (float)g0[j] * (float)cur.o0[j] + (float)g1[j] * (float)cur.o1[j];
    dl += __shfl_xor(dl, 32, 64);
    const float L2 = cur.L2;
    if (lane < 32) delta[(row0 + lane) * NH + head] = dl;
    f32x16_t acc = zero16();
#pragma unroll 2
    for (int kt = 0; kt < NT; ++kt) {
      f32x16_t s = mfma32(frag_rows(Ks, kt * 32, 0, lane), q0, zero16());
      s = mfma32(frag_rows(Ks, kt * 32, 1, lane), q1, s);
      f32x16_t dp = mfma32(frag_rows(Vs, kt * 32, 0, lane), g0, zero16());
      dp = mfma32(frag_rows(Vs, kt * 32, 1, lane), g1, dp);
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = fast_exp2(s[e] * c_log2 - L2) * (dp[e] - dl);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) acc = mfma32(frag_tr(Ks, LDR, kt * 32 + 16 * s2, lane), pack_acc_half(s, s2), acc);
    }
    store_grad_tile(dqkv, hb, frame, head, 0, N, (int)(row0 - frame * N), acc, scale, lane);
  }
}

// ------------------------------------------------------------------------------------- dK, dV
// Q / dO (row-major for the score products, transposed for the gradient products) are staged in TWO
// halves of n/2 query rows, so a workgroup needs 47 KB of LDS instead of 95 KB and two to three are
// resident per CU: with one (measured: 487 us per layer) the staging of a (frame, head) and its MFMA
// phase could not overlap with anything.  Each wave owns up to three 32-key tiles and keeps their dK / dV
// accumulators in registers across both halves.
template <int NT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ d_o,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              uint16_t* __restrict__ dqkv, int64_t frames, float c_log2,
                                                              float scale, int hb) {
  constexpr int N = NT * 32;
  constexpr int NQ = NT / 2;          // query tiles per half
  constexpr int NR = NQ * 32;         // query rows per half
  constexpr int LDV = NR + 4;
  constexpr int KPW = (NT + 3) / 4;   // key tiles per wave (<= 3)
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* Qs = smem;                  // [NR][LDR]
  uint16_t* Gs = Qs + NR * LDR;         // dO, [NR][LDR]
  uint16_t* Qt = Gs + NR * LDR;         // [32][LDV]
  uint16_t* Gt = Qt + 32 * LDV;         // [32][LDV]
  float* L2s = reinterpret_cast<float*>(Gt + 32 * LDV);  // [NR]
  float* Dls = L2s + NR;                                  // [NR]
  int64_t frame; int head;
  decode_block(blockIdx.x, frames, frame, head);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, wrot = wave_rot(wave);
  const uint16_t* base = qkv + frame * N * QKV_LD + head * HD;
  const uint16_t* gbase = d_o + frame * N * DM + head * HD;

  f32x16_t dk[KPW], dv[KPW];
#pragma unroll
  for (int j = 0; j < KPW; ++j) { dk[j] = zero16(); dv[j] = zero16(); }

#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    const int q0 = half * NR;
    {
      static_assert(NR <= 256, "one lse / delta row per thread");
      PairStage<NR> qs, gs;
      qs.load(base + (int64_t)q0 * QKV_LD, QKV_LD, tid);
      gs.load(gbase + (int64_t)q0 * DM, DM, tid);
      const int lr = min(tid, NR - 1);
      const float l2 = lse[(frame * N + q0 + lr) * NH + head];
      const float dl = delta[(frame * N + q0 + lr) * NH + head];
      if (half) __syncthreads();  // everyone is done reading the first half
      qs.store_rows(Qs, tid);
      qs.store_cols(Qt, LDV, tid);
      gs.store_rows(Gs, tid);
      gs.store_cols(Gt, LDV, tid);
      L2s[lr] = l2;
      Dls[lr] = dl;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
      const int kt = wrot + 4 * j;
      if (kt < NT) {
        const uint16_t* kb = base + DM + (int64_t)kt * 32 * QKV_LD;
        const uint16_t* vb = base + 2 * DM + (int64_t)kt * 32 * QKV_LD;
        const bf16x8_t k0 = frag_global(kb, QKV_LD, 0, lane), k1 = frag_global(kb, QKV_LD, 1, lane);
        const bf16x8_t v0 = frag_global(vb, QKV_LD, 0, lane), v1 = frag_global(vb, QKV_LD, 1, lane);
#pragma unroll 1
        for (int qt = 0; qt < NQ; ++qt) {
          // S[q][key]: lane = key, rows = q
          f32x16_t sc = mfma32(frag_rows(Qs, qt * 32, 0, lane), k0, zero16());
          sc = mfma32(frag_rows(Qs, qt * 32, 1, lane), k1, sc);
          f32x16_t dp = mfma32(frag_rows(Gs, qt * 32, 0, lane), v0, zero16());
          dp = mfma32(frag_rows(Gs, qt * 32, 1, lane), v1, dp);
          f32x16_t ds;
#pragma unroll
          for (int g = 0; g < 4; ++g) {  // rows 8 g + 4 hi + {0..3}: one 16-byte LDS read each for lse and delta
            const float4 l4 = *reinterpret_cast<const float4*>(&L2s[qt * 32 + 8 * g + 4 * hi]);
            const float4 d4 = *reinterpret_cast<const float4*>(&Dls[qt * 32 + 8 * g + 4 * hi]);
            const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float pr = fast_exp2(sc[4 * g + e] * c_log2 - lq[e]);
              sc[4 * g + e] = pr;
              ds[4 * g + e] = pr * (dp[4 * g + e] - dq4[e]);
            }
          }
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            dv[j] = mfma32(frag_cols(Gt, LDV, qt * 32, s2, lane), pack_acc_half(sc, s2), dv[j]);
            dk[j] = mfma32(frag_cols(Qt, LDV, qt * 32, s2, lane), pack_acc_half(ds, s2), dk[j]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < KPW; ++j) {
    const int kt = wrot + 4 * j;
    if (kt < NT) {
      const int64_t row0 = frame * N + kt * 32;
      store_grad_tile(dqkv, hb, frame, head, 1, N, kt * 32, dk[j], scale, lane);
      store_grad_tile(dqkv, hb, frame, head, 2, N, kt * 32, dv[j], 1.0f, lane);
    }
  }
}

// ------------------------------------------------------------------------------------- dQ, dK, dV in one kernel
// One persistent workgroup per CU, NT tile waves + 2 dQ waves: tile wave w owns key tile w of a (frame, head) item and keeps its dK / dV accumulators in
// registers.  Q, K, V and dO of the item's n tokens sit in LDS (4 x n x 80 B = 100 KB at n = 320): every operand is read from HBM once
// (the two-kernel form read q / k / v / dO twice: 1039 -> 660 MB per layer at frames = 512), and the score tile S, P = exp(S - lse)
// and dS = P (dP - delta) are formed ONCE per (query tile, key tile) -- 5 products of n^2 d instead of 7, half the exponentials.
// The soft-max arithmetic, not the matrix pipe, bounds these kernels (16 v_exp per 8 MFMAs and lane).  The transposed use of dS (dQ
// contracts over the keys, which are the LANES of the tile a wave has just computed) goes through LDS: every wave writes its dS
// tile as [key][query] rows, and after the step's barrier one of the two dQ waves reads the NT tiles back with transposing reads
// (ds_read_b64_tr_b16, like K^T) and forms dQ of that query tile (2 NT MFMAs) while the tile waves go on with the next query tile
// (two dS buffers).  delta = rowsum(dO * O) is formed while the item is staged; the
// NEXT item's rows are fetched into registers during the compute (one workgroup per CU: nothing else would hide the staging).
#ifndef ATTN_ABL  // debug builds (timing only, results wrong): 1 no lse / delta reads, 2 no dQ job, 4 no exponentials, 8 no dS tile writes, 32 no compute at all (staging + stores), 64 no row loads after the first item, 128 no dqkv stores
#define ATTN_ABL 0
#endif
// ---- LDS image of the fused backward's tiles (ATTN_SWZ, default): UNPADDED 64-byte rows, the 16-byte chunk c of row r stored at
// chunk c ^ ((r >> 2) & 3).  With the padded rows of the other kernels (LDR = 40: 20 dwords) the transposing reads of a 32-lane pass
// -- four consecutive rows, all 64 bytes of each -- wrap around the 64 banks (rows r and r + 3 share 12 of them: 2-way conflicts,
// 34 % of this kernel's LDS cycles in profiles/pmc_sq_r4.txt); with 16-dword rows the four rows of a pass tile the banks exactly, and
// the XOR keeps the row-major 16-byte reads of a 16-lane pass (16 consecutive rows, one chunk) and the staging writes (4 rows x 4
// chunks) conflict-free as well.
// ATTN_SWZ = 2 (round 6; measurement builds -- the default stays the padded layout, see below): the same bank picture WITHOUT the XOR -- unpadded 64-byte rows with 16 bytes of skew in front of every
// group of four rows: offset(row, chunk) = 64 row + 16 (row >> 2) + 16 chunk bytes.  The four rows of a transposing pass are 256
// contiguous bytes (every bank once); the 16 rows of a row-major 16-byte pass start at banks 16 b + 4 a (row = 4 a + b): all different
// within each of the instruction's lane groups.  And the offset is ADDITIVE in the row: with row = row0 + lane part (row0 a multiple of
// 4, a compile-time constant at every call site) the lane part is one register and row0 / chunk fold into the instruction's immediate --
// the XOR form (ATTN_SWZ = 1, removed) needed an address register per call site (70 of them: 300 bytes of scratch in the balanced
// kernel).  Built, parity-green, 239 registers and no scratch -- and 3-4 % SLOWER than the padded layout (standalone, head-blocked
// backward at 512 x 320: 259.7 / 251.1 / 257.4 us padded against 267.8 / 271.1 / 261.3 us; in situ 234 against 242 us,
// profiles/attn_lds_layout_r6.txt): the kernel's limit is its one dQ wave and its stores, not the conflict cycles.
#ifndef ATTN_SWZ
#define ATTN_SWZ 0
#endif
static_assert(ATTN_SWZ == 0 || ATTN_SWZ == 2, "the XOR form (1) is gone: it cost an address register per call site");
constexpr int LDF = ATTN_SWZ == 2 ? 34 : LDR;  // (elements per row of a tile's allocation: N rows take N * LDF)
// element offset of the 16-byte chunk `chunk` of row rc + rl: rc = the part of the row index that is a constant or wave-uniform at the
// call site (a MULTIPLE OF 4), rl = the lane's part -- written as a sum of the two parts' offsets so that the lane part is ONE register
// per access pattern and rc / chunk fold into the instruction's immediate offset ((rc + rl) >> 2 as one expression does not split)
__device__ __forceinline__ int foff(int rc, int rl, int chunk) {
  if (ATTN_SWZ == 2) return rc * 34 + rl * 32 + ((rl >> 2) << 3) + (chunk << 3);
  return (rc + rl) * LDF + (chunk << 3);
}
__device__ __forceinline__ bf16x8_t frag_rows_f(const uint16_t* tile, int row0, int s, int lane) {
  return *reinterpret_cast<const bf16x8_t*>(&tile[foff(row0, lane & 31, 2 * s + (lane >> 5))]);
}
__device__ __forceinline__ bf16x8_t frag_tr_f(const uint16_t* tile, int row0, int lane) {
  const int i16 = lane & 15, g16 = lane >> 4;
  const int rl = 4 * (g16 >> 1) + (i16 >> 2);
  const int chunk = 2 * (g16 & 1) + ((i16 & 3) >> 1), inner = 4 * (i16 & 1);
  const v4s16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((HMA_LDS(v4s16_t)*)(tile + foff(row0, rl, chunk) + inner));
  const v4s16_t hv = __builtin_amdgcn_ds_read_tr16_b64_v4i16((HMA_LDS(v4s16_t)*)(tile + foff(row0 + 8, rl, chunk) + inner));
  typedef short v8s16_t __attribute__((ext_vector_type(8)));
  const v8s16_t v = __builtin_shufflevector(lo, hv, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, v);
}

// LDS-only barrier: waits for this wave's LDS operations, not for its global loads (the next item's rows stay in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int... I, typename F>
__device__ __forceinline__ void static_for_i_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N_, typename F>
__device__ __forceinline__ void static_for_i(F&& f) {
  static_for_i_impl(std::make_integer_sequence<int, N_>{}, static_cast<F&&>(f));
}
template <int NT, int KT = 1, int THREADS_ = (NT / KT + 2) * 64, bool NEG_DELTA = false>
struct FusedStage {
  static constexpr int N = NT * 32, ITEMS = N * 4, THREADS = THREADS_, R = (ITEMS + THREADS - 1) / THREADS;
  // (vectors, not arrays: carried around the item loop as arrays, hipcc leaves three of them in scratch)
  typedef uint32_t vec_t __attribute__((ext_vector_type(4 * R)));
  vec_t rq, rk, rv, rg, ro;
  float rl;
  static __device__ __forceinline__ void put(vec_t& v, int i, const uint4& x) {
    v[4 * i] = x.x; v[4 * i + 1] = x.y; v[4 * i + 2] = x.z; v[4 * i + 3] = x.w;
  }
  static __device__ __forceinline__ uint4 get(const vec_t& v, int i) { return make_uint4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]); }
  // part i of R (i = -1: all of them): a wave blocks at the issue of a load while the memory pipeline is full, and R x 5 loads per
  // thread issued at once by every wave of every CU are such a burst (~9 600 cycles per item, tools/attn_variants.sh -DATTN_ABL=64);
  // the kernel therefore issues one part per step of the item in flight
  __device__ __forceinline__ void fetch(const uint16_t* qkv, const uint16_t* o, const uint16_t* d_o, const float* lse, int64_t item,
                                        int64_t frames, int tid, int part = -1) {
    int64_t frame; int head;
    decode_block(item, frames, frame, head);
    // Wave-uniform bases + 32-bit lane offsets recomputed from an OPAQUE thread index at every call: as loop invariants of the item loop
    // the 64-bit row offsets of the R chunks (and stage()'s LDS offsets) stay in registers -- 27 of them went to scratch, and a scratch
    // reload is a vector-memory load whose wait also drains the previous item's dqkv stores.
    asm volatile("" : "+v"(tid));
    const uint16_t* base = qkv + frame * N * QKV_LD + head * HD;
    const uint16_t* gb = d_o + frame * N * DM + head * HD;
    const uint16_t* ob = o + frame * N * DM + head * HD;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if (part >= 0 && part != i) continue;
      const uint32_t c = (uint32_t)min(tid + i * THREADS, ITEMS - 1);
      const uint32_t row = c >> 2, ch = (c & 3) * 8;
      const uint32_t oq = row * QKV_LD + ch, og = row * DM + ch;
      put(rq, i, *reinterpret_cast<const uint4*>(base + oq));
      put(rk, i, *reinterpret_cast<const uint4*>(base + DM + oq));
      put(rv, i, *reinterpret_cast<const uint4*>(base + 2 * DM + oq));
      put(rg, i, *reinterpret_cast<const uint4*>(gb + og));
      put(ro, i, *reinterpret_cast<const uint4*>(ob + og));
    }
    if (part <= 0) rl = (lse + frame * N * NH + head)[(uint32_t)min(tid, N - 1) * NH];
  }
  __device__ __forceinline__ void stage(uint16_t* Qs, uint16_t* Ks, uint16_t* Vs, uint16_t* Gs, float* L2s, float* Dls, float* delta,
                                        int64_t item, int64_t frames, int tid) const {
    int64_t frame; int head;
    decode_block(item, frames, frame, head);
    asm volatile("" : "+v"(tid));  // (see fetch)
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int c = min(tid + i * THREADS, ITEMS - 1);
      const int off = foff(0, c >> 2, c & 3);
      *reinterpret_cast<uint4*>(&Qs[off]) = get(rq, i);
      *reinterpret_cast<uint4*>(&Ks[off]) = get(rk, i);
      *reinterpret_cast<uint4*>(&Vs[off]) = get(rv, i);
      *reinterpret_cast<uint4*>(&Gs[off]) = get(rg, i);
      // delta of the row: the four 8-column pieces of a row sit in four neighbouring lanes
      const uint32_t gw[4] = {rg[4 * i], rg[4 * i + 1], rg[4 * i + 2], rg[4 * i + 3]}, ow[4] = {ro[4 * i], ro[4 * i + 1], ro[4 * i + 2], ro[4 * i + 3]};
      float dsum = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dsum = __builtin_fmaf(__builtin_bit_cast(float, gw[e] << 16), __builtin_bit_cast(float, ow[e] << 16), dsum);
        dsum = __builtin_fmaf(__builtin_bit_cast(float, gw[e] & 0xffff0000u), __builtin_bit_cast(float, ow[e] & 0xffff0000u), dsum);
      }
      dsum += __shfl_xor(dsum, 1, 64);
      dsum += __shfl_xor(dsum, 2, 64);
      if ((c & 3) == 0) {
        Dls[c >> 2] = NEG_DELTA ? -dsum : dsum;  // (NEG_DELTA: the dP accumulators start from -delta, see attn_bwd_bal_kernel)
        (delta + frame * N * NH + head)[(uint32_t)(c >> 2) * NH] = dsum;
      }
    }
    if (tid < N) L2s[tid] = rl;
  }
};

// KT = key tiles per tile wave.  KT = 2 (round 4, n = 256 / 320): NT / 2 tile waves + 2 dQ waves = 7 waves at n = 320 -- two per SIMD
// instead of three, 256 registers each.  A tile wave forms the scores of TWO independent key tiles per step: the second tile's
// MFMAs run under the first tile's soft-max arithmetic (a wave issues in order, the matrix pipe works behind it), and what the two
// tiles share -- the query tile's Q / dO fragments (row-major and transposed), lse, delta -- is read from LDS once instead of twice.
template <int NT, int KT>
__global__ __launch_bounds__((NT / KT + 2) * 64, 1) void attn_bwd_fused_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ o,
                                                                    const uint16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                                    float* __restrict__ delta, uint16_t* __restrict__ dqkv,
                                                                    int64_t frames, float c_log2, float scale, int hb) {
  static_assert(NT % KT == 0, "whole key tiles per wave");
  constexpr int TW = NT / KT;   // tile waves; waves TW, TW + 1 are the dQ waves
  constexpr int N = NT * 32;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* Qs = smem;                 // [N][LDF]
  uint16_t* Ks = Qs + N * LDF;
  uint16_t* Vs = Ks + N * LDF;
  uint16_t* Gs = Vs + N * LDF;         // dO
  uint16_t* Ts = Gs + N * LDF;         // 2 x [N keys][LDR]: dS of one query tile, [key][query]
  float* L2s = reinterpret_cast<float*>(Ts + 2 * N * LDF);  // [N]
  float* Dls = L2s + N;                                      // [N]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int64_t nitems = frames * NH;

  FusedStage<NT, KT> st;
  int64_t item = blockIdx.x;
  if (item < nitems) st.fetch(qkv, o, d_o, lse, item, frames, tid);
#pragma unroll 1
  for (; item < nitems; item += gridDim.x) {
    st.stage(Qs, Ks, Vs, Gs, L2s, Dls, delta, item, frames, tid);
    lds_barrier();  // (LDS-only barriers throughout: a __syncthreads would also drain the previous item's dqkv stores)
    const int64_t nxt = item + gridDim.x;
    const bool do_fetch = nxt < nitems && !(ATTN_ABL & 64);  // (64: measurement only -- the rows of item 0 again)
#ifndef ATTN_FETCH_SPREAD
    if (do_fetch) st.fetch(qkv, o, d_o, lse, nxt, frames, tid);
#endif
    int64_t frame; int head;
    decode_block(item, frames, frame, head);
    if (wave >= TW) {
      // ---- the two dQ waves: dQ^T[d][q] of query tile qt = sum over the keys of K^T[d][key] dS^T[key][q], wave TW + (qt & 1), while the
      // tile waves work on query tile qt + 1 (which goes to the other dS buffer)
#ifdef ATTN_FETCH_SPREAD
#pragma unroll 1
#else
#pragma unroll 1
#endif
      for (int qt = 0; qt < ((ATTN_ABL & 32) ? 0 : NT); ++qt) {
#ifdef ATTN_FETCH_SPREAD
        if (do_fetch && qt % ATTN_FETCH_SPREAD == 0 && qt / ATTN_FETCH_SPREAD < FusedStage<NT, KT>::R) st.fetch(qkv, o, d_o, lse, nxt, frames, tid, qt / ATTN_FETCH_SPREAD);
#endif
        lds_barrier();
        if ((qt & 1) == wave - TW && !(ATTN_ABL & 2)) {
          const uint16_t* Tq = Ts + (qt & 1) * N * LDF;
          f32x16_t a0 = zero16(), a1 = zero16();
          // 2 NT k-steps of 16 keys in groups of 4, the next group's transposing reads in flight behind this group's MFMAs (one read
          // -> MFMA round trip per k-step made the job longer than the tile waves' step, and they wait for it at the next barrier)
          static_assert((2 * NT) % 4 == 0, "groups of four k-steps");
          bf16x8_t fa[4], fb[4], na[4], nb[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) { fa[i] = frag_tr_f(Ks, 16 * i, lane); fb[i] = frag_tr_f(Tq, 16 * i, lane); }
#pragma unroll
          for (int grp = 0; grp < NT / 2; ++grp) {
            if (grp + 1 < NT / 2) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                na[i] = frag_tr_f(Ks, 16 * (4 * grp + 4 + i), lane);
                nb[i] = frag_tr_f(Tq, 16 * (4 * grp + 4 + i), lane);
              }
            }
            a0 = mfma32(fa[0], fb[0], a0);
            a1 = mfma32(fa[1], fb[1], a1);
            a0 = mfma32(fa[2], fb[2], a0);
            a1 = mfma32(fa[3], fb[3], a1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) { fa[i] = na[i]; fb[i] = nb[i]; }
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) a0[e] += a1[e];
          if (!(ATTN_ABL & 128)) store_grad_tile(dqkv, hb, frame, head, 0, N, qt * 32, a0, scale, lane);
        }
      }
    } else {
      bf16x8_t kf[KT][2], vf[KT][2];
      f32x16_t dk[KT], dv[KT];
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const int kt = wave * KT + t;
        kf[t][0] = frag_rows_f(Ks, kt * 32, 0, lane); kf[t][1] = frag_rows_f(Ks, kt * 32, 1, lane);
        vf[t][0] = frag_rows_f(Vs, kt * 32, 0, lane); vf[t][1] = frag_rows_f(Vs, kt * 32, 1, lane);
        dk[t] = zero16(); dv[t] = zero16();
      }
#pragma unroll 1
      for (int qt = 0; qt < ((ATTN_ABL & 32) ? 0 : NT); ++qt) {
        // S[q][key], dP[q][key]: lane = key, rows = q.  The query tile's fragments serve every key tile of the wave.
#ifdef ATTN_FETCH_SPREAD
        if (do_fetch && qt % ATTN_FETCH_SPREAD == 0 && qt / ATTN_FETCH_SPREAD < FusedStage<NT, KT>::R) st.fetch(qkv, o, d_o, lse, nxt, frames, tid, qt / ATTN_FETCH_SPREAD);
#endif
        const bf16x8_t aq0 = frag_rows_f(Qs, qt * 32, 0, lane), aq1 = frag_rows_f(Qs, qt * 32, 1, lane);
        const bf16x8_t ag0 = frag_rows_f(Gs, qt * 32, 0, lane), ag1 = frag_rows_f(Gs, qt * 32, 1, lane);
        f32x16_t sc[KT], dp[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) {
#ifndef ATTN_KV_REGS  // (the K / V fragments are re-read from LDS every step: 16 registers per key tile would spill -- 376 us against 345)
          const int kt = wave * KT + t;
          sc[t] = mfma32(aq0, frag_rows_f(Ks, kt * 32, 0, lane), zero16());
          sc[t] = mfma32(aq1, frag_rows_f(Ks, kt * 32, 1, lane), sc[t]);
          dp[t] = mfma32(ag0, frag_rows_f(Vs, kt * 32, 0, lane), zero16());
          dp[t] = mfma32(ag1, frag_rows_f(Vs, kt * 32, 1, lane), dp[t]);
#else
          sc[t] = mfma32(aq0, kf[t][0], zero16());
          sc[t] = mfma32(aq1, kf[t][1], sc[t]);
          dp[t] = mfma32(ag0, vf[t][0], zero16());
          dp[t] = mfma32(ag1, vf[t][1], dp[t]);
#endif
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // rows 8 g + 4 hi + {0..3}: one 16-byte LDS read each for lse and delta, shared by the key tiles
          float4 l4 = make_float4(c_log2, scale, c_log2, scale), d4 = l4;
          if (!(ATTN_ABL & 1)) {
            l4 = *reinterpret_cast<const float4*>(&L2s[qt * 32 + 8 * g + 4 * hi]);
            d4 = *reinterpret_cast<const float4*>(&Dls[qt * 32 + 8 * g + 4 * hi]);
          }
          const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int t = 0; t < KT; ++t) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float arg = sc[t][4 * g + e] * c_log2 - lq[e];
              const float pr = (ATTN_ABL & 4) ? arg : fast_exp2(arg);
              sc[t][4 * g + e] = pr;
              dp[t][4 * g + e] = pr * (dp[t][4 * g + e] - dq4[e]);
            }
            // this lane's key row of the dS tile, queries 8 g + 4 hi .. + 3
            if (!(ATTN_ABL & 8)) {
              uint16_t* T = Ts + (qt & 1) * N * LDF + foff((wave * KT + t) * 32, lane & 31, g);
              *reinterpret_cast<uint2*>(T + 4 * hi) =
                  make_uint2(pack_bf16(dp[t][4 * g], dp[t][4 * g + 1]), pack_bf16(dp[t][4 * g + 2], dp[t][4 * g + 3]));
            }
          }
        }
        // (dS tile of query tile qt complete after this barrier; the buffer's previous reader -- the dQ job of qt - 2 -- is past its
        // reads: it arrived at the barrier of qt - 1 only after them)
        lds_barrier();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8_t gt = frag_tr_f(Gs, qt * 32 + 16 * s2, lane), qtr = frag_tr_f(Qs, qt * 32 + 16 * s2, lane);
#pragma unroll
          for (int t = 0; t < KT; ++t) {
            dv[t] = mfma32(gt, pack_acc_half(sc[t], s2), dv[t]);
            dk[t] = mfma32(qtr, pack_acc_half(dp[t], s2), dk[t]);
          }
        }
      }
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const int64_t row0 = frame * N + (wave * KT + t) * 32;
        if (!(ATTN_ABL & 128)) {
          store_grad_tile(dqkv, hb, frame, head, 1, N, (wave * KT + t) * 32, dk[t], scale, lane);
          store_grad_tile(dqkv, hb, frame, head, 2, N, (wave * KT + t) * 32, dv[t], 1.0f, lane);
        }
      }
    }
    lds_barrier();  // every wave is done with this item's LDS before the next one is staged
  }
}

// ------------------------------------------------------------------------------------- the fused backward, balanced over the SIMDs
// Round 5.  The 7-wave form above puts TWO tile waves of two key tiles each on one SIMD (waves w and w + 4 share a SIMD: four tiles
// per step there) and one tile wave alone on another (two tiles): the step -- every wave meets at its barrier -- runs at the pace
// of the four-tile SIMD.  Here n = 320 runs EIGHT waves: waves 0..2 own two key tiles each (tiles 0..5), waves 3..6 one each
// (tiles 6..9), wave 7 forms dQ of EVERY query tile; the SIMD pairs are then (w0, w4) = 3 tiles, (w1, w5) = 3, (w2, w6) = 3,
// (w3, w7) = 1 tile + the dQ job.  Also:
//   * the dP accumulators START from -delta[q] (the MFMA's C operand, read from LDS where the staging put the negated values), which
//     removes the subtraction from dS = P (dP - delta) -- one VALU operation in five per score element;
//   * the next item's rows are fetched by the five LIGHT waves only (waves 3..7: 320 threads x 4 chunks of each of q, k, v, dO, o);
//     the two-tile waves carry no prefetch registers (60 of them, which the 7-wave form had to fit beside two tiles' accumulators)
//     and keep their K / V fragments in registers instead of re-reading them every step.  Each role runs its OWN copy of the item
//     loop (the hardware barrier counts arrivals, not program counters), so the prefetch registers are dead code on the heavy path.
#ifndef ATTN_BAL   // (0: measurement builds -- the 7-wave form for n = 320 as well)
#define ATTN_BAL 1
#endif
template <int NT>
__global__ __launch_bounds__(512, 2) void attn_bwd_bal_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ o,
                                                              const uint16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                              float* __restrict__ delta, uint16_t* __restrict__ dqkv,
                                                              int64_t frames, float c_log2, float scale, int hb) {
  static_assert(NT == 10, "the wave -> key tile table below is for ten key tiles");
  constexpr int N = NT * 32;
  constexpr int HEAVY = 3;  // waves 0 .. HEAVY - 1: two key tiles each; HEAVY .. 6: one each; 7: dQ
  constexpr int FT = (8 - HEAVY) * 64;
  static_assert(FT == N, "one lse value per fetching thread");
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* Qs = smem;                 // [N][LDF]
  uint16_t* Ks = Qs + N * LDF;
  uint16_t* Vs = Ks + N * LDF;
  uint16_t* Gs = Vs + N * LDF;         // dO
  uint16_t* Ts = Gs + N * LDF;         // 2 x [N keys][LDF]: dS of one query tile, [key][query]
  float* L2s = reinterpret_cast<float*>(Ts + 2 * N * LDF);  // [N]
  float* Dls = L2s + N;                                      // [N]: -delta
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int64_t nitems = frames * NH;

  // ---- one (frame, head) item of a tile wave: KT key tiles from tile kt0 on
  auto tile_item = [&](auto kt_c, const int kt0, const int64_t frame, const int head, auto peel_c, auto&& hook) __attribute__((always_inline)) {
    constexpr int KT = decltype(kt_c)::value;
    bf16x8_t kf[KT][2], vf[KT][2];
    f32x16_t dk[KT], dv[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      kf[t][0] = frag_rows_f(Ks, (kt0 + t) * 32, 0, lane); kf[t][1] = frag_rows_f(Ks, (kt0 + t) * 32, 1, lane);
      vf[t][0] = frag_rows_f(Vs, (kt0 + t) * 32, 0, lane); vf[t][1] = frag_rows_f(Vs, (kt0 + t) * 32, 1, lane);
      dk[t] = zero16(); dv[t] = zero16();
    }
    auto step = [&](const int qt) __attribute__((always_inline)) {
      // S[q][key], dP[q][key]: lane = key, rows = q.  The query tile's fragments serve every key tile of the wave.
      const bf16x8_t aq0 = frag_rows_f(Qs, qt * 32, 0, lane), aq1 = frag_rows_f(Qs, qt * 32, 1, lane);
      const bf16x8_t ag0 = frag_rows_f(Gs, qt * 32, 0, lane), ag1 = frag_rows_f(Gs, qt * 32, 1, lane);
      f32x16_t sc[KT], dp[KT];
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        // dP starts from -delta of the tile's query rows 8 g + 4 hi + {0..3} (the accumulator's order), read straight into the
        // accumulator registers
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 d4 = make_float4(c_log2, scale, c_log2, scale);
          if (!(ATTN_ABL & 1)) d4 = *reinterpret_cast<const float4*>(&Dls[qt * 32 + 8 * g + 4 * hi]);
          dp[t][4 * g] = d4.x; dp[t][4 * g + 1] = d4.y; dp[t][4 * g + 2] = d4.z; dp[t][4 * g + 3] = d4.w;
        }
        sc[t] = mfma32(aq0, kf[t][0], zero16());
        sc[t] = mfma32(aq1, kf[t][1], sc[t]);
        dp[t] = mfma32(ag0, vf[t][0], dp[t]);
        dp[t] = mfma32(ag1, vf[t][1], dp[t]);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // rows 8 g + 4 hi + {0..3}: one 16-byte LDS read for lse, shared by the key tiles
        float4 l4 = make_float4(c_log2, scale, c_log2, scale);
        if (!(ATTN_ABL & 1)) l4 = *reinterpret_cast<const float4*>(&L2s[qt * 32 + 8 * g + 4 * hi]);
        const float lq[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
        for (int t = 0; t < KT; ++t) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float arg = sc[t][4 * g + e] * c_log2 - lq[e];
            const float pr = (ATTN_ABL & 4) ? arg : fast_exp2(arg);
            sc[t][4 * g + e] = pr;
            dp[t][4 * g + e] = pr * dp[t][4 * g + e];
          }
          // this lane's key row of the dS tile, queries 8 g + 4 hi .. + 3
          if (!(ATTN_ABL & 8)) {
            uint16_t* T = Ts + (qt & 1) * N * LDF + foff((kt0 + t) * 32, lane & 31, g);
            *reinterpret_cast<uint2*>(T + 4 * hi) =
                make_uint2(pack_bf16(dp[t][4 * g], dp[t][4 * g + 1]), pack_bf16(dp[t][4 * g + 2], dp[t][4 * g + 3]));
          }
        }
      }
      // (dS tile of query tile qt complete after this barrier; the buffer's previous reader -- the dQ job of qt - 2 -- is past its
      // reads: the dQ wave arrived at the barrier of qt - 1 only after them)
      lds_barrier();
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8_t gt = frag_tr_f(Gs, qt * 32 + 16 * s2, lane), qtr = frag_tr_f(Qs, qt * 32 + 16 * s2, lane);
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          dv[t] = mfma32(gt, pack_acc_half(sc[t], s2), dv[t]);
          dk[t] = mfma32(qtr, pack_acc_half(dp[t], s2), dk[t]);
        }
      }
    };
    // the first PEEL steps are straight-line code with the hook's part index a compile-time constant (the fetch registers are indexed
    // by it), the rest is a loop
    constexpr int PEEL = decltype(peel_c)::value;
    if (!(ATTN_ABL & 32)) {
      static_for_i<PEEL>([&](auto i_) __attribute__((always_inline)) {
        hook(i_);
        step(decltype(i_)::value);
      });
#pragma unroll 1
      for (int qt = PEEL; qt < NT; ++qt) step(qt);
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const int64_t row0 = frame * N + (kt0 + t) * 32;
      if (!(ATTN_ABL & 128)) {
        store_grad_tile(dqkv, hb, frame, head, 1, N, (kt0 + t) * 32, dk[t], scale, lane);
        store_grad_tile(dqkv, hb, frame, head, 2, N, (kt0 + t) * 32, dv[t], 1.0f, lane);
      }
    }
  };
  // ---- one item of the dQ wave: dQ^T[d][q] of query tile qt = sum over the keys of K^T[d][key] dS^T[key][q], while the tile waves
  // work on query tile qt + 1 (which goes to the other dS buffer)
  auto dq_item = [&](const int64_t frame, const int head, auto peel_c, auto&& hook) __attribute__((always_inline)) {
    // K^T of the item as A-operand fragments: the first KREG of the 2 NT k-steps of 16 keys stay in registers this wave has (it owns
    // no tile) and are read ONCE per item instead of once per query tile -- the dQ wave is the slowest wave of a step (no dQ job:
    // 325 -> 271 us), and half of its job's read -> MFMA round trips are K^T reads
#ifndef ATTN_DQ_KREG   // k-steps whose K^T fragments stay in registers (a multiple of 4, <= 2 NT)
#define ATTN_DQ_KREG 0
#endif
    constexpr int KREG = ATTN_DQ_KREG;
    bf16x8_t kt[2 * NT];
#pragma unroll
    for (int i = 0; i < KREG; ++i) kt[i] = frag_tr_f(Ks, 16 * i, lane);
    auto step = [&](const int qt) __attribute__((always_inline)) {
      lds_barrier();
      if (!(ATTN_ABL & 2)) {
        const uint16_t* Tq = Ts + (qt & 1) * N * LDF;
        f32x16_t a0 = zero16(), a1 = zero16();
        static_assert((2 * NT) % 4 == 0, "groups of four k-steps");
        bf16x8_t fb[4], nb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fb[i] = frag_tr_f(Tq, 16 * i, lane);
#pragma unroll
        for (int grp = 0; grp < NT / 2; ++grp) {
          if (grp + 1 < NT / 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) nb[i] = frag_tr_f(Tq, 16 * (4 * grp + 4 + i), lane);
          }
          if (4 * grp >= KREG) {
#pragma unroll
            for (int i = 0; i < 4; ++i) kt[4 * grp + i] = frag_tr_f(Ks, 16 * (4 * grp + i), lane);
          }
          a0 = mfma32(kt[4 * grp + 0], fb[0], a0);
          a1 = mfma32(kt[4 * grp + 1], fb[1], a1);
          a0 = mfma32(kt[4 * grp + 2], fb[2], a0);
          a1 = mfma32(kt[4 * grp + 3], fb[3], a1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 4; ++i) fb[i] = nb[i];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) a0[e] += a1[e];
        if (!(ATTN_ABL & 128)) store_grad_tile(dqkv, hb, frame, head, 0, N, qt * 32, a0, scale, lane);
      }
    };
    constexpr int PEEL = decltype(peel_c)::value;
    if (!(ATTN_ABL & 32)) {
      static_for_i<PEEL>([&](auto i_) __attribute__((always_inline)) {
        hook(i_);
        step(decltype(i_)::value);
      });
#pragma unroll 1
      for (int qt = PEEL; qt < NT; ++qt) step(qt);
    }
  };

  if (wave < HEAVY) {
    // ---- the two-tile waves: no staging work (barriers only around it)
#pragma unroll 1
    for (int64_t item = blockIdx.x; item < nitems; item += gridDim.x) {
      lds_barrier();  // the item is staged
      int64_t frame; int head;
      decode_block(item, frames, frame, head);
      tile_item(std::integral_constant<int, 2>{}, 2 * wave, frame, head, std::integral_constant<int, 0>{}, [](auto) {});
      lds_barrier();  // every wave is done with this item's LDS before the next one is staged
    }
  } else {
    // ---- the light waves stage the items: 320 threads x 4 sixteen-byte chunks of each array
    FusedStage<NT, 1, FT, true> st;
    const int ftid = tid - HEAVY * 64;
    int64_t item = blockIdx.x;
    if (item < nitems) st.fetch(qkv, o, d_o, lse, item, frames, ftid);
#pragma unroll 1
    for (; item < nitems; item += gridDim.x) {
      st.stage(Qs, Ks, Vs, Gs, L2s, Dls, delta, item, frames, ftid);
      lds_barrier();  // (LDS-only barriers throughout: a __syncthreads would also drain the previous item's dqkv stores)
      const int64_t nxt = item + gridDim.x;
#ifndef ATTN_PF_SPREAD  // 1: the next item's rows are requested a part (five loads per thread) per step of this item instead of all at its start
#define ATTN_PF_SPREAD 1
#endif
      const bool more = nxt < nitems && !(ATTN_ABL & 64);
      if (more && !ATTN_PF_SPREAD) st.fetch(qkv, o, d_o, lse, nxt, frames, ftid);
      constexpr int PEEL = ATTN_PF_SPREAD ? decltype(st)::R : 0;
      const int64_t nxt_c = more ? nxt : item;  // (unconditional: the last item's loads are repeated rather than the register vectors written under a branch)
      auto hook = [&](auto part_) __attribute__((always_inline)) {
        st.fetch(qkv, o, d_o, lse, nxt_c, frames, ftid, decltype(part_)::value);
      };
      int64_t frame; int head;
      decode_block(item, frames, frame, head);
      if (wave == 7) {
#ifdef ATTN_DQ_PRIO  // (measurement builds: the dQ wave, the slowest of every step, ahead of the tile wave it shares a SIMD with)
        __builtin_amdgcn_s_setprio(ATTN_DQ_PRIO);
#endif
        dq_item(frame, head, std::integral_constant<int, PEEL>{}, hook);
      }
      else tile_item(std::integral_constant<int, 1>{}, 2 * HEAVY + (wave - HEAVY), frame, head, std::integral_constant<int, PEEL>{}, hook);
      lds_barrier();
    }
  }
}

template <auto Kern>
int set_lds(int bytes) {
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return -(int)e;
    done = true;
  }
  return 0;
}

constexpr float LOG2E = 1.4426950408889634f;

int cu_count();

template <int NT>
int launch_fwd(hipStream_t s, const void* qkv, void* o, float* lse, int64_t frames, float scale) {
  constexpr int N = NT * 32, LDV = N + 4;
  constexpr int bytes = (N * LDR + 32 * LDV) * 2;
  int rc = set_lds<attn_fwd_kernel<NT>>(bytes);
  if (rc) return rc;
  // (fewer (frame, head) items than a third of the CUs: one round of query tiles per workgroup)
  const int qsplit = (NT > 4 && frames * NH * 3 <= cu_count()) ? (NT + 3) / 4 : 1;
  hipLaunchKernelGGL(attn_fwd_kernel<NT>, dim3((unsigned)(frames * NH * qsplit)), dim3(256), bytes, s, (const uint16_t*)qkv,
                     (uint16_t*)o, lse, frames, scale * LOG2E, qsplit);
  HMA_CHECK_LAUNCH();
  return 0;
}

int cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}

#ifndef ATTN_KT  // key tiles per tile wave of the fused backward (measurement builds: -DATTN_KT=1, the round-3 form)
#define ATTN_KT 2
#endif
template <int NT>
int launch_bwd_fused(hipStream_t s, const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                     int64_t frames, float scale, int hb) {
  constexpr int N = NT * 32;
#if ATTN_BAL
  if constexpr (NT == 10) {
    constexpr int bytes = 6 * N * LDF * 2 + 2 * N * 4;
    int rc = set_lds<attn_bwd_bal_kernel<NT>>(bytes);
    if (rc) return rc;
    const int64_t items = frames * NH;
    const int grid = (int)(items < cu_count() ? items : cu_count());
    hipLaunchKernelGGL((attn_bwd_bal_kernel<NT>), dim3((unsigned)grid), dim3(512), bytes, s, (const uint16_t*)qkv, (const uint16_t*)o,
                       (const uint16_t*)d_o, lse, delta, (uint16_t*)dqkv, frames, scale * LOG2E, scale, hb);
    HMA_CHECK_LAUNCH();
    return 0;
  }
#endif
  constexpr int KT = (NT >= 8 && NT % ATTN_KT == 0) ? ATTN_KT : 1;
  constexpr int bytes = 6 * N * LDF * 2 + 2 * N * 4;
  int rc = set_lds<attn_bwd_fused_kernel<NT, KT>>(bytes);
  if (rc) return rc;
  const int64_t items = frames * NH;
  const int grid = (int)(items < cu_count() ? items : cu_count());
  hipLaunchKernelGGL((attn_bwd_fused_kernel<NT, KT>), dim3((unsigned)grid), dim3((NT / KT + 2) * 64), bytes, s, (const uint16_t*)qkv, (const uint16_t*)o,
                     (const uint16_t*)d_o, lse, delta, (uint16_t*)dqkv, frames, scale * LOG2E, scale, hb);
  HMA_CHECK_LAUNCH();
  return 0;
}

template <int NT>
int launch_bwd(hipStream_t s, const void* qkv, const void* o, const void* d_o, const float* lse, float* delta,
               void* dqkv, int64_t frames, float scale, int hb) {
#ifndef ATTN_BWD_SPLIT  // (debug builds: the two-kernel form, tools/attn_bench.py compares)
  if (NT >= 8) return launch_bwd_fused<NT>(s, qkv, o, d_o, lse, delta, dqkv, frames, scale, hb);
#endif
  constexpr int N = NT * 32, LDV = N + 4;
  constexpr int bytes_dq = 2 * N * LDR * 2;
  constexpr int NRh = N / 2;
  constexpr int bytes_dkv = (2 * NRh * LDR + 2 * 32 * (NRh + 4)) * 2 + 2 * NRh * 4;
  int rc = set_lds<attn_bwd_dq_kernel<NT>>(bytes_dq);
  if (rc) return rc;
  rc = set_lds<attn_bwd_dkv_kernel<NT>>(bytes_dkv);
  if (rc) return rc;
  hipLaunchKernelGGL(attn_bwd_dq_kernel<NT>, dim3((unsigned)(frames * NH)), dim3(256), bytes_dq, s,
                     (const uint16_t*)qkv, (const uint16_t*)o, (const uint16_t*)d_o, lse, delta, (uint16_t*)dqkv, frames,
                     scale * LOG2E, scale, hb);
  HMA_CHECK_LAUNCH();
  hipLaunchKernelGGL(attn_bwd_dkv_kernel<NT>, dim3((unsigned)(frames * NH)), dim3(256), bytes_dkv, s,
                     (const uint16_t*)qkv, (const uint16_t*)d_o, lse, (const float*)delta, (uint16_t*)dqkv, frames,
                     scale * LOG2E, scale, hb);
  HMA_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int hma_attn_spatial_fwd(void* stream, const void* qkv, void* o, float* lse, int64_t frames, int32_t n,
                                    float scale) {
  if (!qkv || !o || !lse) return HMA_EINVAL;
  if (frames <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  switch (n) {
    case 320: return launch_fwd<10>(s, qkv, o, lse, frames, scale);
    case 256: return launch_fwd<8>(s, qkv, o, lse, frames, scale);
    case 64: return launch_fwd<2>(s, qkv, o, lse, frames, scale);
    default: return HMA_EINVAL;
  }
}

static int attn_spatial_bwd_any(void* stream, const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                                int64_t frames, int32_t n, float scale, int hb) {
  if (!qkv || !o || !d_o || !lse || !delta || !dqkv) return HMA_EINVAL;
  if (frames <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  switch (n) {
    case 320: return launch_bwd<10>(s, qkv, o, d_o, lse, delta, dqkv, frames, scale, hb);
    case 256: return launch_bwd<8>(s, qkv, o, d_o, lse, delta, dqkv, frames, scale, hb);
    case 64: return launch_bwd<2>(s, qkv, o, d_o, lse, delta, dqkv, frames, scale, hb);
    default: return HMA_EINVAL;
  }
}

extern "C" int hma_attn_spatial_bwd(void* stream, const void* qkv, const void* o, const void* d_o, const float* lse,
                                    float* delta, void* dqkv, int64_t frames, int32_t n, float scale) {
  return attn_spatial_bwd_any(stream, qkv, o, d_o, lse, delta, dqkv, frames, n, scale, 0);
}

extern "C" int hma_attn_spatial_bwd_blocked(void* stream, const void* qkv, const void* o, const void* d_o, const float* lse,
                                            float* delta, void* dqkv, int64_t frames, int32_t n, float scale) {
  return attn_spatial_bwd_any(stream, qkv, o, d_o, lse, delta, dqkv, frames, n, scale, 1);
}
