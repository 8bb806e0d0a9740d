// Fused MLP block of the ST-transformer for gfx950:  x += fc2(gelu(fc1(norm2(x))))  and its backward.
// Reference: Mlp.forward (hma/model/st_transformer.py:24-27) as called from STBlock.forward (:112).
//
// The 1024-wide hidden activation never exists in HBM in the forward pass, and in the backward pass the
// pre-activation is RECOMPUTED from the saved LayerNorm output instead of being stored.
//
// Structure shared by both kernels (512 threads = 8 waves, one workgroup per CU, 128 token rows per tile):
//   * four producer / consumer wave PAIRS, each owning 32 token rows.  A producer wave and a consumer wave share every
//     SIMD, so one wave's VALU work (GELU) runs beside the other's MFMAs.
//   * all MFMAs are v_mfma_f32_32x32x16_bf16 with the WEIGHTS as the A operand and the token rows as the B operand
//     ("swapped"): a lane then owns ONE token row.  The weight rows inside a 32-row block are permuted (rowmap) so
//     that the lane (token, hi) ends with 16 CONSECUTIVE output columns -> what the first GEMM leaves in a lane's
//     accumulator registers is, after bf16 packing, exactly the B operand of the second GEMM (no LDS round trip,
//     no shuffles), and every global access of a lane is 32 / 64 contiguous bytes.
//   * the token-row operands (xhat rows; dy rows in backward) live in REGISTERS for the whole tile; the output
//     accumulators (32 tokens x 256 columns = 128 VGPRs) live in the consumer's registers for the whole tile.
//   * only the weights move: they are pre-packed in MFMA-fragment order (hma_mlp_pack), so a 1 KB LDS-DMA piece is
//     1 KB of contiguous global memory and a fragment read is one conflict-free lane-linear ds_read_b128.  The stream
//     (1 MB per tile in forward) comes out of L2 through a ring of bundles, one bundle per 32 hidden units.
//   * per step (32 hidden units) the producer computes u = W1 xhat (+ dgelu inputs in backward), applies the
//     activation and hands the packed bf16 tile to its consumer through 2 KB of LDS; the consumer multiplies it
//     into the output accumulators one step later.  One raw s_barrier per step.
#include "hma_common.h"
#include "../../include/hma_hip.h"

using namespace hma;

namespace {

// MFMA row rho (0..31) of a weight fragment holds logical row rowmap(rho) of its 32-row block: the accumulator
// registers r = 0..15 of lane (n, hi) are then the logical rows 16 hi + r.
__host__ __device__ constexpr int rowmap(int rho) { return (rho & 3) + 4 * (rho >> 3) + 16 * ((rho >> 2) & 1); }

__device__ __forceinline__ bf16x8_t as_frag(const uint4& v) { return __builtin_bit_cast(bf16x8_t, v); }
// one MFMA fragment (or any 16-byte piece) of this lane from LDS
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
#ifndef MLP_ABL
#define MLP_ABL 0
#endif
__device__ __forceinline__ bf16x8_t lds_frag(HMA_LDS(char)* p) {
  if (MLP_ABL & 32) return __builtin_bit_cast(bf16x8_t, make_uint4((uint32_t)(uintptr_t)p, 1u, 2u, 3u));  // (debug: no fragment reads)
  return __builtin_bit_cast(bf16x8_t, *(HMA_LDS(u32x4_t)*)p);
}
__device__ __forceinline__ void lds_put(HMA_LDS(char)* p, const uint4& v) { *(HMA_LDS(u32x4_t)*)p = __builtin_bit_cast(u32x4_t, v); }
__device__ __forceinline__ float4 lds_f4(HMA_LDS(char)* p) {
  const f32x4_t v = *(HMA_LDS(f32x4_t)*)p;
  return make_float4(v[0], v[1], v[2], v[3]);
}

// (the staged exact-erf GELU of 8 / 16 accumulator values -- gelu_n / gelu_bwd_n -- lives in hma_common.h: chain B shares it)

// ------------------------------------------------------------------------------------------------ weight packing
// kind 0 ("K256"): logical A[1024][256]; fragment (mb = 0..31, j = 0..15), lane (rho, hi), element i holds
//                  A[32 mb + rowmap(rho)][32 (j >> 1) + 16 hi + 8 (j & 1) + i]
// kind 1 ("H32") : logical A[256][1024]; fragment (s = 0..31, mb = 0..7, j = 0..1) holds
//                  A[32 mb + rowmap(rho)][32 s + 16 hi + 8 j + i]
// Both: fragment f of 512, 1 KB each, lane-linear.  A[r][c] = src[r * rs + c * cs] * rscale[r] * cscale[c].
__device__ __forceinline__ void mlp_pack_body(const float* __restrict__ src, int64_t rs, int64_t cs, const float* __restrict__ rscale,
                                              const float* __restrict__ cscale, uint16_t* __restrict__ dst, int kind, int64_t sstride,
                                              int64_t dstride, int bx, int64_t bz) {
  src += bz * sstride;
  if (rscale) rscale += bz * sstride;
  if (cscale) cscale += bz * sstride;
  dst += bz * dstride;
  const int idx = bx * 256 + threadIdx.x;
  const int lane = idx & 63, frag = idx >> 6, rho = lane & 31, hi = lane >> 5;
  int row, col0;
  if (kind == 0) {
    const int mb = frag >> 4, j = frag & 15;
    row = 32 * mb + rowmap(rho);
    col0 = 32 * (j >> 1) + 16 * hi + 8 * (j & 1);
  } else {
    const int s = frag >> 4, mb = (frag >> 1) & 7, j = frag & 1;
    row = 32 * mb + rowmap(rho);
    col0 = 32 * s + 16 * hi + 8 * j;
  }
  const float rsc = rscale ? rscale[row] : 1.0f;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = col0 + i;
    float w = src[(int64_t)row * rs + (int64_t)c * cs];
    if (rscale) w *= rsc;
    if (cscale) w *= cscale[c];
    v[i] = w;
  }
  *reinterpret_cast<uint4*>(dst + (int64_t)idx * 8) = pack8(v);
}
__global__ __launch_bounds__(256) void mlp_pack_kernel(const float* __restrict__ src, int64_t rs, int64_t cs,
                                                       const float* __restrict__ rscale, const float* __restrict__ cscale,
                                                       uint16_t* __restrict__ dst, int kind, int64_t sstride, int64_t dstride) {
  mlp_pack_body(src, rs, cs, rscale, cscale, dst, kind, sstride, dstride, blockIdx.x, blockIdx.y);
}
// hma_mlp_pack_multi: blockIdx.y = (job, batch index) through the prefix sums of the jobs' batch counts
constexpr int PACK_JOBS = 24;
struct pack_jobs {
  hma_pack_job_t j[PACK_JOBS];
  int y0[PACK_JOBS + 1];
  int n;
};
__global__ __launch_bounds__(256) void mlp_pack_multi_kernel(pack_jobs jobs) {
  int k = 0;
  while (k + 1 < jobs.n && (int)blockIdx.y >= jobs.y0[k + 1]) ++k;
  const hma_pack_job_t& j = jobs.j[k];
  mlp_pack_body(j.src, j.row_stride, j.col_stride, j.row_scale, j.col_scale, reinterpret_cast<uint16_t*>(j.dst), j.kind,
                j.src_batch_stride, j.dst_batch_stride, blockIdx.x, (int)blockIdx.y - jobs.y0[k]);
}

// Debug builds only (tools/mlp_ablate.sh, -DMLP_ABL=bits): 1 no LDS-DMA, 2 no MFMA, 4 no GELU, 8 no barrier,
// 16 no activation loads / stores (forward kernel), 32 no LDS fragment reads, 64 no backward epilogue, 128 no hg / du stores
__device__ __forceinline__ f32x16_t mfma32a(const bf16x8_t& a, const bf16x8_t& b, const f32x16_t& c) {
  if (MLP_ABL & 2) {
    f32x16_t r = c;
    r[0] += __builtin_bit_cast(float, __builtin_bit_cast(uint4, a).x ^ __builtin_bit_cast(uint4, b).y);
    return r;
  }
  return mfma32(a, b, c);
}

// Debug builds only (-DMLP_PROF): per-wave s_memtime deltas summed per phase, block 0 only (tools/mlp_prof.py)
#ifdef MLP_PROF
__device__ unsigned long long g_mlp_prof[2][8][8];
#define MPROF_DECL unsigned long long pt_ = __builtin_readcyclecounter(), pacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define MPROF_MARK(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); pacc_[i] += n_ - pt_; pt_ = n_; } while (0)
#define MPROF_FLUSH(k) do { if (blockIdx.x == 0 && lane == 0) { for (int i_ = 0; i_ < 8; ++i_) g_mlp_prof[k][wave][i_] = pacc_[i_]; } } while (0)
#else
#define MPROF_DECL
#define MPROF_MARK(i)
#define MPROF_FLUSH(k)
#endif

// ------------------------------------------------------------------------------------------------ forward
// Per SIMD one producer wave and one consumer wave, in ANTI-PHASE within a step: a wave cannot hide much VALU work
// behind its own MFMAs (it issues in order: ~5 instructions per 32-cycle MFMA), but a wave in a VALU phase and a wave
// in an MFMA phase on the same SIMD run concurrently.  So the GELU of a hidden block is split between the pair:
//   producer, step g:      16 MFMAs (u of block g), then the GELU of its accumulator registers 0..7
//   consumer, step g + 1:  the GELU of registers 8..15 (received as fp32), then 16 MFMAs (block g into the outputs)
// and after the step barrier one wave of every SIMD starts on the matrix pipe while the other starts on the VALU.
constexpr int MF_NSLOT = 3;                  // ring slots
constexpr int MF_AHEAD = 2;                  // bundles in flight ahead of the one being used
constexpr int MF_SLOT = 32768;               // bundle g = fc1 fragments of hidden block g (16 KB) | fc2 fragments of block g - 1
constexpr int MF_XCH = MF_NSLOT * MF_SLOT;   // per pair: 2 buffers x 3 planes x 1 KB (gelu half as bf16 | raw half, fp32)
constexpr int MF_XCH_PAIR = 6144;
constexpr int MF_B1 = MF_XCH + 4 * MF_XCH_PAIR;
constexpr int MF_B2 = MF_B1 + 4096;
constexpr int MF_STG = MF_B2 + 1024;         // per pair: 2 staging pieces of 4 KB (32 rows x 128 B) for the consumer's row I/O
constexpr int MF_SMEM = MF_STG + 4 * 8192;   // 160768 B

// A staging piece holds 32 token rows x 128 bytes.  16-byte chunk c of row r sits at chunk c ^ ((r >> 1) & 7): both access
// shapes are then conflict-free -- "a lane owns a row" (the accumulator layout: ds_read/write_b128 of one chunk index by
// 16 lanes with 16 different rows) and "8 lanes per row" (the 1 KB, 8-row pieces of coalesced global traffic).
__device__ __forceinline__ int stg_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// two consecutive LDS-DMA pieces (2 KB) behind one M0 set-up
__device__ __forceinline__ void glds16x2(const void* src, uint32_t dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "global_load_lds_dwordx4 %1, off offset:1024\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(dst)
      : "memory");
}

template <bool LNOUT>
__global__ __launch_bounds__(512, 2) void mlp_fwd_kernel(hma_mlp_fwd_t p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  HMA_LDS(char)* lds = (HMA_LDS(char)*)smem;
  const uint32_t lds_b = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 2, pair = wave & 3;  // waves w and w + 4 share a SIMD: one producer + one consumer each
  const int lr = lane & 31, hi = lane >> 5;
  const int64_t ntiles = (p.M + 127) >> 7;
  const int nt = (int)((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x);
  const int nsteps = nt * 32;
  // (Walking the hidden blocks in a different rotation per workgroup, so that the CUs do not all request the same 32 KB
  // of weights at once, measured 10 % SLOWER: the L2 serves the broadcast well.)
  constexpr int rot = 0;

  {
    HMA_LDS(float)* b1s = (HMA_LDS(float)*)(lds + MF_B1);
    for (int i = tid; i < 1024; i += 512) b1s[i] = p.b1[i];
    if (tid < 256) ((HMA_LDS(float)*)(lds + MF_B2))[tid] = p.b2 ? p.b2[tid] : 0.f;
  }
#ifdef MLP_STAGGER
  {  // experiment: de-phase the workgroups (their tile epilogues otherwise hit HBM all at once)
    const int nsl = (((int)blockIdx.x * 37) & 255) * MLP_STAGGER / 256;
    for (int i = 0; i < nsl; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif
  __syncthreads();

  // LDS-DMA: the four PRODUCER waves move everything (8 of a bundle's 32 pieces each, plus their pair's residual blocks):
  // issuing a piece is ~50-150 cycles of a wave's time, and the consumer wave is the longer one of a pair.
  const char* w1g = reinterpret_cast<const char*>(p.w1p) + pair * 4096 + lane * 16;
  const char* w2g = reinterpret_cast<const char*>(p.w2p) + pair * 4096 + lane * 16;
  auto issue = [&](int b) __attribute__((always_inline)) {
    const uint32_t base = lds_b + (b % MF_NSLOT) * MF_SLOT + pair * 4096;
    const int s1 = (b + rot) & 31, s2 = (b + rot + 31) & 31;
    if (MLP_ABL & 1) return;
    glds16x4(w1g + s1 * 16384, base);
    glds16x4(w2g + s2 * 16384, base + 16384);
  };
  // Bundle g has landed and this wave's LDS writes of the previous step are done.  `newer` = vector-memory loads this wave
  // issued AFTER bundle g's pieces (loads return in order, so vmcnt(newer) means bundle g is complete): the 8 of bundle
  // g + 1, plus 4 when a residual block was issued in the previous step, plus the 16 row loads of the next tile.
  auto step_sync = [&](int newer) __attribute__((always_inline)) {
    if (newer >= 24)
      asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");
    else if (newer >= 12)
      asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    else if (newer >= 8)
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (!(MLP_ABL & 8)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto tile_row0 = [&](int tl) __attribute__((always_inline)) {
    return ((int64_t)blockIdx.x + (int64_t)tl * gridDim.x) * 128 + pair * 32;
  };
  // residual block cb of this pair's rows of tile tl -> staging piece cb & 1 (see the consumer)
  const uint32_t stg_b = lds_b + MF_STG + pair * 8192;
  const int prow_ = lane >> 3, pchunk_ = lane & 7;  // "8 lanes per row" shape: this lane's row within 8 and physical chunk
  auto issue_x = [&](int tl, int cb) __attribute__((always_inline)) {
    if (MLP_ABL & 17) return;
    const int64_t r0 = tile_row0(tl);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 8 * i + prow_;
      int64_t gr = r0 + r;
      gr = gr < p.M ? gr : p.M - 1;
      glds16(p.x + gr * 256 + cb * 32 + ((pchunk_ ^ ((r >> 1) & 7)) << 2), stg_b + (cb & 1) * 4096 + i * 1024);
    }
  };
  if (role == 0) {
#pragma unroll
    for (int b = 0; b < MF_AHEAD; ++b) issue(b);
    // ---------------------------------------------------------------- producer: u = W1f xhat + b1; gelu of registers 0..7
    bf16x8_t xh[16], xn[16];
    auto load_x = [&](int tl, bf16x8_t (&dst)[16]) __attribute__((always_inline)) {
      int64_t row = tile_row0(tl) + lr;
      row = row < p.M ? row : p.M - 1;
      const uint16_t* src = reinterpret_cast<const uint16_t*>(p.xhat) + row * 256 + 16 * hi;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        dst[j] = (MLP_ABL & 16) ? as_frag(make_uint4(j, lane, j, lane)) : as_frag(*reinterpret_cast<const uint4*>(src + 32 * (j >> 1) + 8 * (j & 1)));
    };
    load_x(0, xh);
#pragma unroll
    for (int j = 0; j < 16; ++j) xn[j] = xh[j];
    MPROF_DECL;
    for (int g = 0; g <= nsteps; ++g) {
      MPROF_MARK(0);
      // (the 16 row loads of the next tile, issued at the end of an iteration with s == 16, are newer than bundle g too)
      step_sync((g < nsteps ? 8 : 0) + ((g >= 2 && ((g - 2) & 3) == 0) ? 4 : 0) +
                ((g >= 1 && ((g - 1) & 31) == 16 && ((g - 1) >> 5) + 1 < nt) ? 16 : 0));
      MPROF_MARK(1);
      // residual block for the consumer's hidden block g - 1 when that is a multiple of 4 (added two steps later); issued
      // BEFORE the bundle so that it is older than the bundle whose arrival the consumer's step g + 2 waits for
      if (g >= 1 && g <= nsteps && ((g - 1) & 3) == 0) issue_x((g - 1) >> 5, ((g - 1) & 31) >> 2);
#ifdef MLP_DMA_INTERLEAVE
      // (the bundle's 8 pieces are issued between the MFMA groups below: a piece costs 60-150 cycles of issue time, which
      // an in-order wave otherwise spends before its first MFMA)
      const bool do_issue = g + MF_AHEAD <= nsteps && !(MLP_ABL & 1);
      const int bi = g + MF_AHEAD;
      const uint32_t ibase = lds_b + (bi % MF_NSLOT) * MF_SLOT + pair * 4096;
      const char* isrc1 = w1g + ((bi + rot) & 31) * 16384;
      const char* isrc2 = w2g + ((bi + rot + 31) & 31) * 16384;
#else
      if (g + MF_AHEAD <= nsteps) issue(g + MF_AHEAD);
#endif
      MPROF_MARK(2);
      if (g < nsteps) {
        const int s = g & 31;
        if (s == 0 && g > 0) {
#pragma unroll
          for (int j = 0; j < 16; ++j) xh[j] = xn[j];
        }
        HMA_LDS(char)* wb = lds + (g % MF_NSLOT) * MF_SLOT + lane * 16;
        HMA_LDS(char)* bp = lds + MF_B1 + (32 * ((s + rot) & 31) + 16 * hi) * 4;
        f32x16_t U0, U1;  // U0 starts from the bias
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b = lds_f4(bp + 16 * q);
          U0[4 * q + 0] = b.x; U0[4 * q + 1] = b.y; U0[4 * q + 2] = b.z; U0[4 * q + 3] = b.w;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) U1[e] = 0.f;
#ifndef MLP_NOPRIO
        __builtin_amdgcn_s_setprio(1);  // (this wave's MFMAs ahead of a consumer that is late with its own)
#endif
        {
          bf16x8_t fa[4], fb[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) fa[i] = lds_frag(wb + i * 1024);
#pragma unroll
          for (int grp = 0; grp < 4; ++grp) {
            if (grp < 3) {
#pragma unroll
              for (int i = 0; i < 4; ++i) fb[i] = lds_frag(wb + (4 * grp + 4 + i) * 1024);
            }
            U0 = mfma32a(fa[0], xh[4 * grp + 0], U0);
            U1 = mfma32a(fa[1], xh[4 * grp + 1], U1);
#ifdef MLP_DMA_INTERLEAVE
            if (do_issue) {
              if (grp < 2) glds16x2(isrc1 + grp * 2048, ibase + grp * 2048);
              else glds16x2(isrc2 + (grp - 2) * 2048, ibase + 16384 + (grp - 2) * 2048);
            }
#endif
            U0 = mfma32a(fa[2], xh[4 * grp + 2], U0);
            U1 = mfma32a(fa[3], xh[4 * grp + 3], U1);
            __builtin_amdgcn_sched_barrier(0);  // keeps the scheduler from hoisting every fragment read (it spills)
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = fb[i];
          }
        }
        __builtin_amdgcn_s_setprio(0);
        MPROF_MARK(3);
        HMA_LDS(char)* xc = lds + MF_XCH + pair * MF_XCH_PAIR + (g & 1) * 3072 + lane * 16;
        {  // registers 8..15 (the second k-step of the consumer's product) leave as fp32: the consumer applies the GELU
          float uc[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) uc[e] = U0[8 + e] + U1[8 + e];
          lds_put(xc + 1024, __builtin_bit_cast(uint4, make_float4(uc[0], uc[1], uc[2], uc[3])));
          lds_put(xc + 2048, __builtin_bit_cast(uint4, make_float4(uc[4], uc[5], uc[6], uc[7])));
        }
        float up[8], hp[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) up[e] = U0[e] + U1[e];
        if (MLP_ABL & 4) {
#pragma unroll
          for (int e = 0; e < 8; ++e) hp[e] = up[e];
        } else {
          gelu_n<8>(up, hp);
        }
        lds_put(xc, pack8(hp));
        MPROF_MARK(4);
        if (s == 16 && (g >> 5) + 1 < nt) load_x((g >> 5) + 1, xn);  // next tile's rows, landed long before they are needed
        MPROF_MARK(5);
      }
    }
    MPROF_FLUSH(0);
  } else {
    // ---------------------------------------------------------------- consumer: x += hg W2^T + b2 (+ LayerNorm of the new row)
    // The residual rows reach the accumulators through the staging pieces: one 32 x 32 fp32 block every 4th step by
    // LDS-DMA (1 KB = 8 rows x 128 B per instruction: whole cache lines), added two steps later -- the register-direct
    // alternative (every lane fetching 16-byte pieces of its own row, all at the tile boundary) cost 35 k cycles per tile.
    f32x16_t Y[8];
    HMA_LDS(char)* stg = lds + MF_STG + pair * 8192;
    MPROF_DECL;
    for (int g = 0; g <= nsteps; ++g) {
      MPROF_MARK(0);
      // (this wave has no loads of its own in flight: the producers' vmcnt waits + the barrier cover the LDS-DMA; its
      // own global stores of a tile epilogue need no wait)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!(MLP_ABL & 8)) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      MPROF_MARK(1);
      MPROF_MARK(2);
      if (g >= 1) {
        const int gc = g - 1, s = gc & 31;
        if (s == 0) {
#pragma unroll
          for (int cb = 0; cb < 8; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) Y[cb][e] = 0.f;
        }
        // VALU phase first (beside the producer's MFMAs): the GELU of the half that arrived as fp32
        HMA_LDS(char)* xc = lds + MF_XCH + pair * MF_XCH_PAIR + (gc & 1) * 3072 + lane * 16;
        float uc[8], hc[8];
        {
          const float4 v0 = lds_f4(xc + 1024), v1 = lds_f4(xc + 2048);
          uc[0] = v0.x; uc[1] = v0.y; uc[2] = v0.z; uc[3] = v0.w; uc[4] = v1.x; uc[5] = v1.y; uc[6] = v1.z; uc[7] = v1.w;
        }
        if (MLP_ABL & 4) {
#pragma unroll
          for (int e = 0; e < 8; ++e) hc[e] = uc[e];
        } else {
          gelu_n<8>(uc, hc);
        }
        const bf16x8_t h1 = as_frag(pack8(hc));
        const bf16x8_t h0 = lds_frag(xc);
        MPROF_MARK(3);
        HMA_LDS(char)* wb = lds + (g % MF_NSLOT) * MF_SLOT + 16384 + lane * 16;
        {
          // fragment order in the bundle: (cb, j) -> 2 cb + j; walked as j = 0: cb 0..7, then j = 1: cb 0..7
          bf16x8_t fa[4], fb[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) fa[i] = lds_frag(wb + (2 * i) * 1024);
#pragma unroll
          for (int grp = 0; grp < 4; ++grp) {
            if (grp < 3) {
              const int gn = grp + 1;
#pragma unroll
              for (int i = 0; i < 4; ++i) fb[i] = lds_frag(wb + (2 * (4 * (gn & 1) + i) + (gn >> 1)) * 1024);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int cb = 4 * (grp & 1) + i;
              Y[cb] = mfma32a(fa[i], (grp >> 1) ? h1 : h0, Y[cb]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = fb[i];
          }
        }
        if ((s & 3) == 2 && !(MLP_ABL & 16)) {  // the residual block issued two steps ago (complete: older than this step's bundle)
          const int cbx = s >> 2;
          float xv[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v = lds_f4(stg + (cbx & 1) * 4096 + stg_off(lr, 4 * hi + q));
            xv[4 * q + 0] = v.x; xv[4 * q + 1] = v.y; xv[4 * q + 2] = v.z; xv[4 * q + 3] = v.w;
          }
#pragma unroll
          for (int cb = 0; cb < 8; ++cb)
            if (cb == cbx) {
#pragma unroll
              for (int e = 0; e < 16; ++e) Y[cb][e] += xv[e];
            }
        }
        MPROF_MARK(4);
        if (s == 31) {
          // ---- tile epilogue: + b2, (LayerNorm statistics), rows out through the staging pieces (whole cache lines)
          const int64_t r0 = tile_row0(gc >> 5);
          HMA_LDS(char)* b2p = lds + MF_B2 + 16 * hi * 4;
          // (opaque copies: otherwise every row address of the epilogue is hoisted out of the step loop and spilled)
          int prow = prow_, pchunk = pchunk_;
          asm volatile("" : "+v"(prow), "+v"(pchunk));
          // LayerNorm statistics in one pass over d = x - shift, shift = the row's first element (both lanes of a row use
          // lane hi = 0's): sum d and sum d^2 of values at the scale of the row's spread, no cancellation
          float sum = 0.f, sq = 0.f, shift = 0.f;
#pragma unroll
          for (int cb = 0; cb < 8; ++cb) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 b = lds_f4(b2p + (8 * cb + q) * 16);
              Y[cb][4 * q + 0] += b.x; Y[cb][4 * q + 1] += b.y; Y[cb][4 * q + 2] += b.z; Y[cb][4 * q + 3] += b.w;
            }
            if (LNOUT) {
              if (cb == 0) shift = __shfl(Y[0][0], lr, 64);
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                const float d = Y[cb][e] - shift;
                sum += d;
                sq = __builtin_fmaf(d, d, sq);
              }
            }
            // new residual block -> staging (lane owns a row) -> 8 rows x 128 B per store instruction
            HMA_LDS(char)* sb = stg + (cb & 1) * 4096;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              lds_put(sb + stg_off(lr, 4 * hi + q), __builtin_bit_cast(uint4, make_float4(Y[cb][4 * q], Y[cb][4 * q + 1], Y[cb][4 * q + 2], Y[cb][4 * q + 3])));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int r = 8 * i + prow;
              const float4 v = lds_f4(sb + r * 128 + (pchunk << 4));
              if (r0 + r < p.M && !(MLP_ABL & 16)) *reinterpret_cast<float4*>(p.x + (r0 + r) * 256 + cb * 32 + ((pchunk ^ ((r >> 1) & 7)) << 2)) = v;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          if (LNOUT) {
            sum += __shfl_xor(sum, 32, 64);
            sq += __shfl_xor(sq, 32, 64);
            const float md = sum * (1.0f / 256.0f);                       // mean - shift
            const float var = fmaxf(sq * (1.0f / 256.0f) - md * md, 0.f);
            const float rstd = rsqrtf(var + p.ln_eps);
            const float nb = -(md + shift) * rstd;                       // xhat = x * rstd - mean * rstd
            __builtin_amdgcn_sched_barrier(0);
            // xhat of the next block: two 32-column blocks (64 bf16 = 128 B per row) per staging piece
#pragma unroll
            for (int cp = 0; cp < 4; ++cp) {
              HMA_LDS(char)* sb = stg + (cp & 1) * 4096;
#pragma unroll
              for (int k = 0; k < 2; ++k) {
                float o[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) o[e] = __builtin_fmaf(Y[2 * cp + k][e], rstd, nb);
                lds_put(sb + stg_off(lr, 4 * k + 2 * hi), pack8(o));
                lds_put(sb + stg_off(lr, 4 * k + 2 * hi + 1), pack8(o + 8));
                __builtin_amdgcn_sched_barrier(0);
              }
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const int r = 8 * i + prow;
                const uint4 v = __builtin_bit_cast(uint4, lds_f4(sb + r * 128 + (pchunk << 4)));
                if (r0 + r < p.M && !(MLP_ABL & 16))
                  *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.ln_xhat) + (r0 + r) * 256 + cp * 64 + ((pchunk ^ ((r >> 1) & 7)) << 3)) = v;
              }
              __builtin_amdgcn_sched_barrier(0);
            }
            if (hi == 0 && r0 + lr < p.M) p.ln_rstd[r0 + lr] = rstd;
          }
        }
      }
      MPROF_MARK(5);
    }
    MPROF_FLUSH(0);
  }
}

// ------------------------------------------------------------------------------------------------ backward
// Producer: u = W1f xhat + b1 (recomputed), dhg = W2^T dy, hg = gelu(u), du = dhg * gelu'(u); hg and du go to HBM for the
// two weight-gradient GEMMs (hma_gemm_tn_pair), du also to the consumer.  Consumer: g = dxhat = W1f^T du (gamma already
// folded into W1f), then the LayerNorm backward added to the residual gradient:
//     dx_new = dx_old + rstd (g - mean_k g - xhat mean_k (g xhat)).
// The row statistic mean_k (g xhat) never needs xhat: sum_k g[k] xhat[k] = sum_h du[h] (W1f xhat)[h], i.e. the producer's
// du times its own pre-bias accumulator, summed over the hidden units as they go by.  dx_old reaches the accumulators
// during the main loop (one 32 x 32 fp32 block every 4th step by LDS-DMA, scaled by 1 / rstd when added), so the tile
// epilogue only subtracts the xhat term (4 staged pieces), scales, and writes whole cache lines through the staging
// pieces.  All LDS-DMA is issued by the CONSUMER waves (the producers' queues hold the hg / du stores, which drain slowly
// and out of order with loads: a producer never waits on vmcnt inside the loop).
constexpr int MB_NSLOT = 2;
constexpr int MB_SLOT = 49152;               // bundle g = fc1 frags | fc2^T frags of hidden block g | fc1^T frags of block g - 1
constexpr int MB_XCH = MB_NSLOT * MB_SLOT;   // per pair: 2 buffers x 2 planes x 1 KB (du, bf16)
constexpr int MB_B1 = MB_XCH + 4 * 4096;
constexpr int MB_ST2 = MB_B1 + 4096;         // per pair: 32 floats (sum_h du (W1f xhat) per token row) | 32 floats (rstd)
constexpr int MB_STG = MB_ST2 + 1024;        // per pair: 2 staging pieces of 4 KB
constexpr int MB_SMEM = MB_STG + 4 * 8192;   // 152576 B

// DROP (mlp_drop > 0): the forward's two masks re-created from the counter-based hash (hma_common.h drop_keep; chain B forward or the
// GELU2 / RESID epilogues applied them): dy is masked as it is loaded (and written out for the fc2 weight gradient), hg and the
// gradient entering gelu' carry the activation mask.
template <bool DROP>
__global__ __launch_bounds__(512, 2) void mlp_bwd_kernel(hma_mlp_bwd_t p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  HMA_LDS(char)* lds = (HMA_LDS(char)*)smem;
  const uint32_t lds_b = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 2, pair = wave & 3;
  const int lr = lane & 31, hi = lane >> 5;
  const int64_t ntiles = (p.M + 127) >> 7;
  const int nt = (int)((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x);
  const int nsteps = nt * 32;

  {
    HMA_LDS(float)* b1s = (HMA_LDS(float)*)(lds + MB_B1);
    for (int i = tid; i < 1024; i += 512) b1s[i] = p.b1[i];
  }
#ifdef MLP_STAGGER
  {
    const int nsl = (((int)blockIdx.x * 37) & 255) * MLP_STAGGER / 256;
    for (int i = 0; i < nsl; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif
  __syncthreads();

  auto tile_row0 = [&](int tl) __attribute__((always_inline)) {
    return ((int64_t)blockIdx.x + (int64_t)tl * gridDim.x) * 128 + pair * 32;
  };

  if (role == 0) {
    // ---------------------------------------------------------------- producer
    bf16x8_t xh[16], dy[16];
    uint4 sv[4];             // packed hg | du of the previous step, stored at the start of the next one
    uint16_t* sp_hg = nullptr;
    uint16_t* sp_du = nullptr;
    bool sv_ok = false;
    float s2acc = 0.f;
    uint32_t dseed = 0, dth = 0;
    float dsc = 1.f;
    if constexpr (DROP) {
      dseed = *p.drop_seed;
      dth = drop_thresh(p.drop_p);
      dsc = drop_scale(p.drop_p);
    }
    MPROF_DECL;
    for (int g = 0; g <= nsteps; ++g) {
      MPROF_MARK(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!(MLP_ABL & 8)) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      MPROF_MARK(1);
      if (g > 0 && sv_ok && !(MLP_ABL & 128)) {
        *reinterpret_cast<uint4*>(sp_hg) = sv[0];
        *reinterpret_cast<uint4*>(sp_hg + 512) = sv[1];
        *reinterpret_cast<uint4*>(sp_du) = sv[2];
        *reinterpret_cast<uint4*>(sp_du + 512) = sv[3];
      }
      MPROF_MARK(2);
      if (g < nsteps) {
        const int s = g & 31;
        const int64_t row = tile_row0(g >> 5) + lr;
        const int64_t rowc = row < p.M ? row : p.M - 1;
        if (s == 0 && (!(MLP_ABL & 2048) || g == 0)) {
          const uint16_t* xs = reinterpret_cast<const uint16_t*>(p.xhat) + rowc * 256 + 16 * hi;
          const uint16_t* ds = reinterpret_cast<const uint16_t*>(p.dy) + rowc * 256 + 16 * hi;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            xh[j] = as_frag(*reinterpret_cast<const uint4*>(xs + 32 * (j >> 1) + 8 * (j & 1)));
            dy[j] = as_frag(*reinterpret_cast<const uint4*>(ds + 32 * (j >> 1) + 8 * (j & 1)));
          }
          float rs = p.rstd[rowc];
          // (consumed inside this branch: see the forward kernel's note on hipcc's waitcnt placement)
#pragma unroll
          for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(xh[j]), "+v"(dy[j]));
          asm volatile("" : "+v"(rs));
          if (hi == 0) ((HMA_LDS(float)*)(lds + MB_ST2 + pair * 256 + 128))[lr] = rs;  // the consumer reads it one step later
          if constexpr (DROP) {  // the Dropout behind fc2: dy * keep / (1 - p), also written out for the fc2 weight / bias gradient
            uint16_t* dd = reinterpret_cast<uint16_t*>(p.dy_drop) + row * 256 + 16 * hi;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              const int c0 = 32 * (j >> 1) + 8 * (j & 1);
              float v[8];
              unpack8(__builtin_bit_cast(uint4, dy[j]), v);
#pragma unroll
              for (int e = 0; e < 8; e += 2) {
                bool k0, k1;
                drop_keep2(dseed, p.drop_salt + 1, row * 256 + 16 * hi + c0 + e, dth, k0, k1);
                v[e] = k0 ? v[e] * dsc : 0.f;
                v[e + 1] = k1 ? v[e + 1] * dsc : 0.f;
              }
              const uint4 pk = pack8(v);
              dy[j] = as_frag(pk);
              if (row < p.M) *reinterpret_cast<uint4*>(dd + c0) = pk;
            }
          }
        }
        MPROF_MARK(3);
        HMA_LDS(char)* wb = lds + (g % MB_NSLOT) * MB_SLOT + lane * 16;
        f32x16_t U, D;
#pragma unroll
        for (int e = 0; e < 16; ++e) U[e] = 0.f, D[e] = 0.f;
        {
          bf16x8_t fa[4], fb[4];  // (fc1 j, fc2^T j, fc1 j + 1, fc2^T j + 1)
#pragma unroll
          for (int i = 0; i < 4; ++i) fa[i] = lds_frag(wb + (i & 1) * 16384 + (i >> 1) * 1024);
#pragma unroll
          for (int grp = 0; grp < 8; ++grp) {
            if (grp < 7) {
#pragma unroll
              for (int i = 0; i < 4; ++i) fb[i] = lds_frag(wb + (i & 1) * 16384 + (2 * grp + 2 + (i >> 1)) * 1024);
            }
            U = mfma32a(fa[0], xh[2 * grp], U);
            D = mfma32a(fa[1], dy[2 * grp], D);
            U = mfma32a(fa[2], xh[2 * grp + 1], U);
            D = mfma32a(fa[3], dy[2 * grp + 1], D);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = fb[i];
          }
        }
        MPROF_MARK(4);
        HMA_LDS(char)* bp = lds + MB_B1 + (32 * s + 16 * hi) * 4;
        // (two halves of 8: the staged GELU holds ~8 values per element in flight)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          float u[8], dd[8], hg[8], du[8];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const float4 b = lds_f4(bp + 32 * half + 16 * q);
            u[4 * q + 0] = U[8 * half + 4 * q + 0] + b.x; u[4 * q + 1] = U[8 * half + 4 * q + 1] + b.y;
            u[4 * q + 2] = U[8 * half + 4 * q + 2] + b.z; u[4 * q + 3] = U[8 * half + 4 * q + 3] + b.w;
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) dd[e] = D[8 * half + e];
          bool keep[8];
          if constexpr (DROP) {  // the Dropout behind the GELU: the gradient passes the same mask before gelu'
            const int64_t e0 = row * 1024 + 32 * s + 16 * hi + 8 * half;
#pragma unroll
            for (int e = 0; e < 8; e += 2) drop_keep2(dseed, p.drop_salt, e0 + e, dth, keep[e], keep[e + 1]);
#pragma unroll
            for (int e = 0; e < 8; ++e) dd[e] = keep[e] ? dd[e] * dsc : 0.f;
          }
          if (MLP_ABL & 4) {
#pragma unroll
            for (int e = 0; e < 8; ++e) hg[e] = u[e], du[e] = dd[e];
          } else {
            gelu_bwd_n<8>(u, dd, hg, du);
          }
          if constexpr (DROP) {
#pragma unroll
            for (int e = 0; e < 8; ++e) hg[e] = keep[e] ? hg[e] * dsc : 0.f;
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) s2acc = __builtin_fmaf(du[e], U[8 * half + e], s2acc);  // du (W1f xhat): the pre-bias accumulator
          sv[half] = pack8(hg);
          sv[2 + half] = pack8(du);
        }
        sv_ok = row < p.M;
        {  // HMA_A_BF16_FRAG32: (tile, hidden block, pair) -> 2 KB = [the lane's units 0..7 | 8..15][64 lanes][8 units]: each of
           // the four store instructions of a step writes 1 KB of contiguous memory (row-major, a lane's 16 bytes sit in a
           // 2 KB-pitch row: 32 partial cache lines per instruction, which held up the LDS-DMA issue behind them)
          const int64_t blk = ((((int64_t)blockIdx.x + (int64_t)(g >> 5) * gridDim.x) * 32 + s) * 4 + pair) * 1024 + lane * 8;
          sp_hg = reinterpret_cast<uint16_t*>(p.hg) + blk;
          sp_du = reinterpret_cast<uint16_t*>(p.du) + blk;
        }
        HMA_LDS(char)* xc = lds + MB_XCH + pair * 4096 + (g & 1) * 2048 + lane * 16;
        lds_put(xc, sv[2]);
        lds_put(xc + 1024, sv[3]);
        if (s == 31) {  // the tile's sum_h du (W1f xhat) per token row -> the consumer (read after the next barrier)
          const float t2 = s2acc + __shfl_xor(s2acc, 32, 64);
          if (hi == 0) ((HMA_LDS(float)*)(lds + MB_ST2 + pair * 256))[lr] = t2;
          s2acc = 0.f;
        }
        MPROF_MARK(5);
      }
    }
    MPROF_FLUSH(1);
  } else {
    // ---------------------------------------------------------------- consumer
    f32x16_t G[8];
    HMA_LDS(char)* stg = lds + MB_STG + pair * 8192;
    const uint32_t stg_b = lds_b + MB_STG + pair * 8192;
    const int prow_ = lane >> 3, pchunk_ = lane & 7;
    const char* g1 = reinterpret_cast<const char*>(p.w1p) + pair * 4096 + lane * 16;
    const char* g2 = reinterpret_cast<const char*>(p.w2tp) + pair * 4096 + lane * 16;
    const char* g3 = reinterpret_cast<const char*>(p.w1tp) + pair * 4096 + lane * 16;
    // this wave's 12 KB of bundle b, in three parts (issued between the MFMA groups of the step: a burst of twelve 1 KB
    // pieces right behind the barrier held the wave at the issue for as long as the MFMAs it had not started yet)
    auto issue_part = [&](int b, int part) __attribute__((always_inline)) {
      const uint32_t base = lds_b + (b % MB_NSLOT) * MB_SLOT + pair * 4096;
      const int s1 = b & 31, s2 = (b + 31) & 31;
      if (MLP_ABL & 1) return;
      if (part == 0) glds16x4(g1 + s1 * 16384, base);
      if (part == 1) glds16x4(g2 + s1 * 16384, base + 16384);
      if (part == 2) glds16x4(g3 + s2 * 16384, base + 32768);
    };
    auto issue = [&](int b) __attribute__((always_inline)) {
      issue_part(b, 0);
      issue_part(b, 1);
      issue_part(b, 2);
    };
    // one 32-row x 128-byte block of a row-major matrix (fp32 dx: 32 columns; bf16 xhat: 64 columns) -> staging piece `buf`
    auto issue_rows = [&](const char* base, int64_t row_bytes, int tl, int buf) __attribute__((always_inline)) {
      if (MLP_ABL & 65) return;
      const int64_t r0 = tile_row0(tl);
      int pr = prow_, pc = pchunk_;
      asm volatile("" : "+v"(pr), "+v"(pc));  // (opaque: the address arithmetic of every call site is otherwise hoisted and spilled)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + pr;
        int64_t gr = r0 + r;
        gr = gr < p.M ? gr : p.M - 1;
        glds16(base + gr * row_bytes + ((pc ^ ((r >> 1) & 7)) << 4), stg_b + buf * 4096 + i * 1024);
      }
    };
    auto wait_vm = [&](int newer) __attribute__((always_inline)) {  // all but the `newer` youngest loads of this wave are complete
      if (newer >= 63)
        return;
      else if (newer >= 12)
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (newer >= 9)
        asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      else if (newer >= 8)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (newer >= 5)
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else if (newer >= 4)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) G[kb][e] = 0.f;
    issue(0);
    int pend = 0;            // loads issued after the newest bundle (they are younger than it: vmcnt(pend) = bundle complete)
    float invr = 1.f, rstd = 1.f, sumold = 0.f;
    MPROF_DECL;
    for (int g = 0; g <= nsteps; ++g) {
      MPROF_MARK(0);
      wait_vm(pend);
      MPROF_MARK(1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!(MLP_ABL & 8)) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      MPROF_MARK(2);
      const bool nxt = g + 1 <= nsteps;
      pend = 0;
      if (g == 0 && nxt) issue(g + 1);
      if (g >= 1) {
        const int gc = g - 1, s = gc & 31, tl = gc >> 5;
        if (nxt) issue_part(g + 1, 0);
        if (s == 0) {
          int lro = lr;
          asm volatile("" : "+v"(lro));  // (opaque: hoisted out of the loop this address is spilled, and reloaded behind a vmcnt(0))
          rstd = ((HMA_LDS(float)*)(lds + MB_ST2 + pair * 256 + 128))[lro];  // published by the producer at its s == 0
          invr = 1.0f / rstd;
          sumold = 0.f;
        }
        MPROF_MARK(3);
        HMA_LDS(char)* xc = lds + MB_XCH + pair * 4096 + (gc & 1) * 2048 + lane * 16;
        const bf16x8_t d0 = lds_frag(xc);
        const bf16x8_t d1 = lds_frag(xc + 1024);
        HMA_LDS(char)* wb = lds + (g % MB_NSLOT) * MB_SLOT + 32768 + lane * 16;
        {
          bf16x8_t fa[4], fb[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) fa[i] = lds_frag(wb + (2 * i) * 1024);
#pragma unroll
          for (int grp = 0; grp < 4; ++grp) {
            if (grp < 3) {
              const int gn = grp + 1;
#pragma unroll
              for (int i = 0; i < 4; ++i) fb[i] = lds_frag(wb + (2 * (4 * (gn & 1) + i) + (gn >> 1)) * 1024);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int kb = 4 * (grp & 1) + i;
              G[kb] = mfma32a(fa[i], (grp >> 1) ? d1 : d0, G[kb]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (grp < 2 && nxt) issue_part(g + 1, grp + 1);
            if (grp == 2) {
              // staged row blocks, issued AFTER the bundle (so that the bundle wait of the next step does not wait for HBM):
              // dx block cb at s = 4 cb (added at s = 4 cb + 2), xhat pieces 0 / 1 at s = 28 / 30 (used by the epilogue)
              if ((s & 3) == 0 && !(MLP_ABL & 256)) {
                issue_rows(reinterpret_cast<const char*>(p.dx) + (s >> 2) * 128, 1024, tl, (s >> 2) & 1);
                pend += 4;
              }
              if (s == 28 && !(MLP_ABL & 1024)) {
                issue_rows(reinterpret_cast<const char*>(p.xhat), 512, tl, 0);
                pend += 4;
              }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = fb[i];
          }
        }
        MPROF_MARK(4);
        if ((s & 3) == 2 && !(MLP_ABL & 64)) {  // dx block issued two steps ago (older than the bundle waited for above)
          const int cbx = s >> 2;
          float xv[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v = lds_f4(stg + (cbx & 1) * 4096 + stg_off(lr, 4 * hi + q));
            xv[4 * q + 0] = v.x * invr; xv[4 * q + 1] = v.y * invr; xv[4 * q + 2] = v.z * invr; xv[4 * q + 3] = v.w * invr;
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) sumold += xv[e];
#pragma unroll
          for (int kb = 0; kb < 8; ++kb)
            if (kb == cbx) {
#pragma unroll
              for (int e = 0; e < 16; ++e) G[kb][e] += xv[e];
            }
        }
        if (s == 30 && !(MLP_ABL & 1024)) {  // (after the last dx block left staging piece 1)
          issue_rows(reinterpret_cast<const char*>(p.xhat) + 128, 512, tl, 1);
          pend += 4;
        }
        MPROF_MARK(5);
        if (s == 31 && (MLP_ABL & 64)) {
#pragma unroll
          for (int kb = 0; kb < 8; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) G[kb][e] = 0.f;
        } else if (s == 31) {
          // ---- tile epilogue.  G = g + dx_old / rstd.  Row statistics, the xhat term (4 staged pieces, before any store),
          // then dx_new = rstd (G - mean g - xhat s2) out through the staging pieces.
          const int64_t r0 = tile_row0(tl);
          int prow = prow_, pchunk = pchunk_;
          asm volatile("" : "+v"(prow), "+v"(pchunk));
          float sg = 0.f;
#pragma unroll
          for (int kb = 0; kb < 8; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) sg += G[kb][e];
          sg -= sumold;
          sg += __shfl_xor(sg, 32, 64);
          const float s1 = sg * (1.0f / 256.0f);
          const float s2 = ((HMA_LDS(float)*)(lds + MB_ST2 + pair * 256))[lr] * (1.0f / 256.0f);
          // xhat pieces: piece cp holds columns 64 cp .. 64 cp + 63 (blocks kb = 2 cp, 2 cp + 1); 0 and 1 are in flight since
          // s = 28 / 30, 2 and 3 are issued as the buffers free up.  This wave's queue: [piece 1]? [bundle g + 1 (12)] ...
          wait_vm(nxt ? 12 : 0);  // pieces 0 and 1 (older than this step's bundle; the last step issues none)
#pragma unroll
          for (int cp = 0; cp < 4; ++cp) {
            if (cp == 2) wait_vm(4);   // piece 2 (piece 3 is younger)
            if (cp == 3) wait_vm(0);
            HMA_LDS(char)* sb = stg + (cp & 1) * 4096;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              float xf[16];
              unpack8(__builtin_bit_cast(uint4, lds_f4(sb + stg_off(lr, 4 * k + 2 * hi))), xf);
              unpack8(__builtin_bit_cast(uint4, lds_f4(sb + stg_off(lr, 4 * k + 2 * hi + 1))), xf + 8);
#pragma unroll
              for (int e = 0; e < 16; ++e) G[2 * cp + k][e] = __builtin_fmaf(-s2, xf[e], G[2 * cp + k][e]);
            }
            if (cp < 2 && !(MLP_ABL & 1024)) {
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this wave's reads of the piece are done before it is refilled)
              issue_rows(reinterpret_cast<const char*>(p.xhat) + (cp + 2) * 128, 512, tl, cp & 1);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          const float nb = -s1 * rstd;
#pragma unroll
          for (int cp = 0; cp < 4; ++cp) {  // blocks kb = 2 cp, 2 cp + 1: two fp32 pieces (dx), then their bf16 copy as one piece
            uint4 qb[4];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const int kb = 2 * cp + k;
              float o[16];
#pragma unroll
              for (int e = 0; e < 16; ++e) o[e] = __builtin_fmaf(G[kb][e], rstd, nb);
              HMA_LDS(char)* sb = stg + k * 4096;
#pragma unroll
              for (int q = 0; q < 4; ++q)
                lds_put(sb + stg_off(lr, 4 * hi + q), __builtin_bit_cast(uint4, make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3])));
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const int r = 8 * i + prow;
                const float4 v = lds_f4(sb + r * 128 + (pchunk << 4));
                if (r0 + r < p.M && !(MLP_ABL & 512)) *reinterpret_cast<float4*>(p.dx + (r0 + r) * 256 + kb * 32 + ((pchunk ^ ((r >> 1) & 7)) << 2)) = v;
              }
              qb[2 * k] = pack8(o);
              qb[2 * k + 1] = pack8(o + 8);
              __builtin_amdgcn_sched_barrier(0);
            }
            HMA_LDS(char)* sb = stg;  // (piece 0: its fp32 rows were read back above, LDS operations of a wave execute in order)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              lds_put(sb + stg_off(lr, 4 * k + 2 * hi), qb[2 * k]);
              lds_put(sb + stg_off(lr, 4 * k + 2 * hi + 1), qb[2 * k + 1]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int r = 8 * i + prow;
              const uint4 v = __builtin_bit_cast(uint4, lds_f4(sb + r * 128 + (pchunk << 4)));
              if (r0 + r < p.M && !(MLP_ABL & 512))
                *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.dx_bf16) + (r0 + r) * 256 + cp * 64 + ((pchunk ^ ((r >> 1) & 7)) << 3)) = v;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int kb = 0; kb < 8; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) G[kb][e] = 0.f;
          pend = 63;  // every load of this wave is complete (vmcnt(0) above); the stores need no wait
        }
        MPROF_MARK(6);
      }
    }
    MPROF_FLUSH(1);
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

int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}

}  // namespace

#ifdef MLP_PROF
extern "C" int hma_mlp_debug_prof(unsigned long long* out128) {
  if (hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_mlp_prof), sizeof(unsigned long long) * 128) != hipSuccess) return -1;
  return 0;
}
#endif

extern "C" int hma_mlp_pack(void* stream, const float* src, int64_t row_stride, int64_t col_stride, const float* row_scale,
                            const float* col_scale, void* dst, int32_t kind, int32_t batch, int64_t src_batch_stride,
                            int64_t dst_batch_stride) {
  if (!src || !dst || (kind != 0 && kind != 1) || batch < 1) return HMA_EINVAL;
  hipLaunchKernelGGL(mlp_pack_kernel, dim3(128, batch), dim3(256), 0, (hipStream_t)stream, src, row_stride, col_stride,
                     row_scale, col_scale, reinterpret_cast<uint16_t*>(dst), (int)kind, src_batch_stride, dst_batch_stride);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_mlp_pack_multi(void* stream, const hma_pack_job_t* jobs, int32_t njobs) {
  if (njobs < 0 || (njobs > 0 && !jobs)) return HMA_EINVAL;
  for (int i = 0; i < njobs; ++i)
    if (!jobs[i].src || !jobs[i].dst || (jobs[i].kind != 0 && jobs[i].kind != 1) || jobs[i].batch < 1) return HMA_EINVAL;
  for (int i0 = 0; i0 < njobs; i0 += PACK_JOBS) {
    pack_jobs pj;
    pj.n = njobs - i0 < PACK_JOBS ? njobs - i0 : PACK_JOBS;
    int gy = 0;
    for (int i = 0; i < pj.n; ++i) {
      pj.j[i] = jobs[i0 + i];
      pj.y0[i] = gy;
      gy += pj.j[i].batch;
    }
    pj.y0[pj.n] = gy;
    hipLaunchKernelGGL(mlp_pack_multi_kernel, dim3(128, gy), dim3(256), 0, (hipStream_t)stream, pj);
    HMA_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int hma_mlp_fwd(void* stream, const hma_mlp_fwd_t* p) {
  if (!p || !p->xhat || !p->x || !p->w1p || !p->w2p || !p->b1 || p->M <= 0) return HMA_EINVAL;
  if (p->ln_xhat && !p->ln_rstd) return HMA_EINVAL;
  const int64_t ntiles = (p->M + 127) >> 7;
  const int grid = (int)(ntiles < num_cus() ? ntiles : num_cus());
  if (p->ln_xhat) {
    if (int rc = set_lds<mlp_fwd_kernel<true>>(MF_SMEM)) return rc;
    hipLaunchKernelGGL(mlp_fwd_kernel<true>, dim3(grid), dim3(512), MF_SMEM, (hipStream_t)stream, *p);
  } else {
    if (int rc = set_lds<mlp_fwd_kernel<false>>(MF_SMEM)) return rc;
    hipLaunchKernelGGL(mlp_fwd_kernel<false>, dim3(grid), dim3(512), MF_SMEM, (hipStream_t)stream, *p);
  }
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_mlp_bwd(void* stream, const hma_mlp_bwd_t* p) {
  if (!p || !p->xhat || !p->rstd || !p->dy || !p->dx || !p->dx_bf16 || !p->w1p || !p->w2tp || !p->w1tp || !p->b1 || !p->hg ||
      !p->du || p->M <= 0 || p->dy == p->dx_bf16)
    return HMA_EINVAL;
  const bool drop = p->drop_p != 0.f;
  if (drop && (!(p->drop_p > 0.f && p->drop_p < 1.f) || !p->drop_seed || !p->dy_drop || p->dy_drop == p->dx_bf16)) return HMA_EINVAL;
  const int64_t ntiles = (p->M + 127) >> 7;
  const int grid = (int)(ntiles < num_cus() ? ntiles : num_cus());
  if (drop) {
    if (int rc = set_lds<mlp_bwd_kernel<true>>(MB_SMEM)) return rc;
    hipLaunchKernelGGL(mlp_bwd_kernel<true>, dim3(grid), dim3(512), MB_SMEM, (hipStream_t)stream, *p);
    HMA_CHECK_LAUNCH();
    return 0;
  }
  if (int rc = set_lds<mlp_bwd_kernel<false>>(MB_SMEM)) return rc;
  hipLaunchKernelGGL(mlp_bwd_kernel<false>, dim3(grid), dim3(512), MB_SMEM, (hipStream_t)stream, *p);
  HMA_CHECK_LAUNCH();
  return 0;
}
