// Fused temporal-attention block of the ST-transformer for gfx950 (forward):
//     x += proj(causal_attn_T(qkv(x)))   then   xhat2 = LayerNorm(x)          (the MLP's norm2, no affine here)
// Reference: SelfAttention.forward (hma/model/attention.py:37-61, causal = True) as called from STBlock.forward on the
// "(B S) T C" view of the token grid (hma/model/st_transformer.py:111; no LayerNorm in front of it), followed by norm2
// (:112).  Replaces three launches (qkv GEMM, hma_attn_temporal_fwd, proj GEMM + residual + LayerNorm epilogue): the
// qkv rows and the attention output are still WRITTEN once (the backward kernels read them) but never re-read here.
//
// Structure (512 threads = 4 producer / consumer wave pairs, one workgroup per CU, same skeleton as csrc/mlp.hip):
//   * a workgroup tile is 8 token columns (b, s0 .. s0 + 7) x 16 frames = 128 token rows; a pair owns 2 columns = 32
//     rows, token n = 16 c + t of the pair is global row (b T + t) (S + A) + s0 + 2 pair + c.
//   * all MFMAs are v_mfma_f32_32x32x16_bf16.  Projections run "swapped" (weights = A operand with the rows of a 32-row
//     block permuted by rowmap, token rows = B operand from registers): a lane then owns ONE token and ends with 16
//     consecutive output features in its accumulator registers.
//   * one step = one head.  Producer: q, k, v of the head for its 32 tokens (48 MFMAs), V^T by the same fragments with
//     the operands exchanged (16 MFMAs: lanes = head channels, registers = tokens), S^T = K Q^T of the 32 x 32 token
//     tile (2 MFMAs; the two 16 x 16 causal blocks on the diagonal are the two columns, the rest is masked), softmax in
//     registers (8 scores per lane + one cross-half shuffle), O^T = V^T P^T (2 MFMAs).  Everything chains through
//     accumulator registers: packed to bf16, what one product leaves in a lane is exactly the next product's operand.
//     Consumer: out += Wproj[:, head] o_head (16 MFMAs) into 32 x 256 accumulators, the residual rows added as they
//     arrive (staged by LDS-DMA), and at the tile's end bias + LayerNorm statistics + whole-cache-line stores.
//   * the producers' weight stream (q | k | v fragments of a head, 48 KB) goes through a 2-slot LDS ring filled by the
//     CONSUMER waves (they have the slack: 16 of a step's 84 MFMAs); the consumers take their own 16 projection fragments
//     of the next head straight from L2 into registers a step ahead.
#include "hma_common.h"
#include "../../include/hma_hip.h"

using namespace hma;

namespace {

__host__ __device__ constexpr int rowmap(int rho) { return (rho & 3) + 4 * (rho >> 3) + 16 * ((rho >> 2) & 1); }

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
__device__ __forceinline__ bf16x8_t as_frag(const uint4& v) { return __builtin_bit_cast(bf16x8_t, v); }
__device__ __forceinline__ bf16x8_t lds_frag(HMA_LDS(char)* p) { return __builtin_bit_cast(bf16x8_t, *(HMA_LDS(u32x4_t)*)p); }
__device__ __forceinline__ void lds_put(HMA_LDS(char)* p, const uint4& v) { *(HMA_LDS(u32x4_t)*)p = __builtin_bit_cast(u32x4_t, v); }
__device__ __forceinline__ float4 lds_f4(HMA_LDS(char)* p) {
  const f32x4_t v = *(HMA_LDS(f32x4_t)*)p;
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint4 pack8r(const f32x16_t& a, int o) {
  uint4 v;
  v.x = pack_bf16(a[o + 0], a[o + 1]);
  v.y = pack_bf16(a[o + 2], a[o + 3]);
  v.z = pack_bf16(a[o + 4], a[o + 5]);
  v.w = pack_bf16(a[o + 6], a[o + 7]);
  return v;
}

// ------------------------------------------------------------------------------------------------ weight packing
// Fragment = 1 KB, lane-linear: lane (rho, hi), element i.
// kind 0 (qkv, logical W[768][256]): fragment f = (h * 3 + w) * 16 + j  (head h, w = q / k / v, k-step j) holds
//        W[256 w + 32 h + rowmap(rho)][32 (j >> 1) + 16 hi + 8 (j & 1) + i]                       -> 384 fragments
// kind 1 (proj, logical W[256][256]): fragment f = (h * 8 + mb) * 2 + j holds
//        W[32 mb + rowmap(rho)][32 h + 16 hi + 8 j + i]                                            -> 128 fragments
__global__ __launch_bounds__(256) void tblock_pack_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int kind,
                                                          int64_t sstride, int64_t dstride) {
  src += (int64_t)blockIdx.y * sstride;
  dst += (int64_t)blockIdx.y * dstride;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int lane = idx & 63, frag = idx >> 6, rho = lane & 31, hi = lane >> 5;
  int row, col0;
  if (kind == 0) {
    const int j = frag & 15, hw = frag >> 4, h = hw / 3, w = hw % 3;
    row = 256 * w + 32 * h + rowmap(rho);
    col0 = 32 * (j >> 1) + 16 * hi + 8 * (j & 1);
  } else {
    const int j = frag & 1, mb = (frag >> 1) & 7, h = frag >> 4;
    row = 32 * mb + rowmap(rho);
    col0 = 32 * h + 16 * hi + 8 * j;
  }
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = src[(int64_t)row * 256 + col0 + i];
  *reinterpret_cast<uint4*>(dst + (int64_t)idx * 8) = pack8(v);
}

// ------------------------------------------------------------------------------------------------ forward
// Debug builds only (tools/tblock_variants.sh, -DTB_ABL=bits): 1 no q / k / v / o stores, 2 no ring LDS-DMA, 4 no residual staging, 32 no
// epilogue stores, 64 producer stores to an L2-resident scratch, 8 projection fragments not loaded, 16 no q / k / v MFMA loop
#ifndef TB_ABL
#define TB_ABL 0
#endif
// Debug builds only (-DTB_PROF): per-wave cycle counts per phase, block 0 (tools/tblock_bench.py prints them)
#ifdef TB_PROF
__device__ unsigned long long g_tb_prof[8][8];
#define TPROF_DECL unsigned long long pt_ = __builtin_readcyclecounter(), pacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define TPROF_MARK(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); pacc_[i] += n_ - pt_; pt_ = n_; } while (0)
#define TPROF_FLUSH() do { if (blockIdx.x == 0 && lane == 0) { for (int i_ = 0; i_ < 8; ++i_) g_tb_prof[wave][i_] = pacc_[i_]; } } while (0)
#else
#define TPROF_DECL
#define TPROF_MARK(i)
#define TPROF_FLUSH()
#endif
constexpr int TB_SLOT = 49152;                  // one head's q | k | v fragments
constexpr int TB_OUT = 2 * TB_SLOT;             // per pair: hand-over tiles q | k | v | o of the head (4 x 2 KB)
constexpr int TB_OUT_PAIR = 8192;
constexpr int TB_STG = TB_OUT + 4 * TB_OUT_PAIR; // per pair: ONE staging piece of 4 KB (32 rows x 128 B)
constexpr int TB_BQKV = TB_STG + 4 * 4096;      // 768 floats
constexpr int TB_BPROJ = TB_BQKV + 3072;        // 256 floats
constexpr int TB_FLAG = TB_BPROJ + 1024;        // per pair: the step whose hand-over tiles the consumer has drained
constexpr int TB_SMEM = TB_FLAG + 64;           // 151616 B

// Row-major 2 KB tile of a pair (q, k, v or o of one head): row n (token of the pair) holds the head's 32 channels as four
// 16-byte chunks, chunk c = 2 hi + e (channels 8 c .. 8 c + 7) at position c ^ ((n >> 1) & 3): writes in the accumulator
// shape ("a lane owns a token") and row-major reads (4 lanes per row = 64 contiguous bytes of a global row) are both
// conflict-free.  The producer leaves q, k, v, o of a head in these tiles and never touches global memory inside its loop: a
// store instruction of the compute-critical wave stalls whenever HBM is busy (measured: +60 ... +108 us per launch).  The
// consumer, which has the slack, reads the tiles row-major and stores whole 64-byte row pieces (16 rows per instruction) at
// the START of its step, so that they drain while the producers compute; a flag tells the producer the tiles are free again.
__device__ __forceinline__ int out_off(int n, int c) { return n * 64 + ((c ^ ((n >> 1) & 3)) << 4); }

__device__ __forceinline__ int stg_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

__global__ __launch_bounds__(512, 2) void tblock_fwd_kernel(hma_tblock_fwd_t p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  HMA_LDS(char)* lds = (HMA_LDS(char)*)smem;
  const uint32_t lds_b = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef TB_FORCE_ROLE
  const int role = TB_FORCE_ROLE, pair = wave & 3;
#else
  const int role = wave >> 2, pair = wave & 3;  // waves w and w + 4 share a SIMD
#endif
  const int lr = lane & 31, hi = lane >> 5;
  const int SA = p.SA, nsb = SA >> 3;           // 8-column groups per sample
  const int64_t ntiles = p.B * nsb;
  const int nt = (int)((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x);
  const int nsteps = nt * 8;

  {
    HMA_LDS(float)* bq = (HMA_LDS(float)*)(lds + TB_BQKV);
    for (int i = tid; i < 768; i += 512) bq[i] = p.bqkv ? p.bqkv[i] : 0.f;
    if (tid < 256) ((HMA_LDS(float)*)(lds + TB_BPROJ))[tid] = p.bproj ? p.bproj[tid] : 0.f;
    if (tid < 16) ((HMA_LDS(int)*)(lds + TB_FLAG))[tid] = 0;
  }
  __syncthreads();

  // global row of token r (0..31) of this pair in the wave's tl-th tile = tbase(tl) + (r & 15) SA + (r >> 4)
  auto tbase = [&](int tl) __attribute__((always_inline)) -> int64_t {
    const int64_t tile = (int64_t)blockIdx.x + (int64_t)tl * gridDim.x;
    const int64_t b = tile / nsb;
    const int sb = (int)(tile - b * nsb);
    return b * 16 * SA + sb * 8 + 2 * pair;
  };
  auto grow = [&](int64_t base, int r) __attribute__((always_inline)) -> int64_t { return base + (int64_t)(r & 15) * SA + (r >> 4); };

  if (role == 0) {
    // ================================================================ producer: q, k, v, attention of one head per step
    // The 32 token rows of the pair as B operands (16 k-steps), loaded by ONE asm block that also waits for them.  As
    // compiler-visible loads they put an `s_waitcnt vmcnt(0)` in front of the first MFMA of EVERY step (the pass cannot tell
    // which iteration issued them), and that wait also covers the previous step's q / k / v / o stores: every byte stored was
    // paid for at HBM speed with the matrix pipe idle (+60 us per launch).  Now the producer's loop has no vmcnt wait at all.
    u32x4_t xr[16];
    auto load_x = [&](int64_t base) __attribute__((always_inline)) {
      const uint16_t* src = reinterpret_cast<const uint16_t*>(p.xb) + grow(base, lr) * 256 + 16 * hi;
      asm volatile(
          "global_load_dwordx4 %0, %16, off\n\t"
          "global_load_dwordx4 %1, %16, off offset:16\n\t"
          "global_load_dwordx4 %2, %16, off offset:64\n\t"
          "global_load_dwordx4 %3, %16, off offset:80\n\t"
          "global_load_dwordx4 %4, %16, off offset:128\n\t"
          "global_load_dwordx4 %5, %16, off offset:144\n\t"
          "global_load_dwordx4 %6, %16, off offset:192\n\t"
          "global_load_dwordx4 %7, %16, off offset:208\n\t"
          "global_load_dwordx4 %8, %16, off offset:256\n\t"
          "global_load_dwordx4 %9, %16, off offset:272\n\t"
          "global_load_dwordx4 %10, %16, off offset:320\n\t"
          "global_load_dwordx4 %11, %16, off offset:336\n\t"
          "global_load_dwordx4 %12, %16, off offset:384\n\t"
          "global_load_dwordx4 %13, %16, off offset:400\n\t"
          "global_load_dwordx4 %14, %16, off offset:448\n\t"
          "global_load_dwordx4 %15, %16, off offset:464\n\t"
          "s_waitcnt vmcnt(0)"
          : "=&v"(xr[0]), "=&v"(xr[1]), "=&v"(xr[2]), "=&v"(xr[3]), "=&v"(xr[4]), "=&v"(xr[5]), "=&v"(xr[6]), "=&v"(xr[7]),
            "=&v"(xr[8]), "=&v"(xr[9]), "=&v"(xr[10]), "=&v"(xr[11]), "=&v"(xr[12]), "=&v"(xr[13]), "=&v"(xr[14]), "=&v"(xr[15])
          : "v"(src)
          : "memory");
    };
#define xh(j) __builtin_bit_cast(bf16x8_t, xr[j])
    int64_t base = tbase(0);
    load_x(base);
    const int c = lr >> 4, t = lr & 15;
    const float alpha = p.scale * 1.4426950408889634f;  // scores in the log2 domain
    TPROF_DECL;
    for (int g = 0; g <= nsteps; ++g) {
      TPROF_MARK(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      TPROF_MARK(1);
      if (g == nsteps) break;
      const int h = g & 7, tl = g >> 3;
      if (h == 0 && tl > 0) {  // next tile's rows (once per tile; also drains the previous tile's stores)
        base = tbase(tl);
        load_x(base);
      }
      TPROF_MARK(2);
      HMA_LDS(char)* wb = lds + (g & 1) * TB_SLOT + lane * 16;
      HMA_LDS(char)* bp = lds + TB_BQKV + (32 * h + 16 * hi) * 4;
      f32x16_t Q, K, V, VT;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 a = lds_f4(bp + 16 * q4), b = lds_f4(bp + 1024 + 16 * q4), cc = lds_f4(bp + 2048 + 16 * q4);
        Q[4 * q4] = a.x; Q[4 * q4 + 1] = a.y; Q[4 * q4 + 2] = a.z; Q[4 * q4 + 3] = a.w;
        K[4 * q4] = b.x; K[4 * q4 + 1] = b.y; K[4 * q4 + 2] = b.z; K[4 * q4 + 3] = b.w;
        V[4 * q4] = cc.x; V[4 * q4 + 1] = cc.y; V[4 * q4 + 2] = cc.z; V[4 * q4 + 3] = cc.w;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) VT[e] = 0.f;
      {
        // per k-step j: the q, k, v fragments of the head (fragment (w, j) at (16 w + j) KB of the slot)
        bf16x8_t fa[3], fb[3], fc[3];  // k-steps j, j + 1, j + 2: an LDS read has two MFMA groups (256 cycles) to arrive
#pragma unroll
        for (int w = 0; w < 3; ++w) fa[w] = lds_frag(wb + (16 * w) * 1024);
#pragma unroll
        for (int w = 0; w < 3; ++w) fb[w] = lds_frag(wb + (16 * w + 1) * 1024);
#pragma unroll
        for (int j = 0; j < ((TB_ABL & 16) ? 1 : 16); ++j) {
          if (j < 14) {
#pragma unroll
            for (int w = 0; w < 3; ++w) fc[w] = lds_frag(wb + (16 * w + j + 2) * 1024);
          }
          Q = mfma32(fa[0], xh(j), Q);
          K = mfma32(fa[1], xh(j), K);
          V = mfma32(fa[2], xh(j), V);
          VT = mfma32(xh(j), fa[2], VT);  // operands exchanged: rows = tokens, columns = the fragment's (permuted) channels
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int w = 0; w < 3; ++w) fa[w] = fb[w], fb[w] = fc[w];
        }
      }
      // V^T came out without the bias (its lanes are channels): lane rho holds channel rowmap(rho) of the head
      {
        const float bv = ((HMA_LDS(float)*)(lds + TB_BQKV))[512 + 32 * h + rowmap(lr)];
#pragma unroll
        for (int e = 0; e < 16; ++e) VT[e] += bv;
      }
      TPROF_MARK(3);
      HMA_LDS(char)* oa = lds + TB_OUT + pair * TB_OUT_PAIR;
      {  // tiles free?  (the consumer drains them at the very start of its step; this wave has just spent 64 MFMAs)
        volatile HMA_LDS(int)* flag = (volatile HMA_LDS(int)*)(lds + TB_FLAG + pair * 16);
        while (*flag < g) __builtin_amdgcn_s_sleep(1);
      }
      TPROF_MARK(4);
      {  // q | k | v rows of the head (for the backward kernels)
        lds_put(oa + out_off(lr, 2 * hi), pack8r(Q, 0));
        lds_put(oa + out_off(lr, 2 * hi + 1), pack8r(Q, 8));
        lds_put(oa + 2048 + out_off(lr, 2 * hi), pack8r(K, 0));
        lds_put(oa + 2048 + out_off(lr, 2 * hi + 1), pack8r(K, 8));
        lds_put(oa + 4096 + out_off(lr, 2 * hi), pack8r(V, 0));
        lds_put(oa + 4096 + out_off(lr, 2 * hi + 1), pack8r(V, 8));
      }
      // S^T[key][query] over the pair's 32 tokens; contraction index of k-step e: channels 16 kg + 8 e + i
      f32x16_t ST;
#pragma unroll
      for (int e = 0; e < 16; ++e) ST[e] = 0.f;
      ST = mfma32(as_frag(pack8r(K, 0)), as_frag(pack8r(Q, 0)), ST);
      ST = mfma32(as_frag(pack8r(K, 8)), as_frag(pack8r(Q, 8)), ST);
      // this lane: query token (c, t); its column's keys sit in registers 8 c .. 8 c + 7, key t' = (i & 3) + 8 (i >> 2) + 4 hi
      float s[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v = c ? ST[8 + i] : ST[i];
        const int tk = (i & 3) + 8 * (i >> 2) + 4 * hi;
        s[i] = tk <= t ? v * alpha : -INFINITY;
      }
      float m = s[0];
#pragma unroll
      for (int i = 1; i < 8; ++i) m = fmaxf(m, s[i]);
      m = fmaxf(m, __shfl_xor(m, 32, 64));   // (key 0 is never masked: m is finite)
      float l = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        s[i] = __builtin_amdgcn_exp2f(s[i] - m);
        l += s[i];
      }
      l += __shfl_xor(l, 32, 64);
      const float inv = 1.0f / l;
      float pr[8], z[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) pr[i] = s[i] * inv, z[i] = 0.f;
      // P^T as the B operand of O^T = V^T P^T: k-step e = key tokens of column e (registers 8 e .. 8 e + 7)
      const bf16x8_t pb0 = as_frag(c ? pack8(z) : pack8(pr));
      const bf16x8_t pb1 = as_frag(c ? pack8(pr) : pack8(z));
      f32x16_t OT;
#pragma unroll
      for (int e = 0; e < 16; ++e) OT[e] = 0.f;
      OT = mfma32(as_frag(pack8r(VT, 0)), pb0, OT);
      OT = mfma32(as_frag(pack8r(VT, 8)), pb1, OT);
      // o of the head: lane (token, hi) holds channels 16 hi + r -> the consumer's B operand, and the saved o rows
      lds_put(oa + 6144 + out_off(lr, 2 * hi), pack8r(OT, 0));
      lds_put(oa + 6144 + out_off(lr, 2 * hi + 1), pack8r(OT, 8));
      TPROF_MARK(5);
    }
    TPROF_FLUSH();
  } else {
    // ================================================================ consumer: x += Wproj o + b (+ LayerNorm of the new row)
    f32x16_t Y[8];
    bf16x8_t wp[8];   // projection fragments straight from L2, half a head at a time (output blocks 0..3 are loaded a step ahead)
    HMA_LDS(char)* stg = lds + TB_STG + pair * 4096;
    const uint32_t stg_b = lds_b + TB_STG + pair * 4096;
    HMA_LDS(char)* oa = lds + TB_OUT + pair * TB_OUT_PAIR;
    const int prow_ = lane >> 3, pchunk_ = lane & 7;
    const char* wq = reinterpret_cast<const char*>(p.wqkvp) + pair * 12288 + lane * 16;
    auto issue = [&](int b) __attribute__((always_inline)) {  // this wave's 12 of the 48 pieces of bundle b
      const uint32_t base = lds_b + (b & 1) * TB_SLOT + pair * 12288;
      const char* src = wq + (int64_t)(b & 7) * TB_SLOT;
      if (TB_ABL & 2) return;
      glds16x4(src, base);
      glds16x4(src + 4096, base + 4096);
      glds16x4(src + 8192, base + 8192);
    };
    // residual block cb (32 fp32 columns) of this pair's rows of tile tl -> staging piece cb & 1
    // per-lane ELEMENT offsets (relative to the tile's first row) of the four rows this lane touches in the "8 lanes per row"
    // shape, with the chunk swizzle folded in: constant for the whole kernel
    uint32_t xoff[4], hoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 8 * i + prow_;
      const uint32_t ro = (uint32_t)((r & 15) * SA + (r >> 4)) * 256u;
      xoff[i] = ro + ((pchunk_ ^ ((r >> 1) & 7)) << 2);   // fp32 piece: 32 columns, 4 per chunk
      hoff[i] = ro + ((pchunk_ ^ ((r >> 1) & 7)) << 3);   // bf16 piece: 64 columns, 8 per chunk
    }
    // row-major reads of a 2 KB tile: lane -> row 16 part + (lane >> 2), chunk position lane & 3 (logical chunk = position ^ swizzle)
    uint32_t qoff[2], ooff[2];
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      const int n = 16 * part + (lane >> 2);
      const uint32_t ro = (uint32_t)((n & 15) * SA + (n >> 4));
      const uint32_t ch = (uint32_t)(((lane & 3) ^ ((n >> 1) & 3)) << 3);
      qoff[part] = ro * 768u + ch;
      ooff[part] = ro * 256u + ch;
    }
    int64_t base_h = 0;  // first row of the tile whose heads are being drained
    auto issue_x = [&](int64_t base, int cb) __attribute__((always_inline)) {
      const float* sb = p.x + base * 256 + cb * 32;
      if (TB_ABL & 4) return;
#pragma unroll
      for (int i = 0; i < 4; ++i) glds16s(sb, xoff[i] * 4u, stg_b + i * 1024);
    };
    auto load_w = [&](int h, int half) __attribute__((always_inline)) {
      const char* src = reinterpret_cast<const char*>(p.wprojp) + (int64_t)h * 16384 + half * 8192 + lane * 16;
#pragma unroll
      for (int f = 0; f < 8; ++f)
        wp[f] = (TB_ABL & 8) ? as_frag(make_uint4(f, lane, h, half)) : as_frag(*reinterpret_cast<const uint4*>(src + f * 1024));
    };
    int64_t base = tbase(0);
    TPROF_DECL;
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int g = 0; g <= nsteps; ++g) {
      // No vmcnt wait here: vmcnt also counts stores, loads and stores complete out of order with each other (only vmcnt(0)
      // is safe), and a wait right behind this wave's stores would sit out their acknowledgement at HBM pace -- measured as
      // 11 k cycles per step against 4.6 k without the stores.  So a step issues ALL its loads first, waits once in mid-step
      // (by then the previous step's stores have had half a step to drain), computes, and stores LAST.
      TPROF_MARK(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      TPROF_MARK(1);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      TPROF_MARK(2);
      if (g + 1 < nsteps) issue(g + 1);   // the ring bundle of the next step: landed by the mid-step wait, a barrier before its use
      if (g == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (g >= 1) {
        const int gc = g - 1, h = gc & 7, tl = gc >> 3;
        if (h == 0) {
          base_h = base;
#pragma unroll
          for (int cb = 0; cb < 8; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) Y[cb][e] = 0.f;
        }
        issue_x(base_h, h);               // residual block h of this tile (the piece was read in the previous step)
        load_w(h, 0);                     // projection fragments, output blocks 0..3
        // the head's o as this wave's B operand and its q | k | v | o rows into registers: the producer gets its tiles back now
        const bf16x8_t o0 = lds_frag(oa + 6144 + out_off(lr, 2 * hi)), o1 = lds_frag(oa + 6144 + out_off(lr, 2 * hi + 1));
        uint4 rv[8];
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int part = 0; part < 2; ++part)
            rv[2 * w + part] = __builtin_bit_cast(uint4, lds_f4(oa + w * 2048 + (16 * part + (lane >> 2)) * 64 + ((lane & 3) << 4)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) *(volatile HMA_LDS(int)*)(lds + TB_FLAG + pair * 16) = g;   // drained: the producer may write step g's tiles
        // every load of this step (ring bundle, residual block, fragments) and the previous step's stores.  The builtin, not
        // asm: the compiler must know that its own fragment loads are complete.
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          Y[mb] = mfma32(wp[2 * mb], o0, Y[mb]);
          Y[mb] = mfma32(wp[2 * mb + 1], o1, Y[mb]);
        }
        __builtin_amdgcn_sched_barrier(0);
        load_w(h, 1);                       // output blocks 4..7 (an L2 round trip; nothing else of this wave is in flight)
        __builtin_amdgcn_sched_barrier(0);
        {  // residual block h while the fragments arrive
          float xv[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v = lds_f4(stg + stg_off(lr, 4 * hi + q));
            xv[4 * q + 0] = v.x; xv[4 * q + 1] = v.y; xv[4 * q + 2] = v.z; xv[4 * q + 3] = v.w;
          }
#pragma unroll
          for (int cb = 0; cb < 8; ++cb)
            if (cb == h) {
#pragma unroll
              for (int e = 0; e < 16; ++e) Y[cb][e] += xv[e];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          Y[4 + mb] = mfma32(wp[2 * mb], o0, Y[4 + mb]);
          Y[4 + mb] = mfma32(wp[2 * mb + 1], o1, Y[4 + mb]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(TB_ABL & 1)) {  // the head's q | k | v | o rows: 16 rows x 64 B per store instruction, last thing in the step
#ifdef TB_STORE_LINEAR
          // timing experiment: the same bytes as 1 KB of contiguous memory per store instruction (a fragment-order layout; the
          // results are not readable by the row-major consumers)
          const int64_t tix = (base_h / ((int64_t)p.T * SA)) * (SA >> 3) + (base_h % ((int64_t)p.T * SA)) / 8;
          char* ql = reinterpret_cast<char*>(p.qkv) + ((tix * 8 + h) * 4 + pair) * 6144 + lane * 16;
          char* ol = reinterpret_cast<char*>(p.o) + ((tix * 8 + h) * 4 + pair) * 2048 + lane * 16;
#pragma unroll
          for (int k = 0; k < 6; ++k) *reinterpret_cast<uint4*>(ql + k * 1024) = rv[k];
          *reinterpret_cast<uint4*>(ol) = rv[6];
          *reinterpret_cast<uint4*>(ol + 1024) = rv[7];
#else
          uint16_t* qt = reinterpret_cast<uint16_t*>(p.qkv) + base_h * 768 + 32 * h;
          uint16_t* ot = reinterpret_cast<uint16_t*>(p.o) + base_h * 256 + 32 * h;
#pragma unroll
          for (int part = 0; part < 2; ++part) {
            *reinterpret_cast<uint4*>(qt + qoff[part]) = rv[part];
            *reinterpret_cast<uint4*>(qt + 256 + qoff[part]) = rv[2 + part];
            *reinterpret_cast<uint4*>(qt + 512 + qoff[part]) = rv[4 + part];
            *reinterpret_cast<uint4*>(ot + ooff[part]) = rv[6 + part];
          }
#endif
        }
        TPROF_MARK(3);
        if (h == 7) {
          // ---- tile epilogue: + bias, LayerNorm statistics, rows out through the staging pieces (whole cache lines)
          HMA_LDS(char)* b2p = lds + TB_BPROJ + 16 * hi * 4;
          int prow = prow_, pchunk = pchunk_;
          asm volatile("" : "+v"(prow), "+v"(pchunk));
          float* xt = p.x + base * 256;                                                  // (wave-uniform)
          uint16_t* ht = reinterpret_cast<uint16_t*>(p.ln_xhat) + base * 256;
          float sum = 0.f, sq = 0.f, shift = 0.f;
#pragma unroll
          for (int cb = 0; cb < 8; ++cb) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 b = lds_f4(b2p + (8 * cb + q) * 16);
              Y[cb][4 * q + 0] += b.x; Y[cb][4 * q + 1] += b.y; Y[cb][4 * q + 2] += b.z; Y[cb][4 * q + 3] += b.w;
            }
            if (cb == 0) shift = __shfl(Y[0][0], lr, 64);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const float d = Y[cb][e] - shift;
              sum += d;
              sq = __builtin_fmaf(d, d, sq);
            }
            HMA_LDS(char)* sb = stg;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              lds_put(sb + stg_off(lr, 4 * hi + q), __builtin_bit_cast(uint4, make_float4(Y[cb][4 * q], Y[cb][4 * q + 1], Y[cb][4 * q + 2], Y[cb][4 * q + 3])));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int r = 8 * i + prow;
              const float4 v = lds_f4(sb + r * 128 + (pchunk << 4));
              if (!(TB_ABL & 32)) *reinterpret_cast<float4*>(xt + cb * 32 + xoff[i]) = v;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          if (p.ln_xhat) {
            sum += __shfl_xor(sum, 32, 64);
            sq += __shfl_xor(sq, 32, 64);
            const float md = sum * (1.0f / 256.0f);
            const float var = fmaxf(sq * (1.0f / 256.0f) - md * md, 0.f);
            const float rstd = rsqrtf(var + p.ln_eps);
            const float nb = -(md + shift) * rstd;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cp = 0; cp < 4; ++cp) {
              HMA_LDS(char)* sb = stg;
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
                if (!(TB_ABL & 32)) *reinterpret_cast<uint4*>(ht + cp * 64 + hoff[i]) = v;
              }
              __builtin_amdgcn_sched_barrier(0);
            }
            if (hi == 0) p.ln_rstd[grow(base, lr)] = rstd;
          }
          if (tl + 1 < nt) base = tbase(tl + 1);
          TPROF_MARK(4);
        }
      }
    }
    TPROF_FLUSH();
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

#ifdef TB_PROF
extern "C" int hma_tblock_debug_prof(unsigned long long* out64) {
  if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_tb_prof), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
  return 0;
}
#endif

extern "C" int hma_tblock_pack(void* stream, const float* src, void* dst, int32_t kind, int32_t batch, int64_t src_batch_stride,
                               int64_t dst_batch_stride) {
  if (!src || !dst || (kind != 0 && kind != 1) || batch < 1) return HMA_EINVAL;
  const int frags = kind == 0 ? 384 : 128;
  hipLaunchKernelGGL(tblock_pack_kernel, dim3(frags / 4, batch), dim3(256), 0, (hipStream_t)stream, src, reinterpret_cast<uint16_t*>(dst),
                     (int)kind, src_batch_stride, dst_batch_stride);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_tblock_fwd(void* stream, const hma_tblock_fwd_t* p) {
  if (!p || !p->xb || !p->x || !p->wqkvp || !p->wprojp || !p->qkv || !p->o || p->B <= 0) return HMA_EINVAL;
  if (p->T != 16 || p->SA <= 0 || (p->SA & 7)) return HMA_EINVAL;   // 16 frames, 8-column tiles
  if (p->ln_xhat && !p->ln_rstd) return HMA_EINVAL;
  const int64_t ntiles = p->B * (p->SA >> 3);
  const int grid = (int)(ntiles < num_cus() ? ntiles : num_cus());
  if (int rc = set_lds<tblock_fwd_kernel>(TB_SMEM)) return rc;
  hipLaunchKernelGGL(tblock_fwd_kernel, dim3(grid), dim3(512), TB_SMEM, (hipStream_t)stream, *p);
  HMA_CHECK_LAUNCH();
  return 0;
}
