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
__device__ __forceinline__ bf16x8_t lds_frag(HMA_LDS(char)* p) { return __builtin_bit_cast(bf16x8_t, *(HMA_LDS(u32x4_t)*)p); }
__device__ __forceinline__ void lds_put(HMA_LDS(char)* p, const uint4& v) { *(HMA_LDS(u32x4_t)*)p = __builtin_bit_cast(u32x4_t, v); }
__device__ __forceinline__ float4 lds_f4(HMA_LDS(char)* p) {
  const f32x4_t v = *(HMA_LDS(f32x4_t)*)p;
  return make_float4(v[0], v[1], v[2], v[3]);
}

// exact-erf GELU pieces (A&S 7.1.26, as hma_common.h gelu_parts): h = Phi(-|u|), gauss = exp(-u^2 / 2)
__device__ __forceinline__ void gelu_tail(float u, float& h, float& gauss) {
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(fabsf(u), 0.3275911f * 0.70710678118654752f, 1.0f));
  gauss = __builtin_amdgcn_exp2f(-0.72134752044448170f * u * u);
  float poly = 0.5f * 1.061405429f;
  poly = __builtin_fmaf(poly, t, -0.5f * 1.453152027f);
  poly = __builtin_fmaf(poly, t, 0.5f * 1.421413741f);
  poly = __builtin_fmaf(poly, t, -0.5f * 0.284496736f);
  poly = __builtin_fmaf(poly, t, 0.5f * 0.254829592f);
  h = poly * t * gauss;
}
// gelu(u) = max(u, 0) - |u| Phi(-|u|)
__device__ __forceinline__ float gelu_fused(float u) {
  float h, g;
  gelu_tail(u, h, g);
  return fmaxf(u, 0.f) - fabsf(u) * h;
}

// ------------------------------------------------------------------------------------------------ weight packing
// kind 0 ("K256"): logical A[1024][256]; fragment (mb = 0..31, j = 0..15), lane (rho, hi), element i holds
//                  A[32 mb + rowmap(rho)][32 (j >> 1) + 16 hi + 8 (j & 1) + i]
// kind 1 ("H32") : logical A[256][1024]; fragment (s = 0..31, mb = 0..7, j = 0..1) holds
//                  A[32 mb + rowmap(rho)][32 s + 16 hi + 8 j + i]
// Both: fragment f of 512, 1 KB each, lane-linear.  A[r][c] = src[r * rs + c * cs] * rscale[r] * cscale[c].
__global__ __launch_bounds__(256) void mlp_pack_kernel(const float* __restrict__ src, int64_t rs, int64_t cs,
                                                       const float* __restrict__ rscale, const float* __restrict__ cscale,
                                                       uint16_t* __restrict__ dst, int kind, int64_t sstride, int64_t dstride) {
  const int64_t bz = blockIdx.y;
  src += bz * sstride;
  if (rscale) rscale += bz * sstride;
  if (cscale) cscale += bz * sstride;
  dst += bz * dstride;
  const int idx = blockIdx.x * 256 + threadIdx.x;
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

// ------------------------------------------------------------------------------------------------ forward
constexpr int MF_NSLOT = 4;                  // ring slots
constexpr int MF_AHEAD = 2;                  // bundles in flight ahead of the one being used
constexpr int MF_SLOT = 32768;               // bundle g = fc1 fragments of hidden block g (16 KB) | fc2 fragments of block g - 1
constexpr int MF_XCH = MF_NSLOT * MF_SLOT;   // per pair: 2 buffers x 2 planes x 1 KB
constexpr int MF_B1 = MF_XCH + 4 * 4096;
constexpr int MF_B2 = MF_B1 + 4096;
constexpr int MF_SMEM = MF_B2 + 1024;        // 152576 B

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

  {
    HMA_LDS(float)* b1s = (HMA_LDS(float)*)(lds + MF_B1);
    for (int i = tid; i < 1024; i += 512) b1s[i] = p.b1[i];
    if (tid < 256) ((HMA_LDS(float)*)(lds + MF_B2))[tid] = p.b2 ? p.b2[tid] : 0.f;
  }
  __syncthreads();

  // LDS-DMA: every wave moves 4 of a bundle's 32 pieces (2 of each half).  Always 4, so the vmcnt immediates are fixed.
  const char* w1g = reinterpret_cast<const char*>(p.w1p) + wave * 2048 + lane * 16;
  const char* w2g = reinterpret_cast<const char*>(p.w2p) + wave * 2048 + lane * 16;
  auto issue = [&](int b) __attribute__((always_inline)) {
    const uint32_t base = lds_b + (b % MF_NSLOT) * MF_SLOT + wave * 2048;
    const int s1 = b & 31, s2 = (b + 31) & 31;
    glds16(w1g + s1 * 16384, base);
    glds16(w1g + s1 * 16384 + 1024, base + 1024);
    glds16(w2g + s2 * 16384, base + 16384);
    glds16(w2g + s2 * 16384 + 1024, base + 16384 + 1024);
  };
  // bundle g has landed (all but the newest bundle's 4 pieces of this wave are complete: loads return in order) and
  // this wave's LDS writes of the previous step are done
  auto step_sync = [&](int g) __attribute__((always_inline)) {
    if (g < nsteps)
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
#pragma unroll
  for (int b = 0; b < MF_AHEAD; ++b) issue(b);

  auto tile_row = [&](int tl) __attribute__((always_inline)) {
    return ((int64_t)blockIdx.x + (int64_t)tl * gridDim.x) * 128 + pair * 32 + lr;
  };

  if (role == 0) {
    // ---------------------------------------------------------------- producer: u = W1f xhat + b1, hg = gelu(u)
    bf16x8_t xh[16], xn[16];
    auto load_x = [&](int tl, bf16x8_t (&dst)[16]) __attribute__((always_inline)) {
      int64_t row = tile_row(tl);
      row = row < p.M ? row : p.M - 1;
      const uint16_t* src = reinterpret_cast<const uint16_t*>(p.xhat) + row * 256 + 16 * hi;
#pragma unroll
      for (int j = 0; j < 16; ++j) dst[j] = as_frag(*reinterpret_cast<const uint4*>(src + 32 * (j >> 1) + 8 * (j & 1)));
    };
    load_x(0, xh);
#pragma unroll
    for (int j = 0; j < 16; ++j) xn[j] = xh[j];
    for (int g = 0; g <= nsteps; ++g) {
      step_sync(g);
      if (g + MF_AHEAD <= nsteps) issue(g + MF_AHEAD);
      if (g < nsteps) {
        const int s = g & 31;
        if (s == 0 && g > 0) {
#pragma unroll
          for (int j = 0; j < 16; ++j) xh[j] = xn[j];
        }
        HMA_LDS(char)* wb = lds + (g % MF_NSLOT) * MF_SLOT + lane * 16;
        f32x16_t U0, U1;
#pragma unroll
        for (int e = 0; e < 16; ++e) U0[e] = 0.f, U1[e] = 0.f;
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
            U0 = mfma32(fa[0], xh[4 * grp + 0], U0);
            U1 = mfma32(fa[1], xh[4 * grp + 1], U1);
            U0 = mfma32(fa[2], xh[4 * grp + 2], U0);
            U1 = mfma32(fa[3], xh[4 * grp + 3], U1);
            __builtin_amdgcn_sched_barrier(0);  // keeps the scheduler from hoisting every fragment read (it spills)
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = fb[i];
          }
        }
        HMA_LDS(char)* bp = lds + MF_B1 + (32 * s + 16 * hi) * 4;
        float h[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b = lds_f4(bp + 16 * q);
          h[4 * q + 0] = gelu_fused(U0[4 * q + 0] + U1[4 * q + 0] + b.x);
          h[4 * q + 1] = gelu_fused(U0[4 * q + 1] + U1[4 * q + 1] + b.y);
          h[4 * q + 2] = gelu_fused(U0[4 * q + 2] + U1[4 * q + 2] + b.z);
          h[4 * q + 3] = gelu_fused(U0[4 * q + 3] + U1[4 * q + 3] + b.w);
        }
        HMA_LDS(char)* xc = lds + MF_XCH + pair * 4096 + (g & 1) * 2048 + lane * 16;
        lds_put(xc, pack8(h));
        lds_put(xc + 1024, pack8(h + 8));
        if (s == 16 && (g >> 5) + 1 < nt) load_x((g >> 5) + 1, xn);  // next tile's rows, landed long before they are needed
      }
    }
  } else {
    // ---------------------------------------------------------------- consumer: x += hg W2^T + b2 (+ LayerNorm of the new row)
    f32x16_t Y[8];
    for (int g = 0; g <= nsteps; ++g) {
      step_sync(g);
      if (g + MF_AHEAD <= nsteps) issue(g + MF_AHEAD);
      if (g >= 1) {
        const int gc = g - 1, s = gc & 31;
        const int64_t row = tile_row(gc >> 5);
        const int64_t rowc = row < p.M ? row : p.M - 1;
        float* xrow = p.x + rowc * 256 + 16 * hi;
        if (s == 0) {  // the accumulators start from the residual row: x + (...) needs no separate add
#pragma unroll
          for (int cb = 0; cb < 8; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 v = *reinterpret_cast<const float4*>(xrow + 32 * cb + 4 * q);
              Y[cb][4 * q + 0] = v.x; Y[cb][4 * q + 1] = v.y; Y[cb][4 * q + 2] = v.z; Y[cb][4 * q + 3] = v.w;
            }
          // Consume the loads INSIDE this branch: otherwise hipcc places their s_waitcnt vmcnt(N) chain in front of the
          // MFMAs of every step (the join below), and in the steady state those waits drain the LDS-DMA pieces just issued.
#pragma unroll
          for (int cb = 0; cb < 8; ++cb) asm volatile("" : "+v"(Y[cb]));
        }
        HMA_LDS(char)* xc = lds + MF_XCH + pair * 4096 + (gc & 1) * 2048 + lane * 16;
        const bf16x8_t h0 = lds_frag(xc);
        const bf16x8_t h1 = lds_frag(xc + 1024);
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
              Y[cb] = mfma32(fa[i], (grp >> 1) ? h1 : h0, Y[cb]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = fb[i];
          }
        }
        if (s == 31) {
          const bool ok = row < p.M;
          HMA_LDS(char)* b2p = lds + MF_B2 + 16 * hi * 4;
          float sum = 0.f;
#pragma unroll
          for (int cb = 0; cb < 8; ++cb) {
            __builtin_amdgcn_sched_barrier(0);  // (the bias reads of all 8 blocks hoisted above the stores spill)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 b = lds_f4(b2p + (8 * cb + q) * 16);
              float4 v;
              v.x = Y[cb][4 * q + 0] + b.x; v.y = Y[cb][4 * q + 1] + b.y; v.z = Y[cb][4 * q + 2] + b.z; v.w = Y[cb][4 * q + 3] + b.w;
              if (ok) *reinterpret_cast<float4*>(xrow + 32 * cb + 4 * q) = v;
              if (LNOUT) {
                Y[cb][4 * q + 0] = v.x; Y[cb][4 * q + 1] = v.y; Y[cb][4 * q + 2] = v.z; Y[cb][4 * q + 3] = v.w;
                sum += v.x + v.y + v.z + v.w;
              }
            }
          }
          if (LNOUT) {  // two-pass mean / variance as ln_fwd_kernel; the row's other half sits in lane ^ 32
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * (1.0f / 256.0f);
            float sq = 0.f;
#pragma unroll
            for (int cb = 0; cb < 8; ++cb)
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                Y[cb][e] -= mean;
                sq += Y[cb][e] * Y[cb][e];
              }
            sq += __shfl_xor(sq, 32, 64);
            const float rstd = rsqrtf(sq * (1.0f / 256.0f) + p.ln_eps);
            if (ok) {
              uint16_t* xo = reinterpret_cast<uint16_t*>(p.ln_xhat) + row * 256 + 16 * hi;
#pragma unroll
              for (int cb = 0; cb < 8; ++cb) {
                float o[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) o[e] = Y[cb][e] * rstd;
                *reinterpret_cast<uint4*>(xo + 32 * cb) = pack8(o);
                *reinterpret_cast<uint4*>(xo + 32 * cb + 8) = pack8(o + 8);
              }
              if (hi == 0) p.ln_rstd[row] = rstd;
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
// Producer: u = W1f xhat + b1 (recomputed), dhg = W2^T dy, hg = gelu(u), du = dhg * gelu'(u); hg and du go to HBM for the
// two weight-gradient GEMMs (hma_gemm_tn_pair), du also to the consumer.  Consumer: dxhat = W1f^T du, then the LayerNorm
// backward (gamma already folded into W1f) added to the residual gradient.
constexpr int MB_NSLOT = 2;
constexpr int MB_SLOT = 49152;               // bundle g = fc1 frags | fc2^T frags of hidden block g | fc1^T frags of block g - 1
constexpr int MB_XCH = MB_NSLOT * MB_SLOT;
constexpr int MB_B1 = MB_XCH + 4 * 4096;
constexpr int MB_SMEM = MB_B1 + 4096;        // 118784 B

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
  __syncthreads();

  const char* g1 = reinterpret_cast<const char*>(p.w1p) + wave * 2048 + lane * 16;
  const char* g2 = reinterpret_cast<const char*>(p.w2tp) + wave * 2048 + lane * 16;
  const char* g3 = reinterpret_cast<const char*>(p.w1tp) + wave * 2048 + lane * 16;
  auto issue = [&](int b) __attribute__((always_inline)) {
    const uint32_t base = lds_b + (b % MB_NSLOT) * MB_SLOT + wave * 2048;
    const int s1 = b & 31, s2 = (b + 31) & 31;
    glds16(g1 + s1 * 16384, base);
    glds16(g1 + s1 * 16384 + 1024, base + 1024);
    glds16(g2 + s1 * 16384, base + 16384);
    glds16(g2 + s1 * 16384 + 1024, base + 16384 + 1024);
    glds16(g3 + s2 * 16384, base + 32768);
    glds16(g3 + s2 * 16384 + 1024, base + 32768 + 1024);
  };
  // one bundle ahead: bundle g was issued a whole step ago; everything this wave has in flight is waited for
  auto step_sync = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  issue(0);

  auto tile_row = [&](int tl) __attribute__((always_inline)) {
    return ((int64_t)blockIdx.x + (int64_t)tl * gridDim.x) * 128 + pair * 32 + lr;
  };

  if (role == 0) {
    bf16x8_t xh[16], dy[16];
    uint4 sv[4];             // packed hg | du of the previous step, stored at the start of the next one
    uint16_t* sp_hg = nullptr;
    uint16_t* sp_du = nullptr;
    bool sv_ok = false;
    for (int g = 0; g <= nsteps; ++g) {
      step_sync();
      if (g + 1 <= nsteps) issue(g + 1);
      if (g > 0 && sv_ok) {
        *reinterpret_cast<uint4*>(sp_hg) = sv[0];
        *reinterpret_cast<uint4*>(sp_hg + 8) = sv[1];
        *reinterpret_cast<uint4*>(sp_du) = sv[2];
        *reinterpret_cast<uint4*>(sp_du + 8) = sv[3];
      }
      if (g < nsteps) {
        const int s = g & 31;
        const int64_t row = tile_row(g >> 5);
        const int64_t rowc = row < p.M ? row : p.M - 1;
        if (s == 0) {
          const uint16_t* xs = reinterpret_cast<const uint16_t*>(p.xhat) + rowc * 256 + 16 * hi;
          const uint16_t* ds = reinterpret_cast<const uint16_t*>(p.dy) + rowc * 256 + 16 * hi;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            xh[j] = as_frag(*reinterpret_cast<const uint4*>(xs + 32 * (j >> 1) + 8 * (j & 1)));
            dy[j] = as_frag(*reinterpret_cast<const uint4*>(ds + 32 * (j >> 1) + 8 * (j & 1)));
          }
        }
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
            U = mfma32(fa[0], xh[2 * grp], U);
            D = mfma32(fa[1], dy[2 * grp], D);
            U = mfma32(fa[2], xh[2 * grp + 1], U);
            D = mfma32(fa[3], dy[2 * grp + 1], D);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = fb[i];
          }
        }
        HMA_LDS(char)* bp = lds + MB_B1 + (32 * s + 16 * hi) * 4;
        float hg[16], du[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b = lds_f4(bp + 16 * q);
          const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float u = U[4 * q + e] + bb[e];
            float h, gs;
            gelu_tail(u, h, gs);
            hg[4 * q + e] = fmaxf(u, 0.f) - fabsf(u) * h;
            const float cdf = u >= 0.f ? 1.0f - h : h;
            du[4 * q + e] = D[4 * q + e] * __builtin_fmaf(u * 0.3989422804014327f, gs, cdf);
          }
        }
        sv[0] = pack8(hg); sv[1] = pack8(hg + 8);
        sv[2] = pack8(du); sv[3] = pack8(du + 8);
        sv_ok = row < p.M;
        sp_hg = reinterpret_cast<uint16_t*>(p.hg) + rowc * 1024 + 32 * s + 16 * hi;
        sp_du = reinterpret_cast<uint16_t*>(p.du) + rowc * 1024 + 32 * s + 16 * hi;
        HMA_LDS(char)* xc = lds + MB_XCH + pair * 4096 + (g & 1) * 2048 + lane * 16;
        lds_put(xc, sv[2]);
        lds_put(xc + 1024, sv[3]);
      }
    }
  } else {
    f32x16_t G[8];
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) G[kb][e] = 0.f;
    for (int g = 0; g <= nsteps; ++g) {
      step_sync();
      if (g + 1 <= nsteps) issue(g + 1);
      if (g >= 1) {
        const int gc = g - 1, s = gc & 31;
        const int64_t row = tile_row(gc >> 5);
        const int64_t rowc = row < p.M ? row : p.M - 1;
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
              G[kb] = mfma32(fa[i], (grp >> 1) ? d1 : d0, G[kb]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = fb[i];
          }
        }
        if (s == 31) {
          // dx += rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dxhat (gamma folded into the weights).  xhat is read
          // twice (row statistics, then the update) in register-sized batches: holding the row across the main loop, or
          // letting the scheduler hoist all 48 loads, spills the 128 accumulators.
          const bool ok = row < p.M;
          const uint16_t* xs = reinterpret_cast<const uint16_t*>(p.xhat) + rowc * 256 + 16 * hi;
          const float rstd = p.rstd[rowc];
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int bt = 0; bt < 2; ++bt) {
            uint4 xe[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) xe[i] = *reinterpret_cast<const uint4*>(xs + 32 * (4 * bt + (i >> 1)) + 8 * (i & 1));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int kb = 4 * bt + (i >> 1);
              float xf[8];
              unpack8(xe[i], xf);
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                s1 += G[kb][8 * (i & 1) + e];
                s2 += G[kb][8 * (i & 1) + e] * xf[e];
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          s1 += __shfl_xor(s1, 32, 64);
          s2 += __shfl_xor(s2, 32, 64);
          s1 *= (1.0f / 256.0f);
          s2 *= (1.0f / 256.0f);
          float* dxr = p.dx + rowc * 256 + 16 * hi;
          uint16_t* dbr = reinterpret_cast<uint16_t*>(p.dx_bf16) + rowc * 256 + 16 * hi;
#pragma unroll
          for (int bt = 0; bt < 4; ++bt) {  // two 32-column blocks per batch: 4 + 8 loads in flight
            uint4 xe[4];
            float4 od[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) xe[i] = *reinterpret_cast<const uint4*>(xs + 32 * (2 * bt + (i >> 1)) + 8 * (i & 1));
#pragma unroll
            for (int i = 0; i < 8; ++i) od[i] = *reinterpret_cast<const float4*>(dxr + 32 * (2 * bt + (i >> 2)) + 4 * (i & 3));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int kb = 2 * bt + (i >> 1), hf = i & 1;
              float xf[8], o[8];
              unpack8(xe[i], xf);
              const float old[8] = {od[2 * i].x, od[2 * i].y, od[2 * i].z, od[2 * i].w,
                                    od[2 * i + 1].x, od[2 * i + 1].y, od[2 * i + 1].z, od[2 * i + 1].w};
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = old[e] + rstd * (G[kb][8 * hf + e] - s1 - xf[e] * s2);
              if (ok) {
                *reinterpret_cast<float4*>(dxr + 32 * kb + 8 * hf) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(dxr + 32 * kb + 8 * hf + 4) = make_float4(o[4], o[5], o[6], o[7]);
                *reinterpret_cast<uint4*>(dbr + 32 * kb + 8 * hf) = pack8(o);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int kb = 0; kb < 8; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) G[kb][e] = 0.f;
        }
      }
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

extern "C" int hma_mlp_pack(void* stream, const float* src, int64_t row_stride, int64_t col_stride, const float* row_scale,
                            const float* col_scale, void* dst, int32_t kind, int32_t batch, int64_t src_batch_stride,
                            int64_t dst_batch_stride) {
  if (!src || !dst || (kind != 0 && kind != 1) || batch < 1) return HMA_EINVAL;
  hipLaunchKernelGGL(mlp_pack_kernel, dim3(128, batch), dim3(256), 0, (hipStream_t)stream, src, row_stride, col_stride,
                     row_scale, col_scale, reinterpret_cast<uint16_t*>(dst), (int)kind, src_batch_stride, dst_batch_stride);
  HMA_CHECK_LAUNCH();
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
  const int64_t ntiles = (p->M + 127) >> 7;
  const int grid = (int)(ntiles < num_cus() ? ntiles : num_cus());
  if (int rc = set_lds<mlp_bwd_kernel>(MB_SMEM)) return rc;
  hipLaunchKernelGGL(mlp_bwd_kernel, dim3(grid), dim3(512), MB_SMEM, (hipStream_t)stream, *p);
  HMA_CHECK_LAUNCH();
  return 0;
}
