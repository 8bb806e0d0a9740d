// bf16 MFMA GEMMs of the HMA hot path for gfx950.
//
//   hma_gemm_nt : C[m,n] = sum_k A'[m,k] W[n,k]        (nn.Linear forward / dgrad)
//   hma_gemm_tn : dW[n,k] += sum_m dY[m,n] A'[m,k]     (nn.Linear wgrad, + bias grad)
//
// Both run the same inner product: a 128x128 output tile per 256-thread workgroup, 4 waves in a
// 2x2 grid, each wave 2x2 v_mfma_f32_32x32x16_bf16 tiles (64 fp32 accumulator VGPRs), K-steps of
// 64 double-buffered through LDS (rows padded to 144 B so every ds_read_b128 lane group lands on
// 16 distinct 16-B slots).  The MFMA is issued "swapped" -- weight rows as the A operand, token
// rows as the B operand -- so a lane owns ONE token row and 4 consecutive output columns per
// accumulator quad: epilogues touch 8/16 contiguous bytes per lane (bias, residual, GELU pairs).
//
// Reference call sites replaced: see include/hma_hip.h.
#include "hma_common.h"
#include "../../include/hma_hip.h"

#include <cstdlib>
#include <type_traits>

using namespace hma;

namespace {

// Debug-only phase timers (-DHMA_PROF, tools/phase_prof.py): per-wave s_memtime deltas summed per phase.
#ifdef HMA_PROF
__device__ unsigned long long g_prof[16];
#define PROF_DECL unsigned long long pt_ = __builtin_readcyclecounter(), pacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PROF_MARK(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); pacc_[i] += n_ - pt_; pt_ = n_; } while (0)
#define PROF_FLUSH() do { if ((threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_prof[i_], pacc_[i_]); atomicAdd(&g_prof[8], 1ull); } } while (0)
#else
#define PROF_DECL
#define PROF_MARK(i)
#define PROF_FLUSH()
#endif

constexpr int BM = 128;  // token rows per tile (NT) / k-columns of dW per tile (TN)
constexpr int BN = 128;  // weight rows per tile
constexpr int BK = 64;   // contraction step
constexpr int LDT = BK + 8;
constexpr int TILE_ELEMS = 128 * LDT;
constexpr int SMEM_BYTES = 4 * TILE_ELEMS * 2;  // 2 operands x 2 buffers, 73728 B

__device__ __forceinline__ void mma_tile(const uint16_t* __restrict__ Ts, const uint16_t* __restrict__ Ws,
                                         f32x16_t (&acc)[2][2], int wm, int wn, int lane) {
  const int r = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < BK / 16; ++kk) {
    bf16x8_t wf[2], tf[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      wf[i] = *reinterpret_cast<const bf16x8_t*>(&Ws[(wn * 64 + i * 32 + r) * LDT + kk * 16 + hi * 8]);
      tf[i] = *reinterpret_cast<const bf16x8_t*>(&Ts[(wm * 64 + i * 32 + r) * LDT + kk * 16 + hi * 8]);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = mfma32(wf[nt], tf[mt], acc[nt][mt]);
  }
}

__device__ __forceinline__ int64_t remap_row(int64_t r, int64_t group_rows, int64_t group_stride) {
  return group_rows > 0 ? (r / group_rows) * group_stride + (r % group_rows) : r;
}

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

// One accumulator quad: 4 consecutive output columns n..n+3 of (remapped) row crow; bias already added.
__device__ __forceinline__ void epilogue_quad(const hma_gemm_nt_t& p, const int epi, int64_t bz, int64_t crow, int64_t n, float (&v)[4]) {
  switch (epi) {
    case HMA_EPI_BF16: {
      uint16_t* C = reinterpret_cast<uint16_t*>(p.C) + bz * p.sC + crow * p.ldc + n;
      *reinterpret_cast<uint2*>(C) = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
    } break;
    case HMA_EPI_F32: {
      float* C = reinterpret_cast<float*>(p.C) + bz * p.sC + crow * p.ldc + n;
      *reinterpret_cast<float4*>(C) = make_float4(v[0], v[1], v[2], v[3]);
    } break;
    case HMA_EPI_RESID: {
      float* C = reinterpret_cast<float*>(p.C) + bz * p.sC + crow * p.ldc + n;
      float4 x = *reinterpret_cast<float4*>(C);
      x.x += v[0]; x.y += v[1]; x.z += v[2]; x.w += v[3];
      *reinterpret_cast<float4*>(C) = x;
      if (p.C2) {
        uint16_t* C2 = reinterpret_cast<uint16_t*>(p.C2) + bz * p.sC2 + crow * p.ldc2 + n;
        *reinterpret_cast<uint2*>(C2) = make_uint2(pack_bf16(x.x, x.y), pack_bf16(x.z, x.w));
      }
    } break;
    case HMA_EPI_GELU2:
    case HMA_EPI_SILU2: {
      uint16_t* C = reinterpret_cast<uint16_t*>(p.C) + bz * p.sC + crow * p.ldc + n;
      uint16_t* C2 = reinterpret_cast<uint16_t*>(p.C2) + bz * p.sC2 + crow * p.ldc2 + n;
      // the saved pre-activation is bf16: activate the ROUNDED value so backward sees the same u
      float u[4], a[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        u[e] = from_bf16(to_bf16(v[e]));
        a[e] = epi == HMA_EPI_GELU2 ? gelu_f(u[e]) : silu_f(u[e]);
      }
      *reinterpret_cast<uint2*>(C) = make_uint2(pack_bf16(u[0], u[1]), pack_bf16(u[2], u[3]));
      *reinterpret_cast<uint2*>(C2) = make_uint2(pack_bf16(a[0], a[1]), pack_bf16(a[2], a[3]));
    } break;
    case HMA_EPI_DGELU:
    case HMA_EPI_DSILU: {
      const uint16_t* U = reinterpret_cast<const uint16_t*>(p.U) + bz * p.sU + crow * p.ldu + n;
      const uint2 uu = *reinterpret_cast<const uint2*>(U);
      const float u[4] = {bf16_lo(uu.x), bf16_hi(uu.x), bf16_lo(uu.y), bf16_hi(uu.y)};
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= (epi == HMA_EPI_DGELU ? dgelu_f(u[e]) : dsilu_f(u[e]));
      uint16_t* C = reinterpret_cast<uint16_t*>(p.C) + bz * p.sC + crow * p.ldc + n;
      *reinterpret_cast<uint2*>(C) = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
    } break;
    case HMA_EPI_ATOMIC_F32: {
      float* C = reinterpret_cast<float*>(p.C) + bz * p.sC + crow * p.ldc + n;
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(C + e, v[e]);
    } break;
    default: break;
  }
}

// ------------------------------------------------------------------------------------------ NT
template <int AKIND>
__global__ __launch_bounds__(256) void gemm_nt_kernel(hma_gemm_nt_t p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* As = smem;                   // [2][128][LDT]
  uint16_t* Ws = smem + 2 * TILE_ELEMS;  // [2][128][LDT]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t bm = (int64_t)blockIdx.x * BM;
  const int64_t bn = (int64_t)blockIdx.y * BN;
  const int64_t bz = blockIdx.z;

  const char* Ab = reinterpret_cast<const char*>(p.A) + bz * p.sA * (AKIND == HMA_A_F32 ? 4 : 2);
  const uint16_t* Wb = reinterpret_cast<const uint16_t*>(p.W) + bz * p.sW;

  // per-thread staging coordinates: 4 chunks of 8 elements for each operand
  const int kc = tid & 7;
  int64_t a_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (tid >> 3) + i * 32;
    const int64_t gr = bm + row;
    a_off[i] = gr < p.M ? remap_row(gr, p.a_group_rows, p.a_group_stride) * p.lda : -1;
  }

  uint4 ra[4][AKIND == HMA_A_F32 ? 2 : 1];
  u32x4_t rw[4];  // (an ext vector: the uint4 struct array ended up in scratch, 80 bytes a lane)

  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (tid >> 3) + i * 32;
      rw[i] = *reinterpret_cast<const u32x4_t*>(Wb + (bn + row) * p.ldw + k0 + kc * 8);
      if (a_off[i] >= 0) {
        if (AKIND == HMA_A_F32) {
          const float* s = reinterpret_cast<const float*>(Ab) + a_off[i] + k0 + kc * 8;
          ra[i][0] = *reinterpret_cast<const uint4*>(s);
          ra[i][AKIND == HMA_A_F32 ? 1 : 0] = *reinterpret_cast<const uint4*>(s + 4);
        } else {
          const uint16_t* s = reinterpret_cast<const uint16_t*>(Ab) + a_off[i] + k0 + kc * 8;
          ra[i][0] = *reinterpret_cast<const uint4*>(s);
        }
      } else {
        ra[i][0] = make_uint4(0, 0, 0, 0);
        ra[i][AKIND == HMA_A_F32 ? 1 : 0] = make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto store_tiles = [&](int buf, int k0) {
    float gm[8], bt[8];
    if (AKIND == HMA_A_BF16_AFFINE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        gm[j] = p.gamma[k0 + kc * 8 + j];
        bt[j] = p.beta[k0 + kc * 8 + j];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (tid >> 3) + i * 32;
      uint4 v;
      if (AKIND == HMA_A_F32) {
        const float4 lo = __builtin_bit_cast(float4, ra[i][0]);
        const float4 hi = __builtin_bit_cast(float4, ra[i][AKIND == HMA_A_F32 ? 1 : 0]);
        v.x = pack_bf16(lo.x, lo.y); v.y = pack_bf16(lo.z, lo.w);
        v.z = pack_bf16(hi.x, hi.y); v.w = pack_bf16(hi.z, hi.w);
      } else if (AKIND == HMA_A_BF16_AFFINE) {
        float f[8];
        unpack8(ra[i][0], f);
        if (a_off[i] >= 0) {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = f[j] * gm[j] + bt[j];
        }
        v = pack8(f);
      } else {
        v = ra[i][0];
      }
      *reinterpret_cast<uint4*>(&As[buf * TILE_ELEMS + row * LDT + kc * 8]) = v;
      *reinterpret_cast<u32x4_t*>(&Ws[buf * TILE_ELEMS + row * LDT + kc * 8]) = rw[i];
    }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  const int KT = (int)(p.K / BK);
  load_tiles(0);
  store_tiles(0, 0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) load_tiles((kt + 1) * BK);
    mma_tile(As + cur * TILE_ELEMS, Ws + cur * TILE_ELEMS, acc, wm, wn, lane);
    if (kt + 1 < KT) store_tiles(cur ^ 1, (kt + 1) * BK);
    __syncthreads();
  }

  // ---- epilogue: lane owns token row m and 4 consecutive columns per accumulator quad
  const int r = lane & 31, hi = lane >> 5;
  const float* bias = p.bias ? p.bias + bz * p.sBias : nullptr;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int64_t m = bm + wm * 64 + mt * 32 + r;
    if (m >= p.M) continue;
    const int64_t crow = remap_row(m, p.c_group_rows, p.c_group_stride);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t n = bn + wn * 64 + nt * 32 + 8 * g + 4 * hi;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[nt][mt][4 * g + e];
        if (bias) {
          const float4 b4 = *reinterpret_cast<const float4*>(bias + n);
          v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
        }
        epilogue_quad(p, p.epi, bz, crow, n, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------- NT, persistent
// The layer GEMMs of this model have K = 256..1024: a tile-per-workgroup launch spends most of its
// time filling a 4..16-step pipeline.  Here a workgroup (512 threads, 8 waves, one per CU) walks a
// list of 128 x 256 output tiles and keeps ONE software pipeline running across tile boundaries:
// global loads are issued two 64-deep K-steps ahead into registers, LDS is double buffered with a
// single barrier per step, and consecutive workgroup ids of one XCD share the A rows of an m-tile.
constexpr int PM = 128, PN = 256, PK = 64;
constexpr int P_A = PM * LDT, P_W = PN * LDT, P_STAGE = P_A + P_W;
constexpr int P_SMEM_BYTES = 2 * P_STAGE * 2;  // 110592 B

template <int AKIND>
struct PRegs {
  uint4 a[2][AKIND == HMA_A_F32 ? 2 : 1];
  uint4 w[4];
  bool a_ok[2];
  int k0;
};

// Two adjacent accumulator quads (columns nq + 8*g2 + 4*hi + {0..3} and the same for g2 + 1) of one row.
// Measured (tools/probes/store_pattern.hip): 8-byte-per-lane stores at a row stride reach 3.2-3.6 TB/s,
// 16-byte ones 5.8-7.0 TB/s.  For bf16 outputs the two half-waves therefore trade quads with
// v_permlane32_swap so that every lane owns 8 consecutive bf16 = one 16-byte access: lane (r, hi = 0)
// ends with columns 8*g2 .. +7, lane (r, hi = 1) with 8*(g2+1) .. +7.  The swap is an involution, so the
// same exchange turns a 16-byte load back into the lane's own two quads.
__device__ __forceinline__ void swap_halves(uint32_t& a, uint32_t& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
__device__ __forceinline__ void store_bf16_oct(uint16_t* row_q, int g2, int hi, const float (&v0)[4], const float (&v1)[4]) {
  uint32_t ax = pack_bf16(v0[0], v0[1]), ay = pack_bf16(v0[2], v0[3]);
  uint32_t bx = pack_bf16(v1[0], v1[1]), by = pack_bf16(v1[2], v1[3]);
  swap_halves(ax, bx);
  swap_halves(ay, by);
  *reinterpret_cast<uint4*>(row_q + 8 * (g2 + hi)) = make_uint4(ax, ay, bx, by);
}
__device__ __forceinline__ void load_bf16_oct(const uint16_t* row_q, int g2, int hi, float (&u0)[4], float (&u1)[4]) {
  uint4 q = *reinterpret_cast<const uint4*>(row_q + 8 * (g2 + hi));
  swap_halves(q.x, q.z);
  swap_halves(q.y, q.w);
  u0[0] = bf16_lo(q.x); u0[1] = bf16_hi(q.x); u0[2] = bf16_lo(q.y); u0[3] = bf16_hi(q.y);
  u1[0] = bf16_lo(q.z); u1[1] = bf16_hi(q.z); u1[2] = bf16_lo(q.w); u1[3] = bf16_hi(q.w);
}

// row_q pointers address column nq (the 32-wide MFMA tile's first column) of the output row
template <int EPI>
__device__ __forceinline__ void epilogue_oct(const hma_gemm_nt_t& p, int64_t bz, int64_t crow, int64_t nq, int g2, int hi,
                                             float (&v0)[4], float (&v1)[4]) {
  if (EPI == HMA_EPI_BF16) {
    store_bf16_oct(reinterpret_cast<uint16_t*>(p.C) + bz * p.sC + crow * p.ldc + nq, g2, hi, v0, v1);
  } else if (EPI == HMA_EPI_F32) {
    float* C = reinterpret_cast<float*>(p.C) + bz * p.sC + crow * p.ldc + nq + 8 * g2 + 4 * hi;
    *reinterpret_cast<float4*>(C) = make_float4(v0[0], v0[1], v0[2], v0[3]);
    *reinterpret_cast<float4*>(C + 8) = make_float4(v1[0], v1[1], v1[2], v1[3]);
  } else if (EPI == HMA_EPI_RESID) {
    float* C = reinterpret_cast<float*>(p.C) + bz * p.sC + crow * p.ldc + nq + 8 * g2 + 4 * hi;
    if (p.drop_p > 0.f) {  // Dropout on the branch output before the residual add (st_transformer.py:26)
      const uint32_t seed = *p.drop_seed, th = drop_thresh(p.drop_p);
      const float sc = drop_scale(p.drop_p);
      const int64_t e0 = crow * p.ldc + nq + 8 * g2 + 4 * hi;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v0[e] = drop_keep(seed, p.drop_salt, e0 + e, th) ? v0[e] * sc : 0.f;
        v1[e] = drop_keep(seed, p.drop_salt, e0 + 8 + e, th) ? v1[e] * sc : 0.f;
      }
    }
    float4 x0 = *reinterpret_cast<float4*>(C), x1 = *reinterpret_cast<float4*>(C + 8);
    x0.x += v0[0]; x0.y += v0[1]; x0.z += v0[2]; x0.w += v0[3];
    x1.x += v1[0]; x1.y += v1[1]; x1.z += v1[2]; x1.w += v1[3];
    *reinterpret_cast<float4*>(C) = x0;
    *reinterpret_cast<float4*>(C + 8) = x1;
    if (p.C2) {
      const float a[4] = {x0.x, x0.y, x0.z, x0.w}, b[4] = {x1.x, x1.y, x1.z, x1.w};
      store_bf16_oct(reinterpret_cast<uint16_t*>(p.C2) + bz * p.sC2 + crow * p.ldc2 + nq, g2, hi, a, b);
    }
  } else if (EPI == HMA_EPI_GELU2 || EPI == HMA_EPI_SILU2) {
    // the saved pre-activation is bf16: activate the ROUNDED value so backward sees the same u
    float u0[4], u1[4], a0[4], a1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      u0[e] = from_bf16(to_bf16(v0[e]));
      u1[e] = from_bf16(to_bf16(v1[e]));
      a0[e] = EPI == HMA_EPI_GELU2 ? gelu_f(u0[e]) : silu_f(u0[e]);
      a1[e] = EPI == HMA_EPI_GELU2 ? gelu_f(u1[e]) : silu_f(u1[e]);
    }
    store_bf16_oct(reinterpret_cast<uint16_t*>(p.C) + bz * p.sC + crow * p.ldc + nq, g2, hi, u0, u1);
    store_bf16_oct(reinterpret_cast<uint16_t*>(p.C2) + bz * p.sC2 + crow * p.ldc2 + nq, g2, hi, a0, a1);
  } else if (EPI == HMA_EPI_DGELU || EPI == HMA_EPI_DSILU) {
    float u0[4], u1[4];
    load_bf16_oct(reinterpret_cast<const uint16_t*>(p.U) + bz * p.sU + crow * p.ldu + nq, g2, hi, u0, u1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v0[e] *= (EPI == HMA_EPI_DGELU ? dgelu_f(u0[e]) : dsilu_f(u0[e]));
      v1[e] *= (EPI == HMA_EPI_DGELU ? dgelu_f(u1[e]) : dsilu_f(u1[e]));
    }
    store_bf16_oct(reinterpret_cast<uint16_t*>(p.C) + bz * p.sC + crow * p.ldc + nq, g2, hi, v0, v1);
  } else {
    float* C = reinterpret_cast<float*>(p.C) + bz * p.sC + crow * p.ldc + nq + 8 * g2 + 4 * hi;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      atomicAdd(C + e, v0[e]);
      atomicAdd(C + 8 + e, v1[e]);
    }
  }
}

// ------------------------------------------------------------- NT, persistent, deferred epilogue
// Same 8-wave, one-per-CU pipeline as gemm_nt_persist_kernel, with the epilogue as a template parameter,
// 16-byte bf16 accesses (epilogue_oct) and the deferred, interleaved epilogue described inside.
// SHALLOW (K < 256: fewer than four K-steps per tile -- STMAR's token_embed and the diffusion head's input layer, K = 128): the
// finished tile is written out at once.  The deferred epilogue below needs four steps of the NEXT tile to hide behind; with
// fewer, a workgroup that owned two tiles lost parts of the first one, i.e. from M > 128 x #CUs = 32 768 rows (found in round 3
// by the batch-decomposition property of tests/test_fulldepth_stmar_gpu.py).
template <int AKIND, int EPI, bool SHALLOW = false>
__global__ __launch_bounds__(512, 2) void gemm_nt_p3_kernel(hma_gemm_nt_t p, int tiles_m, int tiles_n, int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int KT = (int)(p.K / PK);
  const int kc = tid & 7;

  // XCD-aware virtual id: the workgroups of one XCD (b % 8) take a contiguous run of tiles
  const int G = gridDim.x;
  const int b = blockIdx.x;
  const int per = G >> 3;
  const int vid = ((G & 7) == 0) ? (b & 7) * per + (b >> 3) : b;
  const int my_tiles = vid < total_tiles ? (total_tiles - vid + G - 1) / G : 0;
  const int total_it = my_tiles * KT;
  if (total_it == 0) return;

  struct Cursor { int tile; int kt; int64_t a_off[2]; int64_t w_off; int64_t bz; };
  auto decode = [&](Cursor& c) {
    const int per_b = tiles_m * tiles_n;
    c.bz = c.tile / per_b;
    const int r = c.tile % per_b;
    const int mt = r / tiles_n, nt = r % tiles_n;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t gr = (int64_t)mt * PM + (tid >> 3) + i * 64;
      c.a_off[i] = gr < p.M ? remap_row(gr, p.a_group_rows, p.a_group_stride) * p.lda : -1;
    }
    c.w_off = ((int64_t)nt * PN + (tid >> 3)) * p.ldw;
  };
  Cursor ld;
  ld.tile = vid; ld.kt = 0;
  decode(ld);

  auto load = [&](PRegs<AKIND>& r) {
    const int k0 = ld.kt * PK + kc * 8;
    r.k0 = k0;
    const char* Ab = reinterpret_cast<const char*>(p.A) + ld.bz * p.sA * (AKIND == HMA_A_F32 ? 4 : 2);
    const uint16_t* Wb = reinterpret_cast<const uint16_t*>(p.W) + ld.bz * p.sW + ld.w_off + k0;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.w[i] = *reinterpret_cast<const uint4*>(Wb + (int64_t)i * 64 * p.ldw);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      r.a_ok[i] = ld.a_off[i] >= 0;
      if (r.a_ok[i]) {
        if (AKIND == HMA_A_F32) {
          const float* s = reinterpret_cast<const float*>(Ab) + ld.a_off[i] + k0;
          r.a[i][0] = *reinterpret_cast<const uint4*>(s);
          r.a[i][AKIND == HMA_A_F32 ? 1 : 0] = *reinterpret_cast<const uint4*>(s + 4);
        } else {
          r.a[i][0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(Ab) + ld.a_off[i] + k0);
        }
      } else {
        r.a[i][0] = make_uint4(0, 0, 0, 0);
        r.a[i][AKIND == HMA_A_F32 ? 1 : 0] = make_uint4(0, 0, 0, 0);
      }
    }
    if (++ld.kt == KT) {
      ld.kt = 0;
      ld.tile += G;
      if (ld.tile < total_tiles) decode(ld);
    }
  };
  auto store = [&](const PRegs<AKIND>& r, int buf) {
    uint16_t* As = smem + buf * P_STAGE;
    uint16_t* Ws = As + P_A;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(&Ws[((tid >> 3) + i * 64) * LDT + kc * 8]) = r.w[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      uint4 v;
      if (AKIND == HMA_A_F32) {
        const float4 lo = __builtin_bit_cast(float4, r.a[i][0]);
        const float4 hi = __builtin_bit_cast(float4, r.a[i][AKIND == HMA_A_F32 ? 1 : 0]);
        v.x = pack_bf16(lo.x, lo.y); v.y = pack_bf16(lo.z, lo.w);
        v.z = pack_bf16(hi.x, hi.y); v.w = pack_bf16(hi.z, hi.w);
      } else if (AKIND == HMA_A_BF16_AFFINE) {
        float f[8];
        unpack8(r.a[i][0], f);
        if (r.a_ok[i]) {
          const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + r.k0), g1 = *reinterpret_cast<const float4*>(p.gamma + r.k0 + 4);
          const float4 b0 = *reinterpret_cast<const float4*>(p.beta + r.k0), b1 = *reinterpret_cast<const float4*>(p.beta + r.k0 + 4);
          f[0] = f[0] * g0.x + b0.x; f[1] = f[1] * g0.y + b0.y; f[2] = f[2] * g0.z + b0.z; f[3] = f[3] * g0.w + b0.w;
          f[4] = f[4] * g1.x + b1.x; f[5] = f[5] * g1.y + b1.y; f[6] = f[6] * g1.z + b1.z; f[7] = f[7] * g1.w + b1.w;
        }
        v = pack8(f);
      } else {
        v = r.a[i][0];
      }
      *reinterpret_cast<uint4*>(&As[((tid >> 3) + i * 64) * LDT + kc * 8]) = v;
    }
  };

  f32x16_t acc[2][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.f;
  };
  zero_acc();

  int cur_tile = vid, cur_kt = 0;
  // Deferred epilogue: a finished tile's accumulators move to `pacc` and are written out in four parts,
  // one after the MFMAs of each of the next tile's first four K-steps, so the store (and residual-load)
  // traffic interleaves with the main loop's loads instead of arriving as one burst that stalls every wave.
  f32x16_t pacc[2][2];
  int ptile = -1;
  const int lr = lane & 31, lhi = lane >> 5;
  auto emit_part = [&](auto mt_tag, auto nt_tag) __attribute__((always_inline)) {
    constexpr int mt = decltype(mt_tag)::value, nt = decltype(nt_tag)::value;
    const int per_b = tiles_m * tiles_n;
    const int64_t bz = ptile / per_b;
    const int rr = ptile % per_b;
    const int64_t bm = (int64_t)(rr / tiles_n) * PM, bn = (int64_t)(rr % tiles_n) * PN;
    const int64_t m = bm + wm * 64 + mt * 32 + lr;
    if (m < p.M) {
      const int64_t crow = remap_row(m, p.c_group_rows, p.c_group_stride);
      const int64_t nq = bn + wn * 64 + nt * 32;
      const float* bias = p.bias ? p.bias + bz * p.sBias : nullptr;
#pragma unroll
      for (int g2 = 0; g2 < 4; g2 += 2) {
        float v0[4], v1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v0[e] = pacc[nt][mt][4 * g2 + e];
          v1[e] = pacc[nt][mt][4 * g2 + 4 + e];
        }
        if (bias) {
          const float4 b0 = *reinterpret_cast<const float4*>(bias + nq + 8 * g2 + 4 * lhi);
          const float4 b1 = *reinterpret_cast<const float4*>(bias + nq + 8 * g2 + 8 + 4 * lhi);
          v0[0] += b0.x; v0[1] += b0.y; v0[2] += b0.z; v0[3] += b0.w;
          v1[0] += b1.x; v1[1] += b1.y; v1[2] += b1.z; v1[3] += b1.w;
        }
        epilogue_oct<EPI>(p, bz, crow, nq, g2, lhi, v0, v1);
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  auto emit_step = [&](int part) __attribute__((always_inline)) {
    if (ptile < 0) return;
    switch (part) {
      case 0: emit_part(I0{}, I0{}); break;
      case 1: emit_part(I0{}, I1{}); break;
      case 2: emit_part(I1{}, I0{}); break;
      case 3: emit_part(I1{}, I1{}); ptile = -1; break;
      default: break;
    }
  };

  PRegs<AKIND> r0, r1;
  load(r0);
  if (total_it > 1) load(r1);
  store(r0, 0);
  __syncthreads();
#ifdef HMA_PROF
  const int ablate = p._pad2;  // debug build only (HMA_GEMM_ABLATE): 1 = skip epilogue, 2 = skip loads, 4 = skip MFMA
#else
  constexpr int ablate = 0;
#endif
  PROF_DECL;
  auto step = [&](int it, PRegs<AKIND>& mine, PRegs<AKIND>& other) {
    PROF_MARK(0);
    // `mine` held step `it` (already in LDS) and is free: refill it with step it + 2
    if (it + 2 < total_it && !(ablate & 2)) load(mine);
    PROF_MARK(1);
    const uint16_t* As = smem + (it & 1) * P_STAGE;
    if (!(ablate & 4)) mma_tile(As, As + P_A, acc, wm, wn, lane);
    PROF_MARK(2);
    if (!SHALLOW && !(ablate & 1)) emit_step(cur_kt);
    PROF_MARK(3);
    if (++cur_kt == KT) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) pacc[a][c] = acc[a][c];
      ptile = cur_tile;
      if (SHALLOW && !(ablate & 1)) {
        emit_part(I0{}, I0{});
        emit_part(I0{}, I1{});
        emit_part(I1{}, I0{});
        emit_part(I1{}, I1{});
        ptile = -1;
      }
      zero_acc();
      cur_kt = 0;
      cur_tile += G;
    }
    PROF_MARK(6);
    if (it + 1 < total_it) store(other, (it + 1) & 1);
    PROF_MARK(4);
    __syncthreads();
    PROF_MARK(5);
  };
  for (int it = 0; it < total_it; it += 2) {
    step(it, r0, r1);
    if (it + 1 < total_it) step(it + 1, r1, r0);
  }
  if (!(ablate & 1)) {  // the last tile's epilogue has no following main loop to hide behind
    for (int part = 0; part < 4; ++part) emit_step(part);
  }
  PROF_MARK(3);
  PROF_FLUSH();
}


// --------------------------------------------------------------------- NT, persistent, 2 per CU
// Measured on MI355X (tools/ablate.sh): with one 8-wave workgroup per CU the epilogue of a tile (its
// HBM stores, at ~5 TB/s aggregate, plus the GELU VALU work) and the MFMA main loop (~1 PFLOP/s without
// it) simply add up, because every wave of the CU is in the same phase.  This variant keeps the same
// 128 x 256 tile and cross-tile software pipeline but runs it with 4 waves (each 64 x 128, 128
// accumulator VGPRs), 32-deep K-steps and 60 KB of LDS, so TWO workgroups are resident per CU and one's
// store/VALU epilogue overlaps the other's MFMA phase.
constexpr int QM = 128, QN = 256, QK = 32, QLD = 40;
constexpr int Q_A = QM * QLD, Q_W = QN * QLD, Q_STAGE = Q_A + Q_W;
constexpr int Q_SMEM_BYTES = 2 * Q_STAGE * 2;  // 61440 B


template <int AKIND, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_p2_kernel(hma_gemm_nt_t p, int tiles_m, int tiles_n, int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int KT = (int)(p.K / QK);
  const int kc = tid & 3;
  const int srow = tid >> 2;  // 0..63

  const int G = gridDim.x;
  const int b = blockIdx.x;
  const int vid = ((G & 7) == 0) ? (b & 7) * (G >> 3) + (b >> 3) : b;
  const int my_tiles = vid < total_tiles ? (total_tiles - vid + G - 1) / G : 0;
  const int total_it = my_tiles * KT;
  if (total_it == 0) return;

  struct Cursor { int tile; int kt; int64_t a_off[2]; int64_t w_off; int64_t bz; };
  auto decode = [&](Cursor& c) {
    const int per_b = tiles_m * tiles_n;
    c.bz = c.tile / per_b;
    const int r = c.tile % per_b;
    const int mt = r / tiles_n, nt = r % tiles_n;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t gr = (int64_t)mt * QM + srow + i * 64;
      c.a_off[i] = gr < p.M ? remap_row(gr, p.a_group_rows, p.a_group_stride) * p.lda : -1;
    }
    c.w_off = ((int64_t)nt * QN + srow) * p.ldw;
  };
  Cursor ld;
  ld.tile = vid; ld.kt = 0;
  decode(ld);

  auto load = [&](PRegs<AKIND>& r) {
    const int k0 = ld.kt * QK + kc * 8;
    r.k0 = k0;
    const char* Ab = reinterpret_cast<const char*>(p.A) + ld.bz * p.sA * (AKIND == HMA_A_F32 ? 4 : 2);
    const uint16_t* Wb = reinterpret_cast<const uint16_t*>(p.W) + ld.bz * p.sW + ld.w_off + k0;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.w[i] = *reinterpret_cast<const uint4*>(Wb + (int64_t)i * 64 * p.ldw);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      r.a_ok[i] = ld.a_off[i] >= 0;
      if (r.a_ok[i]) {
        if (AKIND == HMA_A_F32) {
          const float* s = reinterpret_cast<const float*>(Ab) + ld.a_off[i] + k0;
          r.a[i][0] = *reinterpret_cast<const uint4*>(s);
          r.a[i][AKIND == HMA_A_F32 ? 1 : 0] = *reinterpret_cast<const uint4*>(s + 4);
        } else {
          r.a[i][0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(Ab) + ld.a_off[i] + k0);
        }
      } else {
        r.a[i][0] = make_uint4(0, 0, 0, 0);
        r.a[i][AKIND == HMA_A_F32 ? 1 : 0] = make_uint4(0, 0, 0, 0);
      }
    }
    if (++ld.kt == KT) {
      ld.kt = 0;
      ld.tile += G;
      if (ld.tile < total_tiles) decode(ld);
    }
  };
  auto store = [&](const PRegs<AKIND>& r, int buf) {
    uint16_t* As = smem + buf * Q_STAGE;
    uint16_t* Ws = As + Q_A;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(&Ws[(srow + i * 64) * QLD + kc * 8]) = r.w[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      uint4 v;
      if (AKIND == HMA_A_F32) {
        const float4 lo = __builtin_bit_cast(float4, r.a[i][0]);
        const float4 hi = __builtin_bit_cast(float4, r.a[i][AKIND == HMA_A_F32 ? 1 : 0]);
        v.x = pack_bf16(lo.x, lo.y); v.y = pack_bf16(lo.z, lo.w);
        v.z = pack_bf16(hi.x, hi.y); v.w = pack_bf16(hi.z, hi.w);
      } else if (AKIND == HMA_A_BF16_AFFINE) {
        float f[8];
        unpack8(r.a[i][0], f);
        if (r.a_ok[i]) {
          const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + r.k0), g1 = *reinterpret_cast<const float4*>(p.gamma + r.k0 + 4);
          const float4 b0 = *reinterpret_cast<const float4*>(p.beta + r.k0), b1 = *reinterpret_cast<const float4*>(p.beta + r.k0 + 4);
          f[0] = f[0] * g0.x + b0.x; f[1] = f[1] * g0.y + b0.y; f[2] = f[2] * g0.z + b0.z; f[3] = f[3] * g0.w + b0.w;
          f[4] = f[4] * g1.x + b1.x; f[5] = f[5] * g1.y + b1.y; f[6] = f[6] * g1.z + b1.z; f[7] = f[7] * g1.w + b1.w;
        }
        v = pack8(f);
      } else {
        v = r.a[i][0];
      }
      *reinterpret_cast<uint4*>(&As[(srow + i * 64) * QLD + kc * 8]) = v;
    }
  };

  f32x16_t acc[4][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.f;
  };
  zero_acc();

  int cur_tile = vid, cur_kt = 0;
  auto finish_tile = [&]() {
    const int per_b = tiles_m * tiles_n;
    const int64_t bz = cur_tile / per_b;
    const int rr = cur_tile % per_b;
    const int64_t bm = (int64_t)(rr / tiles_n) * QM, bn = (int64_t)(rr % tiles_n) * QN;
    const int r = lane & 31, hi = lane >> 5;
    const float* bias = p.bias ? p.bias + bz * p.sBias : nullptr;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int64_t m = bm + wm * 64 + mt * 32 + r;
      if (m < p.M) {
        const int64_t crow = remap_row(m, p.c_group_rows, p.c_group_stride);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int g2 = 0; g2 < 4; g2 += 2) {
            const int64_t nq = bn + wn * 128 + nt * 32;
            float v0[4], v1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v0[e] = acc[nt][mt][4 * g2 + e];
              v1[e] = acc[nt][mt][4 * g2 + 4 + e];
            }
            if (bias) {
              const float4 b0 = *reinterpret_cast<const float4*>(bias + nq + 8 * g2 + 4 * hi);
              const float4 b1 = *reinterpret_cast<const float4*>(bias + nq + 8 * g2 + 8 + 4 * hi);
              v0[0] += b0.x; v0[1] += b0.y; v0[2] += b0.z; v0[3] += b0.w;
              v1[0] += b1.x; v1[1] += b1.y; v1[2] += b1.z; v1[3] += b1.w;
            }
            epilogue_oct<EPI>(p, bz, crow, nq, g2, hi, v0, v1);
          }
      }
    }
    zero_acc();
  };
  const int r = lane & 31, hi = lane >> 5;
#ifdef HMA_PROF
  const int ablate = p._pad2 & 7;  // debug build only (HMA_GEMM_ABLATE): 1 = skip epilogue, 2 = skip loads, 4 = skip MFMA
#else
  constexpr int ablate = 0;
#endif
  PRegs<AKIND> r0, r1;
  load(r0);
  if (total_it > 1) load(r1);
  store(r0, 0);
  __syncthreads();
  PROF_DECL;
  auto step = [&](int it, PRegs<AKIND>& mine, PRegs<AKIND>& other) {
    PROF_MARK(0);
    if (it + 2 < total_it && !(ablate & 2)) load(mine);
    PROF_MARK(1);
    const uint16_t* As = smem + (it & 1) * Q_STAGE;
    const uint16_t* Ws = As + Q_A;
    if (!(ablate & 4)) {
#pragma unroll
      for (int kk = 0; kk < QK / 16; ++kk) {
        bf16x8_t wf[4], tf[2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          wf[i] = *reinterpret_cast<const bf16x8_t*>(&Ws[(wn * 128 + i * 32 + r) * QLD + kk * 16 + hi * 8]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
          tf[i] = *reinterpret_cast<const bf16x8_t*>(&As[(wm * 64 + i * 32 + r) * QLD + kk * 16 + hi * 8]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = mfma32(wf[nt], tf[mt], acc[nt][mt]);
      }
    }
    PROF_MARK(2);
    if (++cur_kt == KT) {
      if (!(ablate & 1)) finish_tile();
      cur_kt = 0;
      cur_tile += G;
    }
    PROF_MARK(3);
    if (it + 1 < total_it) store(other, (it + 1) & 1);
    PROF_MARK(4);
    __syncthreads();
    PROF_MARK(5);
  };
  for (int it = 0; it < total_it; it += 2) {
    step(it, r0, r1);
    if (it + 1 < total_it) step(it + 1, r1, r0);
  }
  PROF_FLUSH();
}

// ------------------------------------------------------------- NT, streaming waves (K = 256)
// The kernels above are one (or two) workgroups per CU marching in lock step: a barrier per K-step, every
// wave in the same phase, and -- by the phase timers -- 50-80 % of the wave time spent in the epilogue or
// waiting for operand loads.  At K = 256 these GEMMs are HBM-bound streams (bytes / 5.5 TB/s is 2-2.4x
// below the measured times), so this variant is organised like a streaming kernel instead:
//   * a workgroup parks one 256-column weight slab in LDS (132 KB) once; after that its 8 waves never
//     synchronise again;
//   * a wave owns whole 16-token row tiles: it loads the tile's A rows straight into MFMA B-operand
//     registers (lane = token, 8 x 16 B spread over the row), multiplies against all 256 columns with
//     v_mfma_f32_16x16x32_bf16 (weights as the A operand from LDS), and writes its own outputs; the next
//     tile's A rows are in flight while it computes;
//   * the contraction index and the weight rows are permuted (free: both are only LDS addressing) so that a
//     lane loads full 16-byte chunks and ends up with 8 CONSECUTIVE output columns of its token per pair of
//     16-column MFMA tiles -- 16-byte bf16 / 32-byte fp32 accesses without any cross-lane exchange.
// k permutation: lane (tok, g = lane >> 4) holds chunks c = 4 j + g (j = 0..7) of its row, MFMA step j
// contracts k = 8 c .. 8 c + 7.  Row permutation: tile 2p + o, MFMA row i  <->  n = 32 p + 8 (i >> 2) + (i & 3) + 4 o.
constexpr int TW_LD = 256 + 8;                       // padded weight-slab row (elements)
// exact-erf GELU by table: the activation's argument is a bf16 value, so Phi(u) (forward) or dgelu(u) (backward)
// is a function of 16 bits; every |u| in [2^-16, 20) -- 2592 bf16 values per sign -- gets an fp32 entry in LDS
// (20.7 KB, built per workgroup with erff at start-up), smaller |u| use the first entry (Phi = 0.5 to 6e-6), larger
// the last (0 or 1).  ~9 instructions + one LDS read per element instead of ~23 (the epilogue was VALU-bound), and
// the result is the correctly rounded fp32 gelu of the rounded pre-activation, as in the reference.
constexpr int GT_A0 = 0x3780, GT_A1 = 0x41A0, GT_N = GT_A1 - GT_A0;  // bf16 bit patterns of 2^-16 and 20.0
constexpr int TW_SMEM_BYTES = 256 * TW_LD * 2 + 2 * 256 * 4 + 2 * GT_N * 4;  // slab + gamma/beta + table
__device__ __forceinline__ float gt_lookup(const float* tbl, uint32_t bits16) {
  const uint32_t a = bits16 & 0x7fffu;
  const uint32_t k = min(max(a, (uint32_t)GT_A0), (uint32_t)(GT_A1 - 1));  // -> v_med3_u32
  return tbl[k + (bits16 >> 15) * GT_N];  // tbl points at entry -GT_A0
}

typedef __attribute__((ext_vector_type(4))) float f32x4v_t;

template <int EPI>
__device__ __forceinline__ void epilogue_run8(const hma_gemm_nt_t& p, int64_t bz, int64_t crow, int64_t n8, float (&v)[8]) {
  // v = 8 consecutive columns n8 .. n8 + 7 of output row crow
  if (EPI == HMA_EPI_BF16) {
    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.C) + bz * p.sC + crow * p.ldc + n8) = pack8(v);
  } else if (EPI == HMA_EPI_F32) {
    float* C = reinterpret_cast<float*>(p.C) + bz * p.sC + crow * p.ldc + n8;
    *reinterpret_cast<float4*>(C) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(C + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else if (EPI == HMA_EPI_GELU2 || EPI == HMA_EPI_SILU2) {
    float u[8], a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      u[e] = from_bf16(to_bf16(v[e]));  // activate the ROUNDED pre-activation (what backward will read)
      a[e] = EPI == HMA_EPI_GELU2 ? gelu_f(u[e]) : silu_f(u[e]);
    }
    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.C) + bz * p.sC + crow * p.ldc + n8) = pack8(u);
    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.C2) + bz * p.sC2 + crow * p.ldc2 + n8) = pack8(a);
  } else if (EPI == HMA_EPI_DGELU || EPI == HMA_EPI_DSILU) {
    const uint4 q = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(p.U) + bz * p.sU + crow * p.ldu + n8);
    float u[8];
    unpack8(q, u);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= (EPI == HMA_EPI_DGELU ? dgelu_f(u[e]) : dsilu_f(u[e]));
    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.C) + bz * p.sC + crow * p.ldc + n8) = pack8(v);
  }
}

// NWAVES = 12 (3 per SIMD, <= 168 VGPRs) where the registers allow it, else 8: waves are independent, so more of
// them is more loads in flight and more overlap of one wave's VALU/store epilogue with another's MFMAs.
template <int AKIND, int EPI, int NWAVES, bool LNF = false>
__global__ __launch_bounds__(64 * NWAVES, NWAVES / 4) void gemm_nt_sw_kernel(hma_gemm_nt_t p, int nslabs, int per_slab) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* Wsl = smem;                                             // [256 n][TW_LD], chunk-swizzled
  float* gb = reinterpret_cast<float*>(smem + 256 * TW_LD);         // gamma[256] | beta[256]
  float* gtab = gb + 512;                                           // Phi / dgelu table (see gt_lookup)
  const float* gtl = gtab - GT_A0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tok = lane & 15, g = lane >> 4;

  // Workgroup b runs on XCD b & 7.  Every XCD gets one contiguous run of virtual ids, and the n-slabs of one slot
  // (consecutive ids: they stream the SAME token rows, in the same order) therefore share an L2: at N = 768 / 1024 the
  // A rows otherwise cross the fabric once per slab.
  const int G = gridDim.x;
  const int b = (blockIdx.x & 7) * (G >> 3) + min((int)(blockIdx.x & 7), G & 7) + (blockIdx.x >> 3);
  const int slab_id = b % nslabs, slot = b / nslabs;  // slab_id = (batch, n-slab)
  const int slabs_n = (int)(p.N / 256);
  const int64_t bz = slab_id / slabs_n;
  const int64_t bn = (int64_t)(slab_id % slabs_n) * 256;

  const int64_t tiles = (p.M + 15) / 16;
  const int64_t gw = (int64_t)slot * NWAVES + wave, nw = (int64_t)per_slab * NWAVES;

  const char* Ab = reinterpret_cast<const char*>(p.A) + bz * p.sA * (AKIND == HMA_A_F32 ? 4 : 2);
  // A rows of one tile as MFMA B operands: a[j] = chunk 4 j + g of token row (tile * 16 + tok)
  auto load_a = [&](int64_t tile, bf16x8_t (&a)[8]) __attribute__((always_inline)) {
    int64_t m = tile * 16 + tok;
    m = m < p.M ? m : p.M - 1;
    const int64_t off = remap_row(m, p.a_group_rows, p.a_group_stride) * p.lda;
    if (AKIND == HMA_A_F32) {
      const float* row = reinterpret_cast<const float*>(Ab) + off;
      float4 lo[8], hi[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        lo[j] = *reinterpret_cast<const float4*>(row + (4 * j + g) * 8);
        hi[j] = *reinterpret_cast<const float4*>(row + (4 * j + g) * 8 + 4);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint4 v = make_uint4(pack_bf16(lo[j].x, lo[j].y), pack_bf16(lo[j].z, lo[j].w), pack_bf16(hi[j].x, hi[j].y),
                                   pack_bf16(hi[j].z, hi[j].w));
        a[j] = __builtin_bit_cast(bf16x8_t, v);
      }
    } else {
      const uint16_t* row = reinterpret_cast<const uint16_t*>(Ab) + off;
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(row + (4 * j + g) * 8));
    }
  };
  // the first tile's rows are requested before the slab: their latency passes under the fill
  bf16x8_t a[8], an[8];
  if (gw < tiles) load_a(gw, an);

  // park the weight slab by LDS-DMA (132 pieces of 1 KB = 256 rows x 33 slots; all of a wave's pieces in flight at once --
  // through registers the fill was 16 dependent rounds per thread, most of a small-M launch): slot L = 33 n + pos holds
  // chunk pos ^ (5 * (((n >> 3) ^ (n >> 4)) & 1)) of row n (bank spread, see below); pos = 32 is the row padding.
  {
    const uint16_t* Wb = reinterpret_cast<const uint16_t*>(p.W) + bz * p.sW + bn * p.ldw;
    const uint32_t lds_b = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(HMA_LDS(char)*)smem);
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    for (int piece = wv; piece < 132; piece += NWAVES) {
      const int L = piece * 64 + lane;
      const int n = L / 33, pos = L - 33 * n;
      const int ch = (pos < 32 ? pos : 0) ^ (5 * (((n >> 3) ^ (n >> 4)) & 1));
      glds16(Wb + (int64_t)n * p.ldw + ch * 8, lds_b + piece * 1024);
    }
    if (AKIND == HMA_A_BF16_AFFINE) {
      for (int c = tid; c < 256; c += 64 * NWAVES) { gb[c] = p.gamma[c]; gb[256 + c] = p.beta[c]; }
    }
    if (EPI == HMA_EPI_GELU2 || EPI == HMA_EPI_DGELU) {
      for (int c = tid; c < 2 * GT_N; c += 64 * NWAVES) {
        const uint32_t bits = (uint32_t)(GT_A0 + (c % GT_N)) | (c >= GT_N ? 0x8000u : 0u);
        const float u = __uint_as_float(bits << 16);
        const float cdf = 0.5f * (1.0f + erff(u * 0.70710678118654752f));
        gtab[c] = EPI == HMA_EPI_GELU2 ? cdf : cdf + u * 0.3989422804014327f * __expf(-0.5f * u * u);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // the only barrier

  if (gw >= tiles) return;

  auto affine_a = [&](bf16x8_t (&a)[8]) __attribute__((always_inline)) {
    if (AKIND == HMA_A_BF16_AFFINE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float f[8];
        unpack8(__builtin_bit_cast(uint4, a[j]), f);
        const float* gm = gb + (4 * j + g) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = f[e] * gm[e] + gm[256 + e];
        a[j] = __builtin_bit_cast(bf16x8_t, pack8(f));
      }
    }
  };

  // lane constants of the weight-fragment address: MFMA row i = lane & 15 of tile (p, o) is
  // n = 32 p + 8 (i >> 2) + (i & 3) + 4 o; logical chunk L = 4 j + g sits at L ^ 5 b, b = ((i >> 2) ^ (i >> 3)) & 1.
  // Bank check (row stride 528 B = 33 x 16-B slots, slot = (row + chunk) mod 16): ds_read_b128 is served in the lane
  // groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, {32-35, 44-47, 52-59}, {36-43, 48-51, 60-63} (NOT 16 consecutive
  // lanes); the first swizzle (4 j ^ 4 ((i >> 3) & 1), checked against consecutive groups) left one slot shared in every
  // group -- SQ_LDS_BANK_CONFLICT = SQ_LDS_IDX_ACTIVE / 2 on every launch; this one is exhaustively conflict-free.
  const int i16 = lane & 15;
  const int jsw = ((i16 >> 2) ^ (i16 >> 3)) & 1;
  const uint16_t* wbase = Wsl + (8 * (i16 >> 2) + (i16 & 3)) * TW_LD + (g ^ jsw) * 8;
  const float* bias = p.bias ? p.bias + bz * p.sBias + bn : nullptr;

  for (int64_t tile = gw; tile < tiles; tile += nw) {
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = an[j];
    {
      const int64_t nxt = tile + nw < tiles ? tile + nw : tile;  // clamped re-load on the last trip
      load_a(nxt, an);
    }
    affine_a(a);
    // epilogue operands (residual rows / saved pre-activation) are fetched BEFORE the 128 MFMAs that hide them
    const int64_t m = tile * 16 + tok;
    const int64_t crow = remap_row(m < p.M ? m : p.M - 1, p.c_group_rows, p.c_group_stride);
    float4 rx[EPI == HMA_EPI_RESID ? 16 : 1];
    uint4 ru[(EPI == HMA_EPI_DGELU || EPI == HMA_EPI_DSILU) ? 8 : 1];
    if (EPI == HMA_EPI_RESID) {
      const float* C = reinterpret_cast<const float*>(p.C) + bz * p.sC + crow * p.ldc + bn + 8 * g;
#pragma unroll
      for (int pr = 0; pr < 8; ++pr) {
        rx[2 * pr] = *reinterpret_cast<const float4*>(C + 32 * pr);
        rx[2 * pr + 1] = *reinterpret_cast<const float4*>(C + 32 * pr + 4);
      }
    }
    if (EPI == HMA_EPI_DGELU || EPI == HMA_EPI_DSILU) {
      const uint16_t* U = reinterpret_cast<const uint16_t*>(p.U) + bz * p.sU + crow * p.ldu + bn + 8 * g;
#pragma unroll
      for (int pr = 0; pr < 8; ++pr) ru[pr] = *reinterpret_cast<const uint4*>(U + 32 * pr);
    }
    f32x4v_t acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = f32x4v_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint16_t* wj = wbase + ((j ^ jsw) << 5);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const bf16x8_t wf = *reinterpret_cast<const bf16x8_t*>(wj + (32 * (t >> 1) + 4 * (t & 1)) * TW_LD);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, a[j], acc[t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the scheduler from hoisting all 128 fragment reads (it spills)
    }
    if (m < p.M) {
#pragma unroll
      for (int pr = 0; pr < 8; ++pr) {
        const int64_t nl = 32 * pr + 8 * g;  // this lane's 8 consecutive columns of the slab
        float v[8] = {acc[2 * pr][0], acc[2 * pr][1], acc[2 * pr][2], acc[2 * pr][3],
                      acc[2 * pr + 1][0], acc[2 * pr + 1][1], acc[2 * pr + 1][2], acc[2 * pr + 1][3]};
        if (bias) {
          const float4 b0 = *reinterpret_cast<const float4*>(bias + nl), b1 = *reinterpret_cast<const float4*>(bias + nl + 4);
          v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
          v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if (EPI == HMA_EPI_RESID) {
          float* C = reinterpret_cast<float*>(p.C) + bz * p.sC + crow * p.ldc + bn + nl;
          float4 x0 = rx[2 * pr], x1 = rx[2 * pr + 1];
          x0.x += v[0]; x0.y += v[1]; x0.z += v[2]; x0.w += v[3];
          x1.x += v[4]; x1.y += v[5]; x1.z += v[6]; x1.w += v[7];
          *reinterpret_cast<float4*>(C) = x0;
          *reinterpret_cast<float4*>(C + 4) = x1;
          if (p.C2) {
            const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.C2) + bz * p.sC2 + crow * p.ldc2 + bn + nl) = pack8(xs);
          }
          if (LNF) {  // kept for the fused LayerNorm below
            rx[2 * pr] = x0;
            rx[2 * pr + 1] = x1;
          }
        } else if (EPI == HMA_EPI_DGELU) {
          const uint32_t w[4] = {ru[pr].x, ru[pr].y, ru[pr].z, ru[pr].w};
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= gt_lookup(gtl, (e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu));
          if (p.drop_p > 0.f) {  // the forward's mask on gelu(u): same seed, salt and element index
            const uint32_t seed = *p.drop_seed, th = drop_thresh(p.drop_p);
            const float sc = drop_scale(p.drop_p);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = drop_keep(seed, p.drop_salt, crow * p.ldc + bn + nl + e, th) ? v[e] * sc : 0.f;
          }
          *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.C) + bz * p.sC + crow * p.ldc + bn + nl) = pack8(v);
        } else if (EPI == HMA_EPI_GELU2) {
          const uint4 uq = pack8(v);  // the saved pre-activation; the activation is applied to the ROUNDED value
          const uint32_t w[4] = {uq.x, uq.y, uq.z, uq.w};
          float h[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const uint32_t bits = (e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu);
            h[e] = __uint_as_float(bits << 16) * gt_lookup(gtl, bits);
          }
          if (p.drop_p > 0.f) {  // Dropout on the activation (st_transformer.py:25); u itself is saved undropped
            const uint32_t seed = *p.drop_seed, th = drop_thresh(p.drop_p);
            const float sc = drop_scale(p.drop_p);
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = drop_keep(seed, p.drop_salt, crow * p.ldc2 + bn + nl + e, th) ? h[e] * sc : 0.f;
          }
          *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.C) + bz * p.sC + crow * p.ldc + bn + nl) = uq;
          *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.C2) + bz * p.sC2 + crow * p.ldc2 + bn + nl) = pack8(h);
        } else if (EPI == HMA_EPI_DSILU) {
          float u[8];
          unpack8(ru[pr], u);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= dsilu_f(u[e]);
          *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.C) + bz * p.sC + crow * p.ldc + bn + nl) = pack8(v);
        } else {
          epilogue_run8<EPI>(p, bz, crow, bn + nl, v);
        }
      }
    }
    if constexpr (EPI == HMA_EPI_RESID && LNF) {  // (a separate instantiation: the plain residual epilogue stays lean)
      // Fused LayerNorm of the new residual row (N = 256: the 4 lanes tok, tok+16, tok+32, tok+48 hold the whole
      // row, 64 values each): same two-pass mean / variance as ln_fwd_kernel, reductions = 2 shuffles.
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) sum += rx[q].x + rx[q].y + rx[q].z + rx[q].w;
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float mean = sum * (1.0f / 256.0f);
      float sq = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        rx[q].x -= mean; rx[q].y -= mean; rx[q].z -= mean; rx[q].w -= mean;
        sq += rx[q].x * rx[q].x + rx[q].y * rx[q].y + rx[q].z * rx[q].z + rx[q].w * rx[q].w;
      }
      sq += __shfl_xor(sq, 16, 64);
      sq += __shfl_xor(sq, 32, 64);
      const float rstd = rsqrtf(sq * (1.0f / 256.0f) + p.ln_eps);
      if (m < p.M) {
        uint16_t* xh = reinterpret_cast<uint16_t*>(p.ln_xhat) + crow * 256 + 8 * g;
        const float* ssf = p.ln_ss ? p.ln_ss + (crow / p.ln_rows_per_frame) * 512 + 8 * g : nullptr;
        uint16_t* xm = p.ln_ss ? reinterpret_cast<uint16_t*>(p.ln_xm) + crow * 256 + 8 * g : nullptr;
#pragma unroll
        for (int pr = 0; pr < 8; ++pr) {
          float h[8] = {rx[2 * pr].x * rstd, rx[2 * pr].y * rstd, rx[2 * pr].z * rstd, rx[2 * pr].w * rstd,
                        rx[2 * pr + 1].x * rstd, rx[2 * pr + 1].y * rstd, rx[2 * pr + 1].z * rstd, rx[2 * pr + 1].w * rstd};
          *reinterpret_cast<uint4*>(xh + 32 * pr) = pack8(h);
          if (ssf) {
            const float4 sh0 = *reinterpret_cast<const float4*>(ssf + 32 * pr), sh1 = *reinterpret_cast<const float4*>(ssf + 32 * pr + 4);
            const float4 sc0 = *reinterpret_cast<const float4*>(ssf + 256 + 32 * pr), sc1 = *reinterpret_cast<const float4*>(ssf + 256 + 32 * pr + 4);
            float mm[8] = {h[0] * (1.f + sc0.x) + sh0.x, h[1] * (1.f + sc0.y) + sh0.y, h[2] * (1.f + sc0.z) + sh0.z, h[3] * (1.f + sc0.w) + sh0.w,
                           h[4] * (1.f + sc1.x) + sh1.x, h[5] * (1.f + sc1.y) + sh1.y, h[6] * (1.f + sc1.z) + sh1.z, h[7] * (1.f + sc1.w) + sh1.w};
            *reinterpret_cast<uint4*>(xm + 32 * pr) = pack8(mm);
          }
        }
        if (g == 0) p.ln_rstd[crow] = rstd;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ TN
// Stage a 64(m) x 128(col) slab of a row-major matrix TRANSPOSED into LDS as [128 col][64 m].
// Thread task: rows 4*mi..4*mi+3, columns w*32 + ci*8 .. +8 (mi = lane & 15, ci = lane >> 4):
// a 16-lane ds_write_b64 group then writes 16 consecutive 8-B slots of one LDS row (conflict-free).
template <int KIND>
struct SlabRegs {
  uint4 v[4][KIND == HMA_A_F32 ? 2 : 1];
};

template <int KIND>
__device__ __forceinline__ void slab_load(SlabRegs<KIND>& r, const char* base, int64_t ld, int64_t m0, int64_t m_end,
                                          int64_t group_rows, int64_t group_stride, int64_t col0, int lane, int wave) {
  const int mi = lane & 15, ci = lane >> 4;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int64_t gm = m0 + 4 * mi + rr;
    if (gm < m_end) {
      const int64_t off = remap_row(gm, group_rows, group_stride) * ld + col0 + wave * 32 + ci * 8;
      if (KIND == HMA_A_F32) {
        const float* s = reinterpret_cast<const float*>(base) + off;
        r.v[rr][0] = *reinterpret_cast<const uint4*>(s);
        r.v[rr][KIND == HMA_A_F32 ? 1 : 0] = *reinterpret_cast<const uint4*>(s + 4);
      } else {
        r.v[rr][0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(base) + off);
      }
    } else {
      r.v[rr][0] = make_uint4(0, 0, 0, 0);
      r.v[rr][KIND == HMA_A_F32 ? 1 : 0] = make_uint4(0, 0, 0, 0);
    }
  }
}

// converts to bf16 (applying the LN affine for HMA_A_BF16_AFFINE), optionally accumulates the
// column sums of the fp32 values (bias gradient), transposes 4x8 in registers and writes LDS.
template <int KIND, bool COLSUM>
__device__ __forceinline__ void slab_store(const SlabRegs<KIND>& r, uint16_t* T, int lane, int wave, const float* gm,
                                           const float* bt, float* colsum, int64_t m0, int64_t m_end) {
  const int mi = lane & 15, ci = lane >> 4;
  uint4 q[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    float f[8];
    if (KIND == HMA_A_F32) {
      const float4 lo = __builtin_bit_cast(float4, r.v[rr][0]);
      const float4 hi = __builtin_bit_cast(float4, r.v[rr][KIND == HMA_A_F32 ? 1 : 0]);
      f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w;
      f[4] = hi.x; f[5] = hi.y; f[6] = hi.z; f[7] = hi.w;
    } else {
      unpack8(r.v[rr][0], f);
      if (KIND == HMA_A_BF16_AFFINE) {
        if (m0 + 4 * mi + rr < m_end) {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = f[j] * gm[j] + bt[j];
        }
      }
    }
    if (COLSUM) {
#pragma unroll
      for (int j = 0; j < 8; ++j) colsum[j] += f[j];
    }
    q[rr] = pack8(f);
  }
  const uint32_t* d0 = reinterpret_cast<const uint32_t*>(&q[0]);
  const uint32_t* d1 = reinterpret_cast<const uint32_t*>(&q[1]);
  const uint32_t* d2 = reinterpret_cast<const uint32_t*>(&q[2]);
  const uint32_t* d3 = reinterpret_cast<const uint32_t*>(&q[3]);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int w = j >> 1;
    uint32_t lo, hi;
    if (j & 1) {
      lo = (d0[w] >> 16) | (d1[w] & 0xffff0000u);
      hi = (d2[w] >> 16) | (d3[w] & 0xffff0000u);
    } else {
      lo = (d0[w] & 0xffffu) | (d1[w] << 16);
      hi = (d2[w] & 0xffffu) | (d3[w] << 16);
    }
    *reinterpret_cast<uint2*>(&T[(wave * 32 + ci * 8 + j) * LDT + 4 * mi]) = make_uint2(lo, hi);
  }
}

template <int YKIND, int AKIND>
__global__ __launch_bounds__(256) void gemm_tn_kernel(hma_gemm_tn_t p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* At = smem;                   // [2][128 k][LDT]  (token-side operand of mma_tile)
  uint16_t* Yt = smem + 2 * TILE_ELEMS;  // [2][128 n][LDT]  (weight-side operand of mma_tile)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_k = (int)(p.K / 128);
  const int tile_n = blockIdx.y / tiles_k, tile_k = blockIdx.y % tiles_k;
  const int64_t n0 = (int64_t)tile_n * 128, k0 = (int64_t)tile_k * 128;
  const int64_t bz = blockIdx.z;

  // M range of this split, rounded to whole 64-row slabs
  const int64_t slabs = (p.M + 63) / 64;
  const int64_t per = (slabs + p.splits - 1) / p.splits;
  const int64_t m_begin = (int64_t)blockIdx.x * per * 64;
  int64_t m_end = m_begin + per * 64;
  if (m_end > p.M) m_end = p.M;
  if (m_begin >= m_end) return;

  const char* Yb = reinterpret_cast<const char*>(p.dY) + bz * p.sY * (YKIND == HMA_A_F32 ? 4 : 2);
  const char* Ab = reinterpret_cast<const char*>(p.A) + bz * p.sA * (AKIND == HMA_A_F32 ? 4 : 2);

  const int ci = lane >> 4;
  float gm[8], bt[8];
  if (AKIND == HMA_A_BF16_AFFINE) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      gm[j] = p.gamma[k0 + wave * 32 + ci * 8 + j];
      bt[j] = p.beta[k0 + wave * 32 + ci * 8 + j];
    }
  }
  float colsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) colsum[j] = 0.f;
  const bool do_bias = (p.dBias != nullptr) && (tile_k == 0);

  SlabRegs<YKIND> ry;
  SlabRegs<AKIND> ra;
  f32x16_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  auto load = [&](int64_t m0) {
    slab_load<YKIND>(ry, Yb, p.ldy, m0, m_end, p.y_group_rows, p.y_group_stride, n0, lane, wave);
    slab_load<AKIND>(ra, Ab, p.lda, m0, m_end, p.a_group_rows, p.a_group_stride, k0, lane, wave);
  };
  auto store = [&](int buf, int64_t m0) {
    if (do_bias)
      slab_store<YKIND, true>(ry, Yt + buf * TILE_ELEMS, lane, wave, nullptr, nullptr, colsum, m0, m_end);
    else
      slab_store<YKIND, false>(ry, Yt + buf * TILE_ELEMS, lane, wave, nullptr, nullptr, colsum, m0, m_end);
    slab_store<AKIND, false>(ra, At + buf * TILE_ELEMS, lane, wave, gm, bt, nullptr, m0, m_end);
  };

  const int iters = (int)((m_end - m_begin + 63) / 64);
  load(m_begin);
  store(0, m_begin);
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    const int cur = it & 1;
    const int64_t m_next = m_begin + (int64_t)(it + 1) * 64;
    if (it + 1 < iters) load(m_next);
    mma_tile(At + cur * TILE_ELEMS, Yt + cur * TILE_ELEMS, acc, wm, wn, lane);
    if (it + 1 < iters) store(cur ^ 1, m_next);
    __syncthreads();
  }

  // D rows = n (weight-side rows of Yt), D cols = k (token-side rows of At)
  const int r = lane & 31, hi = lane >> 5;
  float* dW = p.dW + bz * p.sdW;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int64_t k = k0 + wm * 64 + mt * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t n = n0 + wn * 64 + nt * 32 + mfma32_row(e, hi);
        atomicAdd(dW + n * p.lddw + k, acc[nt][mt][e]);
      }
    }
  if (do_bias) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float s = colsum[j];
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      s += __shfl_xor(s, 8, 64);
      if ((lane & 15) == 0) atomicAdd(p.dBias + bz * p.sdBias + n0 + wave * 32 + ci * 8 + j, s);
    }
  }
}

// ------------------------------------------------------------------------------- TN, wide tile
// Weight gradients contract over M = B*T*(S+A) = 163840 rows but produce only N x K <= 1024 x 256
// outputs: the cost is streaming dY and the activations.  A 512-thread workgroup owns a FULL
// 256 (n) x 256 (k) block of dW for its slice of M, so for the d_model-sized layers every byte of dY
// and A is read exactly once; each wave keeps a 128 x 64 corner (8 accumulators, 128 VGPRs).
constexpr int WT = 256;                       // n-group = k-group = 256
constexpr int W_TILE = WT * LDT;              // one transposed operand slab [256][64 + 8]
constexpr int W_SMEM_BYTES = 4 * W_TILE * 2;  // 2 operands x 2 buffers = 147456 B

template <int YKIND, int AKIND>
__global__ __launch_bounds__(512, 2) void gemm_tn_wide_kernel(hma_gemm_tn_t p, int groups_n, int groups_k) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* At = smem;               // [2][256 k][LDT]
  uint16_t* Yt = smem + 2 * W_TILE;  // [2][256 n][LDT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn2 = wave >> 2, wk4 = wave & 3;

  const int G = gridDim.x;
  const int b = blockIdx.x;
  const int vid = ((G & 7) == 0) ? (b & 7) * (G >> 3) + (b >> 3) : b;
  const int groups = groups_n * groups_k;
  const int per_batch = groups * p.splits;
  const int64_t bz = vid / per_batch;
  const int r0 = vid % per_batch;
  const int split = r0 / groups, g = r0 % groups;
  const int64_t n0 = (int64_t)(g / groups_k) * WT, k0 = (int64_t)(g % groups_k) * WT;

  const int64_t slabs = (p.M + 63) / 64;
  const int64_t per = (slabs + p.splits - 1) / p.splits;
  const int64_t m_begin = (int64_t)split * per * 64;
  int64_t m_end = m_begin + per * 64;
  if (m_end > p.M) m_end = p.M;
  if (m_begin >= m_end) return;

  const char* Yb = reinterpret_cast<const char*>(p.dY) + bz * p.sY * (YKIND == HMA_A_F32 ? 4 : 2);
  const char* Ab = reinterpret_cast<const char*>(p.A) + bz * p.sA * (AKIND == HMA_A_F32 ? 4 : 2);
  const int ci = lane >> 4;
  float gm[8], bt[8];
  if (AKIND == HMA_A_BF16_AFFINE) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      gm[j] = p.gamma[k0 + wave * 32 + ci * 8 + j];
      bt[j] = p.beta[k0 + wave * 32 + ci * 8 + j];
    }
  }
  float colsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) colsum[j] = 0.f;
  const bool do_bias = (p.dBias != nullptr) && (k0 == 0);

  // Register stages of the operand prefetch.  Ablation (HMA_GEMM_TN_ABLATE, tools/gemm_bench.py) shows the loads,
  // the transposing LDS writes, the MFMAs and the epilogue of this one-stage loop adding up almost serially; a
  // second stage (loads of slab it + 2 in flight during slab it) was measured and is WORSE: 128 accumulator +
  // 64 staging registers spill inside the loop (bf16 dY: 110 -> 169 us, affine: 124 -> 351 us in situ).
  constexpr int NSETS = 1;
  SlabRegs<YKIND> ry[NSETS];
  SlabRegs<AKIND> ra[NSETS];
  f32x16_t acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.f;

  auto load = [&](int set, int64_t m0) __attribute__((always_inline)) {
    slab_load<YKIND>(ry[set], Yb, p.ldy, m0, m_end, p.y_group_rows, p.y_group_stride, n0, lane, wave);
    slab_load<AKIND>(ra[set], Ab, p.lda, m0, m_end, p.a_group_rows, p.a_group_stride, k0, lane, wave);
  };
  auto store = [&](int set, int buf, int64_t m0) __attribute__((always_inline)) {
    if (do_bias)
      slab_store<YKIND, true>(ry[set], Yt + buf * W_TILE, lane, wave, nullptr, nullptr, colsum, m0, m_end);
    else
      slab_store<YKIND, false>(ry[set], Yt + buf * W_TILE, lane, wave, nullptr, nullptr, colsum, m0, m_end);
    slab_store<AKIND, false>(ra[set], At + buf * W_TILE, lane, wave, gm, bt, nullptr, m0, m_end);
  };

  const int iters = (int)((m_end - m_begin + 63) / 64);
  const int r = lane & 31, hi = lane >> 5;
  PROF_DECL;
#ifdef HMA_PROF
  const int ablate = p._pad0;  // debug build only (HMA_GEMM_TN_ABLATE): 1 = skip MFMA, 2 = skip LDS stores, 4 = skip loads, 8 = skip epilogue
#else
  constexpr int ablate = 0;
#endif
  auto mma = [&](int cur) __attribute__((always_inline)) {
    const uint16_t* Ys = Yt + cur * W_TILE;
    const uint16_t* As = At + cur * W_TILE;
    if (!(ablate & 1))
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8_t yf[4], af[2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        yf[i] = *reinterpret_cast<const bf16x8_t*>(&Ys[(wn2 * 128 + i * 32 + r) * LDT + kk * 16 + hi * 8]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        af[j] = *reinterpret_cast<const bf16x8_t*>(&As[(wk4 * 64 + j * 32 + r) * LDT + kk * 16 + hi * 8]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(af[j], yf[i], acc[i][j]);  // D rows = k, D cols = n
    }
  };
  load(0, m_begin);
  store(0, 0, m_begin);
  if (NSETS == 2 && iters > 1) load(NSETS - 1, m_begin + 64);
  __syncthreads();
  PROF_MARK(0);
  if constexpr (NSETS == 2) {
    // step(it, set_free, set_next): set_free held slab it (already in LDS) -> refill with slab it + 2;
    // set_next holds slab it + 1 (loaded during step it - 1) -> transposed into the other LDS buffer.
    auto step = [&](int it, int set_free, int set_next) __attribute__((always_inline)) {
      if (it + 2 < iters && !(ablate & 4)) load(set_free, m_begin + (int64_t)(it + 2) * 64);
      PROF_MARK(1);
      mma(it & 1);
      PROF_MARK(2);
      if (it + 1 < iters && !(ablate & 2)) store(set_next, (it & 1) ^ 1, m_begin + (int64_t)(it + 1) * 64);
      PROF_MARK(4);
      __syncthreads();
      PROF_MARK(5);
    };
    for (int it = 0; it < iters; it += 2) {
      step(it, 0, NSETS - 1);
      if (it + 1 < iters) step(it + 1, NSETS - 1, 0);
    }
  } else {
    for (int it = 0; it < iters; ++it) {
      const int cur = it & 1;
      const int64_t m_next = m_begin + (int64_t)(it + 1) * 64;
      if (it + 1 < iters && !(ablate & 4)) load(0, m_next);
      PROF_MARK(1);
      mma(cur);
      PROF_MARK(2);
      if (it + 1 < iters && !(ablate & 2)) store(0, cur ^ 1, m_next);
      PROF_MARK(4);
      __syncthreads();
      PROF_MARK(5);
    }
  }

  // D rows = k, D cols = n: a lane owns one n and, per accumulator, four runs of 4 consecutive k -> 16-byte
  // stores (the phase timers showed 40-70 % of this kernel's wave time in an epilogue of 4-byte stores).
  const bool bias_ws = p.ws && p.ws_elems >= (int64_t)gridDim.x * (WT * WT + WT);
  if (ablate & 8) {
  } else if (p.ws) {
    // two-stage reduction: plain stores of this workgroup's 256 x 256 partial; tn_reduce_kernel sums the
    // splits.  (Device-scope fp32 atomics from 256 workgroups onto the same 64 K addresses cost more than
    // the whole main loop for the d_model-sized layers.)
    // Partials are bf16 (the fp32 sum over splits is formed by tn_reduce_kernel): the reference's autocast
    // backward returns the whole weight gradient in bf16, and the partial round trip (write + re-read of
    // 256 x 64 K values per call) is otherwise half as many bytes again as the operands themselves.
    uint16_t* part = reinterpret_cast<uint16_t*>(p.ws) + (int64_t)vid * (WT * WT);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        uint16_t* row = part + (wn2 * 128 + i * 32 + r) * WT + wk4 * 64 + j * 32;
#pragma unroll
        for (int q = 0; q < 4; q += 2) {
          const float v0[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
          const float v1[4] = {acc[i][j][4 * q + 4], acc[i][j][4 * q + 5], acc[i][j][4 * q + 6], acc[i][j][4 * q + 7]};
          store_bf16_oct(row, q, hi, v0, v1);
        }
      }
  } else {
    // Rotate the tile order by the split index so concurrent workgroups hit different addresses with
    // their atomics.
    float* dW = p.dW + bz * p.sdW;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int qq = (q + split) & 7;
      const int i = qq >> 1, j = qq & 1;
      const int64_t n = n0 + wn2 * 128 + i * 32 + r;
      f32x16_t v;
      // static indexing of the accumulator array (a dynamic index would spill it to scratch)
      switch (qq) {
        case 0: v = acc[0][0]; break; case 1: v = acc[0][1]; break;
        case 2: v = acc[1][0]; break; case 3: v = acc[1][1]; break;
        case 4: v = acc[2][0]; break; case 5: v = acc[2][1]; break;
        case 6: v = acc[3][0]; break; default: v = acc[3][1]; break;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t kk = k0 + wk4 * 64 + j * 32 + mfma32_row(e, hi);
        atomicAdd(dW + n * p.lddw + kk, v[e]);
      }
    }
  }
  if (do_bias) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float sj = colsum[j];
      sj += __shfl_xor(sj, 1, 64);
      sj += __shfl_xor(sj, 2, 64);
      sj += __shfl_xor(sj, 4, 64);
      sj += __shfl_xor(sj, 8, 64);
      if ((lane & 15) == 0) {
        if (bias_ws)  // per-workgroup partial, summed by tn_reduce_kernel (no contended atomics)
          p.ws[(int64_t)gridDim.x * (WT * WT) + (int64_t)vid * WT + wave * 32 + ci * 8 + j] = sj;
        else
          atomicAdd(p.dBias + bz * p.sdBias + n0 + wave * 32 + ci * 8 + j, sj);
      }
    }
  }
  PROF_MARK(3);
  PROF_FLUSH();
}

// ------------------------------------------------------------------------------- TN, LDS-DMA ring
// bf16 x bf16 weight gradient whose operand tiles never pass through registers: `global_load_lds` (16 bytes per
// lane, 1 KB per wave instruction) fills a ring of four 32-token stages ([32][256] bf16 of dY, then of A: 32 KB),
// three stages in flight ahead of the MFMAs across raw s_barriers with counted vmcnt waits.  The register-staged
// kernel above keeps one 64-token stage (64 KB per CU) in flight and exposes most of the HBM latency each iteration
// (its time did not move when the staging transposes were replaced by row-major LDS tiles + 2-byte gathers).  The MFMA fragments
// (8 tokens of one column per lane) come out of the row-major image with ds_read_b64_tr_b16 (TR) or eight 2-byte
// reads (!TR, the reference gather).  LDS image of a tile: row r, 16-byte chunk c lives at chunk c ^ ((r & 3) << 2)
// (the DMA writes lane-linear, so the permutation is applied to each lane's SOURCE address and again on the read):
// the four rows of a transposing read then fall in four different 64-byte bank groups.
// The LayerNorm affine of A is not applied here: tn_reduce_kernel finishes dW = gamma * (dY^T xhat) + colsum(dY) beta^T.
constexpr int DM_ROWS = 32;
constexpr int DM_TILE_BYTES = DM_ROWS * 512;
constexpr int DM_STAGE_BYTES = 2 * DM_TILE_BYTES;
#ifndef DM_STAGES_N
#define DM_STAGES_N 4
#endif
#ifndef TN_PHASE
#define TN_PHASE 1
#endif
#ifndef TN_INTERLEAVE
#define TN_INTERLEAVE 0
#endif
#ifndef TN_SPREAD
#define TN_SPREAD 1
#endif
constexpr int DM_STAGES = DM_STAGES_N;  // 4 x 32 KB (5 slots, the whole LDS, measured the same)
constexpr int DM_SMEM_BYTES = DM_STAGES * DM_STAGE_BYTES;

typedef short v4s16_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int dm_off(int row, int col) {  // byte offset of element (row, col) in a swizzled tile
  return row * 512 + ((((col >> 3) ^ ((row & 3) << 2))) << 4) + ((col & 7) << 1);
}

// ABL (debug build only): 1 no DMA, 2 no MFMA, 4 no LDS reads, 8 no partial stores, 16 no barrier
// One launch serves up to TWO weight-gradient problems (hma_gemm_tn_pair): the 256 workgroups are divided between
// them, so each is split over fewer M-slices and the fixed cost of a split-M wgrad -- 128 KB of partials per
// workgroup, written and re-read -- is paid once per launch instead of once per problem.
struct tn_prob {
  const void* dY;
  const void* A;
  float* ws;         // this problem's partials (bf16 blocks, then fp32 column-sum partials)
  int64_t M, ldy, lda, sY, sA;
  int splits, gn, gk;
  int nb;            // workgroups of this problem
  int colsum;        // write the column-sum partials
  int yfrag, afrag;  // operand stored in the fused-MLP fragment order (HMA_A_BF16_FRAG32): only the DMA source address differs
  int yhb, ahb;      // > 0: operand in the head-blocked order of the spatial attention (HMA_A_BF16_HEADBLK), rows per frame
};
constexpr int TN_MAXP = 16;  // (round 6: the weight gradients of TWO blocks in one launch)
struct tn_pair_args {
  tn_prob q[TN_MAXP];  // problem i's workgroup ids follow problem i - 1's; n problems (hma_gemm_tn: 1, _pair: 2, _multi: up to 8)
  int n;
};

template <bool TR, bool COLSUM, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm_tn_dma_kernel(tn_pair_args args) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  HMA_LDS(char)* lds = (HMA_LDS(char)*)smem;
  const uint32_t lds_b = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn2 = wave >> 2, wk4 = wave & 3;

  const int G = gridDim.x;
  const int b = blockIdx.x;
  // workgroup b runs on XCD b & 7: give every XCD one contiguous run of virtual ids, so that the workgroups of one split
  // (consecutive ids: the n / k groups that re-read the same rows of A or dY) share an L2 -- for any G
  const int gvid = (b & 7) * (G >> 3) + min(b & 7, G & 7) + (b >> 3);
  int pi = 0, vbase = 0;
#pragma unroll 1
  while (pi + 1 < args.n && gvid >= vbase + args.q[pi].nb) {
    vbase += args.q[pi].nb;
    ++pi;
  }
  const tn_prob p = args.q[pi];
  const int vid = gvid - vbase;
  const int groups_n = p.gn, groups_k = p.gk;
  const int groups = groups_n * groups_k;
  const int per_batch = groups * p.splits;
  const int64_t bz = vid / per_batch;
  const int r0 = vid % per_batch;
  const int split = r0 / groups, g = r0 % groups;
  const int64_t n0 = (int64_t)(g / groups_k) * WT, k0 = (int64_t)(g % groups_k) * WT;

  const int64_t slabs = (p.M + 63) / 64;
  const int64_t per = (slabs + p.splits - 1) / p.splits;
  const int64_t m_begin = (int64_t)split * per * 64;
  int64_t m_end = m_begin + per * 64;
  if (m_end > p.M) m_end = p.M;
  if (m_begin >= m_end) return;
#if TN_INTERLEAVE
  // split s takes the 32-row stages s, s + splits, s + 2 splits, ...: at any moment the workgroups of a problem read ONE
  // moving window of splits x 32 consecutive rows (whole DRAM pages, every channel) instead of `splits` streams that sit
  // a fixed power-of-two-ish distance apart (163 840 rows / 32 splits = 10 MB of a 1024-wide operand)
  const int64_t nstages = p.M / DM_ROWS;
  const int nst = (int)((nstages - split + p.splits - 1) / p.splits);
  const int64_t st_row0 = (int64_t)split * DM_ROWS, st_pitch = (int64_t)p.splits * DM_ROWS;
#else
  const int nst = (int)((m_end - m_begin) / DM_ROWS);  // host guarantees M % 32 == 0
  const int64_t st_row0 = m_begin, st_pitch = DM_ROWS;
#endif

  const uint16_t* Yb = reinterpret_cast<const uint16_t*>(p.dY) + bz * p.sY + n0;
  const uint16_t* Ab = reinterpret_cast<const uint16_t*>(p.A) + bz * p.sA + k0;
  float colsum = 0.f;
  constexpr int ablate = ABL;

  f32x16_t acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.f;

  // A piece's address = wave-uniform base (stage, row pair) + this lane's byte offset, which is the same for every stage: the lane parts
  // of the three operand orders are computed ONCE here (two row pairs q x two operands), the uniform parts per piece in scalar registers
  // (round 6: the per-piece 64-bit vector arithmetic of the fragment / head-blocked orders was ~170 cycles of issue per piece -- a wave
  // spent a third of a stage computing addresses).
  //   row-major      element (m, c) at m ld + c
  //   HMA_A_BF16_FRAG32: (128-row tile, 32-column block, 32-row group) -> 2 KB = [columns 8..15 / 24..31 ? 1 : 0][column >= 16][row][8 columns]
  //                  -- what hma_mlp_bwd's producers store with one contiguous 1 KB per wave instruction
  //   HMA_A_BF16_HEADBLK: element (m, c) of a [M, 256 W] matrix at (((frame 8 + head) W + c / 256) n + m % n) 32 + c % 32, frame = m / n,
  //                  head = (c % 256) / 32 -- what the spatial attention backward writes as whole contiguous 2 KB tiles (a stage's 32
  //                  rows lie inside one frame)
  // with m = mu + (lane >> 5), mu = the piece's first row (even: m >> 5 = mu >> 5), c = c0 + 8 lc, c0 a multiple of 256.
  uint32_t loff_y[2], loff_a[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const uint32_t hi5 = (uint32_t)lane >> 5;
    const uint32_t rl = q * 2 + hi5;
    const uint32_t lc = ((uint32_t)lane & 31u) ^ (rl << 2);  // logical chunk that lands in this lane's slot
    auto lane_off = [&](int frag, int hb, int64_t ld) __attribute__((always_inline)) -> uint32_t {
      if (frag) return 2u * (((lc >> 2) << 12) + ((lc & 1u) << 9) + (((lc >> 1) & 1u) << 8) + (hi5 << 3));
      if (hb > 0) return 2u * ((((lc >> 2) * (uint32_t)(ld >> 8) * (uint32_t)hb + hi5) << 5) + (lc & 3u) * 8u);
      return 2u * (hi5 * (uint32_t)ld + lc * 8u);
    };
    loff_y[q] = lane_off(p.yfrag, p.yhb, p.ldy);
    loff_a[q] = lane_off(p.afrag, p.ahb, p.lda);
  }
  // wave w fills rows 4 w .. 4 w + 3 of both tiles of a stage: two 1 KB pieces (two rows each) per tile
  // parts: bit mask of the stage's four pieces of this wave (bit 2 q + {0: dY, 1: A}); 15 = all four at once
  auto issue_parts = [&](int st, int slot, int parts, int wv = -1) __attribute__((always_inline)) {
    if (ablate & 1) return;
    if (wv < 0) wv = wave;
    const int64_t m0 = st_row0 + (int64_t)st * st_pitch;
    const uint32_t slot_b = lds_b + slot * DM_STAGE_BYTES;
    auto ubase = [&](const void* mat, const uint16_t* rowmajor, int frag, int hb, int64_t ld, int64_t c0, int64_t mu) __attribute__((always_inline)) {
      const uint16_t* base = reinterpret_cast<const uint16_t*>(mat);
      if (frag) return base + (((((mu >> 7) * (ld >> 5) + (c0 >> 5)) << 2) + ((mu >> 5) & 3)) << 10) + ((mu & 31) << 3);
      if (hb > 0) {
        const uint32_t fr = (uint32_t)m0 / (uint32_t)hb;  // (ONE 32-bit division per stage, by the wave-uniform first row)
        return base + ((((int64_t)fr * 8 * (ld >> 8) + (c0 >> 8)) * hb + (mu - (int64_t)fr * hb)) << 5);
      }
      return rowmajor + mu * ld;  // (host: no row groups on this path; `rowmajor` already carries the batch offset and c0)
    };
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (!((parts >> (2 * q)) & 3)) continue;
      const int64_t mu = m0 + wv * 4 + q * 2;
      if ((parts >> (2 * q)) & 1) glds16s(ubase(p.dY, Yb, p.yfrag, p.yhb, p.ldy, n0, mu), loff_y[q], slot_b + (wv * 4 + q * 2) * 512);
      if ((parts >> (2 * q)) & 2) glds16s(ubase(p.A, Ab, p.afrag, p.ahb, p.lda, k0, mu), loff_a[q], slot_b + DM_TILE_BYTES + (wv * 4 + q * 2) * 512);
    }
  };
  auto issue = [&](int st, int slot) __attribute__((always_inline)) { issue_parts(st, slot, 15); };

  const int r = lane & 31, hi = lane >> 5;
  const int i16 = lane & 15, g16 = lane >> 4;
  // fragment: tokens 16 kk + 8 hi .. + 7 of column c0 + r of the tile at `tile`
  auto frag = [&](HMA_LDS(char)* tile, int c0, int kk) __attribute__((always_inline)) {
    if (ablate & 4) return __builtin_bit_cast(bf16x8_t, make_uint4(c0, kk, c0, kk));
    if (TR) {
      // a 16-lane group reads a [4 token][16 column] block: lane i supplies row i >> 2, columns 4 (i & 3) .. + 3 and
      // receives column i of the block's 4 rows (tools/probes/tr_read_dma.hip)
      const int col = c0 + 16 * (g16 & 1) + 4 * (i16 & 3);
      const int row = kk * 16 + 8 * hi + (i16 >> 2);
      const v4s16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((HMA_LDS(v4s16_t)*)(tile + dm_off(row, col)));
      const v4s16_t hv = __builtin_amdgcn_ds_read_tr16_b64_v4i16((HMA_LDS(v4s16_t)*)(tile + dm_off(row + 4, col)));
      typedef short v8s16_t __attribute__((ext_vector_type(8)));
      const v8s16_t v = __builtin_shufflevector(lo, hv, 0, 1, 2, 3, 4, 5, 6, 7);
      return __builtin_bit_cast(bf16x8_t, v);
    } else {
      uint32_t w[4];
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const uint32_t x0 = *(HMA_LDS(uint16_t)*)(tile + dm_off(kk * 16 + 8 * hi + e, c0 + r));
        const uint32_t x1 = *(HMA_LDS(uint16_t)*)(tile + dm_off(kk * 16 + 8 * hi + e + 1, c0 + r));
        w[e >> 1] = x0 | (x1 << 16);
      }
      return __builtin_bit_cast(bf16x8_t, make_uint4(w[0], w[1], w[2], w[3]));
    }
  };

  struct Frags {
    bf16x8_t y[4], a[2];
  };
  auto read_frags = [&](Frags& f, int slot, int kk) __attribute__((always_inline)) {
    HMA_LDS(char)* Ys = lds + slot * DM_STAGE_BYTES;
    HMA_LDS(char)* As = Ys + DM_TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) f.y[i] = frag(Ys, wn2 * 128 + ((i + wk4) & 3) * 32, kk);  // y[i] = n-block (i + wk4) & 3
#pragma unroll
    for (int j = 0; j < 2; ++j) f.a[j] = frag(As, wk4 * 64 + j * 32, kk);
  };
  // TN_SPREAD: the refill's four pieces are issued ONE AT A TIME between the MFMAs of a stage (piece k behind MFMA 4 k + 3 of the
  // sixteen) instead of as a burst between the two groups: a wave blocks at the issue of a vector-memory instruction while the CU's
  // memory pipeline is full (the kernel is HBM-bound: that is its normal state), and a burst of four makes the wave sit out ~1 000
  // cycles per stage with the matrix pipe idle behind its last MFMA
  auto mma_spread = [&](const Frags& f, int half, int st_next, int slot_next, bool refill) __attribute__((always_inline)) {
    if (COLSUM) {  // (as in mma below)
      const uint4 ys = __builtin_bit_cast(uint4, f.y[0]);
      const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3F803F80u);
      colsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, ys.x), ones, colsum, false);
      colsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, ys.y), ones, colsum, false);
      colsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, ys.z), ones, colsum, false);
      colsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, ys.w), ones, colsum, false);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (ablate & 2)
          acc[i][j][0] += __builtin_bit_cast(float, __builtin_bit_cast(uint4, f.a[j]).x ^ __builtin_bit_cast(uint4, f.y[i]).y);
        else
          acc[i][j] = mfma32(f.a[j], f.y[i], acc[i][j]);
      }
#if TN_SPREAD == 2  // (measurement: the waves of a half issue at different MFMA positions -- wave wk4 behind group (i + wk4) & 3)
      if (((i + wk4) & 1) == 1) {
        __builtin_amdgcn_sched_barrier(0);
        if (refill) issue_parts(st_next, slot_next, 1 << (2 * half + (((i + wk4) & 3) >> 1)));
        __builtin_amdgcn_sched_barrier(0);
      }
#else
      if ((i & 1) == 1) {
        __builtin_amdgcn_sched_barrier(0);  // (the piece stays where it is written: between the MFMA groups)
        if (refill) issue_parts(st_next, slot_next, 1 << (2 * half + (i >> 1)));
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    }
  };
  auto mma = [&](const Frags& f) __attribute__((always_inline)) {
    if (COLSUM) {
      // the four waves of a wn2 half hold the same dY fragments: wave wk4 sums the n-block wk4, which the rotated fragment
      // order (read_frags) puts in y[0] -- four v_dot2_f32_bf16 against (1, 1) per fragment (the select + unpack + add
      // version was 28 VALU operations per eight MFMAs: 10 % of the MLP pair's time)
      const uint4 ys = __builtin_bit_cast(uint4, f.y[0]);
      const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3F803F80u);
      colsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, ys.x), ones, colsum, false);
      colsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, ys.y), ones, colsum, false);
      colsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, ys.z), ones, colsum, false);
      colsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, ys.w), ones, colsum, false);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (ablate & 2)
          acc[i][j][0] += __builtin_bit_cast(float, __builtin_bit_cast(uint4, f.a[j]).x ^ __builtin_bit_cast(uint4, f.y[i]).y);
        else
          acc[i][j] = mfma32(f.a[j], f.y[i], acc[i][j]);  // D rows = k, D cols = n
      }
  };
  // wait until this wave's pieces of stage st have landed: the pieces of the later stages issued so far (4 per stage, at
  // most `behind` stages) may stay outstanding
  auto wait_stage = [&](int st, int behind) __attribute__((always_inline)) {
    int later = nst - 1 - st;
    later = later < behind ? later : behind;
    if (later >= 3)
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (later == 2)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (later == 1)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // (A software-pipelined variant -- fragments read half a stage ahead, barrier in the middle of the stage -- measured
  // 10 % slower: it needs stage st + 1 landed half a stage earlier, and the DMA depth is what this loop lives on.)
#ifdef TN_ISSUER_PROBE
  // MEASUREMENT ONLY (results wrong): wave 7 issues EVERY piece of every stage and multiplies nothing; waves 0..6 never touch the
  // vector-memory pipe inside the loop.  What a ring kernel with one issuing wave could reach (its eighth wave's output tile is
  // simply left out here; the real organisation deals the 64 output blocks over seven waves).
  static_assert(DM_STAGES == 3, "probe: 32 pieces per stage, the wait counter holds 63");
  if (wave == 7) {
    for (int st = 0; st < DM_STAGES - 1; ++st)
      if (st < nst)
        for (int wv = 0; wv < 8; ++wv) issue_parts(st, st, 15, wv);
    int slot_i = 0;
    for (int st = 0; st < nst; ++st) {
      if (st + 1 < nst) asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (st + DM_STAGES - 1 < nst)
        for (int wv = 0; wv < 8; ++wv) issue_parts(st + DM_STAGES - 1, slot_i == 0 ? DM_STAGES - 1 : slot_i - 1, 15, wv);
      slot_i = slot_i + 1 == DM_STAGES ? 0 : slot_i + 1;
    }
    return;
  }
  {
    Frags f0, f1;
    int slot = 0;
    if (wn2 == 0) {
      for (int st = 0; st < nst; ++st) {
        __builtin_amdgcn_s_barrier();
        read_frags(f0, slot, 0);
        read_frags(f1, slot, 1);
        mma(f0);
        mma(f1);
        slot = slot + 1 == DM_STAGES ? 0 : slot + 1;
      }
    } else {
      __builtin_amdgcn_s_barrier();
      read_frags(f0, 0, 0);
      read_frags(f1, 0, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      slot = 1;
      for (int st = 1; st < nst; ++st) {
        __builtin_amdgcn_s_barrier();
        mma(f0);
        mma(f1);
        read_frags(f0, slot, 0);
        read_frags(f1, slot, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        slot = slot + 1 == DM_STAGES ? 0 : slot + 1;
      }
      mma(f0);
      mma(f1);
    }
  }
#else
#pragma unroll
  for (int st = 0; st < DM_STAGES - 1; ++st)
    if (st < nst) issue(st, st);
  Frags f0, f1;
  int slot = 0;
#if TN_PHASE
  // The two waves of a SIMD (w and w + 4: the two wn2 halves) run a stage in opposite order: the first reads its fragments and
  // then multiplies; the second multiplies the PREVIOUS stage's fragments (still in its registers) and then reads this
  // stage's -- so one wave's LDS reads sit under the other's MFMAs instead of all eight waves reading, then all eight
  // multiplying.  The lagging half finishes its reads before the next barrier (the slot is refilled behind it).
  PROF_DECL;
  if (wn2 == 0) {
    for (int st = 0; st < nst; ++st) {
      wait_stage(st, DM_STAGES - 2);
      PROF_MARK(0);
      if (!(ablate & 16)) __builtin_amdgcn_s_barrier();
      PROF_MARK(1);
#if TN_PHASE == 2
      if (st + DM_STAGES - 1 < nst) issue(st + DM_STAGES - 1, slot == 0 ? DM_STAGES - 1 : slot - 1);
#endif
      read_frags(f0, slot, 0);
      read_frags(f1, slot, 1);
#ifdef HMA_PROF
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      PROF_MARK(2);
#if TN_SPREAD
      {
        const bool refill = st + DM_STAGES - 1 < nst;
        const int sn = st + DM_STAGES - 1, sl = slot == 0 ? DM_STAGES - 1 : slot - 1;
        mma_spread(f0, 0, sn, sl, refill);
        mma_spread(f1, 1, sn, sl, refill);
      }
      PROF_MARK(3);
#else
      mma(f0);
      PROF_MARK(3);
#if TN_PHASE != 2
      if (st + DM_STAGES - 1 < nst) issue(st + DM_STAGES - 1, slot == 0 ? DM_STAGES - 1 : slot - 1);
#endif
      PROF_MARK(4);
      mma(f1);
      PROF_MARK(3);
#endif
      slot = slot + 1 == DM_STAGES ? 0 : slot + 1;
    }
    PROF_MARK(5);
#ifdef HMA_PROF
    if (wave == 0) PROF_FLUSH();
#endif
  } else {
    wait_stage(0, DM_STAGES - 2);
    if (!(ablate & 16)) __builtin_amdgcn_s_barrier();
    if (DM_STAGES - 1 < nst) issue(DM_STAGES - 1, DM_STAGES - 1);
    read_frags(f0, 0, 0);
    read_frags(f1, 0, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    slot = 1;
    for (int st = 1; st < nst; ++st) {
      wait_stage(st, DM_STAGES - 2);
      if (!(ablate & 16)) __builtin_amdgcn_s_barrier();
#if TN_SPREAD
      {
        const bool refill = st + DM_STAGES - 1 < nst;
        const int sn = st + DM_STAGES - 1, sl = slot == 0 ? DM_STAGES - 1 : slot - 1;
        mma_spread(f0, 0, sn, sl, refill);
        mma_spread(f1, 1, sn, sl, refill);
      }
#else
      mma(f0);
      mma(f1);
      if (st + DM_STAGES - 1 < nst) issue(st + DM_STAGES - 1, slot == 0 ? DM_STAGES - 1 : slot - 1);
#endif
      read_frags(f0, slot, 0);
      read_frags(f1, slot, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      slot = slot + 1 == DM_STAGES ? 0 : slot + 1;
    }
    mma(f0);
    mma(f1);
  }
#else
  for (int st = 0; st < nst; ++st) {
    wait_stage(st, DM_STAGES - 2);
    // every wave's pieces of stage st are in LDS, and every wave is done reading stage st - 1 (its slot is refilled next)
    if (!(ablate & 16)) __builtin_amdgcn_s_barrier();
    read_frags(f0, slot, 0);
    read_frags(f1, slot, 1);
    mma(f0);
    // the refill goes behind the first eight MFMAs: its address generation and issue overlap the matrix pipe
    if (st + DM_STAGES - 1 < nst) issue(st + DM_STAGES - 1, slot == 0 ? DM_STAGES - 1 : slot - 1);
    mma(f1);
    slot = slot + 1 == DM_STAGES ? 0 : slot + 1;
  }
#endif
#endif  // TN_ISSUER_PROBE
  // bf16 partial of this workgroup's 256 x 256 block, in the order the accumulators sit in the waves: piece
  // ((wave * 8 + i * 2 + j) * 2 + h) * 64 + lane is the lane's 8 values e = 8 h .. 8 h + 7 of acc[i][j], i.e. row
  // n = 128 wn2 + 32 i + r, columns k = 64 wk4 + 32 j + 16 h + 4 hi + {0..3} and the same + 8 (tn_reduce_native_kernel
  // decodes this).  Every store instruction writes 1 KB contiguous; the row-major layout of the register-staged
  // kernels (32 rows x 32 B per instruction) cost 17-22 us per call for these 32 MB (ablation in profiles/).
  uint16_t* part = reinterpret_cast<uint16_t*>(p.ws) + (int64_t)vid * (WT * WT);
  if (!(ablate & 8))
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x16_t& v = acc[i][j];
          const uint4 o = make_uint4(pack_bf16(v[8 * h], v[8 * h + 1]), pack_bf16(v[8 * h + 2], v[8 * h + 3]),
                                     pack_bf16(v[8 * h + 4], v[8 * h + 5]), pack_bf16(v[8 * h + 6], v[8 * h + 7]));
          *reinterpret_cast<uint4*>(part + ((((wave * 8 + ((i + wk4) & 3) * 2 + j) * 2 + h) * 64 + lane) << 3)) = o;
        }
  if (COLSUM && k0 == 0 && p.colsum) {
    colsum += __shfl_xor(colsum, 32);
    if (hi == 0) p.ws[(int64_t)p.nb * (WT * WT) + (int64_t)vid * WT + wn2 * 128 + wk4 * 32 + r] = colsum;
  }
}

// Sum of the LDS-DMA kernel's partials (layout: see its epilogue) into dW.  One block per 32 pieces (512 B of every
// split's partial): the 8 half-waves take every 8th split, 16 bytes per lane; 256 blocks per 256 x 256 output block.
// affine != 0: dW += gamma[k] * P[n][k] + beta[k] * colsum(dY)[n] (the LayerNorm affine of A, which the DMA kernel cannot
// apply on the way into LDS); colsum comes from the bias partials of the n-block's k0 == 0 group.
struct tn_reduce_args {
  hma_gemm_tn_t p[TN_MAXP];  // ws / splits already set per problem
  int gn[TN_MAXP], gk[TN_MAXP], affine[TN_MAXP];
  int gy_end[TN_MAXP];       // grid.y rows of problems 0 .. i (prefix sums)
  int n;
};
__global__ __launch_bounds__(256) void tn_reduce_native_kernel(tn_reduce_args ra) {
  __shared__ float red[8][32][8];
  __shared__ float csum[8][32];
  int pi = 0;
#pragma unroll 1
  while (pi + 1 < ra.n && (int)blockIdx.y >= ra.gy_end[pi]) ++pi;
  const hma_gemm_tn_t& p = ra.p[pi];
  const int groups_n = ra.gn[pi], groups_k = ra.gk[pi];
  const int affine = ra.affine[pi];
  const int by = (int)blockIdx.y - (pi > 0 ? ra.gy_end[pi - 1] : 0);
  const int groups = groups_n * groups_k;
  const int g = by % groups;
  const int64_t bz = by / groups;
  const int64_t n0 = (int64_t)(g / groups_k) * WT, k0 = (int64_t)(g % groups_k) * WT;
  const int tid = threadIdx.x, r = tid & 31, sl = tid >> 5;  // sl: split lane 0..7
  const int grp = blockIdx.x >> 1, hi = blockIdx.x & 1;      // 1 KB group of 64 pieces, and which half of it
  const int h = grp & 1, j = (grp >> 1) & 1, i = (grp >> 2) & 3, wave = grp >> 4;
  const int wn2 = wave >> 2, wk4 = wave & 3;
  const int nl = wn2 * 128 + i * 32 + r;
  const int kl = wk4 * 64 + j * 32 + 16 * h + 4 * hi;  // this piece: columns kl .. kl + 3 and kl + 8 .. kl + 11
  const int64_t nblocks = (int64_t)p.splits * groups * (p.batch > 0 ? p.batch : 1);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const uint16_t* base = reinterpret_cast<const uint16_t*>(p.ws) + (bz * p.splits * groups + g) * (int64_t)(WT * WT) +
                         ((grp * 64 + hi * 32 + r) << 3);
#pragma unroll 8
  for (int sp = sl; sp < p.splits; sp += 8) {
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(base + (int64_t)sp * groups * (WT * WT)), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += f[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[sl][r][e] = acc[e];
  if (affine) {
    const int g0 = (g / groups_k) * groups_k;
    const float* bp = p.ws + nblocks * (WT * WT) + (bz * p.splits * groups + g0) * (int64_t)WT + nl;
    float sum = 0.f;
#pragma unroll 8
    for (int sp = sl; sp < p.splits; sp += 8) sum += bp[(int64_t)sp * groups * WT];
    csum[sl][r] = sum;
  }
  __syncthreads();
  if (tid < 64) {  // lanes 0-31 finish columns kl .. kl + 3, lanes 32-63 kl + 8 .. kl + 11
    const int qd = tid >> 5;
    const int kk = kl + 8 * qd;
    float4* dst = reinterpret_cast<float4*>(p.dW + bz * p.sdW + (n0 + nl) * p.lddw + k0 + kk);
    float4 o = *dst;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      t.x += red[q][r][4 * qd]; t.y += red[q][r][4 * qd + 1]; t.z += red[q][r][4 * qd + 2]; t.w += red[q][r][4 * qd + 3];
    }
    if (affine) {
      const float4 gm = *reinterpret_cast<const float4*>(p.gamma + k0 + kk);
      const float4 bt = *reinterpret_cast<const float4*>(p.beta + k0 + kk);
      float cs = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) cs += csum[q][r];
      if (p.dgamma) {
        // gradients of the folded LayerNorm affine from the un-scaled product t = (dY^T xhat)[n][k]:
        // dgamma[k] += sum_n W[n][k] t[n][k], dbeta[k] += sum_n W[n][k] colsum(dY)[n]; the 32 lanes of a half-wave hold
        // 32 different rows n of the same 4 columns
        const float4 wm = *reinterpret_cast<const float4*>(p.w_master + bz * p.sdW + (n0 + nl) * p.lddw + k0 + kk);
        float dg[4] = {wm.x * t.x, wm.y * t.y, wm.z * t.z, wm.w * t.w};
        float db[4] = {wm.x * cs, wm.y * cs, wm.z * cs, wm.w * cs};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) {
            dg[e] += __shfl_xor(dg[e], o, 64);
            db[e] += __shfl_xor(db[e], o, 64);
          }
        }
        if (r == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            atomicAdd(p.dgamma + k0 + kk + e, dg[e]);
            atomicAdd(p.dbeta + k0 + kk + e, db[e]);
          }
        }
      }
      t.x = gm.x * t.x + bt.x * cs; t.y = gm.y * t.y + bt.y * cs; t.z = gm.z * t.z + bt.z * cs; t.w = gm.w * t.w + bt.w * cs;
    }
    o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
    *dst = o;
  }
  // bias gradient: blocks 0..7 of a k0 == 0 group sum the column-sum partials of columns 32 x .. 32 x + 31
  if (p.dBias && k0 == 0 && blockIdx.x < 8) {
    __syncthreads();
    const int col = blockIdx.x * 32 + r;
    const float* bp = p.ws + nblocks * (WT * WT) + (bz * p.splits * groups + g) * (int64_t)WT + col;
    float sum = 0.f;
#pragma unroll 8
    for (int sp = sl; sp < p.splits; sp += 8) sum += bp[(int64_t)sp * groups * WT];
    csum[sl][r] = sum;
    __syncthreads();
    if (tid < 32) {
      float tot = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) tot += csum[q][r];
      p.dBias[bz * p.sdBias + n0 + col] += tot;
    }
  }
}

// dW[bz][n0 + nl][k0 + kl] += sum over splits of ws[((bz * splits + s) * groups + g)][nl][kl]
// 128 workgroups per 256 x 256 block: 64 lanes x 8 outputs each, the splits dealt over the 4 waves.
__global__ __launch_bounds__(256) void tn_reduce_kernel(hma_gemm_tn_t p, int groups_n, int groups_k) {
  __shared__ float4 red[4][64][2];
  const int groups = groups_n * groups_k;
  const int g = blockIdx.y % groups;
  const int64_t bz = blockIdx.y / groups;
  const int64_t n0 = (int64_t)(g / groups_k) * WT, k0 = (int64_t)(g % groups_k) * WT;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int idx = (blockIdx.x * 64 + lane) * 8;  // 8 consecutive k of one n row: one 16-byte bf16 load per split
  const int nl = idx / WT, kl = idx % WT;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const uint16_t* base = reinterpret_cast<const uint16_t*>(p.ws) + (bz * p.splits * groups + g) * (int64_t)(WT * WT) + idx;
#pragma unroll 8
  for (int sp = w; sp < p.splits; sp += 4) {
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(base + (int64_t)sp * groups * (WT * WT)), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += f[e];
  }
  red[w][lane][0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  red[w][lane][1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
  __syncthreads();
  if (w < 2) {  // wave 0 finishes k .. k+3, wave 1 k+4 .. k+7
    float4* dst = reinterpret_cast<float4*>(p.dW + bz * p.sdW + (n0 + nl) * p.lddw + k0 + kl + 4 * w);
    float4 o = *dst;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 a = red[q][lane][w];
      o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    }
    *dst = o;
  }
  const int64_t nblocks = (int64_t)p.splits * groups * (p.batch > 0 ? p.batch : 1);
  // bias partials (written when the workspace has room for them): one block per group sums them
  if (p.dBias && k0 == 0 && blockIdx.x < 4 && p.ws_elems >= nblocks * (WT * WT + WT)) {
    // block x sums columns 64 x .. 64 x + 63; its 4 waves take every 4th split (independent loads), then LDS
    __syncthreads();
    const int col = blockIdx.x * 64 + lane;
    const float* bp = p.ws + nblocks * (WT * WT) + (bz * p.splits * groups + g) * (int64_t)WT + col;
    float sum = 0.f;
#pragma unroll 8
    for (int sp = w; sp < p.splits; sp += 4) sum += bp[(int64_t)sp * groups * WT];
    red[w][lane][0].x = sum;
    __syncthreads();
    if (w == 0) p.dBias[bz * p.sdBias + n0 + col] += red[0][lane][0].x + red[1][lane][0].x + red[2][lane][0].x + red[3][lane][0].x;
  }
}

// ------------------------------------------------------------------------------- NT, LDS-DMA ring (K > 256, N = 256)
// C[m][0..255] = A[m][:] . W[0..255][:]^T for the deep-K layer GEMMs (fc2, dfc1, dqkv: K = 768 / 1024).  The lock-step
// kernel above re-streams the 256 x K weight slab through each CU's vector-memory path once per 128 token rows, and that
// path (~11-12 B/clk per CU, L2 hits included) is what bounds it.  Here a workgroup owns a contiguous run of token rows
// and works through it in 256-row tiles (half the weight re-reads per row), both operands arriving by LDS-DMA in 32-deep
// K stages ([256][32] bf16 of A, then of W: 32 KB, four-slot ring, three stages in flight; or 64-deep stages in two
// slots) that keep streaming across tile boundaries, so the next tile's first stages land while this tile's epilogue
// runs.  Rows are 64 (128) B in LDS; 16-byte chunk c of row r sits at chunk c ^ swz(r) (applied to the DMA source
// address and again on the read), which makes the ds_read_b128 fragment reads conflict-free.  Accumulators and epilogue as gemm_nt_p3_kernel (a lane owns one
// token row and runs of 4 output columns).
constexpr int NR_SMEM_BYTES = 128 * 1024;  // NS slots x (A tile + W tile) x 256 rows x 2 KD bytes

// ABL (debug build only): 1 no A DMA, 2 no W DMA, 4 no epilogue memory traffic, 8 no MFMA
// KD = K depth of a stage (32: 64-byte LDS rows, 4 slots, 3 stages in flight; 64: 128-byte rows = whole cache lines per
// DMA row piece, 2 slots, 1 stage in flight), NS = slots.
template <int EPI, int ABL = 0, int KD = 32, int NS = 4>
__global__ __launch_bounds__(512, 2) void gemm_nt_ring_kernel(hma_gemm_nt_t p, int64_t chunk) {
  constexpr int RB = KD * 2;                 // LDS row bytes
  constexpr int NR_TILE_BYTES = 256 * RB;    // one operand's K stage
  constexpr int NR_STAGE_BYTES = 2 * NR_TILE_BYTES;
  constexpr int NR_STAGES = NS;
  static_assert(NS * NR_STAGE_BYTES == NR_SMEM_BYTES, "ring size");
  constexpr int LPR = RB / 16;               // lanes (16-byte chunks) per row
  constexpr int RPP = 64 / LPR;              // rows per 1 KB DMA piece
  constexpr int QN = 32 / RPP;               // pieces per wave, tile and stage (a wave fills rows 32 w .. 32 w + 31)
  constexpr int KK = KD / 16;                // MFMA k-steps per stage
  // chunk swizzle of a row (conflict-free ds_read_b128 fragments for both row sizes)
  auto swz = [](int row) { return KD == 32 ? (row >> 2) & 3 : (row >> 1) & 7; };
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  HMA_LDS(char)* lds = (HMA_LDS(char)*)smem;
  const uint32_t lds_b = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int G = gridDim.x, b = blockIdx.x;
  const int vid = (b & 7) * (G >> 3) + min(b & 7, G & 7) + (b >> 3);  // contiguous row runs per XCD
  const int64_t m_begin = (int64_t)vid * chunk;
  int64_t m_end = m_begin + chunk;
  if (m_end > p.M) m_end = p.M;
  if (m_begin >= m_end) return;
  const int KT = (int)(p.K / KD);
  // Every other workgroup starts with a 128-row tile: neighbouring CUs are then half a tile out of phase, and one's
  // epilogue (residual read + write, HBM) overlaps the other's K loop instead of the whole chip alternating between the two.
  const int64_t first = ((vid & 1) && m_end - m_begin > 128) ? 128 : 256;
  const int64_t rest = m_end - m_begin - first;
  const int ntiles = 1 + (rest > 0 ? (int)((rest + 255) >> 8) : 0);
  const int nst = ntiles * KT;

  const uint16_t* A = reinterpret_cast<const uint16_t*>(p.A);
  const uint16_t* W = reinterpret_cast<const uint16_t*>(p.W);
  // DMA pieces of this wave: rows 32 wave + RPP q + lane / LPR of both tiles, q = 0 .. QN - 1
  const int prow = wave * 32 + lane / LPR;
  // issue cursor (runs three stages ahead of the compute cursor)
  int i_kt = 0;
  int64_t i_m0 = m_begin, i_end = m_begin + first < m_end ? m_begin + first : m_end;
  const uint16_t* a_src[QN];
  const uint16_t* w_src[QN];
#pragma unroll
  for (int q = 0; q < QN; ++q) {
    const int row = prow + RPP * q;
    const int c = (lane % LPR) ^ swz(row);
    w_src[q] = W + (int64_t)row * p.ldw + c * 8;
  }
  auto set_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      const int row = prow + RPP * q;
      const int c = (lane % LPR) ^ swz(row);
      int64_t m = i_m0 + row;
      m = m < i_end ? m : i_end - 1;  // rows past the tile are clamped (their outputs are never stored)
      a_src[q] = A + m * p.lda + c * 8;
    }
  };
  set_tile();
  auto issue = [&](int slot) __attribute__((always_inline)) {
    const uint32_t sb = lds_b + slot * NR_STAGE_BYTES + wave * 32 * RB;
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      if (!(ABL & 1)) glds16(a_src[q] + i_kt * KD, sb + q * 1024);
      if (!(ABL & 2)) glds16(w_src[q] + i_kt * KD, sb + NR_TILE_BYTES + q * 1024);
    }
    if (++i_kt == KT) {
      i_kt = 0;
      i_m0 = i_end;
      i_end = i_m0 + 256 < m_end ? i_m0 + 256 : m_end;
      if (i_m0 >= m_end) i_m0 = m_end - 1, i_end = m_end;  // (no tile follows: nothing more is issued)
      set_tile();
    }
  };

  f32x16_t acc[4][2];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.f;
  };
  zero_acc();

  const int lr = lane & 31, lhi = lane >> 5;
  // fragment of row `row` of a tile: k = 16 kk + 8 lhi .. + 7
  auto frag = [&](HMA_LDS(char)* tile, int row, int kk) __attribute__((always_inline)) {
    const int c = (kk * 2 + lhi) ^ swz(row);
    return __builtin_bit_cast(bf16x8_t, *(HMA_LDS(uint4)*)(tile + row * RB + c * 16));
  };
  auto wait_stage = [&](int st) __attribute__((always_inline)) {
    int later = nst - 1 - st;  // stages issued behind st: at most NS - 2 at this point, 2 QN pieces each
    later = later < NS - 2 ? later : NS - 2;
    static_assert(2 * QN == 4 || NS == 2, "vmcnt immediates below: 4 pieces per stage and wave, or nothing in flight behind st");
    if (NS > 2 && later >= 2)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (NS > 2 && later == 1)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

#pragma unroll
  for (int st = 0; st < NR_STAGES - 1; ++st)
    if (st < nst) issue(st);
  int kt = 0;
  int64_t m0 = m_begin, t_end = m_begin + first < m_end ? m_begin + first : m_end;
  for (int st = 0; st < nst; ++st) {
    const int slot = st % NR_STAGES;
    wait_stage(st);
    // every wave's pieces of stage st are in LDS, and every wave is done reading stage st - 1 (its slot is refilled below)
    __builtin_amdgcn_s_barrier();
    HMA_LDS(char)* As = lds + slot * NR_STAGE_BYTES;
    HMA_LDS(char)* Ws = As + NR_TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      bf16x8_t af[4], wf[2];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) af[mt] = frag(As, wm * 128 + mt * 32 + lr, kk);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) wf[nt] = frag(Ws, wn * 64 + nt * 32 + lr, kk);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          if (ABL & 8)
            acc[mt][nt][0] += __builtin_bit_cast(float, __builtin_bit_cast(uint4, wf[nt]).x ^ __builtin_bit_cast(uint4, af[mt]).y);
          else
            acc[mt][nt] = mfma32(wf[nt], af[mt], acc[mt][nt]);  // D rows = n, D cols = token
        }
      if (kk == 0 && st + NR_STAGES - 1 < nst) issue((st + NR_STAGES - 1) % NR_STAGES);
    }
    if (++kt == KT) {
      // tile done: epilogue (the ring keeps filling with the next tile's stages meanwhile)
      const float* bias = p.bias;
      if (EPI == HMA_EPI_RESID) {
        // The residual rows come from HBM: all 16 float4 loads of a 64-row half are issued before the first add (16 KB
        // per wave in flight).  One 32-byte round trip per accumulator pair, as epilogue_oct does it, left the tile
        // epilogue latency-bound: 70 us of fc2's 178 us (ablation in profiles/).
        float* Cb = reinterpret_cast<float*>(p.C);
        const uint32_t seed = p.drop_p > 0.f ? *p.drop_seed : 0u;
        const uint32_t th = drop_thresh(p.drop_p);
        const float sc = p.drop_p > 0.f ? drop_scale(p.drop_p) : 1.0f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          float4 x[2][2][2][2];
          int64_t off[2], crow[2];
          bool ok[2];
#pragma unroll
          for (int ml = 0; ml < 2; ++ml) {
            int64_t m = m0 + wm * 128 + (half * 2 + ml) * 32 + lr;
            ok[ml] = m < t_end;
            m = ok[ml] ? m : t_end - 1;  // clamped, never stored
            crow[ml] = remap_row(m, p.c_group_rows, p.c_group_stride);
            off[ml] = crow[ml] * p.ldc;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int gi = 0; gi < 2; ++gi) {
                const float* c = Cb + off[ml] + wn * 64 + nt * 32 + 16 * gi + 4 * lhi;
                x[ml][nt][gi][0] = *reinterpret_cast<const float4*>(c);
                x[ml][nt][gi][1] = *reinterpret_cast<const float4*>(c + 8);
              }
          }
#pragma unroll
          for (int ml = 0; ml < 2; ++ml)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int gi = 0; gi < 2; ++gi) {
                const int mt = half * 2 + ml, g2 = 2 * gi;
                const int64_t nq = wn * 64 + nt * 32;
                float v0[4], v1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  v0[e] = (ABL & 4) ? 0.f : acc[mt][nt][4 * g2 + e];
                  v1[e] = (ABL & 4) ? 0.f : acc[mt][nt][4 * g2 + 4 + e];
                }
                if (bias) {
                  const float4 b0 = *reinterpret_cast<const float4*>(bias + nq + 8 * g2 + 4 * lhi);
                  const float4 b1 = *reinterpret_cast<const float4*>(bias + nq + 8 * g2 + 8 + 4 * lhi);
                  v0[0] += b0.x; v0[1] += b0.y; v0[2] += b0.z; v0[3] += b0.w;
                  v1[0] += b1.x; v1[1] += b1.y; v1[2] += b1.z; v1[3] += b1.w;
                }
                if (p.drop_p > 0.f) {  // Dropout on the branch output before the residual add (st_transformer.py:26)
                  const int64_t e0 = off[ml] + nq + 8 * g2 + 4 * lhi;
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    v0[e] = drop_keep(seed, p.drop_salt, e0 + e, th) ? v0[e] * sc : 0.f;
                    v1[e] = drop_keep(seed, p.drop_salt, e0 + 8 + e, th) ? v1[e] * sc : 0.f;
                  }
                }
                float4 x0 = x[ml][nt][gi][0], x1 = x[ml][nt][gi][1];
                x0.x += v0[0]; x0.y += v0[1]; x0.z += v0[2]; x0.w += v0[3];
                x1.x += v1[0]; x1.y += v1[1]; x1.z += v1[2]; x1.w += v1[3];
                if (ok[ml] && !(ABL & 4)) {
                  float* c = Cb + off[ml] + nq + 8 * g2 + 4 * lhi;
                  *reinterpret_cast<float4*>(c) = x0;
                  *reinterpret_cast<float4*>(c + 8) = x1;
                  if (p.C2) {
                    const float ca[4] = {x0.x, x0.y, x0.z, x0.w}, cb[4] = {x1.x, x1.y, x1.z, x1.w};
                    store_bf16_oct(reinterpret_cast<uint16_t*>(p.C2) + crow[ml] * p.ldc2 + nq, g2, lhi, ca, cb);
                  }
                }
              }
        }
      } else {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int64_t m = m0 + wm * 128 + mt * 32 + lr;
          if (m < t_end) {
            const int64_t crow = remap_row(m, p.c_group_rows, p.c_group_stride);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              const int64_t nq = wn * 64 + nt * 32;
#pragma unroll
              for (int g2 = 0; g2 < 4; g2 += 2) {
                float v0[4], v1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  v0[e] = acc[mt][nt][4 * g2 + e];
                  v1[e] = acc[mt][nt][4 * g2 + 4 + e];
                }
                if (bias) {
                  const float4 b0 = *reinterpret_cast<const float4*>(bias + nq + 8 * g2 + 4 * lhi);
                  const float4 b1 = *reinterpret_cast<const float4*>(bias + nq + 8 * g2 + 8 + 4 * lhi);
                  v0[0] += b0.x; v0[1] += b0.y; v0[2] += b0.z; v0[3] += b0.w;
                  v1[0] += b1.x; v1[1] += b1.y; v1[2] += b1.z; v1[3] += b1.w;
                }
                if (!(ABL & 4) || v0[0] + v1[3] == 1.2345e-30f) epilogue_oct<EPI>(p, 0, crow, nq, g2, lhi, v0, v1);
              }
            }
          }
        }
      }
      zero_acc();
      kt = 0;
      m0 = t_end;
      t_end = m0 + 256 < m_end ? m0 + 256 : m_end;
    }
  }
}

template <auto Kern>
int set_smem_bytes(int bytes) {
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return -(int)e;
    done = true;
  }
  return 0;
}

template <auto Kern>
int set_smem() {
  static bool done = false;  // one flag per kernel instantiation
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(Kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return -(int)e;
    done = true;
  }
  return 0;
}

}  // namespace

extern "C" int hma_gemm_nt(void* stream, const hma_gemm_nt_t* p) {
  if (!p || !p->A || !p->W || !p->C) return HMA_EINVAL;
  if (p->M <= 0) return 0;
  if (p->N % BN != 0 || p->K % BK != 0) return HMA_EINVAL;
  if (p->a_kind == HMA_A_BF16_AFFINE && (!p->gamma || !p->beta)) return HMA_EINVAL;
  if ((p->epi == HMA_EPI_GELU2 || p->epi == HMA_EPI_SILU2) && !p->C2) return HMA_EINVAL;
  if ((p->epi == HMA_EPI_DGELU || p->epi == HMA_EPI_DSILU) && !p->U) return HMA_EINVAL;
  if (p->drop_p != 0.f) {  // dropout: streaming GELU2 / DGELU (K = 256), ring RESID (K > 256)
    if (!(p->drop_p > 0.f && p->drop_p < 1.f) || !p->drop_seed || (p->batch > 1)) return HMA_EINVAL;
    const bool act = (p->epi == HMA_EPI_GELU2 || p->epi == HMA_EPI_DGELU) && p->K == 256;
    const bool res = p->epi == HMA_EPI_RESID && p->K > 256 && p->K % 64 == 0 && p->N % 256 == 0 && !p->ln_xhat;
    if (!act && !res) return HMA_EINVAL;
    if (p->epi == HMA_EPI_DGELU && p->ldc != p->ldu) return HMA_EINVAL;  // the mask is indexed like the forward's C2
  }
  if (p->ln_xhat) {  // fused LayerNorm: only on the streaming residual path
    if (p->epi != HMA_EPI_RESID || p->N != 256 || p->K != 256 || p->batch > 1 || !p->ln_rstd) return HMA_EINVAL;
    if (p->ln_ss && (!p->ln_xm || p->ln_rows_per_frame <= 0)) return HMA_EINVAL;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc;
  hma_gemm_nt_t pa = *p;
  pa._pad2 = 0;
#ifdef HMA_PROF
  static const int ablate = getenv("HMA_GEMM_ABLATE") ? atoi(getenv("HMA_GEMM_ABLATE")) : 0;  // debug build only
  pa._pad2 = ablate;
#endif
  // (K < 256 on the persistent 8-wave kernel needs its undeferred variant, instantiated for the plain epilogues of bf16 / fp32
  // operands; anything else that shallow takes the tile-per-workgroup kernel at the bottom)
  const bool plain_epi = p->epi == HMA_EPI_BF16 || p->epi == HMA_EPI_F32 || p->epi == HMA_EPI_RESID || p->epi == HMA_EPI_ATOMIC_F32;
  const bool shallow_other = p->K < 4 * PK && plain_epi &&
                             !((p->a_kind == HMA_A_BF16 || p->a_kind == HMA_A_F32) && p->epi != HMA_EPI_ATOMIC_F32);
  if (p->N % PN == 0 && !shallow_other) {
    const int tiles_m = (int)((p->M + PM - 1) / PM), tiles_n = (int)(p->N / PN);
    const int total = tiles_m * tiles_n * (p->batch > 0 ? p->batch : 1);
    static int n_cu = 0;
    if (n_cu == 0) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return HMA_EINVAL;
      n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    // K = 256 (qkv, proj, modulate linear, fc1 and the dgrads of proj / fc2): streaming waves
    if (p->K == 256 && p->epi != HMA_EPI_ATOMIC_F32) {
      const int nslabs = (int)(p->N / 256) * (p->batch > 0 ? p->batch : 1);
      const int64_t tiles16 = (p->M + 15) / 16;
      if (p->ln_xhat) {  // validated above: RESID, N = K = 256
        int per_slab = n_cu;
        if ((int64_t)per_slab * 8 > tiles16) per_slab = (int)((tiles16 + 7) / 8);
#define HMA_NTW_LN_CASE(AK)                                                                           \
  if (p->a_kind == AK) {                                                                              \
    if ((rc = set_smem_bytes<gemm_nt_sw_kernel<AK, HMA_EPI_RESID, 8, true>>(TW_SMEM_BYTES))) return rc; \
    hipLaunchKernelGGL((gemm_nt_sw_kernel<AK, HMA_EPI_RESID, 8, true>), dim3((unsigned)per_slab), dim3(512), TW_SMEM_BYTES, s, pa, \
                       1, per_slab);                                                                  \
    HMA_CHECK_LAUNCH();                                                                               \
    return 0;                                                                                         \
  }
        HMA_NTW_LN_CASE(HMA_A_BF16)
        HMA_NTW_LN_CASE(HMA_A_F32)
        HMA_NTW_LN_CASE(HMA_A_BF16_AFFINE)
        return HMA_EINVAL;
      }
#define HMA_NTW_CASE(AK, EP)                                                                          \
  if (p->a_kind == AK && p->epi == EP) {                                                              \
    constexpr int NWV = 8; /* 12 waves (168 VGPRs) spill and measured 1.05-2.7x slower */             \
    int per_slab = n_cu / nslabs;                                                                     \
    if (per_slab < 1) per_slab = 1;                                                                   \
    if ((int64_t)per_slab * NWV > tiles16) per_slab = (int)((tiles16 + NWV - 1) / NWV);               \
    if ((rc = set_smem_bytes<gemm_nt_sw_kernel<AK, EP, NWV>>(TW_SMEM_BYTES))) return rc;              \
    hipLaunchKernelGGL((gemm_nt_sw_kernel<AK, EP, NWV>), dim3((unsigned)(nslabs * per_slab)), dim3(64 * NWV), TW_SMEM_BYTES, s, pa, \
                       nslabs, per_slab);                                                             \
    HMA_CHECK_LAUNCH();                                                                               \
    return 0;                                                                                         \
  }
#define HMA_NTW_ALL(AK)                                                                               \
  HMA_NTW_CASE(AK, HMA_EPI_BF16) HMA_NTW_CASE(AK, HMA_EPI_F32) HMA_NTW_CASE(AK, HMA_EPI_RESID)        \
  HMA_NTW_CASE(AK, HMA_EPI_GELU2) HMA_NTW_CASE(AK, HMA_EPI_SILU2) HMA_NTW_CASE(AK, HMA_EPI_DGELU)     \
  HMA_NTW_CASE(AK, HMA_EPI_DSILU)
      HMA_NTW_ALL(HMA_A_BF16)
      HMA_NTW_ALL(HMA_A_F32)
      HMA_NTW_ALL(HMA_A_BF16_AFFINE)
      return HMA_EINVAL;
    }
    // deep-K, N = 256, plain bf16 operands (fc2, dfc1, dqkv): the LDS-DMA ring kernel
    if (p->K > 256 && (p->K & 31) == 0 && p->N == 256 && p->a_kind == HMA_A_BF16 &&
        (p->batch <= 1) && p->a_group_rows <= 0 && (p->epi == HMA_EPI_BF16 || p->epi == HMA_EPI_RESID) &&
        (p->lda & 7) == 0 && (p->ldw & 7) == 0 && (reinterpret_cast<uintptr_t>(p->A) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(p->W) & 15) == 0 && p->M >= 256) {
      int64_t chunk = (p->M + n_cu - 1) / n_cu;
      chunk = (chunk + 31) & ~(int64_t)31;
      const unsigned g = (unsigned)((p->M + chunk - 1) / chunk);
#ifdef HMA_PROF
#define HMA_NTR_ABL(EP, A)                                                                            \
  if (p->epi == EP && ablate == A) {                                                                  \
    if ((rc = set_smem_bytes<gemm_nt_ring_kernel<EP, A>>(NR_SMEM_BYTES))) return rc;                  \
    hipLaunchKernelGGL((gemm_nt_ring_kernel<EP, A>), dim3(g), dim3(512), NR_SMEM_BYTES, s, pa, chunk); \
    HMA_CHECK_LAUNCH();                                                                               \
    return 0;                                                                                         \
  }
#define HMA_NTR_ABLS(EP) HMA_NTR_ABL(EP, 1) HMA_NTR_ABL(EP, 2) HMA_NTR_ABL(EP, 3) HMA_NTR_ABL(EP, 4) HMA_NTR_ABL(EP, 8) \
  HMA_NTR_ABL(EP, 7) HMA_NTR_ABL(EP, 12) HMA_NTR_ABL(EP, 11) HMA_NTR_ABL(EP, 15)
      HMA_NTR_ABLS(HMA_EPI_BF16) HMA_NTR_ABLS(HMA_EPI_RESID)
#endif
      // stage depth: 64 (two slots) measured 5 % faster for the bf16 epilogue, 0-3 % slower for the residual one (which is
      // at the register cap with its batched epilogue)
      if (p->epi == HMA_EPI_BF16 && (p->K & 63) == 0) {
        if ((rc = set_smem_bytes<gemm_nt_ring_kernel<HMA_EPI_BF16, 0, 64, 2>>(NR_SMEM_BYTES))) return rc;
        hipLaunchKernelGGL((gemm_nt_ring_kernel<HMA_EPI_BF16, 0, 64, 2>), dim3(g), dim3(512), NR_SMEM_BYTES, s, pa, chunk);
      } else if (p->epi == HMA_EPI_BF16) {
        if ((rc = set_smem_bytes<gemm_nt_ring_kernel<HMA_EPI_BF16>>(NR_SMEM_BYTES))) return rc;
        hipLaunchKernelGGL((gemm_nt_ring_kernel<HMA_EPI_BF16>), dim3(g), dim3(512), NR_SMEM_BYTES, s, pa, chunk);
      } else {
        if ((rc = set_smem_bytes<gemm_nt_ring_kernel<HMA_EPI_RESID>>(NR_SMEM_BYTES))) return rc;
        hipLaunchKernelGGL((gemm_nt_ring_kernel<HMA_EPI_RESID>), dim3(g), dim3(512), NR_SMEM_BYTES, s, pa, chunk);
      }
      HMA_CHECK_LAUNCH();
      return 0;
    }
    if (p->drop_p != 0.f && p->epi != HMA_EPI_RESID) return HMA_EINVAL;  // (activation dropout: streaming kernel only)
    // everything else with N % 256 == 0: lock-step persistent tiles; plain epilogues on the 8-wave kernel,
    // activation epilogues on the two-per-CU 4-wave kernel
    const bool valu_epi = p->epi == HMA_EPI_GELU2 || p->epi == HMA_EPI_SILU2 || p->epi == HMA_EPI_DGELU || p->epi == HMA_EPI_DSILU;
    if (!valu_epi && p->K % PK == 0) {
      const dim3 p3grid((unsigned)(total < n_cu ? total : n_cu));
#define HMA_NT3_CASE(AK, EP)                                                                          \
  if (p->a_kind == AK && p->epi == EP) {                                                              \
    if ((rc = set_smem_bytes<gemm_nt_p3_kernel<AK, EP>>(P_SMEM_BYTES))) return rc;                    \
    hipLaunchKernelGGL((gemm_nt_p3_kernel<AK, EP>), p3grid, dim3(512), P_SMEM_BYTES, s, pa, tiles_m, tiles_n, total); \
    HMA_CHECK_LAUNCH();                                                                               \
    return 0;                                                                                         \
  }
#define HMA_NT3S_CASE(AK, EP)                                                                         \
  if (p->a_kind == AK && p->epi == EP) {                                                              \
    if ((rc = set_smem_bytes<gemm_nt_p3_kernel<AK, EP, true>>(P_SMEM_BYTES))) return rc;              \
    hipLaunchKernelGGL((gemm_nt_p3_kernel<AK, EP, true>), p3grid, dim3(512), P_SMEM_BYTES, s, pa, tiles_m, tiles_n, total); \
    HMA_CHECK_LAUNCH();                                                                               \
    return 0;                                                                                         \
  }
#define HMA_NT3_ALL(AK)                                                                               \
  HMA_NT3_CASE(AK, HMA_EPI_BF16) HMA_NT3_CASE(AK, HMA_EPI_F32) HMA_NT3_CASE(AK, HMA_EPI_RESID)        \
  HMA_NT3_CASE(AK, HMA_EPI_ATOMIC_F32)
      if (p->K < 4 * PK) {  // fewer than four K-steps per tile: the undeferred variant (see the kernel)
        HMA_NT3S_CASE(HMA_A_BF16, HMA_EPI_BF16) HMA_NT3S_CASE(HMA_A_BF16, HMA_EPI_F32) HMA_NT3S_CASE(HMA_A_BF16, HMA_EPI_RESID)
        HMA_NT3S_CASE(HMA_A_F32, HMA_EPI_BF16) HMA_NT3S_CASE(HMA_A_F32, HMA_EPI_F32) HMA_NT3S_CASE(HMA_A_F32, HMA_EPI_RESID)
        return HMA_EINVAL;
      }
      HMA_NT3_ALL(HMA_A_BF16)
      HMA_NT3_ALL(HMA_A_F32)
      HMA_NT3_ALL(HMA_A_BF16_AFFINE)
      return HMA_EINVAL;
    }
    if (p->K % QK == 0) {
      const dim3 qgrid((unsigned)(total < 2 * n_cu ? total : 2 * n_cu));
#define HMA_NTQ_CASE(AK, EP)                                                                          \
  if (p->a_kind == AK && p->epi == EP) {                                                              \
    if ((rc = set_smem_bytes<gemm_nt_p2_kernel<AK, EP>>(Q_SMEM_BYTES))) return rc;                    \
    hipLaunchKernelGGL((gemm_nt_p2_kernel<AK, EP>), qgrid, dim3(256), Q_SMEM_BYTES, s, pa, tiles_m, tiles_n, total); \
    HMA_CHECK_LAUNCH();                                                                               \
    return 0;                                                                                         \
  }
#define HMA_NTQ_ALL(AK)                                                                               \
  HMA_NTQ_CASE(AK, HMA_EPI_GELU2) HMA_NTQ_CASE(AK, HMA_EPI_SILU2) HMA_NTQ_CASE(AK, HMA_EPI_DGELU)     \
  HMA_NTQ_CASE(AK, HMA_EPI_DSILU)
      HMA_NTQ_ALL(HMA_A_BF16)
      HMA_NTQ_ALL(HMA_A_F32)
      HMA_NTQ_ALL(HMA_A_BF16_AFFINE)
      return HMA_EINVAL;
    }
    return HMA_EINVAL;
  }
  if (p->drop_p != 0.f || p->ln_xhat) return HMA_EINVAL;
  // N % 256 != 0 (N = 128: the diffusion head's padded output layer): one 128 x 128 tile per workgroup
  const dim3 grid((unsigned)((p->M + BM - 1) / BM), (unsigned)(p->N / BN), (unsigned)(p->batch > 0 ? p->batch : 1));
  switch (p->a_kind) {
    case HMA_A_BF16:
      if ((rc = set_smem<gemm_nt_kernel<HMA_A_BF16>>())) return rc;
      hipLaunchKernelGGL(gemm_nt_kernel<HMA_A_BF16>, grid, dim3(256), SMEM_BYTES, s, pa);
      break;
    case HMA_A_F32:
      if ((rc = set_smem<gemm_nt_kernel<HMA_A_F32>>())) return rc;
      hipLaunchKernelGGL(gemm_nt_kernel<HMA_A_F32>, grid, dim3(256), SMEM_BYTES, s, pa);
      break;
    case HMA_A_BF16_AFFINE:
      if ((rc = set_smem<gemm_nt_kernel<HMA_A_BF16_AFFINE>>())) return rc;
      hipLaunchKernelGGL(gemm_nt_kernel<HMA_A_BF16_AFFINE>, grid, dim3(256), SMEM_BYTES, s, pa);
      break;
    default:
      return HMA_EINVAL;
  }
  HMA_CHECK_LAUNCH();
  return 0;
}

#define HMA_TN_CASE(YK, AK)                                                               \
  if (p->y_kind == YK && p->a_kind == AK) {                                               \
    if ((rc = set_smem<gemm_tn_kernel<YK, AK>>())) return rc;                               \
    hipLaunchKernelGGL((gemm_tn_kernel<YK, AK>), grid, dim3(256), SMEM_BYTES, s, q);      \
    HMA_CHECK_LAUNCH();                                                                   \
    return 0;                                                                             \
  }

// ---- LDS-DMA ring weight gradients: eligibility, split planning and launch of one or two problems
static bool tn_dma_eligible(const hma_gemm_tn_t& q) {
  const bool yf = q.y_kind == HMA_A_BF16_FRAG32, af = q.a_kind == HMA_A_BF16_FRAG32;
  const bool yh = q.y_kind == HMA_A_BF16_HEADBLK, ah = q.a_kind == HMA_A_BF16_HEADBLK;
  if ((yf || af || yh || ah) && q.batch > 1) return false;
  if ((yf && q.ldy != q.N) || (af && q.lda != q.K)) return false;  // the fragment order has no row pitch: the operand is the whole matrix
  // head-blocked: *_group_rows carries the rows per frame (a stage's 32 rows lie inside one frame), ld = 256 W says how many 256-column parts
  if (yh && (q.y_group_rows <= 0 || q.y_group_rows % DM_ROWS || q.M % q.y_group_rows || q.ldy % 256 || q.ldy != q.N)) return false;
  if (ah && (q.a_group_rows <= 0 || q.a_group_rows % DM_ROWS || q.M % q.a_group_rows || q.lda % 256 || q.lda != q.K)) return false;
  return q.N % WT == 0 && q.K % WT == 0 && (q.y_kind == HMA_A_BF16 || yf || yh) &&
         (q.a_kind == HMA_A_BF16 || q.a_kind == HMA_A_BF16_AFFINE || af || ah) && q.M > 0 && q.M % DM_ROWS == 0 && (yh || q.y_group_rows <= 0) && (ah || q.a_group_rows <= 0) && q.ldy % 8 == 0 && q.lda % 8 == 0 &&
         q.sY % 8 == 0 && q.sA % 8 == 0 && (reinterpret_cast<uintptr_t>(q.dY) & 15) == 0 && (reinterpret_cast<uintptr_t>(q.A) & 15) == 0;
}
// splits for a problem that may use at most `budget` workgroups; 0 if it cannot take the two-stage path
static int tn_plan_splits(const hma_gemm_tn_t& q, int budget) {
  const int64_t slabs = (q.M + 63) / 64;
  const int groups = (int)(q.N / WT) * (int)(q.K / WT) * (q.batch > 0 ? q.batch : 1);
  int splits = budget / groups;
  if (splits < 1) splits = 1;
  if (splits > slabs) splits = (int)slabs;
  while (splits > 1 && ((slabs + splits - 1) / splits) * (splits - 1) >= slabs) --splits;
  return splits > 1 ? splits : 0;
}
static tn_prob tn_make_prob(const hma_gemm_tn_t& q, float* ws, int splits) {
  tn_prob t;
  t.dY = q.dY; t.A = q.A; t.ws = ws;
  t.M = q.M; t.ldy = q.ldy; t.lda = q.lda; t.sY = q.sY; t.sA = q.sA;
  t.splits = splits; t.gn = (int)(q.N / WT); t.gk = (int)(q.K / WT);
  t.nb = splits * t.gn * t.gk * (q.batch > 0 ? q.batch : 1);
  t.colsum = (q.dBias != nullptr) || q.a_kind == HMA_A_BF16_AFFINE;
  t.yfrag = q.y_kind == HMA_A_BF16_FRAG32;
  t.afrag = q.a_kind == HMA_A_BF16_FRAG32;
  t.yhb = q.y_kind == HMA_A_BF16_HEADBLK ? (int)q.y_group_rows : 0;
  t.ahb = q.a_kind == HMA_A_BF16_HEADBLK ? (int)q.a_group_rows : 0;
  return t;
}
template <bool TR>
static int tn_dma_launch(hipStream_t s, const tn_pair_args& a) {
  bool cs = false;
  unsigned nb = 0;
  for (int i = 0; i < a.n; ++i) {
    cs = cs || a.q[i].colsum;
    nb += (unsigned)a.q[i].nb;
  }
  const dim3 grid(nb);
  int rc;
#ifdef HMA_PROF
  {
    static const int abl = getenv("HMA_GEMM_TN_ABLATE") ? atoi(getenv("HMA_GEMM_TN_ABLATE")) : 0;  // debug build only
#define HMA_TNM_ABL(A)                                                                                  \
  case A:                                                                                               \
    if ((rc = set_smem_bytes<gemm_tn_dma_kernel<true, true, A>>(DM_SMEM_BYTES))) return rc;             \
    hipLaunchKernelGGL((gemm_tn_dma_kernel<true, true, A>), grid, dim3(512), DM_SMEM_BYTES, s, a);      \
    HMA_CHECK_LAUNCH();                                                                                 \
    return 0;
    switch (abl) {
      HMA_TNM_ABL(1) HMA_TNM_ABL(2) HMA_TNM_ABL(4) HMA_TNM_ABL(8) HMA_TNM_ABL(16) HMA_TNM_ABL(6) HMA_TNM_ABL(7) HMA_TNM_ABL(15)
      HMA_TNM_ABL(3) HMA_TNM_ABL(9) HMA_TNM_ABL(24)
      default: break;
    }
  }
#endif
  if (cs) {
    if ((rc = set_smem_bytes<gemm_tn_dma_kernel<TR, true>>(DM_SMEM_BYTES))) return rc;
    hipLaunchKernelGGL((gemm_tn_dma_kernel<TR, true>), grid, dim3(512), DM_SMEM_BYTES, s, a);
  } else {
    if ((rc = set_smem_bytes<gemm_tn_dma_kernel<TR, false>>(DM_SMEM_BYTES))) return rc;
    hipLaunchKernelGGL((gemm_tn_dma_kernel<TR, false>), grid, dim3(512), DM_SMEM_BYTES, s, a);
  }
  HMA_CHECK_LAUNCH();
  return 0;
}
// one reduction launch for the n problems of a ring-kernel launch
static int tn_dma_reduce(hipStream_t s, const hma_gemm_tn_t* const* probs, const tn_pair_args& a) {
  tn_reduce_args ra;
  ra.n = a.n;
  unsigned gy = 0;
  for (int i = 0; i < TN_MAXP; ++i) {
    const int j = i < a.n ? i : 0;
    ra.p[i] = *probs[j];
    ra.p[i].splits = a.q[j].splits;
    ra.p[i].ws = a.q[j].ws;
    ra.gn[i] = a.q[j].gn; ra.gk[i] = a.q[j].gk; ra.affine[i] = probs[j]->a_kind == HMA_A_BF16_AFFINE;
    if (i < a.n) gy += (unsigned)(a.q[j].gn * a.q[j].gk * (probs[j]->batch > 0 ? probs[j]->batch : 1));
    ra.gy_end[i] = (int)gy;
  }
  hipLaunchKernelGGL(tn_reduce_native_kernel, dim3((unsigned)(WT * WT / 256), gy), dim3(256), 0, s, ra);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_gemm_tn(void* stream, const hma_gemm_tn_t* p) {
  if (!p || !p->dY || !p->A || !p->dW) return HMA_EINVAL;
  if (p->M <= 0) return 0;
  if (p->N % 128 != 0 || p->K % 128 != 0) return HMA_EINVAL;
  if (p->y_kind == HMA_A_BF16_AFFINE) return HMA_EINVAL;
  if (p->a_kind == HMA_A_BF16_AFFINE && (!p->gamma || !p->beta)) return HMA_EINVAL;
  if (p->dgamma && (p->a_kind != HMA_A_BF16_AFFINE || !p->dbeta || !p->w_master)) return HMA_EINVAL;
  hma_gemm_tn_t q = *p;
  q._pad0 = 0;
#ifdef HMA_PROF
  static const int tn_ablate = getenv("HMA_GEMM_TN_ABLATE") ? atoi(getenv("HMA_GEMM_TN_ABLATE")) : 0;  // debug build only
  q._pad0 = tn_ablate;
#endif
  const int64_t slabs = (q.M + 63) / 64;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc;
  if (q.N % WT == 0 && q.K % WT == 0) {
    const int gn = (int)(q.N / WT), gk = (int)(q.K / WT);
    const int nb = q.batch > 0 ? q.batch : 1;
    int splits = 256 / (gn * gk * nb);  // one resident workgroup per CU
    if (splits < 1) splits = 1;
    if (splits > slabs) splits = (int)slabs;
    // every split must own at least one 64-row slab (the two-stage reduction reads every partial)
    while (splits > 1 && ((slabs + splits - 1) / splits) * (splits - 1) >= slabs) --splits;
    q.splits = splits;
    const int nblocks = gn * gk * nb * splits;
    const dim3 wgrid((unsigned)nblocks);
    // the workspace path needs every split to own a full slice (no early-exit workgroups) and a grid the
    // XCD remap leaves in (batch, split, group) order
    const int64_t per = (slabs + splits - 1) / splits;
    if (q.ws && (q.ws_elems < (int64_t)nblocks * WT * WT || per * (splits - 1) >= slabs || splits == 1)) q.ws = nullptr;
    const dim3 rgrid((unsigned)(WT * WT / 512), (unsigned)(gn * gk * nb));
    // bf16 x bf16 with a workspace: the LDS-DMA ring kernel
    const bool bias_room = q.ws && q.ws_elems >= (int64_t)nblocks * (WT * WT + WT);
    if (q.ws && tn_dma_eligible(q) && (bias_room || (!q.dBias && q.a_kind != HMA_A_BF16_AFFINE))) {
      tn_pair_args args;
      args.n = 1;
      for (int i = 0; i < TN_MAXP; ++i) args.q[i] = tn_make_prob(q, q.ws, splits);
#ifdef HMA_PROF
#define HMA_TND_ABL(A)                                                                                        \
  case A:                                                                                                     \
    if ((rc = set_smem_bytes<gemm_tn_dma_kernel<true, true, A>>(DM_SMEM_BYTES))) return rc;                   \
    hipLaunchKernelGGL((gemm_tn_dma_kernel<true, true, A>), wgrid, dim3(512), DM_SMEM_BYTES, s, args);        \
    break;
      if (tn_ablate) {
        args.q[0].colsum = 1;
        switch (tn_ablate) {
          HMA_TND_ABL(1) HMA_TND_ABL(2) HMA_TND_ABL(4) HMA_TND_ABL(8) HMA_TND_ABL(16) HMA_TND_ABL(6) HMA_TND_ABL(7) HMA_TND_ABL(15)
          HMA_TND_ABL(3) HMA_TND_ABL(9) HMA_TND_ABL(24)
          default: return HMA_EINVAL;
        }
        HMA_CHECK_LAUNCH();
      } else
#endif
      if ((rc = tn_dma_launch<true>(s, args))) return rc;
      const hma_gemm_tn_t* one[1] = {&q};
      return tn_dma_reduce(s, one, args);
    }
    if (q.dgamma) return HMA_EINVAL;  // the folded-affine gradients exist on the LDS-DMA path only
#define HMA_TNW_CASE(YK, AK)                                                                        \
  if (q.y_kind == YK && q.a_kind == AK) {                                                            \
    if ((rc = set_smem_bytes<gemm_tn_wide_kernel<YK, AK>>(W_SMEM_BYTES))) return rc;                 \
    hipLaunchKernelGGL((gemm_tn_wide_kernel<YK, AK>), wgrid, dim3(512), W_SMEM_BYTES, s, q, gn, gk); \
    HMA_CHECK_LAUNCH();                                                                              \
    if (q.ws) {                                                                                      \
      hipLaunchKernelGGL(tn_reduce_kernel, rgrid, dim3(256), 0, s, q, gn, gk);                       \
      HMA_CHECK_LAUNCH();                                                                            \
    }                                                                                                \
    return 0;                                                                                        \
  }
    HMA_TNW_CASE(HMA_A_BF16, HMA_A_BF16)
    HMA_TNW_CASE(HMA_A_BF16, HMA_A_F32)
    HMA_TNW_CASE(HMA_A_BF16, HMA_A_BF16_AFFINE)
    HMA_TNW_CASE(HMA_A_F32, HMA_A_BF16)
    HMA_TNW_CASE(HMA_A_F32, HMA_A_F32)
    HMA_TNW_CASE(HMA_A_F32, HMA_A_BF16_AFFINE)
    return HMA_EINVAL;
  }
  if (q.dgamma) return HMA_EINVAL;
  if (q.splits <= 0) q.splits = 1;
  if (q.splits > slabs) q.splits = (int32_t)slabs;
  const dim3 grid((unsigned)q.splits, (unsigned)((q.N / 128) * (q.K / 128)), (unsigned)(q.batch > 0 ? q.batch : 1));
  HMA_TN_CASE(HMA_A_BF16, HMA_A_BF16)
  HMA_TN_CASE(HMA_A_BF16, HMA_A_F32)
  HMA_TN_CASE(HMA_A_BF16, HMA_A_BF16_AFFINE)
  HMA_TN_CASE(HMA_A_F32, HMA_A_BF16)
  HMA_TN_CASE(HMA_A_F32, HMA_A_F32)
  HMA_TN_CASE(HMA_A_F32, HMA_A_BF16_AFFINE)
  return HMA_EINVAL;
}

#ifndef TN_MULTI_EQUAL_ROWS
#define TN_MULTI_EQUAL_ROWS 1
#endif
static double tn_weight(const hma_gemm_tn_t& q, int n) {
  const double b = q.batch > 0 ? q.batch : 1;
  if (n > 2 && TN_MULTI_EQUAL_ROWS) return (double)q.M * (double)((q.N / WT) * (q.K / WT)) * b;
  return (double)q.M * (double)(q.N + q.K) * b;
}

extern "C" int hma_gemm_tn_multi(void* stream, const hma_gemm_tn_t* const* probs, int32_t n) {
  if (!probs || n < 1 || n > TN_MAXP) return HMA_EINVAL;
  for (int i = 0; i < n; ++i)
    if (!probs[i]) return HMA_EINVAL;
  const int64_t need = (int64_t)256 * (WT * WT + WT);
  bool ok = n >= 2;
  double wsum = 0.0;
  for (int i = 0; i < n && ok; ++i) {
    const hma_gemm_tn_t& q = *probs[i];
    ok = q.dY && q.A && q.dW && q.ws && q.ws == probs[0]->ws && q.ws_elems >= need && tn_dma_eligible(q) &&
         !(q.a_kind == HMA_A_BF16_AFFINE && (!q.gamma || !q.beta)) && !(q.dgamma && (!q.dbeta || !q.w_master));
    // workgroups in proportion to the operand bytes of each problem (two problems: hma_gemm_tn_pair's historical split) or -- more
    // than two, TN_MULTI_EQUAL_ROWS -- to rows x 256 x 256 output blocks: a workgroup streams the rows of ONE block, so equal rows per
    // workgroup means every workgroup of the launch runs the same number of 32-row stages
    wsum += tn_weight(q, n);
  }
#ifdef HMA_PROF
  if (getenv("HMA_GEMM_TN_NOPAIR")) ok = false;  // debug build only
#endif
  int splits[TN_MAXP] = {0};
  if (ok) {
    int total = 0, budget0 = 0;
    for (int i = 0; i < n && ok; ++i) {
      const hma_gemm_tn_t& q = *probs[i];
      const int groups = (int)(q.N / WT) * (int)(q.K / WT) * (q.batch > 0 ? q.batch : 1);
      const double share = 256.0 * tn_weight(q, n) / wsum;
      int budget = (int)share;
      if (n == 2) {  // (two problems: rounded, the second takes the rest -- the split hma_gemm_tn_pair has always made)
        if (i == 0) {
          budget0 = (int)(share + 0.5);
          budget0 = budget0 < 1 ? 1 : (budget0 > 255 ? 255 : budget0);
          budget = budget0;
        } else {
          budget = 256 - budget0;
        }
      }
      if (budget < groups) budget = groups;
      splits[i] = tn_plan_splits(q, budget);
      ok = splits[i] > 0;
      total += splits[i] * groups;
    }
    ok = ok && total <= 256;
  }
  if (!ok) {
    for (int i = 0; i < n; ++i)
      if (int rc = hma_gemm_tn(stream, probs[i])) return rc;
    return 0;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  tn_pair_args args;
  args.n = n;
  int64_t off = 0;
  for (int i = 0; i < TN_MAXP; ++i) {
    const int j = i < n ? i : 0;
    args.q[i] = tn_make_prob(*probs[j], probs[0]->ws + (i < n ? off : 0) * (WT * WT + WT), splits[j]);
    if (i < n) off += args.q[i].nb;
  }
  int rc;
  if ((rc = tn_dma_launch<true>(s, args))) return rc;
  return tn_dma_reduce(s, probs, args);
}

extern "C" int hma_gemm_tn_pair(void* stream, const hma_gemm_tn_t* a, const hma_gemm_tn_t* b) {
  if (!a || !b) return HMA_EINVAL;
  const hma_gemm_tn_t* two[2] = {a, b};
  return hma_gemm_tn_multi(stream, two, 2);
}

#ifdef HMA_PROF
// debug build only (never part of the shipped ABI): read and clear the phase timers
extern "C" int hma_debug_prof(unsigned long long* out16) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  unsigned long long z[16] = {0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
#endif
