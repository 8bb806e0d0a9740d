// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the HMA hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // one MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define HMA_WAVE 64

#define HMA_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return -(int)e__;                 \
  } while (0)

namespace hma {

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// round-to-nearest-even pack of two floats into one dword of bf16 (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  bf16x2_t r;
  r[0] = (__bf16)a;
  r[1] = (__bf16)b;
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint16_t to_bf16(float a) {
  __bf16 r = (__bf16)a;
  return __builtin_bit_cast(uint16_t, r);
}
__device__ __forceinline__ float from_bf16(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = bf16_lo(v.x); f[1] = bf16_hi(v.x);
  f[2] = bf16_lo(v.y); f[3] = bf16_hi(v.y);
  f[4] = bf16_lo(v.z); f[5] = bf16_hi(v.z);
  f[6] = bf16_lo(v.w); f[7] = bf16_hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack_bf16(f[0], f[1]);
  v.y = pack_bf16(f[2], f[3]);
  v.z = pack_bf16(f[4], f[5]);
  v.w = pack_bf16(f[6], f[7]);
  return v;
}

// Sum / maximum over the 64 lanes, in every lane: four DPP steps inside the rows of 16 (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror,
// row_mirror) and the four row results through v_readlane.  (Six __shfl_xor = six ds_bpermute round trips of ~120 cycles each were
// half the time of the row kernels that reduce once or twice per row -- the action stem's forward 83 -> 37 us, round 6.)  Call with
// all 64 lanes active.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float readlane_f32(float x, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f32<0xB1>(v);
  v += dpp_f32<0x4E>(v);
  v += dpp_f32<0x141>(v);
  v += dpp_f32<0x140>(v);
  return (readlane_f32(v, 0) + readlane_f32(v, 16)) + (readlane_f32(v, 32) + readlane_f32(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v));
  v = fmaxf(v, dpp_f32<0x4E>(v));
  v = fmaxf(v, dpp_f32<0x141>(v));
  v = fmaxf(v, dpp_f32<0x140>(v));
  return fmaxf(fmaxf(readlane_f32(v, 0), readlane_f32(v, 16)), fmaxf(readlane_f32(v, 32), readlane_f32(v, 48)));
}

// exact-erf GELU (nn.GELU() default, hma/model/st_transformer.py:20) and its derivative.
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 rounding of the outputs):
// one v_rcp + one v_exp + 7 FMAs instead of ocml's branchy erff -- the GELU epilogues were VALU-bound.
// The exp(-u^2/2) it needs is also the Gaussian of the derivative, so dgelu costs no second exp.
__device__ __forceinline__ void gelu_parts(float u, float& cdf, float& gauss) {
  const float x = fabsf(u) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * x);
  gauss = __builtin_amdgcn_exp2f(-0.72134752044448170f * u * u);  // exp(-u^2 / 2)
  float poly = 1.061405429f;
  poly = poly * t - 1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t - 0.284496736f;
  poly = poly * t + 0.254829592f;
  const float erfc_half = 0.5f * poly * t * gauss;  // 0.5 * erfc(|u| / sqrt 2)
  cdf = u >= 0.f ? 1.0f - erfc_half : erfc_half;      // Phi(u)
}
__device__ __forceinline__ float gelu_f(float u) {
  float cdf, gauss;
  gelu_parts(u, cdf, gauss);
  return u * cdf;
}
__device__ __forceinline__ float dgelu_f(float u) {
  float cdf, gauss;
  gelu_parts(u, cdf, gauss);
  return cdf + u * 0.3989422804014327f * gauss;
}
// ---- GELU (exact-erf, nn.GELU() of st_transformer.py:20) for 8 accumulator values at a time ---------------------------
// gelu(u) = max(u, 0) - |u| h(|u|), h(a) = Phi(-a) = exp2(q(a)) with q a degree-6 polynomial fit of log2(Phi(-a)) on [0, 6]
// (|error of h| <= 2e-5, |error of gelu| <= 7e-6: 1/500 of a bf16 ulp of the values it produces) and a clamped to 6
// (a h(a) < 6e-9 beyond).  Per element: 1 v_med3, 6 FMAs (in pairs: v_pk_fma_f32), ONE transcendental, 1 max, 1 FMA.
// The A&S 7.1.26 form used by the unfused epilogues costs a v_rcp and a v_exp (quarter rate) plus 11 plain operations,
// and on gfx950 a wave64 VALU instruction is 4 cycles (packed: 8): the GELU is the VALU budget of these kernels.
// Written stage by stage over all values, with an empty asm that takes and returns a stage's values as a join point:
// left alone, the compiler serialises element pairs with dependency nops.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#define HMA_PIN4(a, o) asm volatile("" : "+v"(a[o]), "+v"(a[o + 1]), "+v"(a[o + 2]), "+v"(a[o + 3]))
#define HMA_PIN8(a, o)                                                                                                   \
  asm volatile("" : "+v"(a[o]), "+v"(a[o + 1]), "+v"(a[o + 2]), "+v"(a[o + 3]), "+v"(a[o + 4]), "+v"(a[o + 5]), "+v"(a[o + 6]), \
               "+v"(a[o + 7]))
#define HMA_PIN16(a) do { HMA_PIN8(a, 0); HMA_PIN8(a, 8); } while (0)
constexpr float GQ6 = 2.2999249267741106e-05f, GQ5 = -0.0006114901625551283f, GQ4 = 0.007200188934803009f,
                GQ3 = -0.05120821297168732f, GQ2 = -0.46122226119041443f, GQ1 = -1.150214433670044f, GQ0 = -1.000058889389038f;

// ph[e] = Phi(-|u[e]|), N values (N = 8 or 16)
template <int N>
__device__ __forceinline__ void gelu_phi_neg(const float (&u)[N], float (&ph)[N]) {
  f32x2_t a2[N / 2], q2[N / 2];
#pragma unroll
  for (int i = 0; i < N / 2; ++i)
    a2[i] = f32x2_t{__builtin_amdgcn_fmed3f(fabsf(u[2 * i]), 0.f, 6.0f), __builtin_amdgcn_fmed3f(fabsf(u[2 * i + 1]), 0.f, 6.0f)};
  if constexpr (N == 8) HMA_PIN4(a2, 0); else HMA_PIN8(a2, 0);
#define HMA_HORNER(c)                                                                                                 \
  _Pragma("unroll") for (int i = 0; i < N / 2; ++i) q2[i] = __builtin_elementwise_fma(q2[i], a2[i], (f32x2_t{c, c})); \
  if constexpr (N == 8) HMA_PIN4(q2, 0); else HMA_PIN8(q2, 0);
#pragma unroll
  for (int i = 0; i < N / 2; ++i) q2[i] = __builtin_elementwise_fma(f32x2_t{GQ6, GQ6}, a2[i], f32x2_t{GQ5, GQ5});
  if constexpr (N == 8) HMA_PIN4(q2, 0); else HMA_PIN8(q2, 0);
  HMA_HORNER(GQ4) HMA_HORNER(GQ3) HMA_HORNER(GQ2) HMA_HORNER(GQ1) HMA_HORNER(GQ0)
#undef HMA_HORNER
#pragma unroll
  for (int e = 0; e < N; ++e) ph[e] = __builtin_amdgcn_exp2f(q2[e >> 1][e & 1]);
  if constexpr (N == 8) HMA_PIN8(ph, 0); else HMA_PIN16(ph);
}
// h[e] = gelu(u[e])
template <int N>
__device__ __forceinline__ void gelu_n(const float (&u)[N], float (&h)[N]) {
  float ph[N], r[N];
  gelu_phi_neg<N>(u, ph);
#pragma unroll
  for (int e = 0; e < N; ++e) r[e] = fmaxf(u[e], 0.f);
  if constexpr (N == 8) HMA_PIN8(r, 0); else HMA_PIN16(r);
#pragma unroll
  for (int e = 0; e < N; ++e) h[e] = __builtin_fmaf(-fabsf(u[e]), ph[e], r[e]);
  if constexpr (N == 8) HMA_PIN8(h, 0); else HMA_PIN16(h);
}
// hg[e] = gelu(u[e]) (the forward's values, bit for bit) and du[e] = d[e] * gelu'(u[e]), gelu'(u) = Phi(u) + u phi(u)
template <int N>
__device__ __forceinline__ void gelu_bwd_n(const float (&u)[N], const float (&d)[N], float (&hg)[N], float (&du)[N]) {
  float ph[N], gs[N], r[N];
  gelu_phi_neg<N>(u, ph);
#pragma unroll
  for (int e = 0; e < N; ++e) gs[e] = -0.72134752044448170f * u[e] * u[e];
  if constexpr (N == 8) HMA_PIN8(gs, 0); else HMA_PIN16(gs);
#pragma unroll
  for (int e = 0; e < N; ++e) gs[e] = __builtin_amdgcn_exp2f(gs[e]);
  if constexpr (N == 8) HMA_PIN8(gs, 0); else HMA_PIN16(gs);
#pragma unroll
  for (int e = 0; e < N; ++e) r[e] = fmaxf(u[e], 0.f);
  if constexpr (N == 8) HMA_PIN8(r, 0); else HMA_PIN16(r);
#pragma unroll
  for (int e = 0; e < N; ++e) hg[e] = __builtin_fmaf(-fabsf(u[e]), ph[e], r[e]);
  if constexpr (N == 8) HMA_PIN8(hg, 0); else HMA_PIN16(hg);
  // Phi(u) = 0.5 + copysign(0.5 - h, u)
#pragma unroll
  for (int e = 0; e < N; ++e) ph[e] = 0.5f + __builtin_copysignf(0.5f - ph[e], u[e]);
  if constexpr (N == 8) HMA_PIN8(ph, 0); else HMA_PIN16(ph);
#pragma unroll
  for (int e = 0; e < N; ++e) r[e] = __builtin_fmaf(u[e] * 0.3989422804014327f, gs[e], ph[e]);
  if constexpr (N == 8) HMA_PIN8(r, 0); else HMA_PIN16(r);
#pragma unroll
  for (int e = 0; e < N; ++e) du[e] = d[e] * r[e];
  if constexpr (N == 8) HMA_PIN8(du, 0); else HMA_PIN16(du);
}

// ---- order-independent loss sums.  A launch's per-wave fp32 partials are added as 64-bit FIXED-POINT integers (2^-32 units; integer
// addition is associative, so the total does not depend on the order the waves arrive in -- fp32 atomics moved the reported loss by
// ~1e-5 between two replays of one graph), and the last wave of the launch to arrive (a ticket counter) writes the total as stats[0].
// Layout of a `stats` buffer (HMA_CE_STATS_FLOATS = 8 floats, zeroed by the caller before the first launch that adds to it):
// [0] loss sum (fp32, written by the last wave) [1] hits (integer-valued: fp32 atomics are exact) [2] masked rows [3] ticket (u32)
// [4..5] the fixed-point accumulator (i64) [6..7] spare.  `waves` = how many waves of the launch call this (each exactly once).
__device__ __forceinline__ void det_loss_add(float* stats, float v, int lane, unsigned waves) {
  if (lane != 0) return;
  if (v != 0.f) {
    const long long q = __double2ll_rn((double)v * 4294967296.0);
    atomicAdd(reinterpret_cast<unsigned long long*>(stats + 4), (unsigned long long)q);
  }
  __threadfence();
  const unsigned t = atomicAdd(reinterpret_cast<unsigned*>(stats + 3), 1u);
  if (t == waves - 1) {
    __threadfence();
    const long long tot = (long long)atomicAdd(reinterpret_cast<unsigned long long*>(stats + 4), 0ull);
    stats[0] = (float)((double)tot * (1.0 / 4294967296.0));
    *reinterpret_cast<unsigned*>(stats + 3) = 0u;  // (the next launch into the same buffer counts from zero again)
  }
}

__device__ __forceinline__ float silu_f(float u) { return u / (1.0f + __expf(-u)); }
__device__ __forceinline__ float dsilu_f(float u) {
  const float s = 1.0f / (1.0f + __expf(-u));
  return s * (1.0f + u * (1.0f - s));
}

// Counter-based dropout mask: one 32-bit hash (murmur3 finaliser over seed, salt and the element PAIR index) decides two neighbouring
// elements, 16 bits each: keep iff that half >= p * 2^16 (the drop probability is p rounded down to a multiple of 2^-16).  Forward and
// backward call it with the same arguments instead of storing the mask.  The integer multiplies are quarter rate on gfx950, so the
// kernels that mask whole rows (chain B, hma_mlp_bwd) hash once per pair with drop_keep2.
__device__ __forceinline__ uint32_t drop_hash(uint32_t seed, int salt, int64_t pair) {
  uint32_t x = (uint32_t)pair ^ ((uint32_t)((uint64_t)pair >> 32) * 0x9E3779B9u);
  x ^= seed + 0x7F4A7C15u * (uint32_t)(salt + 1);
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bool drop_keep(uint32_t seed, int salt, int64_t idx, uint32_t thresh) {
  const uint32_t x = drop_hash(seed, salt, idx >> 1);
  return ((idx & 1) ? (x >> 16) : (x & 0xFFFFu)) >= (thresh >> 16);
}
// elements idx (even) and idx + 1
__device__ __forceinline__ void drop_keep2(uint32_t seed, int salt, int64_t idx, uint32_t thresh, bool& k0, bool& k1) {
  const uint32_t x = drop_hash(seed, salt, idx >> 1);
  k0 = (x & 0xFFFFu) >= (thresh >> 16);
  k1 = (x >> 16) >= (thresh >> 16);
}
__device__ __forceinline__ uint32_t drop_thresh(float p) { return (uint32_t)fminf(p * 4294967296.0f, 4294967040.0f); }
// the rescale that goes with the mask: 1 / (1 - the probability the 16-bit compare actually drops with) -- p rounded down to a multiple
// of 2^-16 (for p < 2^-16 nothing is dropped and nothing is scaled)
__device__ __forceinline__ float drop_scale(float p) { return 1.0f / (1.0f - (float)(drop_thresh(p) >> 16) * (1.0f / 65536.0f)); }

__device__ __forceinline__ f32x16_t mfma32(const bf16x8_t& a, const bf16x8_t& b, const f32x16_t& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// row of element r (0..15) of a 32x32 MFMA accumulator held by a lane with hi = lane >> 5
__device__ __forceinline__ int mfma32_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }


#define HMA_LDS(T) __attribute__((address_space(3))) T

// One LDS-DMA piece: 64 lanes x 16 bytes from each lane's `src` to LDS bytes [dst, dst + 1024) in lane order.  Issued
// from inline asm so that hipcc does not count it: with the builtin the compiler drains vmcnt(0) before the next LDS
// read and nothing stays in flight.  The waits are counted `s_waitcnt vmcnt(N)` written by hand at the use sites.
__device__ __forceinline__ void glds16(const void* src, uint32_t dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(dst)
      : "memory");
}

// The same piece addressed as a wave-uniform base (an SGPR pair) + this lane's 32-bit byte offset: no vector arithmetic per piece when
// the lane's offset is a loop invariant (the weight-gradient ring: a piece's address is (stage, row pair) -> uniform, (lane) -> constant)
__device__ __forceinline__ void glds16s(const void* sbase, uint32_t voff, uint32_t dst) {
  const uint64_t a = (uint64_t)(uintptr_t)sbase;
  const uint64_t au = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(au), "s"(dst)
      : "memory");
}

// Four consecutive pieces (4 KB of global memory -> 4 KB of LDS): the instruction's immediate offset advances the global
// AND the LDS address, so one M0 set-up serves all four.
__device__ __forceinline__ void glds16x4(const void* src, uint32_t dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "global_load_lds_dwordx4 %1, off offset:1024\n\t"
      "global_load_lds_dwordx4 %1, off offset:2048\n\t"
      "global_load_lds_dwordx4 %1, off offset:3072\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(dst)
      : "memory");
}

}  // namespace hma
