// Probe for DESIGN section 9 item 0 (second candidate): chain B's MLP steps -- fc1 | fc2 bundle pairs, one barrier per pair; an fc1 step
// is 16 MFMAs then ~100 VALU instructions per wave (8 GELUs per lane), an fc2 step 16 MFMAs -- with the two compute waves of a SIMD
//   LOCK : in the same order                        (fc1 MFMAs, GELU, fc2 MFMAs)            -- the shipped organisation
//   LAG  : waves 4..6 half a pair behind waves 0..3 (fc2 MFMAs of the PREVIOUS pair, fc1 MFMAs, GELU), which needs the previous
//          pair's fc2 bundle one barrier longer: a ring of 8 slots (4 pairs) instead of 6.
// Skeleton only (ring, fragment reads, MFMAs, a GELU-sized VALU chain, barriers), 7 compute waves x 16 rows + loader, 64 steps per
// tile, 6 rounds.   hipcc --offload-arch=gfx950 -O3 -o lag_wave lag_wave.hip && ./lag_wave
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
#define LDSP(T) __attribute__((address_space(3))) T
constexpr int SLOT = 16384;

__device__ __forceinline__ void glds16(const void* g, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory");
}
// a GELU-sized dependent-but-wide VALU block on the 8 values of a lane: ~12 operations per value
__device__ __forceinline__ void valu_block(f32x4_t& c0, f32x4_t& c1) {
  float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], v[e] * 0.25f, 0.5f) - 0.125f * v[e];
#pragma unroll
  for (int e = 0; e < 4; ++e) { c0[e] = v[e]; c1[e] = v[4 + e]; }
}
__device__ __forceinline__ void mma16(LDSP(char)* wb, const bf16x8_t (&b)[8], f32x4_t& c0, f32x4_t& c1) {
  bf16x8_t a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = __builtin_bit_cast(bf16x8_t, *(LDSP(u32x4_t)*)(wb + i * 1024));
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * k], b[k], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * k + 1], b[k], c1, 0, 0, 0);
  }
}

template <int NG, bool LAG, bool VALU>   // NG = pairs the ring holds
__global__ __launch_bounds__(512, 1) void skel(const char* __restrict__ w, float* __restrict__ out, int pairs, int nbundles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  LDSP(char)* lds = (LDSP(char)*)smem;
  const uint32_t lds_b = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave == 7) {
    auto issue = [&](int g) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const char* src = w + (size_t)((g * 2 + s) % nbundles) * SLOT + lane * 16;
        const uint32_t dst = lds_b + ((g % NG) * 2 + s) * SLOT;
#pragma unroll
        for (int i = 0; i < 16; ++i) glds16(src + i * 1024, dst + i * 1024);
      }
    };
    issue(0);
    if (pairs > 1) issue(1);
    for (int g = 0; g < pairs; ++g) {
      if (g + 1 < pairs) asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (g + 2 < pairs) issue(g + 2);  // NG = 3: the slots of pair g - 1; NG = 4: of pair g - 2 (pair g - 1 stays readable)
    }
    if (LAG) __builtin_amdgcn_s_barrier();  // (the lagging waves' last half pair)
    return;
  }
  bf16x8_t b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) b[i] = __builtin_bit_cast(bf16x8_t, u32x4_t{(uint32_t)lane, (uint32_t)i, 0x3f803f80u, 0x3f803f80u});
  LDSP(char)* ring = lds + lane * 16;
  f32x4_t c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
  const bool lag = LAG && wave >= 4;  // waves w and w + 4 share a SIMD
  if (!lag) {
    for (int g = 0; g < pairs; ++g) {
      __builtin_amdgcn_s_barrier();
      LDSP(char)* wb = ring + (g % NG) * 2 * SLOT;
      mma16(wb, b, c0, c1);                       // fc1 block g
      if (VALU) valu_block(c0, c1);               // GELU
      mma16(wb + SLOT, b, d0, d1);                // fc2 block g
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (LAG) __builtin_amdgcn_s_barrier();
  } else {
    for (int g = 0; g <= pairs; ++g) {
      __builtin_amdgcn_s_barrier();
      if (g > 0) mma16(ring + ((g - 1) % NG) * 2 * SLOT + SLOT, b, d0, d1);   // fc2 block g - 1 (its bundle is still in the ring)
      if (g < pairs) {
        mma16(ring + (g % NG) * 2 * SLOT, b, c0, c1);                         // fc1 block g
        if (VALU) valu_block(c0, c1);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  const float sum = c0[0] + c1[1] + d0[2] + d1[3];
  if (sum == 123.456f) out[blockIdx.x * 512 + threadIdx.x] = sum;
}

template <int NG, bool LAG, bool VALU>
float run(const char* w, float* out, int pairs, int nb) {
  const int smem = NG * 2 * SLOT;
  hipFuncSetAttribute(reinterpret_cast<const void*>(skel<NG, LAG, VALU>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((skel<NG, LAG, VALU>), dim3(256), dim3(512), smem, 0, w, out, pairs, nb);
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((skel<NG, LAG, VALU>), dim3(256), dim3(512), smem, 0, w, out, pairs, nb);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  return ms * 1e3f / reps;
}

int main() {
  const int nb = 64;
  char* w;
  float* out;
  hipMalloc(&w, (size_t)nb * SLOT);
  hipMalloc(&out, 256 * 512 * 4);
  std::vector<uint16_t> h((size_t)nb * SLOT / 2, 0x3c00);
  hipMemcpy(w, h.data(), (size_t)nb * SLOT, hipMemcpyHostToDevice);
  const int pairs = 6 * 32;  // 6 rounds of a tile's 32 fc1 | fc2 pairs
  const float a = run<3, false, false>(w, out, pairs, nb), b = run<3, false, true>(w, out, pairs, nb);
  const float c = run<4, false, true>(w, out, pairs, nb), d = run<4, true, true>(w, out, pairs, nb), e = run<4, true, false>(w, out, pairs, nb);
  printf("chain B's 64 MLP steps per tile x 6 rounds, skeleton only (us; cycles per fc1 | fc2 pair at 2.3 GHz):\n");
  printf("  lock step, no VALU block, 6-slot ring : %7.1f  (%5.0f)\n", a, a * 2300 / pairs);
  printf("  lock step, GELU-sized VALU block      : %7.1f  (%5.0f)\n", b, b * 2300 / pairs);
  printf("  the same with an 8-slot ring          : %7.1f  (%5.0f)\n", c, c * 2300 / pairs);
  printf("  waves 4..6 half a pair behind, 8 slots: %7.1f  (%5.0f)\n", d, d * 2300 / pairs);
  printf("  the same without the VALU block       : %7.1f  (%5.0f)\n", e, e * 2300 / pairs);
  return 0;
}
