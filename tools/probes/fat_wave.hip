// Probe for DESIGN section 9 item 0: what does the LDS + MFMA + barrier skeleton of a row-local chain cost with FAT waves?
// Both forms stream the same 16 KB weight bundles through the same 6-slot LDS-DMA ring (loader wave, counted vmcnt, one s_barrier per
// two bundles) and read every fragment of a bundle once per compute wave and step; no row loads, no stores, no GELU / LayerNorm work:
//   THIN: 7 compute waves x 16 token rows, 16 x v_mfma_f32_16x16x32_bf16 per step (two accumulator chains)        -> 112 rows per tile
//   FAT : 3 compute waves x 32 token rows, 16 x v_mfma_f32_32x32x16_bf16 per step (two accumulator chains), 4 waves -> 96 rows per tile
// The workload is chain B's: 96 steps per tile, M = 163 840 rows -> 1 463 thin tiles (6 rounds on 256 CUs) or 1 707 fat ones (7 rounds).
//   hipcc --offload-arch=gfx950 -O3 -o fat_wave fat_wave.hip && ./fat_wave
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
#define LDSP(T) __attribute__((address_space(3))) T

constexpr int SLOT = 16384, NS = 6, PB = 2;

__device__ __forceinline__ void glds16(const void* g, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory");
}

template <int NCW, bool FAT, bool PIPE = false, int TMODE = 0>
__global__ __launch_bounds__((NCW + 1) * 64, 1) void skel(const char* __restrict__ w, float* __restrict__ out, int rounds, int steps,
                                                         int nbundles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  LDSP(char)* lds = (LDSP(char)*)smem;
  const uint32_t lds_b = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int total = rounds * steps, groups = total / PB;
  if (wave == NCW) {
    // ---- loader: group g = bundles g * PB .. + PB - 1 into slots (g % 3) * PB ..; two groups ahead of the compute waves
    auto issue = [&](int g) {
#pragma unroll
      for (int s = 0; s < PB; ++s) {
        const int b = (g * PB + s) % nbundles;
        const char* src = w + (size_t)b * SLOT + lane * 16;
        const uint32_t dst = lds_b + ((g % 3) * PB + s) * SLOT;
#pragma unroll
        for (int i = 0; i < 16; ++i) glds16(src + i * 1024, dst + i * 1024);
      }
    };
    issue(0);
    if (groups > 1) issue(1);
    for (int g = 0; g < groups; ++g) {
      if (g + 1 < groups)
        asm volatile("s_waitcnt vmcnt(32)" ::: "memory");  // group g landed (group g + 1 may be in flight)
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (g + 2 < groups) issue(g + 2);  // into the slots of group g - 1: every compute wave is past them
    }
    return;
  }
  // ---- compute waves
  bf16x8_t b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) b[i] = __builtin_bit_cast(bf16x8_t, u32x4_t{(uint32_t)lane, (uint32_t)i, 0x3f803f80u, 0x3f803f80u});
  LDSP(char)* ring = lds + lane * 16;
  float sum = 0.f;
  if constexpr (!FAT && TMODE == 3) {
    f32x4_t c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    for (int g = 0; g < groups; ++g) {
      __builtin_amdgcn_s_barrier();
      bf16x8_t a[2][16];
#pragma unroll
      for (int s = 0; s < PB; ++s) {
        LDSP(char)* wb = ring + ((g % 3) * PB + s) * SLOT;
#pragma unroll
        for (int i = 0; i < 16; ++i) a[s][i] = __builtin_bit_cast(bf16x8_t, *(LDSP(u32x4_t)*)(wb + i * 1024));
      }
#pragma unroll
      for (int s = 0; s < PB; ++s)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][2 * k], b[k], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][2 * k + 1], b[k], c1, 0, 0, 0);
        }
    }
    sum = c0[0] + c1[1] + c0[2] + c1[3];
  } else if constexpr (!FAT) {
    f32x4_t c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    uint32_t x = 0;
    for (int g = 0; g < groups; ++g) {
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int s = 0; s < PB; ++s) {
        LDSP(char)* wb = ring + ((g % 3) * PB + s) * SLOT;
        bf16x8_t a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
          a[i] = TMODE == 2 ? __builtin_bit_cast(bf16x8_t, u32x4_t{(uint32_t)g, (uint32_t)i, (uint32_t)s, 1u})
                            : __builtin_bit_cast(bf16x8_t, *(LDSP(u32x4_t)*)(wb + i * 1024));
        if (TMODE == 1) {
#pragma unroll
          for (int i = 0; i < 16; ++i) x ^= __builtin_bit_cast(u32x4_t, a[i])[0];
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * k], b[k], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * k + 1], b[k], c1, 0, 0, 0);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    sum = c0[0] + c1[1] + c0[2] + c1[3] + (float)x;
  } else if constexpr (PIPE) {
    f32x16_t c0, c1;
#pragma unroll
    for (int e = 0; e < 16; ++e) c0[e] = c1[e] = 0.f;
    for (int g = 0; g < groups; ++g) {
      __builtin_amdgcn_s_barrier();
      bf16x8_t a[2][16];
#pragma unroll
      for (int s = 0; s < PB; ++s) {
        LDSP(char)* wb = ring + ((g % 3) * PB + s) * SLOT;
#pragma unroll
        for (int i = 0; i < 16; ++i) a[s][i] = __builtin_bit_cast(bf16x8_t, *(LDSP(u32x4_t)*)(wb + i * 1024));
      }
#pragma unroll
      for (int s = 0; s < PB; ++s) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][2 * k], b[k], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][2 * k + 1], b[k], c1, 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) sum += c0[e] + c1[e];
  } else {
    f32x16_t c0, c1;
#pragma unroll
    for (int e = 0; e < 16; ++e) c0[e] = c1[e] = 0.f;
    for (int g = 0; g < groups; ++g) {
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int s = 0; s < PB; ++s) {
        LDSP(char)* wb = ring + ((g % 3) * PB + s) * SLOT;
        bf16x8_t a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_bit_cast(bf16x8_t, *(LDSP(u32x4_t)*)(wb + i * 1024));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 * k], b[k], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 * k + 1], b[k], c1, 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) sum += c0[e] + c1[e];
  }
  if (sum == 123.456f) out[blockIdx.x * 512 + threadIdx.x] = sum;
}

template <int NCW, bool FAT, bool PIPE = false, int TMODE = 0>
float run(const char* w, float* out, int rounds, int steps, int nb) {
  const int smem = NS * SLOT;
  hipFuncSetAttribute(reinterpret_cast<const void*>(skel<NCW, FAT, PIPE, TMODE>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((skel<NCW, FAT, PIPE, TMODE>), dim3(256), dim3((NCW + 1) * 64), smem, 0, w, out, rounds, steps, nb);
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((skel<NCW, FAT, PIPE, TMODE>), dim3(256), dim3((NCW + 1) * 64), smem, 0, w, out, rounds, steps, nb);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  return ms * 1e3f / reps;
}

int main() {
  const int nb = 96;
  char* w;
  float* out;
  hipMalloc(&w, (size_t)nb * SLOT);
  hipMalloc(&out, 256 * 512 * 4);
  std::vector<uint16_t> h((size_t)nb * SLOT / 2, 0x3c00);
  hipMemcpy(w, h.data(), (size_t)nb * SLOT, hipMemcpyHostToDevice);
  const int steps = 96;
  // M = 163840 rows: thin tiles of 112 rows -> 1463 tiles -> 6 rounds on 256 CUs; fat tiles of 96 rows -> 1707 -> 7 rounds
  const float thin7 = run<7, false>(w, out, 6, steps, nb);
  const float fat3 = run<3, true>(w, out, 7, steps, nb);
  const float thin5 = run<5, false>(w, out, 8, steps, nb);  // 80-row tiles: 2048 -> 8 rounds
  const float fat3_6 = run<3, true>(w, out, 6, steps, nb);  // (what a 7th fat round costs)
  const float fat3p = run<3, true, true>(w, out, 7, steps, nb);  // both steps' fragments of a barrier group read before its MFMAs
  const float t_nomfma = run<7, false, false, 1>(w, out, 6, steps, nb);
  const float t_nolds = run<7, false, false, 2>(w, out, 6, steps, nb);
  const float t_pipe = run<7, false, false, 3>(w, out, 6, steps, nb);
  printf("chain B skeleton at M = 163840 (LDS-DMA ring + fragment reads + MFMA + barriers only, 96 steps per tile):\n");
  printf("  thin, 7 x 16 rows, 16x16x32, 6 rounds: %7.1f us   (%5.0f cycles per step at 2.3 GHz)\n", thin7, thin7 * 2300 / (6 * steps));
  printf("  thin, 5 x 16 rows, 16x16x32, 8 rounds: %7.1f us   (%5.0f)\n", thin5, thin5 * 2300 / (8 * steps));
  printf("  fat,  3 x 32 rows, 32x32x16, 7 rounds: %7.1f us   (%5.0f)\n", fat3, fat3 * 2300 / (7 * steps));
  printf("  fat,  3 x 32 rows, 32x32x16, 6 rounds: %7.1f us\n", fat3_6);
  printf("  fat,  the same, a barrier group's 32 fragments read before its 32 MFMAs, 7 rounds: %7.1f us   (%5.0f)\n", fat3p, fat3p * 2300 / (7 * steps));
  printf("  thin 7 x 16, fragment reads only (no MFMA):          %7.1f us   (%5.0f)\n", t_nomfma, t_nomfma * 2300 / (6 * steps));
  printf("  thin 7 x 16, MFMAs only (no fragment reads):         %7.1f us   (%5.0f)\n", t_nolds, t_nolds * 2300 / (6 * steps));
  printf("  thin 7 x 16, a barrier group's 32 fragments read before its 32 MFMAs: %7.1f us   (%5.0f)\n", t_pipe, t_pipe * 2300 / (6 * steps));
  return 0;
}
