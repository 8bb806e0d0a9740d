// Probe: do a wave's MFMAs overlap (a) VALU work of the OTHER wave on the same SIMD, (b) its own interleaved VALU work?
// 512-thread workgroups, one per CU: waves w and w + 4 share a SIMD.  hipcc --offload-arch=gfx950 -O3 -o probe probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// mode 0: waves 0-3 MFMA only; 1: waves 4-7 VALU only; 2: both; 3: every wave interleaves 1 MFMA + KV VALU; 4: all 8 waves MFMA only;
// 5: all 8 waves VALU only; 6: waves 0-3 MFMA + waves 4-7 transcendental
template <int MODE, int KV>
__global__ __launch_bounds__(512, 2) void probe(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) a[i] = (__bf16)(0.001f * (threadIdx.x + i)), b[i] = (__bf16)(0.002f * (i + 1));
  f32x16_t acc[4];
  for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
  float v[16];
  for (int e = 0; e < 16; ++e) v[e] = 0.001f * (threadIdx.x + e);
  const bool do_mfma = MODE == 0 ? wave < 4 : MODE == 1 ? false : MODE == 2 ? wave < 4 : MODE == 3 ? true : MODE == 4 ? true : MODE == 5 ? false : wave < 4;
  const bool do_valu = MODE == 0 ? false : MODE == 1 ? wave >= 4 : MODE == 2 ? wave >= 4 : MODE == 3 ? false : MODE == 4 ? false : MODE == 5 ? true : false;
  const bool do_trans = MODE == 6 && wave >= 4;
  if (MODE == 3) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < KV; ++e) v[(m * KV + e) & 15] = __builtin_fmaf(v[(m * KV + e) & 15], 1.0001f, 0.5f);
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
      }
    }
  } else if (do_mfma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 16; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
    }
  } else if (do_valu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < KV; ++r) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = __builtin_fmaf(v[e], 1.0001f, 0.5f);
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
      }
    }
  } else if (do_trans) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < KV; ++r) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = __builtin_amdgcn_exp2f(v[e]);
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
      }
    }
  }
  float s = 0.f;
  for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) s += acc[k][e];
  for (int e = 0; e < 16; ++e) s += v[e];
  if (s == 1.2345f) out[threadIdx.x] = s;
}

template <int MODE, int KV>
float run(float* d, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MODE, KV>), dim3(256), dim3(512), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MODE, KV>), dim3(256), dim3(512), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main() {
  float* d; hipMalloc(&d, 4096);
  const int iters = 2000;  // 16 MFMAs per iteration per wave: 32000 MFMAs
  printf("per iteration (16 MFMAs = 512 pipe cycles per wave): times in us for %d iterations\n", iters);
  printf("mode0 4 waves MFMA only            : %8.1f\n", run<0, 1>(d, iters));
  printf("mode4 8 waves MFMA only            : %8.1f\n", run<4, 1>(d, iters));
  printf("mode1 4 waves VALU only (8x16 fma) : %8.1f\n", run<1, 8>(d, iters));
  printf("mode5 8 waves VALU only (8x16 fma) : %8.1f\n", run<5, 8>(d, iters));
  printf("mode2 MFMA waves + VALU waves      : %8.1f\n", run<2, 8>(d, iters));
  printf("mode6 MFMA waves + 2x16 exp waves  : %8.1f   (exp alone not measured)\n", run<6, 2>(d, iters));
  printf("mode3 8 waves, 1 MFMA + 4 fma      : %8.1f\n", run<3, 4>(d, iters));
  printf("mode3 8 waves, 1 MFMA + 8 fma      : %8.1f\n", run<3, 8>(d, iters));
  printf("mode3 8 waves, 1 MFMA + 2 fma      : %8.1f\n", run<3, 2>(d, iters));
  return 0;
}
