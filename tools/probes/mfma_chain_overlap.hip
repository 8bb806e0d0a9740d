// Probe: does a wave's own VALU stream overlap its MFMAs when the MFMAs form ONE dependent accumulator chain (the fc2^T product
// of the fused MLP backward) as well as when they rotate over 4 accumulators?  Per iteration: 1 MFMA (32x32x16 bf16, 32 matrix-pipe
// cycles) + KV independent v_fma_f32 written in inline asm (5 cycles each).  Cycles per iteration of wave 0, 1 and 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o mfma_chain_overlap mfma_chain_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
#define FMA8 asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t" \
                          "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\t" \
                          : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(c), "v"(d));
template <int CHAINS, int KV8, bool MFMA>
__global__ __launch_bounds__(512) void probe(unsigned long long* out, int iters, int nwaves) {
  const int wave = threadIdx.x >> 6;
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) a[i] = (__bf16)(0.001f * (threadIdx.x + i)), b[i] = (__bf16)(0.002f * (i + 1));
  f32x16_t acc[4];
  for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
  float v0 = threadIdx.x * 0.001f, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
  float c = 1.0001f, d = 0.5f;
  unsigned long long t0 = 0, t1 = 0;
  if (wave < nwaves) {
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        if (MFMA) {
          acc[m % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % CHAINS], 0, 0, 0);
          asm volatile("" : "+v"(a));  // keeps the MFMA at its place among the asm FMAs
        }
#pragma unroll
        for (int r = 0; r < KV8; ++r) { FMA8 }
      }
    }
    t1 = __builtin_readcyclecounter();
  }
  float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) s += acc[k][e];
  if (s == 1.2345f) out[8 + threadIdx.x] = (unsigned long long)s;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}
template <int CHAINS, int KV8, bool MFMA>
void run(const char* name, unsigned long long* d) {
  const int iters = 200;
  for (int nw = 4; nw <= 8; nw += 4) {
    unsigned long long h[8];
    hipLaunchKernelGGL((probe<CHAINS, KV8, MFMA>), dim3(256), dim3(512), 0, 0, d, iters, nw);
    hipLaunchKernelGGL((probe<CHAINS, KV8, MFMA>), dim3(256), dim3(512), 0, 0, d, iters, nw);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s %d wave(s) per SIMD: %7.1f cycles per (MFMA + VALU group)\n", name, nw / 4, (double)h[0] / (iters * 16.0));
  }
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 8192);
  run<1, 0, true>("MFMA only, 1 chain", d);
  run<4, 0, true>("MFMA only, 4 chains", d);
  run<1, 1, false>("8 FMA only", d);
  run<1, 2, false>("16 FMA only", d);
  run<1, 1, true>("1 chain: MFMA + 8 FMA", d);
  run<4, 1, true>("4 chains: MFMA + 8 FMA", d);
  run<1, 2, true>("1 chain: MFMA + 16 FMA", d);
  run<4, 2, true>("4 chains: MFMA + 16 FMA", d);
  run<2, 2, true>("2 chains: MFMA + 16 FMA", d);
  return 0;
}
