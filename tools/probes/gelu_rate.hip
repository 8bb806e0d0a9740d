// Probe: cycles per call of the fused MLP backward's GELU section (gelu_bwd_n<8>, csrc/mlp.hip) on register data, alone on a SIMD
// and beside a second wave -- is its 7-8 cycles per instruction inside the kernel the code's own (dependencies, register banks) or
// the surroundings'?   hipcc --offload-arch=gfx950 -O3 -Ihma_amd/csrc -o gelu_rate tools/probes/gelu_rate.hip
#include "../../hma_amd/csrc/mlp.hip"
#include <cstdio>
namespace {
__global__ __launch_bounds__(512) void gelu_probe(unsigned long long* out, float* sink, int iters, int nwaves) {
  const int wave = threadIdx.x >> 6;
  float u[8], d[8], hg[8], du[8];
  for (int e = 0; e < 8; ++e) u[e] = 0.01f * (threadIdx.x & 63) - 0.3f + 0.1f * e, d[e] = 0.5f + 0.01f * e;
  unsigned long long t0 = 0, t1 = 0;
  float acc = 0.f;
  if (wave < nwaves) {
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
      gelu_bwd_n<8>(u, d, hg, du);
#pragma unroll
      for (int e = 0; e < 8; ++e) u[e] += 1e-6f * du[e], d[e] += 1e-6f * hg[e];  // (loop-carried, so that nothing is hoisted)
    }
    t1 = __builtin_readcyclecounter();
  }
  for (int e = 0; e < 8; ++e) acc += u[e] + d[e];
  if (acc == 1.2345f) sink[threadIdx.x] = acc;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}
}  // namespace
int main() {
  unsigned long long* d; float* s;
  hipMalloc(&d, 4096); hipMalloc(&s, 4096);
  for (int nw = 4; nw <= 8; nw += 4) {
    unsigned long long h[8];
    hipLaunchKernelGGL(gelu_probe, dim3(256), dim3(512), 0, 0, d, s, 2000, nw);
    hipLaunchKernelGGL(gelu_probe, dim3(256), dim3(512), 0, 0, d, s, 2000, nw);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("gelu_bwd_n<8> + 16 fma, %d wave(s) per SIMD: %7.1f cycles per call (8 values)\n", nw / 4, (double)h[0] / 2000.0);
  }
  return 0;
}
