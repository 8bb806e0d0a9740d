// Micro-probe: HBM write bandwidth of the GEMM epilogue store pattern vs a row-contiguous pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// pattern A: the swapped-MFMA layout: lane (r = l & 31, hi) owns row r, 4 consecutive bf16 (8 B) at col 8g + 4hi
__global__ __launch_bounds__(256) void scat8(uint16_t* C, int64_t M, int N, int tiles_n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, hi = lane >> 5;
  for (int64_t t = blockIdx.x; t < (M / 128) * tiles_n; t += gridDim.x) {
    const int64_t bm = (t / tiles_n) * 128, bn = (t % tiles_n) * 256;
    for (int mt = 0; mt < 2; ++mt) for (int nt = 0; nt < 4; ++nt) for (int g = 0; g < 4; ++g) {
      const int64_t m = bm + wm * 64 + mt * 32 + r, n = bn + wn * 128 + nt * 32 + 8 * g + 4 * hi;
      *reinterpret_cast<uint2*>(C + m * N + n) = make_uint2(lane, t);
    }
  }
}
// pattern B: 16 lanes cover 128 consecutive bf16 (256 B) of one row, 16 B per lane; a wave writes 4 rows per instruction
__global__ __launch_bounds__(256) void rows16(uint16_t* C, int64_t M, int N, int tiles_n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  for (int64_t t = blockIdx.x; t < (M / 128) * tiles_n; t += gridDim.x) {
    const int64_t bm = (t / tiles_n) * 128, bn = (t % tiles_n) * 256;
    for (int j = 0; j < 16; ++j) {
      const int64_t m = bm + wm * 64 + j * 4 + (lane >> 4), n = bn + wn * 128 + (lane & 15) * 8;
      *reinterpret_cast<uint4*>(C + m * N + n) = make_uint4(lane, t, j, 0);
    }
  }
}
// pattern C: like A but fp32 (16 B per lane at col 8g + 4hi)
__global__ __launch_bounds__(256) void scat16f(float* C, int64_t M, int N, int tiles_n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, hi = lane >> 5;
  for (int64_t t = blockIdx.x; t < (M / 128) * tiles_n; t += gridDim.x) {
    const int64_t bm = (t / tiles_n) * 128, bn = (t % tiles_n) * 256;
    for (int mt = 0; mt < 2; ++mt) for (int nt = 0; nt < 4; ++nt) for (int g = 0; g < 4; ++g) {
      const int64_t m = bm + wm * 64 + mt * 32 + r, n = bn + wn * 128 + nt * 32 + 8 * g + 4 * hi;
      *reinterpret_cast<float4*>(C + m * N + n) = make_float4(lane, t, 0, 0);
    }
  }
}
__global__ __launch_bounds__(256) void rows16f(float* C, int64_t M, int N, int tiles_n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  for (int64_t t = blockIdx.x; t < (M / 128) * tiles_n; t += gridDim.x) {
    const int64_t bm = (t / tiles_n) * 128, bn = (t % tiles_n) * 256;
    for (int j = 0; j < 32; ++j) {
      const int64_t m = bm + wm * 64 + j * 2 + (lane >> 5), n = bn + wn * 128 + (lane & 31) * 4;
      *reinterpret_cast<float4*>(C + m * N + n) = make_float4(lane, t, j, 0);
    }
  }
}
template <typename F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a); for (int i = 0; i < 20; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 20;
}
int main() {
  const int64_t M = 163840;
  void* buf; hipMalloc(&buf, M * 1024 * 4);
  for (int N : {256, 768, 1024}) {
    const int tn = N / 256;
    for (int grid : {256, 512, 2048}) {
      float a = timeit([&] { hipLaunchKernelGGL(scat8, dim3(grid), dim3(256), 0, 0, (uint16_t*)buf, M, N, tn); });
      float b = timeit([&] { hipLaunchKernelGGL(rows16, dim3(grid), dim3(256), 0, 0, (uint16_t*)buf, M, N, tn); });
      float c = timeit([&] { hipLaunchKernelGGL(scat16f, dim3(grid), dim3(256), 0, 0, (float*)buf, M, N, tn); });
      float d = timeit([&] { hipLaunchKernelGGL(rows16f, dim3(grid), dim3(256), 0, 0, (float*)buf, M, N, tn); });
      printf("N=%4d grid=%4d  bf16 scattered-8B %6.2f TB/s | bf16 row-16B %6.2f TB/s | f32 scattered-16B %6.2f TB/s | f32 row-16B %6.2f TB/s\n",
             N, grid, M * N * 2.0 / a / 1e9, M * N * 2.0 / b / 1e9, M * N * 4.0 / c / 1e9, M * N * 4.0 / d / 1e9);
    }
  }
  return 0;
}
