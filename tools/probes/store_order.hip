// Probe: HBM write rate of the chain kernels' store pattern by ORDER.  256 workgroups x 7 waves, a wave owns 16 consecutive rows of
// five row-major arrays (fp32 [M,256], three bf16 [M,256], one bf16 [M,768]) per tile and writes them with instructions that each
// cover 8 rows x 128 bytes (whole lines, stride = the row pitch).  Orders:
//   0  array after array, each array's rows in one burst                      (xhat | xm | x | xb | qkv)
//   1  the same bytes, one 2-instruction piece per array in rotation           (fine interleave of the five arrays)
//   2  array after array, but every burst split in 2-instruction pieces with a ~1500-cycle pause between pieces
//   3  one array only (fp32), bursts
//   4  order 0 with instructions that write 1 KB contiguous (2 rows x 512 B / 1 row x 1 KB): upper bound of the row-major layout
// hipcc --offload-arch=gfx950 -O3 -o store_order store_order.hip && ./store_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

struct args_t {
  char* a[5];
  int pitch[5];  // bytes per row
  int64_t M;
};

__device__ __forceinline__ void pause(int n) {
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);
}

template <int ORDER>
__global__ __launch_bounds__(512) void k(args_t p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= 7) return;
  const int64_t ntiles = p.M / 112;
  const int tok = lane & 15, g = lane >> 4;
  const bool lo = tok < 8;
  const uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t r0 = (t * 7 + wave) * 16;
    // piece (array i, q): two instructions = 16 rows x 128 bytes at byte offset 128 q of the rows
    auto piece = [&](int i, int q) __attribute__((always_inline)) {
      if (ORDER == 4) {  // contiguous 1 KB per instruction: 128-byte column block q -> rows-major remap of the same byte count
        char* b = p.a[i] + r0 * p.pitch[i] + (int64_t)q * 2048 + lane * 16;
        *reinterpret_cast<uint4*>(b) = v;
        *reinterpret_cast<uint4*>(b + 1024) = v;
        return;
      }
      char* b = p.a[i] + r0 * p.pitch[i] + 128 * q + 16 * g + (lo ? 0 : 64);
      *reinterpret_cast<uint4*>(b + (int64_t)(tok & 7) * p.pitch[i]) = v;
      *reinterpret_cast<uint4*>(b + (int64_t)((tok & 7) + 8) * p.pitch[i]) = v;
    };
    const int npieces[5] = {8, 4, 4, 4, 12};  // 128-byte column blocks per row: fp32 x 8, bf16 256 -> 4, qkv -> 12
    if (ORDER == 0 || ORDER == 4) {
#pragma unroll
      for (int i = 1; i < 5; ++i) {
        if (i == 3) {
#pragma unroll
          for (int q = 0; q < 8; ++q) piece(0, q);
        }
#pragma unroll
        for (int q = 0; q < 12; ++q)
          if (q < npieces[i]) piece(i, q);
      }
    } else if (ORDER == 1) {
#pragma unroll
      for (int q = 0; q < 12; ++q) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
          if (q < npieces[i]) piece(i, q);
      }
    } else if (ORDER == 2) {
#pragma unroll
      for (int i = 0; i < 5; ++i) {
#pragma unroll
        for (int q = 0; q < 12; ++q)
          if (q < npieces[i]) {
            piece(i, q);
            pause(3);
          }
      }
    } else if (ORDER == 3) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int q = 0; q < 8; ++q) piece(0, q);
    }
  }
}

template <int ORDER>
void run(const char* name, const args_t& a, double bytes) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<ORDER>, dim3(256), dim3(512), 0, 0, a);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<ORDER>, dim3(256), dim3(512), 0, 0, a);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 10;
  printf("%-62s %7.1f us  %5.2f TB/s\n", name, ms * 1e3, bytes / ms * 1e-9);
}

int main() {
  args_t a;
  a.M = 163744;  // 1462 tiles of 112 rows
  const int pitch[5] = {1024, 512, 512, 512, 1536};
  double bytes = 0;
  for (int i = 0; i < 5; ++i) {
    a.pitch[i] = pitch[i];
    hipMalloc(&a.a[i], (size_t)163840 * pitch[i]);
    bytes += (double)a.M * pitch[i];
  }
  run<0>("0 array after array, bursts", a, bytes);
  run<1>("1 fine interleave of the five arrays", a, bytes);
  run<2>("2 array after array, 2-instruction pieces with pauses", a, bytes);
  run<3>("3 one fp32 array only, bursts (same instruction count)", a, bytes);
  run<4>("4 order 0, 1 KB contiguous per instruction", a, bytes);
  return 0;
}
