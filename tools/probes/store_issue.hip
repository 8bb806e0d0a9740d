// Probe: how long does a wave spend ISSUING a burst of N 1-KB store instructions (global_store_dwordx4, 8 whole lines per
// instruction), when every CU does the same at once, and how long until they have completed (vmcnt(0))?  One to seven waves per
// CU issue.  The issue time per instruction tells how much store data the CU's memory pipeline takes before the wave blocks.
// hipcc --offload-arch=gfx950 -O3 -o store_issue store_issue.hip && ./store_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int N>
__global__ __launch_bounds__(512) void k(char* buf, unsigned long long* out, int nwaves, int rounds, int gap) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= nwaves) return;
  const uint4 v = make_uint4(lane, wave, blockIdx.x, 3);
  unsigned long long t_issue = 0, t_done = 0;
  char* base = buf + ((size_t)blockIdx.x * 8 + wave) * (size_t)(N * 1024) * rounds + lane * 16;
  for (int r = 0; r < rounds; ++r) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int i = 0; i < gap; ++i) __builtin_amdgcn_s_sleep(32);
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < N; ++i) *reinterpret_cast<uint4*>(base + (size_t)(r * N + i) * 1024) = v;
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();
    t_issue += t1 - t0;
    t_done += t2 - t0;
  }
  if (blockIdx.x == 0 && lane == 0) {
    out[wave * 2] = t_issue / rounds;
    out[wave * 2 + 1] = t_done / rounds;
  }
}

template <int N>
void run(char* buf, unsigned long long* out) {
  for (int nw : {1, 4, 7}) {
    for (int gap : {0, 40}) {
      unsigned long long h[16];
      hipLaunchKernelGGL(k<N>, dim3(256), dim3(512), 0, 0, buf, out, nw, 50, gap);
      hipDeviceSynchronize();
      hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
      printf("N = %2d stores/burst, %d wave(s)/CU, gap %4d cycles: issue %6llu cycles (%5.0f per store), complete %6llu cycles -> %5.1f B/clk/CU\n", N,
             nw, gap * 32 * 64, h[0], (double)h[0] / N, h[1], (double)N * 1024 * nw / (double)(h[1] + gap * 32 * 64));
    }
  }
}

int main() {
  char* buf;
  unsigned long long* out;
  hipMalloc(&buf, (size_t)256 * 8 * 64 * 1024 * 50);
  hipMalloc(&out, 256);
  run<1>(buf, out);
  run<4>(buf, out);
  run<8>(buf, out);
  run<16>(buf, out);
  run<32>(buf, out);
  run<64>(buf, out);
  return 0;
}
