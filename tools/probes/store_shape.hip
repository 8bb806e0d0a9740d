// Probe: what does the SHAPE of a 1 KB store instruction cost?  A chain kernel's store writes 8 rows x 128 B (eight separate cache
// lines, rows 0.5-1.5 KB apart); a row-contiguous layout would write 1 row x 1 KB or 2 rows x 512 B.  Every CU issues bursts of N
// instructions of a given shape from W waves, back to back: issue cycles per instruction (wave 0 of workgroup 0) and bytes per clock
// and CU until everything has completed.  R = rows per instruction (64 / R lanes x 16 B contiguous per row), rows `pitch` bytes apart.
// hipcc --offload-arch=gfx950 -O3 -o store_shape store_shape.hip && ./store_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

template <int N, int R>
__global__ __launch_bounds__(512) void k(char* buf, unsigned long long* out, int nwaves, int rounds, int gap, int pitch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= nwaves) return;
  const u32x4_t v = {(uint32_t)lane, (uint32_t)wave, blockIdx.x, 3u};
  unsigned long long t_issue = 0, t_done = 0;
  constexpr int LPR = 64 / R;                       // lanes per row
  // a wave's region per round: N instructions; instruction i covers rows (i * R .. i * R + R - 1) of a [N * R rows x pitch] panel
  // (pitch >= LPR * 16); the same bytes per instruction for every shape
  const size_t panel = (size_t)N * R * pitch;
  char* base = buf + ((size_t)blockIdx.x * 8 + wave) * panel * rounds + (size_t)(lane / LPR) * pitch + (lane % LPR) * 16;
  for (int r = 0; r < rounds; ++r) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int i = 0; i < gap; ++i) __builtin_amdgcn_s_sleep(32);
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < N; ++i) __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(base + r * panel + (size_t)i * R * pitch));
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

template <int N, int R>
void run(char* buf, unsigned long long* out, int pitch) {
  for (int nw : {1, 2, 7}) {
    for (int gap : {0, 40}) {
      unsigned long long h[16];
      hipLaunchKernelGGL((k<N, R>), dim3(256), dim3(512), 0, 0, buf, out, nw, 20, gap, pitch);
      hipDeviceSynchronize();
      hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
      printf("R = %d rows x %4d B per instruction, pitch %5d, N = %2d per burst, %d wave(s)/CU, gap %5d: issue %6.0f cycles per instruction, %5.1f B/clk/CU\n", R,
             1024 / R, pitch, N, nw, gap * 32 * 64, (double)h[0] / N, (double)N * 1024 * nw / (double)(h[1] + gap * 32 * 64));
    }
  }
}

int main() {
  char* buf;
  unsigned long long* out;
  const size_t bytes = (size_t)256 * 8 * 32 * 8 * 2048 * 20;   // 256 WGs x 8 waves x N x R x pitch x rounds
  if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, 256);
  run<32, 1>(buf, out, 1024);
  run<32, 2>(buf, out, 512);
  run<32, 2>(buf, out, 1536);
  run<32, 4>(buf, out, 512);
  run<32, 8>(buf, out, 512);
  run<32, 8>(buf, out, 1024);
  run<32, 8>(buf, out, 1536);
  return 0;
}
