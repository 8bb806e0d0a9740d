// Probe: how fast can one CU pull a SHARED, L2-resident weight stream (every workgroup reads the same 1 MB again and again,
// the access pattern of the fused MLP / chain kernels' weight ring), by path and by the number of waves issuing:
//   mode 0  global_load_lds_dwordx4 (LDS-DMA), 1 KB pieces, 8-16 pieces in flight per wave
//   mode 1  global_load_dwordx4 into registers, then ds_write_b128 into LDS
//   mode 2  global_load_dwordx4 into registers only (xor-reduced)
// hipcc --offload-arch=gfx950 -O3 -o l2_stream l2_stream.hip && ./l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define LDSP(T) __attribute__((address_space(3))) T
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
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

template <int MODE>
__global__ __launch_bounds__(512) void stream(const char* __restrict__ buf, int buf_kb, int iters, int nwaves, uint32_t* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= nwaves) return;
  const uint32_t lds_b = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(LDSP(char)*)smem) + wave * 16384;
  LDSP(char)* lds = (LDSP(char)*)smem + wave * 16384;
  uint32_t acc = 0;
  // the workgroup's waves interleave 4 KB groups of the buffer; one iteration = 2 groups (8 KB) per wave
  const int groups = buf_kb / 4;
  int gidx = wave;
  auto next = [&]() __attribute__((always_inline)) {
    const char* p = buf + (size_t)gidx * 4096 + lane * 16;
    gidx += nwaves;
    if (gidx >= groups) gidx -= groups;
    return p;
  };
  if (MODE == 0) {
    glds16x4(next(), lds_b);
    glds16x4(next(), lds_b + 4096);
    for (int it = 0; it < iters; ++it) {
      glds16x4(next(), lds_b + 8192);
      glds16x4(next(), lds_b + 12288);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      acc ^= *(LDSP(uint32_t)*)(lds + lane * 4);
      glds16x4(next(), lds_b);
      glds16x4(next(), lds_b + 4096);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      acc ^= *(LDSP(uint32_t)*)(lds + 8192 + lane * 4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    u32x4_t r0[8], r1[8];
    auto ld8 = [&](u32x4_t (&r)[8]) __attribute__((always_inline)) {
      const char* p = next();
      const char* q = next();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        r[i] = *reinterpret_cast<const u32x4_t*>(p + i * 1024);
        r[4 + i] = *reinterpret_cast<const u32x4_t*>(q + i * 1024);
      }
    };
    auto use8 = [&](u32x4_t (&r)[8], int half) __attribute__((always_inline)) {
      if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *(LDSP(u32x4_t)*)(lds + half * 8192 + i * 1024 + lane * 16) = r[i];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc ^= r[i][0] ^ r[i][1] ^ r[i][2] ^ r[i][3];
      }
    };
    ld8(r0);
    for (int it = 0; it < iters; ++it) {
      ld8(r1);
      use8(r0, 0);
      ld8(r0);
      use8(r1, 1);
    }
    use8(r0, 0);
    if (MODE == 1) acc ^= *(LDSP(uint32_t)*)(lds + lane * 4);
  }
  if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

template <int MODE>
void run(const char* name, const char* buf, int buf_kb, uint32_t* out) {
  const int iters = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(stream<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int nw : {1, 2, 4, 8}) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(stream<MODE>, dim3(256), dim3(512), 131072, 0, buf, buf_kb, 50, nw, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(stream<MODE>, dim3(256), dim3(512), 131072, 0, buf, buf_kb, iters, nw, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 256.0 * nw * (double)iters * 16384.0;
    printf("%-28s buf %5d KB  %d wave(s) issuing: %7.2f TB/s chip, %6.1f GB/s per CU, %5.1f B/clk/CU at 2.4 GHz\n", name, buf_kb, nw,
           bytes / ms * 1e-9, bytes / ms * 1e-6 / 256.0, bytes / ms * 1e-6 / 256.0 / 2.4);
  }
}

int main() {
  char* buf;
  uint32_t* out;
  const int maxkb = 65536;
  hipMalloc(&buf, (size_t)maxkb * 1024);
  hipMemset(buf, 1, (size_t)maxkb * 1024);
  hipMalloc(&out, 4096);
  for (int kb : {1024, 4096, 65536}) {
    run<0>("LDS-DMA", buf, kb, out);
    run<1>("load + ds_write_b128", buf, kb, out);
    run<2>("load to registers", buf, kb, out);
  }
  return 0;
}
