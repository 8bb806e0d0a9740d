#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t* out, int stride_b) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  // lane i of group g supplies row (i>>2) of block g, 4 elems at col 4(i&3); blocks are 16 cols apart
  const int byte = (i >> 2) * stride_b + (g * 16 + 4 * (i & 3)) * 2;
  auto* p = (__attribute__((address_space(3))) v4s*)((__attribute__((address_space(3))) char*)lds + byte);
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
__global__ void dma(const uint4* src, uint4* out) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  // write 1 KB at LDS byte offset 100 KB to test addressing beyond 64 KB
  auto* dst = (__attribute__((address_space(3))) void*)((__attribute__((address_space(3))) char*)lds + 100 * 1024);
  __builtin_amdgcn_global_load_lds(src + threadIdx.x, dst, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[threadIdx.x] = *reinterpret_cast<uint4*>(reinterpret_cast<char*>(lds) + 100 * 1024 + threadIdx.x * 16);
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int stride : {512, 64}) {
    probe<<<1, 64, 16384>>>(d, stride);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("stride %d B (=%d elems)\n", stride, stride / 2);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) { int e = h[l*4+j]; printf(" (r%d,c%d)", e / (stride/2), e % (stride/2)); } printf("\n"); if (l == 19) l = 43; }
  }
  uint4 *s, *o; hipMalloc(&s, 1024); hipMalloc(&o, 1024);
  uint32_t hs[256]; for (int i = 0; i < 256; ++i) hs[i] = i * 3 + 1;
  hipMemcpy(s, hs, 1024, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)dma, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  dma<<<1, 64, 140 * 1024>>>(s, o);
  uint32_t ho[256]; hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256; ++i) bad += ho[i] != hs[i];
  printf("dma at 100KB: %d mismatches (%s)\n", bad, hipGetErrorString(hipGetLastError()));
}
