// Probe: issue cost (shader-clock cycles per wave64 instruction) of the VALU operations the GELU epilogues are made of, one
// wave per SIMD and two waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(512) void probe(unsigned long long* out, int iters, int nwaves) {
  const int wave = threadIdx.x >> 6;
  float v0 = threadIdx.x * 0.001f, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
  float c = 1.0001f, d = 0.5f;
  unsigned long long t0 = 0, t1 = 0;
  if (wave < nwaves) {
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
      // 8 independent chains x 16 = 128 instructions per iteration
#define OP8(ins) asm volatile(ins(0) ins(1) ins(2) ins(3) ins(4) ins(5) ins(6) ins(7) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(c), "v"(d));
#define I_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n\t"
#define I_MUL(k) "v_mul_f32 %" #k ", %" #k ", %8\n\t"
#define I_EXP(k) "v_exp_f32 %" #k ", %" #k "\n\t"
#define I_RCP(k) "v_rcp_f32 %" #k ", %" #k "\n\t"
#define I_MED(k) "v_med3_f32 %" #k ", %" #k ", %8, %9\n\t"
#define I_MAX(k) "v_max_f32 %" #k ", %" #k ", %8\n\t"
#define I_BFI(k) "v_bfi_b32 %" #k ", %8, %" #k ", %9\n\t"
#define I_CVT(k) "v_cvt_pk_bf16_f32 %" #k ", %" #k ", %8\n\t"
#define I_MOV(k) "v_mov_b32 %" #k ", %8\n\t"
      if (OP == 0) { REP16(OP8(I_FMA)) }
      if (OP == 1) { REP16(OP8(I_MUL)) }
      if (OP == 2) { REP16(OP8(I_EXP)) }
      if (OP == 3) { REP16(OP8(I_RCP)) }
      if (OP == 4) { REP16(OP8(I_MED)) }
      if (OP == 5) { REP16(OP8(I_MAX)) }
      if (OP == 6) { REP16(OP8(I_BFI)) }
      if (OP == 7) { REP16(OP8(I_CVT)) }
      if (OP == 8) { REP16(OP8(I_MOV)) }
    }
    t1 = __builtin_readcyclecounter();
  }
  float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  if (s == 1.2345f) out[8 + threadIdx.x] = (unsigned long long)s;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}
// packed fp32 FMA: 4 independent 2-wide chains x 32
__global__ __launch_bounds__(512) void probe_pk(unsigned long long* out, int iters, int nwaves) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  const int wave = threadIdx.x >> 6;
  f2 a0 = {threadIdx.x * 0.001f, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, c = {1.0001f, 1.0002f}, d = {0.5f, 0.25f};
  unsigned long long t0 = 0, t1 = 0;
  if (wave < nwaves) {
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define PK4 asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d));
      REP16(PK4 PK4)
    }
    t1 = __builtin_readcyclecounter();
  }
  f2 s = a0 + a1 + a2 + a3;
  if (s[0] + s[1] == 1.2345f) out[8 + threadIdx.x] = 1;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}
template <int OP>
void run(const char* name, unsigned long long* d) {
  const int iters = 200;
  for (int nw = 4; nw <= 8; nw += 4) {
    unsigned long long h[8];
    hipLaunchKernelGGL(probe<OP>, dim3(256), dim3(512), 0, 0, d, iters, nw);
    hipLaunchKernelGGL(probe<OP>, dim3(256), dim3(512), 0, 0, d, iters, nw);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-18s %d wave(s) per SIMD: %6.2f cycles per instruction (wave 0), %6.2f per SIMD issue slot\n", name, nw / 4,
           (double)h[0] / (iters * 128.0), (double)h[0] / (iters * 128.0) / (nw / 4));
  }
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 8192);
  run<0>("v_fma_f32", d); run<1>("v_mul_f32", d); run<2>("v_exp_f32", d); run<3>("v_rcp_f32", d); run<4>("v_med3_f32", d);
  run<5>("v_max_f32", d); run<6>("v_bfi_b32", d); run<7>("v_cvt_pk_bf16_f32", d); run<8>("v_mov_b32", d);
  for (int nw = 4; nw <= 8; nw += 4) {
    unsigned long long h[8];
    hipLaunchKernelGGL(probe_pk, dim3(256), dim3(512), 0, 0, d, 200, nw);
    hipLaunchKernelGGL(probe_pk, dim3(256), dim3(512), 0, 0, d, 200, nw);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-18s %d wave(s) per SIMD: %6.2f cycles per instruction (wave 0)\n", "v_pk_fma_f32", nw / 4, (double)h[0] / (200 * 128.0));
  }
  return 0;
}
