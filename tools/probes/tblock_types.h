// Argument block of the fused temporal-block probe (tools/probes/tblock_fwd_experiment.hip); not part of the shipped ABI.
#pragma once
#include <stdint.h>
typedef struct {
  const void* xb; float* x;
  const void* wqkvp; const void* wprojp; const float* bqkv; const float* bproj;
  void* qkv; void* o;
  void* ln_xhat; float* ln_rstd; float ln_eps; float scale;
  int64_t B; int32_t T; int32_t SA;
} hma_tblock_fwd_t;

#ifdef __HIPCC__
#include "hma_common.h"
namespace hma {
// One LDS-DMA piece with the global address as a wave-uniform base (SGPR pair) + a 32-bit per-lane byte offset: no 64-bit vector
// address arithmetic at the call site (only the probe uses it).
__device__ __forceinline__ void glds16s(const void* sbase, uint32_t voff, uint32_t dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(sbase), "v"(voff), "s"(dst)
      : "memory");
}
}  // namespace hma
#endif
