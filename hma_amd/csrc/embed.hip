// Token / action / positional embedding (forward + backward) and the per-domain action stem.
//
// Reference: FactorizedEmbedding.forward hma/model/factorization_utils.py:31-54 (boolean-index
// gather of two 512 x d tables, mask-token fill), the concat of 64 action tokens per frame and the
// additive pos_embed_TSC at hma/model/st_mask_git.py:651-672, ActionStat + BasicMLP at
// hma/model/st_mask_git.py:134-138, 90-102.  One fused HBM-bound pass instead of ~10 ATen kernels
// and two host syncs (the boolean indexing at factorization_utils.py:45,53).
#include "hma_common.h"
#include "../../include/hma_hip.h"

using namespace hma;

namespace {

constexpr int D = 256;

__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ E0,
                                                        const float* __restrict__ E1, const float* __restrict__ mask_embed,
                                                        const float* __restrict__ pos, const float* __restrict__ a_emb,
                                                        float* __restrict__ x, int64_t rows, int T, int S, int A,
                                                        int pos_frame_rows, int V, int64_t mask_id) {
  const int lane = threadIdx.x & 63;
  const int SA = S + A;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += nw) {
    const int64_t bt = row / SA;
    const int s = (int)(row % SA);
    const int t = (int)(bt % T);
    float4 e;
    if (s < S) {
      const int64_t id = ids[bt * S + s];
      if (id == mask_id) {
        e = *reinterpret_cast<const float4*>(mask_embed + lane * 4);
      } else {
        const int64_t i0 = id % V, i1 = (id / V) % V;
        e = add4(*reinterpret_cast<const float4*>(E0 + i0 * D + lane * 4),
                 *reinterpret_cast<const float4*>(E1 + i1 * D + lane * 4));
      }
    } else {
      e = *reinterpret_cast<const float4*>(a_emb + bt * D + lane * 4);
    }
    const float4 p = *reinterpret_cast<const float4*>(pos + ((int64_t)t * pos_frame_rows + s) * D + lane * 4);
    *reinterpret_cast<float4*>(x + row * D + lane * 4) = add4(e, p);
  }
}

// dpos[t][s][:] += sum_b dx[b][t][s][:]
__global__ __launch_bounds__(256) void embed_bwd_pos_kernel(const float* __restrict__ dx, float* __restrict__ dpos,
                                                            int64_t B, int T, int SA, int pos_frame_rows) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // float4 index over (t, s, c/4)
  const int64_t total = (int64_t)T * SA * (D / 4);
  if (i >= total) return;
  const int c4 = (int)(i % (D / 4));
  const int64_t ts = i / (D / 4);
  const int t = (int)(ts / SA), s = (int)(ts % SA);
  float4 acc = make_float4(0, 0, 0, 0);
  for (int64_t b = 0; b < B; ++b)
    acc = add4(acc, *reinterpret_cast<const float4*>(dx + ((b * T + t) * SA + s) * D + c4 * 4));
  float4* dst = reinterpret_cast<float4*>(dpos + ((int64_t)t * pos_frame_rows + s) * D + c4 * 4);
  *dst = add4(*dst, acc);
}

// table gradients: scatter-add rows of dx into dE0 / dE1 (fp32 atomics), mask rows reduced per wave first
__global__ __launch_bounds__(256) void embed_bwd_tok_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dx,
                                                            float* __restrict__ dE0, float* __restrict__ dE1,
                                                            float* __restrict__ dmask, int64_t img_rows, int S, int SA, int V,
                                                            int64_t mask_id) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  float4 macc = make_float4(0, 0, 0, 0);
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < img_rows; r += nw) {
    const int64_t bt = r / S;
    const int s = (int)(r % S);
    const int64_t id = ids[r];
    const float4 g = *reinterpret_cast<const float4*>(dx + (bt * SA + s) * D + lane * 4);
    if (id == mask_id) {
      macc = add4(macc, g);
    } else {
      float* p0 = dE0 + (id % V) * D + lane * 4;
      float* p1 = dE1 + ((id / V) % V) * D + lane * 4;
      atomicAdd(p0, g.x); atomicAdd(p0 + 1, g.y); atomicAdd(p0 + 2, g.z); atomicAdd(p0 + 3, g.w);
      atomicAdd(p1, g.x); atomicAdd(p1 + 1, g.y); atomicAdd(p1 + 2, g.z); atomicAdd(p1 + 3, g.w);
    }
  }
  float* pm = dmask + lane * 4;
  atomicAdd(pm, macc.x); atomicAdd(pm + 1, macc.y); atomicAdd(pm + 2, macc.z); atomicAdd(pm + 3, macc.w);
}

// Same gradients accumulated in REGISTERS.  Both tables have only V (512) rows, so ~256 token rows land on every table row and the
// global-atomic version above serialises on them (1.05 ms per step at the bench shape, 40x its HBM time); rounds 2-5 moved the atomics
// into LDS (a V x 64 fp32 slice per workgroup, one ds_add_f32 per token and wave), which a round-6 ablation showed to be the whole of
// that kernel's time: a 64-lane ds_add_f32 occupies the LDS for ~87 cycles (217 us with, 46 us without the adds; 16 waves instead of 4
// moved it by 15 %).  Now a workgroup of 16 waves takes one table, one 64-column slice and one contiguous run of token rows; wave w
// OWNS the table rows r with r % 16 == w and keeps their 32 x 64 sums in 32 registers per lane (lane = column), so nothing is shared
// and nothing is atomic until the flush.  Every wave scans all ids of the run (64 per step, the next step's ids in flight), queues
// the tokens that are its own -- (token, register index) words in a 128-entry ring of its own in LDS -- and whenever 64 are queued
// fetches their 256-byte dx pieces eight at a time and adds each to the register the (wave-uniform) index names.  Mask-token rows
// (table 0 workgroups) are dealt over the waves by step and summed in a 33rd register.
__global__ __launch_bounds__(1024) void embed_bwd_tok_reg_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dx,
                                                                float* __restrict__ dE0, float* __restrict__ dE1,
                                                                float* __restrict__ dmask, int64_t img_rows, int S, int SA, int V,
                                                                int64_t mask_id, int64_t rows_per_group) {
  __shared__ uint32_t queue[16][128];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int variant = blockIdx.x & 7, table = variant >> 2, cs = variant & 3;
  const int64_t r_begin = (int64_t)(blockIdx.x >> 3) * rows_per_group;
  int64_t r_end = r_begin + rows_per_group;
  if (r_end > img_rows) r_end = img_rows;
  if (r_begin >= r_end) return;
  typedef float f32x32_t __attribute__((ext_vector_type(32)));
  f32x32_t acc = 0.f;  // (a vector, not an array: indexed by an SGPR it stays in registers)
  float macc = 0.f;
  uint32_t* q = queue[wave];
  int qh = 0, qt = 0;  // ring head / tail (wave-uniform, free running)
  const float* dxc = dx + cs * 64 + lane;
  const uint32_t uS = (uint32_t)S, uSA = (uint32_t)SA, uV = (uint32_t)V;

  auto consume = [&](int n) {  // the oldest n <= 64 queued tokens
    __builtin_amdgcn_wave_barrier();
    const uint32_t e = lane < n ? q[(qh + lane) & 127] : 0xffu;
    const uint32_t rr = (uint32_t)r_begin + (e >> 8);
    const uint32_t bt = rr / uS;
    const uint32_t off = (bt * uSA + (rr - bt * uS)) * (D / 4);  // in 16-byte units (host: fits 32 bits)
    const uint32_t idx = e & 0xffu;
#pragma unroll
    for (int j0 = 0; j0 < 64; j0 += 8) {
      if (j0 < n) {
        float g[8];
        int ij[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          ij[j] = __builtin_amdgcn_readlane((int)idx, j0 + j);
          const uint32_t oj = (uint32_t)__builtin_amdgcn_readlane((int)off, j0 + j);
          g[j] = dxc[(int64_t)oj * 4];  // (lanes past n: the run's first row, loaded and dropped)
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (ij[j] < 32)
            acc[ij[j]] += g[j];
          else if (ij[j] == 32)
            macc += g[j];
      }
    }
    qh += n;
    __builtin_amdgcn_wave_barrier();
  };

  auto id_at = [&](int64_t c0) {
    const int64_t rr = c0 + lane < r_end ? c0 + lane : r_end - 1;
    return ids[rr];
  };
  int64_t id_next = id_at(r_begin);
  for (int64_t c0 = r_begin; c0 < r_end; c0 += 64) {
    const int64_t id = id_next;
    if (c0 + 64 < r_end) id_next = id_at(c0 + 64);
    const uint32_t tok = (uint32_t)(c0 - r_begin) + lane;
    bool sel;
    uint32_t idx;
    if (id == mask_id) {
      sel = table == 0 && ((tok >> 6) & 15) == (uint32_t)wave;
      idx = 32;
    } else {
      const uint32_t u = (uint32_t)id;
      const uint32_t row = table == 0 ? u % uV : (u / uV) % uV;
      sel = (row & 15) == (uint32_t)wave;
      idx = row >> 4;
    }
    sel = sel && c0 + lane < r_end;
    const uint64_t m = __ballot(sel);
    const int pos = qt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
    if (sel) q[pos & 127] = (tok << 8) | idx;
    qt += __builtin_popcountll(m);
    if (qt - qh >= 64) consume(64);
  }
  if (qt > qh) consume(qt - qh);

  float* dE = (table == 0 ? dE0 : dE1) + cs * 64 + lane;
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const int row = 16 * k + wave;
    if (row < V && acc[k] != 0.f) atomicAdd(dE + (int64_t)row * D, acc[k]);
  }
  if (table == 0 && macc != 0.f) atomicAdd(dmask + cs * 64 + lane, macc);
}

// da_emb[bt][:] += sum over the A action-token rows of frame bt
__global__ __launch_bounds__(256) void embed_bwd_act_kernel(const float* __restrict__ dx, float* __restrict__ da_emb, int S,
                                                            int A) {
  const int64_t bt = blockIdx.x;
  const int c = threadIdx.x;
  float acc = 0.f;
  for (int s = S; s < S + A; ++s) acc += dx[(bt * (S + A) + s) * D + c];
  da_emb[bt * D + c] += acc;
}

// ------------------------------------------------------------------------------- action stem
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// sum of one value per thread of the FIRST FOUR waves of a workgroup of SW waves (the others pass anything)
constexpr int SW = 16;
__device__ __forceinline__ float block_sum_first256(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0 && w < 4) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// One workgroup of 16 waves per row (a row's two mat-vecs are a chain of dependent weight-row loads: the time is the outputs per wave
// times a load latency whatever the number of rows -- 83 us with 4 waves and ds_bpermute sums, round 6: 64 outputs per wave -> 16).
__global__ __launch_bounds__(64 * SW) void stem_fwd_kernel(const float* __restrict__ a, const float* __restrict__ mean,
                                                          const float* __restrict__ stdv, int action_dim,
                                                          const float* __restrict__ W1, const float* __restrict__ b1,
                                                          const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                          const float* __restrict__ W2, const float* __restrict__ b2,
                                                          float* __restrict__ an_out, float* __restrict__ xhat_out,
                                                          float* __restrict__ rstd_out, float* __restrict__ h_out,
                                                          float* __restrict__ out, int d_a, int skip_norm) {
  extern __shared__ float sm[];  // an[d_a] | h[256] | red[4]
  float* an = sm;
  float* hs = sm + d_a;
  float* red = hs + D;
  const int64_t r = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  constexpr int PER = D / SW;  // outputs per wave
  for (int i = tid; i < d_a; i += 64 * SW) {
    float v = a[r * d_a + i];
    if (!skip_norm) v = (v - mean[i % action_dim]) / (stdv[i % action_dim] + 1e-10f);
    an[i] = v;
    an_out[r * d_a + i] = v;
  }
  __syncthreads();
  // pre[j]: wave w covers j = w * PER .. + PER - 1, lanes split the d_a inputs (8 outputs' weight rows in flight)
#pragma unroll 8
  for (int jj = 0; jj < PER; ++jj) {
    const int j = w * PER + jj;
    float p = 0.f;
    for (int i = lane; i < d_a; i += 64) p += an[i] * W1[(int64_t)j * d_a + i];
    p = wave_sum(p);
    if (lane == 0) hs[j] = p + b1[j];
  }
  __syncthreads();
  const int c = tid & (D - 1);
  const bool first = tid < D;
  const float pre = hs[c];
  const float mu = block_sum_first256(pre, red) * (1.0f / D);
  const float dv = pre - mu;
  const float var = block_sum_first256(dv * dv, red) * (1.0f / D);
  const float rstd = rsqrtf(var + 1e-5f);
  const float xh = dv * rstd;
  const float hv = fmaxf(xh * ln_w[c] + ln_b[c], 0.f);
  __syncthreads();
  if (first) {
    xhat_out[r * D + c] = xh;
    h_out[r * D + c] = hv;
    hs[c] = hv;
    if (tid == 0) rstd_out[r] = rstd;
  }
  __syncthreads();
  const float4 h4 = *reinterpret_cast<const float4*>(hs + lane * 4);
#pragma unroll 8
  for (int jj = 0; jj < PER; ++jj) {
    const int j = w * PER + jj;
    const float4 w4 = *reinterpret_cast<const float4*>(W2 + (int64_t)j * D + lane * 4);
    float p = h4.x * w4.x + h4.y * w4.y + h4.z * w4.z + h4.w * w4.w;
    p = wave_sum(p);
    if (lane == 0) out[r * D + j] = p + b2[j];
  }
}

// per row: dh = dout W2 ; relu' ; LN affine grads ; LN backward -> dpre (scratch)
__global__ __launch_bounds__(256) void stem_bwd_row_kernel(const float* __restrict__ dout, const float* __restrict__ xhat,
                                                           const float* __restrict__ rstd, const float* __restrict__ h,
                                                           const float* __restrict__ ln_w, const float* __restrict__ W2,
                                                           float* __restrict__ dln_w, float* __restrict__ dln_b,
                                                           float* __restrict__ dpre) {
  __shared__ float ds[D];
  __shared__ float red[4];
  const int64_t r = blockIdx.x;
  const int i = threadIdx.x;
  ds[i] = dout[r * D + i];
  __syncthreads();
  float dh = 0.f;
#pragma unroll 16
  for (int j = 0; j < D; ++j) dh += ds[j] * W2[(int64_t)j * D + i];
  const float xh = xhat[r * D + i];
  const float dz = h[r * D + i] > 0.f ? dh : 0.f;
  atomicAdd(dln_w + i, dz * xh);
  atomicAdd(dln_b + i, dz);
  const float g = dz * ln_w[i];
  const float s1 = block_sum_256(g, red) * (1.0f / D);
  const float s2 = block_sum_256(g * xh, red) * (1.0f / D);
  dpre[r * D + i] = rstd[r] * (g - s1 - xh * s2);
}

// dW[j][i] += sum_r dy[r][j] * x[r][i];  db[j] += sum_r dy[r][j].  A workgroup owns 256 (j, i) outputs; its four groups of 256 threads
// each sum a quarter of the rows (eight rows' loads in flight) and the quarters are added in a fixed order through LDS.  (One thread
// per output over all 512 rows of the bench shape, one row in flight: 138 us a launch, round 6.)
__global__ __launch_bounds__(1024) void stem_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          float* __restrict__ dW, float* __restrict__ db, int64_t rows,
                                                          int in_dim) {
  __shared__ float part[2][3][256];
  const int t = threadIdx.x & 255, q = threadIdx.x >> 8;
  const int64_t idx = (int64_t)blockIdx.x * 256 + t;
  const bool in = idx < (int64_t)D * in_dim;
  const int j = in ? (int)(idx / in_dim) : 0, i = in ? (int)(idx % in_dim) : 0;
  const int64_t per = (rows + 3) / 4, r0 = q * per, r1 = r0 + per < rows ? r0 + per : rows;
  float acc = 0.f, bacc = 0.f;
#pragma unroll 8
  for (int64_t r = r0; r < r1; ++r) {
    const float g = dy[r * D + j];
    acc += g * x[r * in_dim + i];
    bacc += g;
  }
  if (q > 0) {
    part[0][q - 1][t] = acc;
    part[1][q - 1][t] = bacc;
  }
  __syncthreads();
  if (q == 0 && in) {
    dW[idx] += ((acc + part[0][0][t]) + part[0][1][t]) + part[0][2][t];
    if (i == 0) db[j] += ((bacc + part[1][0][t]) + part[1][1][t]) + part[1][2][t];
  }
}

inline unsigned wave_grid(int64_t rows) {
  int64_t b = (rows + 3) / 4;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int hma_embed_fwd(void* stream, const int64_t* ids, const float* E0, const float* E1, const float* mask_embed,
                             const float* pos, const float* a_emb, float* x, int64_t B, int32_t T, int32_t S, int32_t A,
                             int32_t pos_frame_rows, int32_t V, int64_t mask_id) {
  if (!ids || !E0 || !E1 || !mask_embed || !pos || !x) return HMA_EINVAL;
  if (A > 0 && !a_emb) return HMA_EINVAL;
  if (S + A > pos_frame_rows) return HMA_EINVAL;
  const int64_t rows = B * T * (S + A);
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(wave_grid(rows)), dim3(256), 0, (hipStream_t)stream, ids, E0, E1, mask_embed,
                     pos, a_emb, x, rows, (int)T, (int)S, (int)A, (int)pos_frame_rows, (int)V, mask_id);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_embed_bwd(void* stream, const int64_t* ids, const float* dx, float* dE0, float* dE1, float* dmask_embed,
                             float* dpos, float* da_emb, int64_t B, int32_t T, int32_t S, int32_t A,
                             int32_t pos_frame_rows, int32_t V, int64_t mask_id) {
  if (!ids || !dx || !dE0 || !dE1 || !dmask_embed || !dpos) return HMA_EINVAL;
  if (A > 0 && !da_emb) return HMA_EINVAL;
  if (B <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int SA = S + A;
  const int64_t total = (int64_t)T * SA * (D / 4);
  hipLaunchKernelGGL(embed_bwd_pos_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dx, dpos, B, (int)T, SA,
                     (int)pos_frame_rows);
  HMA_CHECK_LAUNCH();
  const int64_t img_rows = B * T * S;
  if (V <= 512 && img_rows >= 4096 && B * T * (int64_t)SA * (D / 4) < ((int64_t)1 << 32)) {
    // 8 (table, column slice) variants x token groups: one workgroup per CU (a group's token index has 24 bits in the queue words)
    int64_t groups = 32;
    while ((img_rows + groups - 1) / groups > ((int64_t)1 << 24)) groups *= 2;
    const int64_t per = (((img_rows + groups - 1) / groups) + 63) & ~(int64_t)63;
    groups = (img_rows + per - 1) / per;
    hipLaunchKernelGGL(embed_bwd_tok_reg_kernel, dim3((unsigned)(groups * 8)), dim3(1024), 0, s, ids, dx, dE0, dE1, dmask_embed,
                       img_rows, (int)S, SA, (int)V, mask_id, per);
  } else {
    int64_t blocks = (img_rows + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(embed_bwd_tok_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ids, dx, dE0, dE1, dmask_embed, img_rows,
                       (int)S, SA, (int)V, mask_id);
  }
  HMA_CHECK_LAUNCH();
  if (A > 0) {
    hipLaunchKernelGGL(embed_bwd_act_kernel, dim3((unsigned)(B * T)), dim3(256), 0, s, dx, da_emb, (int)S, (int)A);
    HMA_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int hma_action_stem_fwd(void* stream, const float* a, const float* mean, const float* stdv, int32_t action_dim,
                                   const float* W1, const float* b1, const float* ln_w, const float* ln_b, const float* W2,
                                   const float* b2, float* an, float* xhat, float* rstd, float* h, float* out, int64_t rows,
                                   int32_t d_a, int32_t skip_norm) {
  if (!a || !W1 || !b1 || !ln_w || !ln_b || !W2 || !b2 || !an || !xhat || !rstd || !h || !out) return HMA_EINVAL;
  if (!skip_norm && (!mean || !stdv || action_dim <= 0)) return HMA_EINVAL;
  if (d_a <= 0 || d_a > 4096) return HMA_EINVAL;
  if (rows <= 0) return 0;
  const size_t smem = (size_t)(d_a + D + 4) * sizeof(float);
  hipLaunchKernelGGL(stem_fwd_kernel, dim3((unsigned)rows), dim3(64 * SW), smem, (hipStream_t)stream, a, mean, stdv,
                     (int)(action_dim > 0 ? action_dim : 1), W1, b1, ln_w, ln_b, W2, b2, an, xhat, rstd, h, out, (int)d_a,
                     (int)skip_norm);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_action_stem_bwd(void* stream, const float* dout, const float* an, const float* xhat, const float* rstd,
                                   const float* h, const float* ln_w, const float* W2, float* dW1, float* db1, float* dln_w,
                                   float* dln_b, float* dW2, float* db2, float* scratch, int64_t rows, int32_t d_a) {
  if (!dout || !an || !xhat || !rstd || !h || !ln_w || !W2 || !dW1 || !db1 || !dln_w || !dln_b || !dW2 || !db2 || !scratch)
    return HMA_EINVAL;
  if (rows <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(stem_bwd_row_kernel, dim3((unsigned)rows), dim3(256), 0, s, dout, xhat, rstd, h, ln_w, W2, dln_w, dln_b,
                     scratch);
  HMA_CHECK_LAUNCH();
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3((unsigned)((D * D + 255) / 256)), dim3(1024), 0, s, dout, h, dW2, db2, rows, D);
  HMA_CHECK_LAUNCH();
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3((unsigned)(((int64_t)D * d_a + 255) / 256)), dim3(1024), 0, s,
                     (const float*)scratch, an, dW1, db1, rows, (int)d_a);
  HMA_CHECK_LAUNCH();
  return 0;
}
