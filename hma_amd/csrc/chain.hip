// Row-local chains of an STBlock for gfx950: everything between two attentions in ONE launch.
// Reference: STBlock.forward (hma/model/st_transformer.py:85-112), ModulateLayer.forward (hma/model/st_mask_git.py:66-76),
// SelfAttention's qkv / proj Linears (hma/model/attention.py:39,60) and their autograd mirrors.
//
// Structure:
//   * 512 threads: seven COMPUTE waves and one LOADER wave.  A compute wave owns 16 token rows of the 112-row tile for the whole
//     chain.  Lane (tok = lane & 15, g = lane >> 4) holds chunks 4 j + g (8 columns each) of its token row: as the B operand of
//     v_mfma_f32_16x16x32_bf16 (weights = A operand) and -- because the weight rows of a 32-column block are permuted -- as
//     what the MFMAs leave in its accumulators: acc[2 pr + o][r] = column 32 pr + 8 g + 4 o + r.  So the output of one GEMM,
//     packed to bf16, IS the B operand of the next one: no LDS round trip, no shuffles, and every global access of a lane is
//     16 (bf16) or 32 (fp32) contiguous bytes, 64 contiguous bytes per token row and instruction.
//   * Only the weights move through LDS: pre-packed in fragment order (hma_chain_pack: a 16 KB bundle = 32 output columns x 256 k
//     = 16 lane-linear 1 KB fragments), streamed by the loader wave through a four-slot ring with LDS-DMA, one raw s_barrier per
//     bundle.  The loader has no stores in its queue, so its counted vmcnt waits see the bundles only; the compute waves issue no
//     loads inside the steps (biases and the per-frame shift / scale rows sit in LDS), so they never wait behind their own stores:
//     their only loads are the next tile's rows, requested a stage ahead straight into the registers that will hold them.
//   * The L2 -> LDS path is not the limit of this organisation (tools/probes/l2_stream.hip: 56 B/clk per CU by LDS-DMA from two
//     waves on, against ~10 B/clk that a chain needs).
#include <type_traits>
#include <utility>

#include "hma_common.h"
#include "../../include/hma_hip.h"

using namespace hma;

// Debug builds only (tools/chain_variants.sh, -DCH_ABL=bits): 1 no global stores, 2 no row loads, 4 no MFMA, 8 no barrier,
// 16 no weight DMA
#ifndef CH_ABL
#define CH_ABL 0
#endif
namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4v_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

// CH_ST = 1 (measurement builds, tools/chain_variants.sh "st:-DCH_ST=1"): STORER mode -- round 4's test of VERDICT item 6.  A chain's
// stores and its other work ADD (chain A forward: default 250 us, without its global stores 91 us), so here compute waves never store:
// they put every finished 16-row x 128-byte block into a small per-wave LDS ring (`stage_block`) and dedicated STORER waves drain the
// rings into HBM.  s_barrier would tie the storers to the others (every live wave of a workgroup has to arrive), so the weight ring is
// synchronised by LDS words instead: the loader publishes `ready` = bundles landed, each compute wave `done` = steps whose fragments
// it has read; the staging rings carry a sequence number in each block's descriptor.  512 threads = 7 - CH_NSTW compute waves +
// loader + CH_NSTW storers.  Parity-green (tests/test_chain_gpu.py) and SLOWER (profiles/chain_storer_r4.txt): a storer wave needs
// ~650 cycles per 2 KB block whatever feeds it (two storers per CU: 3.2 TB/s at best), the 8 KB rings cannot take a tile's bursts,
// and the flag hand-offs add ~500 cycles to a step -- chain A forward 343-400 us against 245, chain B forward 520 against 410.
// CH_ST = 0 (default): 7 compute waves + loader, one s_barrier per CH_PB steps, stores from the compute waves.
#ifndef CH_ST
#define CH_ST 0
#endif
#ifndef CH_NSTW   // storer waves (one wave has at most 63 stores in flight: ~2.3 TB/s over the chip at the write latency under load)
#define CH_NSTW 2
#endif
#ifndef CH_NCW
#define CH_NCW (CH_ST ? 7 - CH_NSTW : 7)
#endif
#ifndef CH_PB   // CH_ST = 0 only: bundles per barrier (the ring is refilled, and the compute waves synchronise, every CH_PB steps)
#define CH_PB (CH_ST ? 1 : 2)
#endif
#ifndef CH_NS
#define CH_NS (CH_PB == 1 ? 4 : 3 * CH_PB)
#endif
#ifndef CH_R    // CH_ST = 1: blocks per compute wave in its staging ring
#define CH_R 4
#endif
constexpr bool ST = CH_ST != 0;
// A wave BLOCKS at the issue of a vector-memory instruction while the CU's memory pipeline is full, so a burst of row loads (24 per
// lane for a tile's o and x rows) costs about its whole transfer time in issue stalls -- the loads are asynchronous only up to the
// queue.  CH_PF_SPREAD = 1 (default, round 4): a tile's row loads are issued a pair per step over the steps in front of their use
// instead of all at once: chain A forward 256 -> 229 us (tools/chain_variants.sh "pf0:-DCH_PF_SPREAD=0" is the burst form).
#ifndef CH_PF_SPREAD
#define CH_PF_SPREAD 1
#endif
#ifndef CH_PF_FIRST   // chain A forward: the qkv step the spread prefetch starts at (the stores of x / bf16(x) go out at the last step before)
#define CH_PF_FIRST 0
#endif
#ifndef CH_BSPREAD  // chain B forward: the tile's x / next-LN1 / qkv stores dealt over the 24 qkv steps instead of one 1.5 KB-per-row burst + three
#define CH_BSPREAD 0
#endif
#ifndef CH_SPREAD   // measurement builds: CH_ST = 0 with the storer mode's even store schedule (chain A forward)
#define CH_SPREAD 0
#endif
constexpr bool SPREAD = ST || CH_SPREAD != 0;
constexpr int NCW = CH_NCW;                  // compute waves; wave NCW is the loader, wave NCW + 1 (storer mode) the storer
constexpr int NSTW = ST ? CH_NSTW : 0;       // storer k takes the staging rings of the compute waves w with w % NSTW == k
constexpr int CH_THREADS = 64 * (NCW + 1 + NSTW);
constexpr int WGS_PER_CU = NCW <= 3 ? 2 : 1;
constexpr int NS = CH_NS;                    // ring slots
constexpr int PB = CH_PB;                    // steps per barrier (every chain has an even number of steps per tile)
static_assert(NS % PB == 0 && NS >= 2 * PB, "the ring holds whole barrier groups");
static_assert(!ST || PB == 1, "storer mode: per-step flags");
constexpr int SLOT = 16384;                  // one bundle
constexpr int TILE_ROWS = 16 * NCW;
constexpr int L_BIAS = NS * SLOT;            // up to 2304 floats of bias vectors (chain B)
constexpr int L_SS = L_BIAS + (ST ? 9216 : 8192);  // 2 tiles x NCW waves x 2 KB: the frames' shift | scale rows (CH_ST = 0: chain B's biases run into it)
#ifndef CH_SS_BYTES  // (measurement builds of chain B alone with a deeper ring, -DCH_NS=8 -DCH_SS_BYTES=1024: chain B has no shift / scale rows)
#define CH_SS_BYTES (2 * NCW * 2048)
#endif
constexpr int L_SYNC = L_SS + CH_SS_BYTES;         // storer mode: [0] ready | [8 + w] done | [16 + w] freed (32-bit words)
constexpr int RB = CH_R;                           // staging blocks per compute wave
#ifndef CH_RG   // blocks reserved per wait: small groups let a wave refill its ring while the storer drains the rest of it
#define CH_RG 2
#endif
constexpr int RG = CH_RG;
static_assert(RG <= RB && (RG & (RG - 1)) == 0, "reserve groups divide the ring");
static_assert((RB & (RB - 1)) == 0, "power of two");
constexpr int L_DESC = L_SYNC + 256;               // NCW x RB descriptors of 16 bytes: {address lo, hi, row pitch, sequence number}
constexpr int L_STG = L_DESC + NCW * RB * 16;      // NCW x RB blocks of 16 rows x 128 bytes
constexpr int SMEM = ST ? L_STG + NCW * RB * 2048 : L_SYNC;

__device__ __forceinline__ bf16x8_t lds_frag(HMA_LDS(char)* p) { return __builtin_bit_cast(bf16x8_t, *(HMA_LDS(u32x4_t)*)p); }
__device__ __forceinline__ float4 lds_f4(HMA_LDS(char)* p) {
  const f32x4v_t v = *(HMA_LDS(f32x4v_t)*)p;
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ f32x4v_t lds_f4v(HMA_LDS(char)* p) { return *(HMA_LDS(f32x4v_t)*)p; }
__device__ __forceinline__ int64_t remap_row(int64_t r, int64_t group_rows, int64_t group_stride) {
  return group_rows > 0 ? (r / group_rows) * group_stride + (r % group_rows) : r;
}
__device__ __forceinline__ f32x4v_t mfma16(const bf16x8_t& a, const bf16x8_t& b, const f32x4v_t& c) {
  if (CH_ABL & 4) {
    f32x4v_t r = c;
    r[0] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4_t, a)[0] ^ __builtin_bit_cast(u32x4_t, b)[1]);
    return r;
  }
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8_t as_frag(const uint4& v) { return __builtin_bit_cast(bf16x8_t, v); }

// Debug builds only (-DCH_PROF): per-wave cycle counters summed per phase, workgroup 0 (tools/chain_bench.py CH_PROF=1)
#ifdef CH_PROF
__device__ unsigned long long g_ch_prof[2][8][8];
#define CPROF_DECL unsigned long long pt_ = __builtin_readcyclecounter(), pacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define CPROF_MARK(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); pacc_[i] += n_ - pt_; pt_ = n_; } while (0)
#define CPROF_FLUSH(k, wv) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 8; ++i_) g_ch_prof[k][wv][i_] = pacc_[i_]; } } while (0)
#else
#define CPROF_DECL
#define CPROF_MARK(i)
#define CPROF_FLUSH(k, wv)
#endif
#ifdef CH_FENCE
#define CH_SCHED_FENCE __builtin_amdgcn_sched_barrier(0);
#else
#define CH_SCHED_FENCE
#endif
#define CH_BARRIER()                                      \
  do {                                                    \
    if (!(CH_ABL & 8)) __builtin_amdgcn_s_barrier();      \
    asm volatile("" ::: "memory");                        \
    CH_SCHED_FENCE                                        \
  } while (0)

// ------------------------------------------------------------------------------------------------ storer mode: flags and staging
__device__ __forceinline__ uint32_t lds_poll(HMA_LDS(char)* p) {
  return __builtin_amdgcn_readfirstlane(*(volatile HMA_LDS(uint32_t)*)p);
}
// wait until the LDS word at p is >= need (`seen` caches the last value read: the producer usually runs ahead)
__device__ __forceinline__ void wait_word(HMA_LDS(char)* p, uint32_t need, uint32_t& seen) {
  while ((int32_t)(seen - need) < 0) {
    seen = lds_poll(p);
    if ((int32_t)(seen - need) >= 0) break;
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void post_word(HMA_LDS(char)* p, uint32_t v, int lane) {
  asm volatile("" ::: "memory");  // (LDS operations of a wave execute in order: everything it read or wrote before is done first)
  if (lane == 0) *(volatile HMA_LDS(uint32_t)*)p = v;
}
// what a compute wave carries through a kernel in storer mode
struct stage_t {
  HMA_LDS(char)* stg;    // this wave's RB blocks
  HMA_LDS(char)* desc;   // their descriptors
  HMA_LDS(char)* freed;  // blocks of this wave the storer has taken out of LDS
  HMA_LDS(char)* ready;  // bundles landed (loader)
  HMA_LDS(char)* done;   // steps this wave is through with
  uint32_t nb, freed_seen, ready_seen, sg;  // blocks staged; cached words; steps begun
  int lane;
};
__device__ __forceinline__ stage_t make_stage(HMA_LDS(char)* lds, int wave, int lane) {
  stage_t st;
  st.stg = lds + L_STG + wave * RB * 2048;
  st.desc = lds + L_DESC + wave * RB * 16;
  st.freed = lds + L_SYNC + 64 + 4 * wave;
  st.ready = lds + L_SYNC;
  st.done = lds + L_SYNC + 32 + 4 * wave;
  st.nb = st.freed_seen = st.ready_seen = st.sg = 0;
  st.lane = lane;
  return st;
}
// one 16-row x 128-byte block (rows `pitch` bytes apart from `base`): this lane's two 16-byte pieces go to LDS byte offsets s0 / s1
// of the block (row tok, chunk c at c ^ (tok & 7): conflict-free here and for the storer's row-linear reads)
// n <= RB blocks may be staged without further checks (one wait loop per burst instead of one per block)
__device__ __forceinline__ void stage_reserve(stage_t& st, int n) {
  if constexpr (ST) wait_word(st.freed, st.nb + (uint32_t)n - RB, st.freed_seen);
}
template <bool WAIT = true>
__device__ __forceinline__ void stage_block(stage_t& st, const void* base, uint32_t pitch, int s0, int s1, const uint4& q0, const uint4& q1) {
  if constexpr (WAIT) wait_word(st.freed, st.nb + 1 - RB, st.freed_seen);
  HMA_LDS(char)* d = st.stg + (st.nb & (RB - 1)) * 2048;
  *(HMA_LDS(u32x4_t)*)(d + s0) = __builtin_bit_cast(u32x4_t, q0);
  *(HMA_LDS(u32x4_t)*)(d + s1) = __builtin_bit_cast(u32x4_t, q1);
  asm volatile("" ::: "memory");
  const uint64_t a = (uint64_t)(uintptr_t)base;
  if (st.lane == 0)
    *(volatile HMA_LDS(u32x4_t)*)(st.desc + (st.nb & (RB - 1)) * 16) = u32x4_t{(uint32_t)a, (uint32_t)(a >> 32), pitch, st.nb + 1};
  ++st.nb;
}
// the step protocol of a compute wave: begin -> fragments of bundle sg may be read; end -> they have been
__device__ __forceinline__ void step_begin(stage_t& st) {
  if constexpr (ST) wait_word(st.ready, st.sg + 1, st.ready_seen);
}
__device__ __forceinline__ void step_end(stage_t& st) {
  if constexpr (ST) {
    ++st.sg;
    post_word(st.done, st.sg, st.lane);
  }
}
__device__ __forceinline__ void steps_skip(stage_t& st, int n) {  // a wave whose rows lie past the matrix
  if constexpr (ST) {
    st.sg += n;
    post_word(st.done, st.sg, st.lane);
  }
}
__device__ __forceinline__ void stage_finish(stage_t& st) {  // the wave's last block: tells the storer it is through
  if constexpr (ST) stage_block(st, nullptr, 0xffffffffu, st.lane * 16, 1024 + st.lane * 16, make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0));
}
// The storer wave.  One round: lane l < NW x RB reads the descriptor of staging block l (wave l / RB, slot l % RB) -- one LDS
// instruction sees every ring -- and a block is complete when its sequence number is the one its slot holds next.  A wave fills
// its slots in order, so the complete blocks of a wave are a run behind the ones already taken.  They go to HBM in batches of up to
// eight: all sixteen LDS reads first, then sixteen whole-line store instructions (8 rows x 128 bytes each) -- a block at a time
// the loop was bound by the LDS round trip (~400 cycles per block against the ~230 the write bandwidth leaves).
template <int NW>
__device__ __forceinline__ void storer_run(HMA_LDS(char)* lds, int lane, int k) {
  __builtin_amdgcn_s_setprio(3);
  static_assert(NW * RB <= 64, "one lane per staging block");
  const int w_ = lane < NW * RB ? lane / RB : 0, j_ = lane & (RB - 1);
  const bool mine = lane < NW * RB && w_ % (NSTW > 0 ? NSTW : 1) == k;
  uint32_t all = 0;  // the compute waves this storer serves
  for (int w = k; w < NW; w += (NSTW > 0 ? NSTW : 1)) all |= 1u << w;
  uint32_t sent = 0;  // blocks of wave w_ taken (the same value in the RB lanes of a wave)
  uint32_t fin = 0;
  const int r8 = lane >> 3, lc = (lane & 7) ^ (r8 & 7);
  HMA_LDS(char)* blk = lds + L_STG + lane * 16;
  CPROF_DECL;
  while (true) {
    const u32x4_t d = *(volatile HMA_LDS(u32x4_t)*)(lds + L_DESC + (mine ? lane : 0) * 16);
    const uint32_t nxt = sent + ((uint32_t)(j_ - (int)sent) & (RB - 1));  // the block this slot holds next
    uint64_t mask = __ballot(mine && d[3] == nxt + 1);
    if (mask == 0) {
      if (fin == all) break;
      __builtin_amdgcn_s_sleep(1);
      CPROF_MARK(0);
      continue;
    }
    CPROF_MARK(1);
#ifdef CH_PROF
    pacc_[5] += 1;
    pacc_[6] += __builtin_popcountll(mask);
#endif
    const uint32_t took = (uint32_t)__builtin_popcountll((mask >> (w_ * RB)) & ((1ull << RB) - 1));
    while (mask) {
      int idx[8];
      u32x4_t v0[8], v1[8];
      uint32_t lo[8], hi[8], pitch[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        idx[k] = -1;
        if (mask) {
          const int b = __builtin_amdgcn_readfirstlane(__builtin_ctzll(mask));
          mask &= mask - 1;
          lo[k] = __builtin_amdgcn_readlane(d[0], b);
          hi[k] = __builtin_amdgcn_readlane(d[1], b);
          pitch[k] = __builtin_amdgcn_readlane(d[2], b);
          if (pitch[k] == 0xffffffffu) {
            fin |= 1u << (b / RB);
          } else {
            idx[k] = b;
            v0[k] = *(HMA_LDS(u32x4_t)*)(blk + b * 2048);
            v1[k] = *(HMA_LDS(u32x4_t)*)(blk + b * 2048 + 1024);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (idx[k] >= 0 && !(CH_ABL & 1)) {
          char* g = reinterpret_cast<char*>((uintptr_t)(((uint64_t)hi[k] << 32) | lo[k])) + (uint64_t)r8 * pitch[k] + lc * 16;
          __builtin_nontemporal_store(v0[k], reinterpret_cast<u32x4_t*>(g));
          __builtin_nontemporal_store(v1[k], reinterpret_cast<u32x4_t*>(g + 8 * (uint64_t)pitch[k]));
        }
      }
    }
    sent += took;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this wave's reads of the blocks are done before they are handed back)
    if (mine && j_ == 0) *(volatile HMA_LDS(uint32_t)*)(lds + L_SYNC + 64 + 4 * w_) = sent;
    CPROF_MARK(2);
#ifdef CH_ST_DEPTH  // (measurement builds: at most this many store instructions in flight)
    __builtin_amdgcn_s_waitcnt(0x0f70 | (CH_ST_DEPTH & 15) | ((CH_ST_DEPTH >> 4) << 14));
#endif
  }
  CPROF_FLUSH(0, NCW + 1 + k);
}
// Measurement builds (-DCH_STAGGER=n): workgroups start up to n x 3.7 us apart, so that the CUs are not all in the same step of
// their tiles (every CU storing at once, then none)
__device__ __forceinline__ void start_stagger() {
#ifdef CH_STAGGER
  const int nsl = (((int)blockIdx.x * 37) & 255) * CH_STAGGER / 256;
  for (int i = 0; i < nsl; ++i) __builtin_amdgcn_s_sleep(127);
#endif
}
// zero the flag words and descriptors (before the kernel's first __syncthreads)
__device__ __forceinline__ void sync_init(HMA_LDS(char)* lds, int tid) {
  if constexpr (ST) {
    for (int i = tid; i < (L_STG - L_SYNC) / 4; i += CH_THREADS) ((HMA_LDS(uint32_t)*)(lds + L_SYNC))[i] = 0;
  }
}

// ------------------------------------------------------------------------------------------------ weight packing
__device__ __forceinline__ void chain_pack_body(const float* __restrict__ src, int64_t rs, int64_t cs,
                                                const float* __restrict__ rscale, const float* __restrict__ cscale,
                                                uint16_t* __restrict__ dst, int kind, int nbundles, int64_t sstride, int64_t dstride,
                                                int bundle_stride, int bx, int64_t bz) {
  src += bz * sstride;
  if (rscale) rscale += bz * sstride;
  if (cscale) cscale += bz * sstride;
  dst += bz * dstride;
  const int idx = bx * 256 + threadIdx.x;  // (bundle, fragment, lane)
  if (idx >= nbundles * 1024) return;
  const int lane = idx & 63, frag = (idx >> 6) & 15, b = idx >> 10;
  const int i = lane & 15, g = lane >> 4;
  int row, col0;
  if (kind == 0) {
    const int j = frag >> 1, o = frag & 1;
    row = 32 * b + 8 * (i >> 2) + (i & 3) + 4 * o;
    col0 = 8 * (4 * j + g);
  } else {
    row = 32 * (frag >> 1) + 8 * (i >> 2) + (i & 3) + 4 * (frag & 1);
    col0 = 32 * b + 8 * g;
  }
  const float rsc = rscale ? rscale[row] : 1.0f;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = col0 + e;
    float w = src[(int64_t)row * rs + (int64_t)c * cs] * rsc;
    if (cscale) w *= cscale[c];
    v[e] = w;
  }
  *reinterpret_cast<uint4*>(dst + ((int64_t)b * bundle_stride * 1024 + (idx & 1023)) * 8) = pack8(v);
}
__global__ __launch_bounds__(256) void chain_pack_kernel(const float* __restrict__ src, int64_t rs, int64_t cs,
                                                         const float* __restrict__ rscale, const float* __restrict__ cscale,
                                                         uint16_t* __restrict__ dst, int kind, int nbundles, int64_t sstride,
                                                         int64_t dstride, int bundle_stride) {
  chain_pack_body(src, rs, cs, rscale, cscale, dst, kind, nbundles, sstride, dstride, bundle_stride, blockIdx.x, blockIdx.y);
}
// hma_chain_pack_multi: the jobs' (nb * 4) x batch grids laid end to end in blockIdx.x (b0 = prefix sums)
constexpr int PACK_JOBS = 24;
struct pack_jobs {
  hma_pack_job_t j[PACK_JOBS];
  int b0[PACK_JOBS + 1];
  int n;
};
__global__ __launch_bounds__(256) void chain_pack_multi_kernel(pack_jobs jobs) {
  int k = 0;
  while (k + 1 < jobs.n && (int)blockIdx.x >= jobs.b0[k + 1]) ++k;
  const hma_pack_job_t& j = jobs.j[k];
  const int nb = (j.kind == 0 ? j.rows : j.cols) / 32, rem = (int)blockIdx.x - jobs.b0[k];
  chain_pack_body(j.src, j.row_stride, j.col_stride, j.row_scale, j.col_scale, reinterpret_cast<uint16_t*>(j.dst), j.kind, nb,
                  j.src_batch_stride, j.dst_batch_stride, j.bundle_stride, rem % (nb * 4), rem / (nb * 4));
}

// ------------------------------------------------------------------------------------------------ rows of a workgroup
// CH_RAGGED = 1 (round 5): a workgroup owns ONE contiguous run of rows -- ceil(M / workgroups) rounded up to 16 -- and walks it in
// tiles of 16 NW rows; the LAST tile of every workgroup is ragged (some waves past the run's end only keep the barriers company).  With
// whole tiles dealt round-robin (CH_RAGGED = 0), M = 163 840 is 1 463 tiles of 112 rows on 256 workgroups: 183 of them run six tiles, 73
// five, and the launch takes six tile times for 5.71 tiles of work per CU; now every workgroup has 640 rows = five tiles + one of 80 rows,
// whose loads and stores -- what a chain's time is made of -- are 5 / 7 of a full tile's.
#ifndef CH_RAGGED
#define CH_RAGGED 1
#endif
struct tile_map {
  int64_t base, end;  // this workgroup's rows [base, end)
  int nt;             // its tiles
};
__device__ __forceinline__ tile_map make_tile_map(int64_t M, int nw) {
  tile_map t;
  if (CH_RAGGED) {
    const int64_t per = (((M + gridDim.x - 1) / gridDim.x) + 15) / 16 * 16;
    t.base = (int64_t)blockIdx.x * per;
    t.end = t.base + per < M ? t.base + per : M;
    t.nt = t.end > t.base ? (int)((t.end - t.base + 16 * nw - 1) / (16 * nw)) : 0;
  } else {
    const int64_t ntiles = (M + 16 * nw - 1) / (16 * nw);
    t.base = 0;
    t.end = M;
    t.nt = (int)((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x);
  }
  return t;
}
// first row of wave `wave` in the workgroup's tile tl
__device__ __forceinline__ int64_t tile_row0(const tile_map& t, int tl, int nw, int wave) {
  if (CH_RAGGED) return t.base + ((int64_t)tl * nw + wave) * 16;
  return (((int64_t)blockIdx.x + (int64_t)tl * gridDim.x) * nw + wave) * 16;
}

// ------------------------------------------------------------------------------------------------ the loader wave
// Streams the chain's bundles, `per_tile` per tile and `nt` tiles, through the ring: bundle s goes to slot s % NS and is
// complete (this wave's vmcnt) before the wave arrives at barrier s; the slot it frees -- the compute waves are past their
// reads of bundle s - 1 when they arrive at barrier s -- is refilled right behind the barrier.  In front of a tile's first
// bundle go the shift / scale rows of the frames its seven 16-row groups lie in (2 KB each).
struct ring_src {
  const char *s0, *s1, *s2, *s3;
  int n0, n1, n2, n3;
  const char *s4 = nullptr, *s5 = nullptr;  // (the fused chain A + temporal attention + chain B kernel streams six segments)
  int n4 = 0, n5 = 0;
};
template <int PROF_K, int NW = NCW>
__device__ __forceinline__ void loader_run(const ring_src& ws, int per_tile, int nt, uint32_t lds_b, int lane, const float* ss,
                                           int64_t M, int rows_per_frame, HMA_LDS(char)* lds = nullptr, const tile_map* tmap = nullptr,
                                           int64_t col_base = -1, int col_sa = 0, uint32_t col_lds = 0) {
  if constexpr (ST) __builtin_amdgcn_s_setprio(2);
  const int total = per_tile * nt;
  int issued = 0, seg = 0, left = ws.n0, tl_issue = 0, slot_issue = 0;
  const char* cur = ws.s0;
  auto issue = [&]() __attribute__((always_inline)) {
    if (seg == 0 && left == ws.n0 && ss && col_base >= 0) {
      // column tiles (hma_chain_ab_fwd): the 16 frames' shift | scale rows of the sample of the tile's FIRST column, 2 KB each, rows
      // 2064 bytes apart in LDS (a lane reads its own frame's row: the 16-byte skew spreads the 16 rows over the banks)
      const int64_t b0 = (col_base + (int64_t)tl_issue * NW) / col_sa;
      const char* s = reinterpret_cast<const char*>(ss) + b0 * 16 * 2048 + lane * 16;
#pragma unroll 1
      for (int f = 0; f < 16; ++f) {
        const uint32_t d = __builtin_amdgcn_readfirstlane(lds_b + col_lds + f * 2064);
        glds16(s + f * 2048, d);
        glds16(s + f * 2048 + 1024, d + 1024);
      }
    } else if (seg == 0 && left == ws.n0 && ss) {  // a tile's first bundle: its shift / scale rows first
#pragma unroll 1
      for (int w = 0; w < NW; ++w) {
        int64_t r0 = tile_row0(*tmap, tl_issue, NW, w);
        r0 = r0 < M ? r0 : M - 1;
        const char* s = reinterpret_cast<const char*>(ss) + (r0 / rows_per_frame) * 2048 + lane * 16;
        const uint32_t d = __builtin_amdgcn_readfirstlane(lds_b + L_SS + ((tl_issue & 1) * NCW + w) * 2048);
        glds16(s, d);
        glds16(s + 1024, d + 1024);
      }
    }
    const char* s = cur + lane * 16;
    const uint32_t d = __builtin_amdgcn_readfirstlane(lds_b + slot_issue * SLOT);
    if (!(CH_ABL & 16)) {
      glds16x4(s, d);
      glds16x4(s + 4096, d + 4096);
      glds16x4(s + 8192, d + 8192);
      glds16x4(s + 12288, d + 12288);
    }
    ++issued;
    slot_issue = slot_issue + 1 == NS ? 0 : slot_issue + 1;
    cur += SLOT;
    if (--left == 0) {
      ++seg;
      int nn = seg == 1 ? ws.n1 : seg == 2 ? ws.n2 : seg == 3 ? ws.n3 : seg == 4 ? ws.n4 : seg == 5 ? ws.n5 : 0;
      if (nn == 0) {
        seg = 0;
        nn = ws.n0;
        ++tl_issue;
      }
      cur = seg == 0 ? ws.s0 : seg == 1 ? ws.s1 : seg == 2 ? ws.s2 : seg == 3 ? ws.s3 : seg == 4 ? ws.s4 : ws.s5;
      left = nn;
    }
  };
  // With PB bundles per barrier the group [s, s + PB) is complete before barrier s / PB, and the PB slots the compute waves have
  // left -- they are past group s / PB - 1 when they arrive -- are refilled behind it.
#pragma unroll 1
  for (int b = 0; b < NS - PB && b < total; ++b) issue();
  CPROF_DECL;
#pragma unroll 1
  for (int s = 0; s < total; s += PB) {
    CPROF_MARK(2);
    const int ahead = issued - PB - s;  // bundles issued after the group (each 16 pieces; shift / scale pieces only make the wait stricter)
    if (ahead >= 3)  // (rings deeper than four slots; the counter has 6 bits, so more than three bundles behind s cannot be told apart)
      asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    else if (ahead == 2)
      asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if (ahead == 1)
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CPROF_MARK(0);
    if constexpr (ST) {
      // bundle s has landed: publish it; the slot of bundle s - 1 is refilled once every compute wave is through with step s - 1
      post_word(lds + L_SYNC, (uint32_t)(s + 1), lane);
      if (issued < total) {
        const uint32_t need = (uint32_t)(issued - NS + 1);
        while (true) {
          const uint32_t dv = *(volatile HMA_LDS(uint32_t)*)(lds + L_SYNC + 32 + 4 * (lane < NW ? lane : 0));
          if (__ballot((int32_t)(dv - need) >= 0) == ~0ull) break;
          __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        CPROF_MARK(1);
        issue();
      }
    } else {
      CH_BARRIER();
      CPROF_MARK(1);
#pragma unroll
      for (int k = 0; k < PB; ++k)
        if (issued < total) issue();
    }
  }
  CPROF_MARK(2);
  CPROF_FLUSH(PROF_K, NCW);
}

// one N-block bundle against the wave's rows: c0 / c1 += W[32 columns] . a   (16 MFMAs)
__device__ __forceinline__ void nb_mma(HMA_LDS(char)* wb, const bf16x8_t (&a)[8], f32x4v_t& c0, f32x4v_t& c1) {
  bf16x8_t f[4], fn[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) f[q] = lds_frag(wb + q * 1024);
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    if (h < 3) {
#pragma unroll
      for (int q = 0; q < 4; ++q) fn[q] = lds_frag(wb + (4 * h + 4 + q) * 1024);
    }
    c0 = mfma16(f[0], a[2 * h], c0);
    c1 = mfma16(f[1], a[2 * h], c1);
    c0 = mfma16(f[2], a[2 * h + 1], c0);
    c1 = mfma16(f[3], a[2 * h + 1], c1);
    __builtin_amdgcn_sched_barrier(0);  // (keeps the scheduler from hoisting every fragment read: it spills)
#pragma unroll
    for (int q = 0; q < 4; ++q) f[q] = fn[q];
  }
}

__device__ __forceinline__ void add4(f32x4v_t& c, const float4& b) { c[0] += b.x; c[1] += b.y; c[2] += b.z; c[3] += b.w; }
__device__ __forceinline__ uint4 pack_pair(const f32x4v_t& c0, const f32x4v_t& c1) {
  return make_uint4(pack_bf16(c0[0], c0[1]), pack_bf16(c0[2], c0[3]), pack_bf16(c1[0], c1[1]), pack_bf16(c1[2], c1[3]));
}
__device__ __forceinline__ f32x4v_t ld4(const float* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  return f32x4v_t{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void st4(float* p, const f32x4v_t& v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }

// ---- whole-line stores.  In the accumulator layout a store instruction writes, per token row, the 64 bytes of the row's four
// lanes (bf16) or four 16-byte pieces 32 bytes apart (fp32): sixteen half-filled 128-byte lines per instruction, and stores of
// that shape were what the first version of these kernels spent two thirds of its time on (ablation: 266 -> 85 us without them).
// Two 16-byte pieces q0 | q1 of a lane that are 64 (bf16: the blocks pr, pr + 1) or 16 (fp32: the block's halves) bytes apart are
// therefore traded across the two halves of the lane group first (DPP row_ror:8: lane tok <-> tok ^ 8): instruction A then writes
// rows 0..7 of the wave's tile, instruction B rows 8..15, each row as ONE full 128-byte line from eight lanes.
__device__ __forceinline__ uint4 dpp_swap8(const uint4& v) {
  uint4 r;
  r.x = __builtin_amdgcn_mov_dpp(v.x, 0x128, 0xf, 0xf, true);
  r.y = __builtin_amdgcn_mov_dpp(v.y, 0x128, 0xf, 0xf, true);
  r.z = __builtin_amdgcn_mov_dpp(v.z, 0x128, 0xf, 0xf, true);
  r.w = __builtin_amdgcn_mov_dpp(v.w, 0x128, 0xf, 0xf, true);
  return r;
}
struct line_offs {
  int a, b;  // byte offsets from the tile's first row: this lane's piece in rows 0..7 (instruction A) / rows 8..15 (instruction B)
  int ra;    // the row (0..7) this lane writes with instruction A; instruction B: ra + 8
  bool lo;
  int pitch, s0, s1;  // storer mode: the row pitch and the LDS byte offsets of q0 / q1 inside a staged block
};
// pitch = row pitch in bytes; g16 = byte offset of the lane group's q0 inside the 128-byte line; d = distance of q1 behind q0
__device__ __forceinline__ line_offs make_lines(int pitch, int tok, int g16, int d) {
  line_offs L;
  L.lo = tok < 8;
  L.ra = tok & 7;
  L.a = (tok & 7) * pitch + g16 + (L.lo ? 0 : d);
  L.b = ((tok & 7) + 8) * pitch + g16 + (L.lo ? d : 0);
  L.pitch = pitch;
  L.s0 = tok * 128 + ((((g16 >> 4)) ^ (tok & 7)) << 4);
  L.s1 = tok * 128 + (((((g16 + d) >> 4)) ^ (tok & 7)) << 4);
  return L;
}
// a 16-row x 128-byte block of an output array: rows from tile_base + off, the lane's pieces q0 | q1.  Storer mode: into the wave's
// staging ring; otherwise straight to HBM as two whole-line instructions.
// MASK (column tiles of windows with fewer than 16 frames): instruction A writes the tile's rows 0..7 (this lane: row L.ra), instruction B
// rows 8..15 (row L.ra + 8); rows >= nrows do not exist and their lanes are switched off (exec-masked stores in straight-line code)
template <bool WAIT = true, bool MASK = false>
__device__ __forceinline__ void store_lines(stage_t& st, void* tile_base, const line_offs& L, int off, const uint4& q0, const uint4& q1,
                                            int nrows = 16) {
  if constexpr (ST) {
    stage_block<WAIT>(st, reinterpret_cast<char*>(tile_base) + off, (uint32_t)L.pitch, L.s0, L.s1, q0, q1);
    return;
  }
  if (CH_ABL & 1) {  // (measurement: no store instruction, but the values stay live -- without this the compiler also drops the GEMMs
                     // whose results are only stored: chain A's qkv, chain B's next-block qkv)
    asm volatile("" ::"v"(q0.x), "v"(q0.y), "v"(q0.z), "v"(q0.w), "v"(q1.x), "v"(q1.y), "v"(q1.z), "v"(q1.w));
    return;
  }
  const uint4 r = dpp_swap8(q1);
  uint4 da, db;
  da.x = L.lo ? q0.x : r.x; da.y = L.lo ? q0.y : r.y; da.z = L.lo ? q0.z : r.z; da.w = L.lo ? q0.w : r.w;
  db.x = L.lo ? r.x : q0.x; db.y = L.lo ? r.y : q0.y; db.z = L.lo ? r.z : q0.z; db.w = L.lo ? r.w : q0.w;
  char* b = reinterpret_cast<char*>(tile_base);
  if constexpr (MASK) {
    if (L.ra < nrows) __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, da), reinterpret_cast<u32x4_t*>(b + (L.a + off)));
    if (L.ra + 8 < nrows) __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, db), reinterpret_cast<u32x4_t*>(b + (L.b + off)));
    return;
  }
#ifndef CH_NO_NT  // (non-temporal: 8 % faster than the default policy in the forward chain, the lines are not read again soon)
  __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, da), reinterpret_cast<u32x4_t*>(b + (L.a + off)));
  __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, db), reinterpret_cast<u32x4_t*>(b + (L.b + off)));
#else
  *reinterpret_cast<uint4*>(b + (L.a + off)) = da;
  *reinterpret_cast<uint4*>(b + (L.b + off)) = db;
#endif
}
__device__ __forceinline__ uint4 as_u4(const f32x4v_t& v) { return __builtin_bit_cast(uint4, v); }

// (opaque use of prefetched registers at the end of a loop body: hipcc then places the counted vmcnt wait there, in straight-line
// code behind the stores it can count, instead of a vmcnt(0) at the loop head)
// compile-time step loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}) -- the step index has to be
// a constant EXPRESSION wherever it selects a register (an index that is only constant after unrolling leaves the arrays in scratch)
template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

#define CH_TOUCH_A(a) _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) asm volatile("" : "+v"(a[j_]))
#define CH_TOUCH_ACC(c) _Pragma("unroll") for (int j_ = 0; j_ < 16; ++j_) asm volatile("" : "+v"(c[j_]))

// ------------------------------------------------------------------------------------------------ chain A, forward
// M % 16 == 0 (checked by the launcher): a wave's 16 rows are all inside the matrix or all outside it, so the tile body has
// no per-lane predication at all -- every store is unconditional straight-line code, which lets hipcc COUNT the stores
// issued behind the next tile's row loads (counted s_waitcnt vmcnt instead of a drain of the store queue).  A wave whose rows
// lie past M only keeps the barriers company.
//
// Anti-phase stores.  Measured on the first version (tools/chain_variants.sh): loads + stores alone 188 us, MFMAs + LDS + barriers
// alone 85 us, together 280 us -- the SUM.  A wave issues in order, so a store that waits for room in the memory pipeline (the
// normal state of an HBM-bound kernel) holds the wave's MFMAs behind it, and with one barrier per step every wave of the chip is
// in the same place.  The two waves of a SIMD therefore run the steps in opposite order: waves 0..3 ("early") multiply, then
// store what the step produced; waves 4..6 ("late") first store what their PREVIOUS step produced, then multiply.  Behind
// every barrier one wave of a SIMD is on the matrix pipe while the other one's stores drain.
// NW = compute waves per workgroup (7; 5 for passes of fewer than 7 x 16 x #CUs rows -- a decode frame pass of 20 480 rows is 256
// tiles of 80 rows instead of 183 of 112: every CU gets one).  Waves NW .. 6 leave at once; wave 7 is the loader.
template <bool MOD, bool SAVE, int NW = NCW>
__global__ __launch_bounds__(CH_THREADS, 2) void chain_a_fwd_kernel(hma_chain_a_fwd_t p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  HMA_LDS(char)* lds = (HMA_LDS(char)*)smem;
  const uint32_t lds_b = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const tile_map tmap = make_tile_map(p.M, NW);
  const int nt = tmap.nt;
  {
    HMA_LDS(float)* bl = (HMA_LDS(float)*)(lds + L_BIAS);  // proj 0..255 | lin 256..511 | qkv 512..1279
    for (int i = tid; i < 1280; i += CH_THREADS) {
      float v = 0.f;
      if (i < 256) v = p.b_proj ? p.b_proj[i] : 0.f;
      else if (i < 512) v = (MOD && p.b_lin) ? p.b_lin[i - 256] : 0.f;
      else v = p.b_qkv ? p.b_qkv[i - 512] : 0.f;
      bl[i] = v;
    }
  }
  sync_init(lds, tid);
  start_stagger();
  __syncthreads();
  constexpr int S3 = MOD ? 16 : 8;       // first qkv step
  constexpr int PER_TILE = S3 + 24;
  static_assert(PER_TILE % PB == 0, "whole barrier groups per tile (the loader refills PB slots per barrier)");
  if (ST && wave > NCW) {
    storer_run<NW>(lds, lane, wave - NCW - 1);
    return;
  }
  if (wave >= NW && wave != NCW) return;
  if (wave == NCW) {
    const ring_src ws = {reinterpret_cast<const char*>(p.w.seg[0]), reinterpret_cast<const char*>(p.w.seg[1]),
                         reinterpret_cast<const char*>(p.w.seg[2]), reinterpret_cast<const char*>(p.w.seg[3]),
                         p.w.bundles[0], p.w.bundles[1], p.w.bundles[2], p.w.bundles[3]};
    loader_run<0, NW>(ws, PER_TILE, nt, lds_b, lane, MOD ? p.ss : nullptr, p.M, p.rows_per_frame, lds, &tmap);
    return;
  }
  stage_t stg_ = make_stage(lds, wave, lane);
  const int tok = lane & 15, g = lane >> 4;
  auto row0_of = [&](int tl) __attribute__((always_inline)) {
    return tile_row0(tmap, tl, NW, wave);
  };
  bf16x8_t a0[8], a1[8];
  f32x4v_t acc[16];
  uint4 hqs[8];   // packed xhat of the tile in flight
  uint4 qb[8];    // qkv blocks waiting for their burst (block pq in qb[pq & 7])
  // the next tile's rows: o (8 loads) and x (16 loads).  `part` < 0: all of them; 0..11: loads 2 part, 2 part + 1 -- measurement builds
  // (-DCH_PF_SPREAD) issue a pair per step instead of 24 at once
  auto prefetch = [&](int tl, auto part_) __attribute__((always_inline)) {
    constexpr int part = decltype(part_)::value;
    if (CH_ABL & 2) return;
    int64_t m = row0_of(tl) + tok;
    m = m < p.M ? m : p.M - 1;
    const uint16_t* orow = reinterpret_cast<const uint16_t*>(p.o) + m * 256 + 8 * g;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (part < 0 || j / 2 == part) a1[j] = as_frag(*reinterpret_cast<const uint4*>(orow + 32 * j));
    const float* xrow = p.x + m * 256 + 8 * g;
#pragma unroll
    for (int pr = 0; pr < 8; ++pr) {
      if (part < 0 || pr + 4 == part) {
        acc[2 * pr] = ld4(xrow + 32 * pr);
        acc[2 * pr + 1] = ld4(xrow + 32 * pr + 4);
      }
    }
  };
  if (CH_ABL & 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) a1[j] = as_frag(make_uint4(lane, j, lane, j));
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = f32x4v_t{0.f, 1.f, 2.f, 3.f};
  }
  prefetch(0, std::integral_constant<int, -1>{});
  CH_TOUCH_A(a1);     // (waited for HERE: a load still pending at the loop head would make hipcc drain the stores of every tile there)
  CH_TOUCH_ACC(acc);
  HMA_LDS(char)* ring = lds + lane * 16;
  HMA_LDS(char)* bias = lds + L_BIAS + 32 * g;
  const line_offs Lb = make_lines(512, tok, 16 * g, 64);                 // bf16 [., 256] outputs: the blocks pr | pr + 1
  const line_offs Lf = make_lines(1024, tok, 32 * g, 16);                // fp32 [., 256]: the two halves of a block
  const line_offs Lq = make_lines((int)p.ldq * 2, tok, 16 * g, 64);      // qkv
  int slot = 0;

  CPROF_DECL;
  auto tile = [&](int tl, int64_t r0) __attribute__((always_inline)) {
    float* xt = p.x + r0 * 256;
    uint16_t* xb = reinterpret_cast<uint16_t*>(p.x_bf16) + r0 * 256;
    uint16_t* xh = reinterpret_cast<uint16_t*>(p.xhat) + r0 * 256;
    uint16_t* xm = reinterpret_cast<uint16_t*>(p.xm) + r0 * 256;
    // (r0 is a multiple of 16 and so is a row group: the tile's 16 rows stay consecutive under the remap)
    uint16_t* qt = reinterpret_cast<uint16_t*>(p.qkv) + remap_row(r0, p.q_group_rows, p.q_group_stride) * p.ldq;
    // Output rows leave in BURSTS, one array at a time (xhat | xm rows after the LayerNorm, x | bf16(x) rows after linear_out,
    // qkv in 512-byte pieces per row): the same stores issued a pair per step, arrays alternating, ran at 2.1 TB/s against
    // 5.1 TB/s in bursts (tools/chain_variants.sh, stores alone).  The late waves issue each burst four steps after the early ones.
    auto burst_ln = [&]() __attribute__((always_inline)) {
      if constexpr (MOD && SAVE) {
#pragma unroll
        for (int pp = 0; pp < (ST ? 0 : 4); ++pp) {
          if (pp % RG == 0) stage_reserve(stg_, RG);
          store_lines<false>(stg_, xh, Lb, 128 * pp, hqs[2 * pp], hqs[2 * pp + 1]);
        }
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
          if (pp % RG == 0) stage_reserve(stg_, RG);
          store_lines<false>(stg_, xm, Lb, 128 * pp, __builtin_bit_cast(uint4, a1[2 * pp]), __builtin_bit_cast(uint4, a1[2 * pp + 1]));
        }
      }
    };
    auto burst_x = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int pr = 0; pr < 8; ++pr) {
        if (pr % RG == 0) stage_reserve(stg_, RG);
        store_lines<false>(stg_, xt, Lf, 128 * pr, as_u4(acc[2 * pr]), as_u4(acc[2 * pr + 1]));
      }
      if constexpr (SAVE) {
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
          if (pp % RG == 0) stage_reserve(stg_, RG);
          store_lines<false>(stg_, xb, Lb, 128 * pp, __builtin_bit_cast(uint4, a0[2 * pp]), __builtin_bit_cast(uint4, a0[2 * pp + 1]));
        }
      }
    };
    // qkv blocks first .. first + 2 * npairs - 1 (block pq lives in qb[pq & 7])
    auto burst_q = [&](auto first_, auto npairs_) __attribute__((always_inline)) {
      constexpr int first = decltype(first_)::value, npairs = decltype(npairs_)::value;
      static_for<npairs>([&](auto k_) __attribute__((always_inline)) {
        constexpr int b = first + 2 * decltype(k_)::value;
        store_lines(stg_, qt, Lq, 64 * b, qb[b & 7], qb[(b + 1) & 7]);
      });
    };
#pragma unroll
    for (int j = 0; j < 8; ++j) a0[j] = a1[j];
    static_for<PER_TILE>([&](auto sc_) __attribute__((always_inline)) {
      constexpr int s = decltype(sc_)::value;
      CPROF_MARK(3);
      if constexpr (ST) step_begin(stg_); else if constexpr (s % PB == 0) CH_BARRIER();
      CPROF_MARK(0);
      HMA_LDS(char)* wb = ring + slot * SLOT;
      slot = slot + 1 == NS ? 0 : slot + 1;
      if constexpr (s < 8) {
        // ---- x1 = x + o Wproj^T + b
        nb_mma(wb, a0, acc[2 * s], acc[2 * s + 1]);
        step_end(stg_);
        add4(acc[2 * s], lds_f4(bias + 128 * s));
        add4(acc[2 * s + 1], lds_f4(bias + 128 * s + 16));
        if constexpr (s == 7 && !MOD) {
#pragma unroll
          for (int pr = 0; pr < 8; ++pr) a0[pr] = as_frag(pack_pair(acc[2 * pr], acc[2 * pr + 1]));
        }
        if constexpr (s == 7 && MOD) {
          // ---- LayerNorm (no affine) of the row the four lanes tok, tok + 16, tok + 32, tok + 48 hold, then the modulation
          float sum = 0.f;
#pragma unroll
          for (int t = 0; t < 16; ++t) sum += (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
          sum += __shfl_xor(sum, 16, 64);
          sum += __shfl_xor(sum, 32, 64);
          const float mean = sum * (1.0f / 256.0f);
          float sq = 0.f;
#pragma unroll
          for (int t = 0; t < 16; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float d = acc[t][r] - mean;
              sq = __builtin_fmaf(d, d, sq);
            }
          }
          sq += __shfl_xor(sq, 16, 64);
          sq += __shfl_xor(sq, 32, 64);
          const float rstd = rsqrtf(sq * (1.0f / 256.0f) + 1e-6f);
          const float nb = -mean * rstd;
          HMA_LDS(char)* ssl = lds + L_SS + ((tl & 1) * NCW + wave) * 2048 + 32 * g;
          static_assert(!ST || RB >= 4, "the four xhat blocks of a tile are staged without a wait in between");
          if constexpr (SAVE && ST) stage_reserve(stg_, 4);  // (in front of the loop: no control flow between its loads and their uses)
#pragma unroll
          for (int pr = 0; pr < 8; ++pr) {
            float h[8], mm[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = __builtin_fmaf(acc[2 * pr + (e >> 2)][e & 3], rstd, nb);
            const float4 sh0 = lds_f4(ssl + 128 * pr), sh1 = lds_f4(ssl + 128 * pr + 16);
            const float4 sc0 = lds_f4(ssl + 1024 + 128 * pr), sc1 = lds_f4(ssl + 1024 + 128 * pr + 16);
            const float sh[8] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w};
            const float sc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
            const uint4 hq = pack8(h);  // (saved for the backward; the modulation itself uses the fp32 value, like the reference)
#pragma unroll
            for (int e = 0; e < 8; ++e) mm[e] = __builtin_fmaf(h[e], 1.0f + sc[e], sh[e]);
            a1[pr] = as_frag(pack8(mm));
            if constexpr (SAVE && SPREAD) {  // (staged pair by pair: the packed xhat of the tile is not kept)
              if ((pr & 1) == 0) {
                hqs[0] = hq;
              } else {
                store_lines<false>(stg_, xh, Lb, 64 * (pr - 1), hqs[0], hq);
              }
            } else if (SAVE) {
              hqs[pr] = hq;
            }
          }
          if (SAVE && !(CH_ABL & 1)) p.rstd[r0 + tok] = rstd;  // (the row's four lanes write the same value)
        }
      } else if constexpr (MOD && s < 16) {
        // ---- x2 = x1 + xm Wlin^T + b: the new residual row; its bf16 copy is the qkv GEMM's operand
        constexpr int pr = s - 8;
        nb_mma(wb, a1, acc[2 * pr], acc[2 * pr + 1]);
        step_end(stg_);
        add4(acc[2 * pr], lds_f4(bias + 1024 + 128 * pr));
        add4(acc[2 * pr + 1], lds_f4(bias + 1024 + 128 * pr + 16));
        a0[pr] = as_frag(pack_pair(acc[2 * pr], acc[2 * pr + 1]));
      } else {
        // ---- qkv = bf16(x2) Wqkv^T + b
        constexpr int pq = s - S3;
        f32x4v_t c0 = lds_f4v(bias + 2048 + 128 * pq), c1 = lds_f4v(bias + 2048 + 128 * pq + 16);
        nb_mma(wb, a0, c0, c1);
        step_end(stg_);
        qb[pq & 7] = pack_pair(c0, c1);
      }
      CPROF_MARK(2);
      using I4 = std::integral_constant<int, 4>;
      if constexpr (SPREAD) {
        // Storer mode: a wave's blocks leave EVENLY over the tile's steps (at most two per step), so that the staging ring --
        // four blocks -- never has to take a burst: xhat pairs inside the LayerNorm (step 7), xm over the linear_out steps, the
        // fp32 rows over the first eight qkv steps (the next tile's rows are requested into those registers right behind),
        // their bf16 copy over the rest, a qkv block as soon as its two 64-byte halves exist.
        if constexpr (MOD && SAVE && s >= 8 && s < 16 && (s & 1) == 0)
          store_lines(stg_, xm, Lb, 64 * (s - 8), __builtin_bit_cast(uint4, a1[s - 8]), __builtin_bit_cast(uint4, a1[s - 7]));
        if constexpr (s >= S3 && s < S3 + 8) store_lines(stg_, xt, Lf, 128 * (s - S3), as_u4(acc[2 * (s - S3)]), as_u4(acc[2 * (s - S3) + 1]));
        if constexpr (SAVE && s >= S3 + 8 && ((s - S3) & 3) == 0)
          store_lines(stg_, xb, Lb, 32 * (s - S3 - 8), __builtin_bit_cast(uint4, a0[(s - S3 - 8) / 2]), __builtin_bit_cast(uint4, a0[(s - S3 - 8) / 2 + 1]));
        if constexpr (s >= S3 && ((s - S3) & 1) == 1) burst_q(std::integral_constant<int, s - S3 - 1>{}, std::integral_constant<int, 1>{});
      } else {
        if constexpr (MOD && s == 7) burst_ln();
        if constexpr (s == S3 - 1) burst_x();
        if constexpr (s >= S3 && ((s - S3) & 7) == 7) burst_q(std::integral_constant<int, s - S3 - 7>{}, I4{});
      }
      CPROF_MARK(1);
      // the next tile's rows (their registers are free from the first qkv step on: x and xm have been stored), a pair of loads per step
      if constexpr (CH_PF_SPREAD && !SPREAD) {
        if constexpr (s >= S3 + CH_PF_FIRST && s < S3 + CH_PF_FIRST + 12) prefetch(tl + 1 < nt ? tl + 1 : tl, std::integral_constant<int, s - S3 - CH_PF_FIRST>{});
      } else if constexpr (s == S3 + 8) {
        prefetch(tl + 1 < nt ? tl + 1 : tl, std::integral_constant<int, -1>{});
      }
    });
    CPROF_MARK(3);
    CH_TOUCH_A(a1);
    CH_TOUCH_ACC(acc);
    CPROF_MARK(4);
  };

#pragma unroll 1
  for (int tl = 0; tl < nt; ++tl) {
    const int64_t r0 = row0_of(tl);
    if (r0 >= tmap.end) {  // (whole wave past the matrix: possible in a workgroup's last tile only)
      if constexpr (ST) {
        steps_skip(stg_, PER_TILE);
      } else {
#pragma unroll 1
        for (int s = 0; s < PER_TILE; s += PB) CH_BARRIER();
      }
      slot = (slot + PER_TILE) % NS;  // (the ring position moves on with the loader whether or not this wave reads the bundles)
      continue;
    }
    tile(tl, r0);
  }
  stage_finish(stg_);
  CPROF_FLUSH(0, wave);
}

// ------------------------------------------------------------------------------------------------ chain A, backward
// Sum over the 16 token lanes of a lane group of 32 per-lane values, transposed on the way: step b exchanges halves with lane
// tok ^ (1 << b) and keeps the half its bit b selects, so after four steps lane tok holds the 16-lane sums of the TWO values
// c = q + 2 b3 + 4 b2 + 8 b1 + 16 b0 (b_i = bit i of tok): 30 exchanges instead of 128, and every lane ends with its own columns.
template <typename F>
__device__ __forceinline__ void colsum16(F&& val, int tok, float (&out)[2]) {
  float w[16];
  const bool b0 = tok & 1, b1 = tok & 2, b2 = tok & 4, b3 = tok & 8;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const float lo = val(c), hi = val(c + 16);
    w[c] = (b0 ? hi : lo) + __shfl_xor(b0 ? lo : hi, 1, 64);
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) w[c] = (b1 ? w[c + 8] : w[c]) + __shfl_xor(b1 ? w[c] : w[c + 8], 2, 64);
#pragma unroll
  for (int c = 0; c < 4; ++c) w[c] = (b2 ? w[c + 4] : w[c]) + __shfl_xor(b2 ? w[c] : w[c + 4], 4, 64);
#pragma unroll
  for (int c = 0; c < 2; ++c) out[c] = (b3 ? w[c + 2] : w[c]) + __shfl_xor(b3 ? w[c] : w[c + 2], 8, 64);
}

template <bool MOD>
__global__ __launch_bounds__(CH_THREADS, 2) void chain_a_bwd_kernel(hma_chain_a_bwd_t p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  HMA_LDS(char)* lds = (HMA_LDS(char)*)smem;
  const uint32_t lds_b = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const tile_map tmap = make_tile_map(p.M, NCW);
  const int nt = tmap.nt;
  constexpr int S3 = MOD ? 32 : 24;      // first d_o step
  constexpr int PER_TILE = S3 + 8;
  static_assert(PER_TILE % PB == 0, "whole barrier groups per tile (the loader refills PB slots per barrier)");
  start_stagger();
  if constexpr (ST) {
    sync_init(lds, tid);
    __syncthreads();
    if (wave > NCW) {
      storer_run<NCW>(lds, lane, wave - NCW - 1);
      return;
    }
  }
  if (wave == NCW) {
    const ring_src ws = {reinterpret_cast<const char*>(p.w.seg[0]), reinterpret_cast<const char*>(p.w.seg[1]),
                         reinterpret_cast<const char*>(p.w.seg[2]), reinterpret_cast<const char*>(p.w.seg[3]),
                         p.w.bundles[0], p.w.bundles[1], p.w.bundles[2], p.w.bundles[3]};
    loader_run<1>(ws, PER_TILE, nt, lds_b, lane, MOD ? p.ss : nullptr, p.M, p.rows_per_frame, lds, &tmap);
    return;
  }
  stage_t stg_ = make_stage(lds, wave, lane);
  const int tok = lane & 15, g = lane >> 4;
  auto row0_of = [&](int tl) __attribute__((always_inline)) {
    return tile_row0(tmap, tl, NCW, wave);
  };
  bf16x8_t dq[3][8], xr[8], a1[8], a2[8];
  f32x4v_t acc[16], dxr[16], dm[16];
  uint4 qb[8];  // d_o blocks waiting for their burst
  float rs = 1.f;
  auto load_chunk = [&](int64_t mc, int c, bf16x8_t (&d)[8]) __attribute__((always_inline)) {
    if (CH_ABL & 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) d[j] = as_frag(make_uint4(lane, j, c, j));
      return;
    }
    const uint16_t* row = reinterpret_cast<const uint16_t*>(p.dqkv) + mc * p.ldq + 256 * c + 8 * g;
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = as_frag(*reinterpret_cast<const uint4*>(row + 32 * j));
  };
  // loads j0 .. j0 + n - 1 of a k-chunk (CH_PF_SPREAD: a row's loads are issued one or two per step in front of their use)
  auto load_part = [&](int64_t mc, int c, bf16x8_t (&d)[8], auto j0_, auto n_) __attribute__((always_inline)) {
    constexpr int j0 = decltype(j0_)::value, n = decltype(n_)::value;
    if (CH_ABL & 2) return;
    const uint16_t* row = reinterpret_cast<const uint16_t*>(p.dqkv) + mc * p.ldq + 256 * c + 8 * g;
#pragma unroll
    for (int j = j0; j < j0 + n; ++j) d[j] = as_frag(*reinterpret_cast<const uint4*>(row + 32 * j));
  };
  auto next_row = [&](int tl) __attribute__((always_inline)) {
    int64_t m = row0_of(tl + 1 < nt ? tl + 1 : tl) + tok;
    return m < p.M ? m : p.M - 1;
  };
  auto prefetch = [&](int tl) __attribute__((always_inline)) {  // the first k-chunk of the next tile's dqkv rows
    int64_t m = row0_of(tl) + tok;
    m = m < p.M ? m : p.M - 1;
    load_chunk(m, 0, dq[0]);
  };
  if ((CH_PF_SPREAD != 0) & ((CH_ABL & 2) != 0)) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) dq[c][j] = as_frag(make_uint4(lane, j, c, j));
#pragma unroll
    for (int j = 0; j < 8; ++j) xr[j] = as_frag(make_uint4(lane, j, lane, j));
#pragma unroll
    for (int t = 0; t < 16; ++t) dxr[t] = f32x4v_t{0.f, 1.f, 2.f, 3.f};
  }
  prefetch(0);
  CH_TOUCH_A(dq[0]);
  int slot = 0;
  HMA_LDS(char)* ring = lds + lane * 16;
  const line_offs Lb = make_lines(512, tok, 16 * g, 64);
  const line_offs Lf = make_lines(1024, tok, 32 * g, 16);

  auto tile = [&](int tl, int64_t r0) __attribute__((always_inline)) {
    const int64_t m = r0 + tok;
    float* xt = p.dx + r0 * 256;
    uint16_t* d1 = reinterpret_cast<uint16_t*>(p.dx1_bf16) + r0 * 256;
    uint16_t* d2 = reinterpret_cast<uint16_t*>(p.dx2_bf16) + r0 * 256;
    uint16_t* ot = reinterpret_cast<uint16_t*>(p.d_o) + r0 * 256;
    auto burst_d2 = [&]() __attribute__((always_inline)) {  // bf16(dx2), the dY of linear_out's weight gradient
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        if (pp % RG == 0) stage_reserve(stg_, RG);
        store_lines<false>(stg_, d2, Lb, 128 * pp, __builtin_bit_cast(uint4, a1[2 * pp]), __builtin_bit_cast(uint4, a1[2 * pp + 1]));
      }
    };
    auto burst_dx = [&]() __attribute__((always_inline)) {  // dx1, the new residual gradient, and its bf16 copy
#pragma unroll
      for (int pr = 0; pr < 8; ++pr) {
        if (pr % RG == 0) stage_reserve(stg_, RG);
        store_lines<false>(stg_, xt, Lf, 128 * pr, as_u4(acc[2 * pr]), as_u4(acc[2 * pr + 1]));
      }
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        if (pp % RG == 0) stage_reserve(stg_, RG);
        store_lines<false>(stg_, d1, Lb, 128 * pp, __builtin_bit_cast(uint4, a2[2 * pp]), __builtin_bit_cast(uint4, a2[2 * pp + 1]));
      }
    };
    auto burst_o = [&](auto first_) __attribute__((always_inline)) {  // d_o blocks first .. first + 3
      constexpr int first = decltype(first_)::value;
      store_lines(stg_, ot, Lb, 64 * first, qb[first], qb[first + 1]);
      store_lines(stg_, ot, Lb, 64 * first + 128, qb[first + 2], qb[first + 3]);
    };
    if constexpr (!CH_PF_SPREAD) load_chunk(m, 1, dq[1]);  // this tile's second k-chunk of dqkv
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = f32x4v_t{0.f, 0.f, 0.f, 0.f};
    static_for<PER_TILE>([&](auto sc_) __attribute__((always_inline)) {
      constexpr int s = decltype(sc_)::value;
      if constexpr (ST) step_begin(stg_); else if constexpr (s % PB == 0) CH_BARRIER();
      HMA_LDS(char)* wb = ring + slot * SLOT;
      slot = slot + 1 == NS ? 0 : slot + 1;
      if constexpr (s < 24) {
        // ---- dx2 = dx + dqkv Wqkv (k = 768 in three chunks; chunk c + 1 is requested while chunk c is multiplied)
        constexpr int c = s >> 3, pr = s & 7;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        if constexpr (CH_PF_SPREAD) {
          // this tile's k-chunks 1 (used from step 8) and 2 (from 16), the residual gradient's row (step 23) and -- MOD -- the saved
          // xhat row (step 31): one or two loads per step
          if constexpr (s < 4) load_part(m, 1, dq[1], std::integral_constant<int, 2 * s>{}, I2{});
          if constexpr (s >= 4 && s < 12) load_part(m, 2, dq[2], std::integral_constant<int, s - 4>{}, I1{});
          if constexpr (s >= 8 && s < 16) {
            if (!(CH_ABL & 2)) {
              const float* xrow = p.dx + m * 256 + 8 * g;
              dxr[2 * (s - 8)] = ld4(xrow + 32 * (s - 8));
              dxr[2 * (s - 8) + 1] = ld4(xrow + 32 * (s - 8) + 4);
            }
          }
          if constexpr (MOD && s >= 16) {
            if (!(CH_ABL & 2)) {
              const uint16_t* hrow = reinterpret_cast<const uint16_t*>(p.xhat) + m * 256 + 8 * g;
              xr[s - 16] = as_frag(*reinterpret_cast<const uint4*>(hrow + 32 * (s - 16)));
              if constexpr (s == 16) rs = p.rstd[m];
            }
          }
        }
        if constexpr (s == 8 && !CH_PF_SPREAD) load_chunk(m, 2, dq[2]);
        if constexpr (s == 16 && !CH_PF_SPREAD) {  // the residual gradient itself, added behind the product (eight steps from here)
          if (!(CH_ABL & 2)) {
            const float* xrow = p.dx + m * 256 + 8 * g;
#pragma unroll
            for (int pr2 = 0; pr2 < 8; ++pr2) {
              dxr[2 * pr2] = ld4(xrow + 32 * pr2);
              dxr[2 * pr2 + 1] = ld4(xrow + 32 * pr2 + 4);
            }
          } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) dxr[t] = f32x4v_t{0.f, 1.f, 2.f, 3.f};
          }
        }
        nb_mma(wb, dq[c], acc[2 * pr], acc[2 * pr + 1]);
        step_end(stg_);
        if constexpr (s == 23) {
#pragma unroll
          for (int t = 0; t < 16; ++t) acc[t] += dxr[t];
#pragma unroll
          for (int pr2 = 0; pr2 < 8; ++pr2) {
            const bf16x8_t q = as_frag(pack_pair(acc[2 * pr2], acc[2 * pr2 + 1]));
            if constexpr (MOD) a1[pr2] = q;
            else a2[pr2] = q;
          }
        }
      } else if constexpr (MOD && s < 32) {
        // ---- dxm = bf16(dx2) Wlin (the saved xhat row and 1 / sigma are requested at its first step, eight steps before use)
        constexpr int pr = s - 24;
        if constexpr (s == 24) {
          if constexpr (!CH_PF_SPREAD) {
            if (!(CH_ABL & 2)) {
              const uint16_t* hrow = reinterpret_cast<const uint16_t*>(p.xhat) + m * 256 + 8 * g;
#pragma unroll
              for (int j = 0; j < 8; ++j) xr[j] = as_frag(*reinterpret_cast<const uint4*>(hrow + 32 * j));
              rs = p.rstd[m];
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) xr[j] = as_frag(make_uint4(lane, j, lane, j));
            }
          }
#pragma unroll
          for (int t = 0; t < 16; ++t) dm[t] = f32x4v_t{0.f, 0.f, 0.f, 0.f};
        }
        nb_mma(wb, a1, dm[2 * pr], dm[2 * pr + 1]);
        step_end(stg_);
        if constexpr (s == 31) {
          // ---- modulate + LayerNorm backward
          HMA_LDS(char)* scl = lds + L_SS + ((tl & 1) * NCW + wave) * 2048 + 1024 + 32 * g;
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int pr2 = 0; pr2 < 8; ++pr2) {
            float xh[8];
            unpack8(__builtin_bit_cast(uint4, xr[pr2]), xh);
            const float4 sc0 = lds_f4(scl + 128 * pr2), sc1 = lds_f4(scl + 128 * pr2 + 16);
            const float sc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float gq = dm[2 * pr2 + (e >> 2)][e & 3] * (1.0f + sc[e]);
              s1 += gq;
              s2 = __builtin_fmaf(gq, xh[e], s2);
            }
          }
          s1 += __shfl_xor(s1, 16, 64);
          s1 += __shfl_xor(s1, 32, 64);
          s2 += __shfl_xor(s2, 16, 64);
          s2 += __shfl_xor(s2, 32, 64);
          s1 *= (1.0f / 256.0f);
          s2 *= (1.0f / 256.0f);
          // (opaque: otherwise the unpacked xhat and the scaled gradient of this pass -- 128 registers -- are kept for the passes below)
          CH_TOUCH_A(xr);
          CH_TOUCH_ACC(dm);
          asm volatile("" ::: "memory");
          {  // d shift = column sums of dxm, d scale = column sums of dxm xhat over the frame's rows (two halves of 128 columns)
            float* dssf = p.dss + (r0 / p.rows_per_frame) * 512 + 8 * g;
            const int cb = ((tok >> 3) & 1) * 2 + ((tok >> 2) & 1) * 4 + ((tok >> 1) & 1) * 8 + (tok & 1) * 16;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              float o2[2];
              colsum16([&](int c) __attribute__((always_inline)) { return dm[8 * h + 2 * (c >> 3) + ((c >> 2) & 1)][c & 3]; }, tok, o2);
              if (!(CH_ABL & 1)) {
#pragma unroll
                for (int q = 0; q < 2; ++q) unsafeAtomicAdd(dssf + 128 * h + 32 * ((cb + q) >> 3) + ((cb + q) & 7), o2[q]);
              }
              colsum16([&](int c) __attribute__((always_inline)) {
                const uint32_t wv = __builtin_bit_cast(u32x4_t, xr[4 * h + (c >> 3)])[(c >> 1) & 3];
                return dm[8 * h + 2 * (c >> 3) + ((c >> 2) & 1)][c & 3] * ((c & 1) ? bf16_hi(wv) : bf16_lo(wv));
              }, tok, o2);
              if (!(CH_ABL & 1)) {
#pragma unroll
                for (int q = 0; q < 2; ++q) unsafeAtomicAdd(dssf + 256 + 128 * h + 32 * ((cb + q) >> 3) + ((cb + q) & 7), o2[q]);
              }
            }
          }
          CH_TOUCH_A(xr);
          CH_TOUCH_ACC(dm);
          asm volatile("" ::: "memory");
#pragma unroll
          for (int pr2 = 0; pr2 < 8; ++pr2) {
            float xh[8];
            unpack8(__builtin_bit_cast(uint4, xr[pr2]), xh);
            const float4 sc0 = lds_f4(scl + 128 * pr2), sc1 = lds_f4(scl + 128 * pr2 + 16);
            const float sc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float gq = dm[2 * pr2 + (e >> 2)][e & 3] * (1.0f + sc[e]);
              acc[2 * pr2 + (e >> 2)][e & 3] += rs * (gq - s1 - xh[e] * s2);
            }
            a2[pr2] = as_frag(pack_pair(acc[2 * pr2], acc[2 * pr2 + 1]));
          }
        }
      } else {
        // ---- d_o = bf16(dx1) Wproj
        constexpr int pr = s - S3;
        if constexpr (CH_PF_SPREAD) {  // the next tile's first k-chunk, a load per step
          load_part(next_row(tl), 0, dq[0], std::integral_constant<int, pr>{}, std::integral_constant<int, 1>{});
        } else if constexpr (s == S3) {
          prefetch(tl + 1 < nt ? tl + 1 : tl);
        }
        f32x4v_t c0 = f32x4v_t{0.f, 0.f, 0.f, 0.f}, c1 = f32x4v_t{0.f, 0.f, 0.f, 0.f};
        nb_mma(wb, a2, c0, c1);
        step_end(stg_);
        qb[pr] = pack_pair(c0, c1);
      }
      if constexpr (MOD && s == 23) burst_d2();
      if constexpr (s == S3 - 1) burst_dx();
      if constexpr (ST) {
        if constexpr (s >= S3 && ((s - S3) & 1) == 1) store_lines(stg_, ot, Lb, 64 * (s - S3 - 1), qb[s - S3 - 1], qb[s - S3]);
      } else if constexpr (s == S3 + 7) {
        burst_o(std::integral_constant<int, 0>{});
        burst_o(std::integral_constant<int, 4>{});
      }
    });
    CH_TOUCH_A(dq[0]);
  };

#pragma unroll 1
  for (int tl = 0; tl < nt; ++tl) {
    const int64_t r0 = row0_of(tl);
    if (r0 >= tmap.end) {
      if constexpr (ST) {
        steps_skip(stg_, PER_TILE);
      } else {
#pragma unroll 1
        for (int s = 0; s < PER_TILE; s += PB) CH_BARRIER();
      }
      slot = (slot + PER_TILE) % NS;  // (the ring position moves on with the loader whether or not this wave reads the bundles)
      continue;
    }
    tile(tl, r0);
  }
  stage_finish(stg_);
}

// ------------------------------------------------------------------------------------------------ chain S, backward
// The spatial side of a block's backward between the attention backward and the previous block (st_transformer.py:85-86 norm1 + qkv of
// attention.py:39, autograd mirror): g = dqkv Wqkv diag(gamma) (gamma folded into the packed weight's output rows), the LayerNorm
// backward of the row and the residual add, dx <- dx + rstd (g - mean(g) - xhat mean(g xhat)), and bf16(dx) for the next consumer --
// what hma_gemm_nt (dqkv -> t256) + hma_ln_bwd did in two launches with a bf16 round trip of g through HBM.  24 steps per tile (the
// three k-chunks of dqkv); a tile's rows arrive a load or two per step (CH_PF_SPREAD's schedule): k-chunks 1 and 2 over steps 0..11,
// the saved xhat row over 8..15, the fp32 dx row over 12..19, the next tile's first k-chunk behind the last step's MFMAs.  norm1's dgamma / dbeta come
// out of the qkv weight-gradient reduction (hma_gemm_tn: w_master / dgamma / dbeta), as norm2's do.
__global__ __launch_bounds__(CH_THREADS, 2) void chain_s_bwd_kernel(hma_chain_s_bwd_t p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  HMA_LDS(char)* lds = (HMA_LDS(char)*)smem;
  const uint32_t lds_b = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const tile_map tmap = make_tile_map(p.M, NCW);
  const int nt = tmap.nt;
  constexpr int PER_TILE = 24;
  static_assert(PER_TILE % PB == 0, "whole barrier groups per tile");
#ifndef CH_S_DX0
#define CH_S_DX0 16
#endif
  constexpr int DX0 = CH_S_DX0;  // the step the fp32 dx row's loads start at (two per step)
  start_stagger();
  if constexpr (ST) {
    sync_init(lds, tid);
    __syncthreads();
    if (wave > NCW) {
      storer_run<NCW>(lds, lane, wave - NCW - 1);
      return;
    }
  }
  if (wave == NCW) {
    const ring_src ws = {reinterpret_cast<const char*>(p.w.seg[0]), reinterpret_cast<const char*>(p.w.seg[1]),
                         reinterpret_cast<const char*>(p.w.seg[2]), reinterpret_cast<const char*>(p.w.seg[3]),
                         p.w.bundles[0], p.w.bundles[1], p.w.bundles[2], p.w.bundles[3]};
    loader_run<1>(ws, PER_TILE, nt, lds_b, lane, nullptr, p.M, 1, lds, &tmap);
    return;
  }
  stage_t stg_ = make_stage(lds, wave, lane);
  const int tok = lane & 15, g = lane >> 4;
  auto row0_of = [&](int tl) __attribute__((always_inline)) {
    return tile_row0(tmap, tl, NCW, wave);
  };
  bf16x8_t dq[3][8], xr[8];
  f32x4v_t acc[16], dxr[16];
  float rs = 1.f;
  // fragment j of k-chunk c = columns 256 c + 32 j + 8 g .. + 7 of a row: head j of part c.  Head-blocked dqkv (hb_rows = rows per
  // frame n > 0, HMA_A_BF16_HEADBLK): element at (((frame 8 + j) 3 + c) n + row in frame) 32 + 8 g -- a load instruction then reads
  // 16 rows x 64 bytes = 1 KB of contiguous memory instead of sixteen 64-byte pieces 1536 bytes apart.  `qrow` = this lane's row
  // pointer (row_ptr: ONE integer division per tile, by the wave-uniform first row), cs / js = element strides of a k-chunk / a head.
  const int hbn = (int)p.hb_rows;
  const int64_t cs = hbn > 0 ? (int64_t)hbn * 32 : 256, js = hbn > 0 ? (int64_t)hbn * 96 : 32;
  auto row_ptr = [&](int64_t r0w) __attribute__((always_inline)) {  // r0w: the wave's first row (16 rows never straddle a frame)
    const uint16_t* b = reinterpret_cast<const uint16_t*>(p.dqkv);
    if (hbn > 0) {
      const uint32_t fr = (uint32_t)r0w / (uint32_t)hbn, rr = (uint32_t)r0w - fr * (uint32_t)hbn;
      return b + (((int64_t)fr * 24 * hbn + rr + tok) << 5) + 8 * g;
    }
    return b + (r0w + tok) * p.ldq + 8 * g;
  };
  auto load_part = [&](const uint16_t* qrow, int c, bf16x8_t (&d)[8], auto j0_, auto n_) __attribute__((always_inline)) {
    constexpr int j0 = decltype(j0_)::value, n = decltype(n_)::value;
    if (CH_ABL & 2) return;
#pragma unroll
    for (int j = j0; j < j0 + n; ++j) d[j] = as_frag(*reinterpret_cast<const uint4*>(qrow + c * cs + j * js));
  };
  auto next_r0 = [&](int tl) __attribute__((always_inline)) {
    const int64_t r = row0_of(tl + 1 < nt ? tl + 1 : tl);
    return r < p.M ? r : p.M - 16;
  };
  if (CH_ABL & 2) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) dq[c][j] = as_frag(make_uint4(lane, j, c, j));
#pragma unroll
    for (int j = 0; j < 8; ++j) xr[j] = as_frag(make_uint4(lane, j, lane, j));
#pragma unroll
    for (int t = 0; t < 16; ++t) dxr[t] = f32x4v_t{0.f, 1.f, 2.f, 3.f};
  }
  {
    const int64_t r = row0_of(0);
    load_part(row_ptr(r < p.M ? r : p.M - 16), 0, dq[0], std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
  }
  CH_TOUCH_A(dq[0]);
  int slot = 0;
  HMA_LDS(char)* ring = lds + lane * 16;
  const line_offs Lb = make_lines(512, tok, 16 * g, 64);
  const line_offs Lf = make_lines(1024, tok, 32 * g, 16);

  auto tile = [&](int tl, int64_t r0) __attribute__((always_inline)) {
    const int64_t m = r0 + tok;
    const uint16_t* qrow = row_ptr(r0);
    float* xt = p.dx + r0 * 256;
    uint16_t* d1 = reinterpret_cast<uint16_t*>(p.dx_bf16) + r0 * 256;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = f32x4v_t{0.f, 0.f, 0.f, 0.f};
    static_for<PER_TILE>([&](auto sc_) __attribute__((always_inline)) {
      constexpr int s = decltype(sc_)::value;
      if constexpr (ST) step_begin(stg_); else if constexpr (s % PB == 0) CH_BARRIER();
      HMA_LDS(char)* wb = ring + slot * SLOT;
      slot = slot + 1 == NS ? 0 : slot + 1;
      constexpr int c = s >> 3, pr = s & 7;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      if constexpr (s < 4) load_part(qrow, 1, dq[1], std::integral_constant<int, 2 * s>{}, I2{});
      if constexpr (s >= 4 && s < 12) load_part(qrow, 2, dq[2], std::integral_constant<int, s - 4>{}, I1{});
      if constexpr (s >= 8 && s < 16) {
        if (!(CH_ABL & 2)) {
          const uint16_t* hrow = reinterpret_cast<const uint16_t*>(p.xhat) + m * 256 + 8 * g;
          xr[s - 8] = as_frag(*reinterpret_cast<const uint4*>(hrow + 32 * (s - 8)));
          if constexpr (s == 8) rs = p.rstd[m];
        }
      }
      if constexpr (s >= DX0 && s < DX0 + 8) {
        if (!(CH_ABL & 2)) {
          const float* xrow = p.dx + m * 256 + 8 * g;
          dxr[2 * (s - DX0)] = ld4(xrow + 32 * (s - DX0));
          dxr[2 * (s - DX0) + 1] = ld4(xrow + 32 * (s - DX0) + 4);
        }
      }
      nb_mma(wb, dq[c], acc[2 * pr], acc[2 * pr + 1]);
      step_end(stg_);
      if constexpr (s == 23) {
        // the next tile's first k-chunk: requested here, in front of this tile's 24 store instructions (the queue is in order), into the
        // registers the last k-chunk has just left -- eight steps earlier they would not fit beside dx, xhat and the accumulators
        load_part(row_ptr(next_r0(tl)), 0, dq[0], std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
        // ---- LayerNorm backward of the row (its four lanes tok, tok + 16, tok + 32, tok + 48 hold it) + residual
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int pr2 = 0; pr2 < 8; ++pr2) {
          float xh[8];
          unpack8(__builtin_bit_cast(uint4, xr[pr2]), xh);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float gq = acc[2 * pr2 + (e >> 2)][e & 3];
            s1 += gq;
            s2 = __builtin_fmaf(gq, xh[e], s2);
          }
        }
        s1 += __shfl_xor(s1, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 16, 64);
        s2 += __shfl_xor(s2, 32, 64);
        s1 *= (1.0f / 256.0f);
        s2 *= (1.0f / 256.0f);
        CH_TOUCH_A(xr);  // (opaque: otherwise the unpacked xhat of the pass above -- 64 registers -- is kept for the pass below)
#pragma unroll
        for (int pr2 = 0; pr2 < 8; ++pr2) {
          float xh[8];
          unpack8(__builtin_bit_cast(uint4, xr[pr2]), xh);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float gq = acc[2 * pr2 + (e >> 2)][e & 3];
            dxr[2 * pr2 + (e >> 2)][e & 3] += rs * (gq - s1 - xh[e] * s2);
          }
        }
#pragma unroll
        for (int pr2 = 0; pr2 < 8; ++pr2) {
          if (pr2 % RG == 0) stage_reserve(stg_, RG);
          store_lines<false>(stg_, xt, Lf, 128 * pr2, as_u4(dxr[2 * pr2]), as_u4(dxr[2 * pr2 + 1]));
        }
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
          if (pp % RG == 0) stage_reserve(stg_, RG);
          store_lines<false>(stg_, d1, Lb, 128 * pp, pack_pair(dxr[4 * pp], dxr[4 * pp + 1]), pack_pair(dxr[4 * pp + 2], dxr[4 * pp + 3]));
        }
      }
    });
    CH_TOUCH_A(dq[0]);
  };

#pragma unroll 1
  for (int tl = 0; tl < nt; ++tl) {
    const int64_t r0 = row0_of(tl);
    if (r0 >= tmap.end) {
      if constexpr (ST) {
        steps_skip(stg_, PER_TILE);
      } else {
#pragma unroll 1
        for (int s = 0; s < PER_TILE; s += PB) CH_BARRIER();
      }
      slot = (slot + PER_TILE) % NS;
      continue;
    }
    tile(tl, r0);
  }
  stage_finish(stg_);
}

// ------------------------------------------------------------------------------------------------ chain B, forward (inference)
// one K-slice bundle against the wave's rows: acc[t] += W[16 columns of tile t][32 k] . h   (16 MFMAs, 16 accumulators)
__device__ __forceinline__ void ks_mma(HMA_LDS(char)* wb, const bf16x8_t& h, f32x4v_t (&acc)[16]) {
  bf16x8_t f[4], fn[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) f[q] = lds_frag(wb + q * 1024);
#pragma unroll
  for (int hq = 0; hq < 4; ++hq) {
    if (hq < 3) {
#pragma unroll
      for (int q = 0; q < 4; ++q) fn[q] = lds_frag(wb + (4 * hq + 4 + q) * 1024);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[4 * hq + q] = mfma16(f[q], h, acc[4 * hq + q]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) f[q] = fn[q];
  }
}

// proj_t + residual -> LayerNorm (norm2, affine folded into fc1) -> fc1 -> GELU -> fc2 + residual -> [LayerNorm (the next block's norm1,
// folded into its qkv) -> qkv]: the second row-local chain of a block (st_transformer.py:111-112 and :85-86 of the next block), for
// passes that save nothing (inference / decode).  Steps per tile: 8 (proj) + 2 x 32 (a hidden block of 32 units: its fc1 rows,
// then its fc2 columns) + 24 (qkv).  The hidden activation exists as ONE B-operand fragment per step.
template <bool QKV, int NW = NCW, bool SAVE = false, bool DROP = false>
__global__ __launch_bounds__(CH_THREADS, 2) void chain_b_fwd_kernel(hma_chain_b_fwd_t p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  HMA_LDS(char)* lds = (HMA_LDS(char)*)smem;
  const uint32_t lds_b = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const tile_map tmap = make_tile_map(p.M, NW);
  const int nt = tmap.nt;
  {
    HMA_LDS(float)* bl = (HMA_LDS(float)*)(lds + L_BIAS);  // proj 0..255 | fc2 256..511 | fc1 512..1535 | qkv 1536..2303
    for (int i = tid; i < 2304; i += CH_THREADS) {
      float v = 0.f;
      if (i < 256) v = p.b_proj ? p.b_proj[i] : 0.f;
      else if (i < 512) v = p.b2 ? p.b2[i - 256] : 0.f;
      else if (i < 1536) v = p.b1[i - 512];
      else v = (QKV && p.b_qkv) ? p.b_qkv[i - 1536] : 0.f;
      bl[i] = v;
    }
  }
  sync_init(lds, tid);
  start_stagger();
  __syncthreads();
  constexpr int SM = 8, SQ = 8 + 64;           // first MLP step, first qkv step
  constexpr int PER_TILE = SQ + (QKV ? 24 : 0);
  static_assert(PER_TILE % PB == 0, "whole barrier groups per tile (the loader refills PB slots per barrier)");
  if (ST && wave > NCW) {
    storer_run<NW>(lds, lane, wave - NCW - 1);
    return;
  }
  if (wave >= NW && wave != NCW) return;
  if (wave == NCW) {
    const ring_src ws = {reinterpret_cast<const char*>(p.w.seg[0]), reinterpret_cast<const char*>(p.w.seg[1]),
                         reinterpret_cast<const char*>(p.w.seg[2]), reinterpret_cast<const char*>(p.w.seg[3]),
                         p.w.bundles[0], p.w.bundles[1], p.w.bundles[2], p.w.bundles[3]};
    loader_run<0, NW>(ws, PER_TILE, nt, lds_b, lane, nullptr, p.M, 1, lds, &tmap);
    return;
  }
  stage_t stg_ = make_stage(lds, wave, lane);
  const int tok = lane & 15, g = lane >> 4;
  auto row0_of = [&](int tl) __attribute__((always_inline)) {
    return tile_row0(tmap, tl, NW, wave);
  };
  bf16x8_t a0[8], a1[8];
  f32x4v_t acc[16];
  uint4 qb[8];
  auto prefetch = [&](int tl, auto part_) __attribute__((always_inline)) {  // (part < 0: all 24 loads; 0..11: a pair)
    constexpr int part = decltype(part_)::value;
    if (CH_ABL & 2) return;
    // (a wave's 16 rows are inside the matrix or past it together -- M % 16 == 0 -- so the clamp is wave-uniform; uniform base + an opaque
    // 32-bit lane offset: `array + lane part` as a 64-bit pointer per lane is a loop invariant the compiler keeps, and in the dropout
    // form spilled -- a scratch reload drains the store queue)
    int64_t m0 = row0_of(tl);
    m0 = m0 < p.M ? m0 : p.M - 16;
    uint32_t lo = (uint32_t)tok * 256u + 8u * (uint32_t)g;
    asm volatile("" : "+v"(lo));
    const uint16_t* orow = reinterpret_cast<const uint16_t*>(p.o) + m0 * 256 + lo;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (part < 0 || j / 2 == part) a1[j] = as_frag(*reinterpret_cast<const uint4*>(orow + 32 * j));
    const float* xrow = p.x + m0 * 256 + lo;
#pragma unroll
    for (int pr = 0; pr < 8; ++pr) {
      if (part < 0 || pr + 4 == part) {
        acc[2 * pr] = ld4(xrow + 32 * pr);
        acc[2 * pr + 1] = ld4(xrow + 32 * pr + 4);
      }
    }
  };
  if (CH_ABL & 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) a1[j] = as_frag(make_uint4(lane, j, lane, j));
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = f32x4v_t{0.f, 1.f, 2.f, 3.f};
  }
  prefetch(0, std::integral_constant<int, -1>{});
  CH_TOUCH_A(a1);
  CH_TOUCH_ACC(acc);
  HMA_LDS(char)* ring = lds + lane * 16;
  HMA_LDS(char)* bias = lds + L_BIAS + 32 * g;
  const line_offs Lf = make_lines(1024, tok, 32 * g, 16);
  const line_offs Lq = make_lines((int)p.ldq * 2, tok, 16 * g, 64);
  int slot = 0;
  const line_offs Lb = make_lines(512, tok, 16 * g, 64);
  // mlp_drop > 0 (training): the two nn.Dropout masks of Mlp.forward, counter-based like the GEMM epilogues' (hma_common.h drop_keep)
  uint32_t dseed = 0, dth = 0;
  float dsc = 1.f;
  f32x4v_t x1[DROP ? 16 : 1];  // (DROP: the residual is kept apart from the branch output, which is masked before the add)
  if constexpr (DROP) {
    dseed = *p.drop_seed;
    dth = drop_thresh(p.drop_p);
    dsc = drop_scale(p.drop_p);
  }
  // LayerNorm (no affine) of the rows in acc -> packed bf16 B operand (training: also saved, with 1 / sigma, for the backward)
  auto ln_pack = [&](bf16x8_t (&dst)[8], void* xhat_out, float* rstd_out, int64_t r0, bool defer = false) __attribute__((always_inline)) {
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) sum += (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / 256.0f);
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = acc[t][r] - mean;
        sq = __builtin_fmaf(d, d, sq);
      }
    }
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    const float rstd = rsqrtf(sq * (1.0f / 256.0f) + p.ln_eps);
    const float nb = -mean * rstd;
#pragma unroll
    for (int pr = 0; pr < 8; ++pr) {
      float h[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = __builtin_fmaf(acc[2 * pr + (e >> 2)][e & 3], rstd, nb);
      dst[pr] = as_frag(pack8(h));
    }
    if constexpr (SAVE) {
      uint16_t* xo = reinterpret_cast<uint16_t*>(xhat_out) + r0 * 256;
      if (!defer) {
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
          if (pp % RG == 0) stage_reserve(stg_, RG);
          store_lines<false>(stg_, xo, Lb, 128 * pp, __builtin_bit_cast(uint4, dst[2 * pp]), __builtin_bit_cast(uint4, dst[2 * pp + 1]));
        }
      }
      if (!(CH_ABL & 1)) {
        uint32_t to = (uint32_t)tok;
        asm volatile("" : "+v"(to));
        (rstd_out + r0)[to] = rstd;
      }
    }
  };
  constexpr bool BSP = CH_BSPREAD && !ST && QKV;
#pragma unroll 1
  for (int tl = 0; tl < nt; ++tl) {
    const int64_t r0 = row0_of(tl);
    if (r0 >= tmap.end) {
      if constexpr (ST) {
        steps_skip(stg_, PER_TILE);
      } else {
#pragma unroll 1
        for (int s = 0; s < PER_TILE; s += PB) CH_BARRIER();
      }
      slot = (slot + PER_TILE) % NS;  // (the ring position moves on with the loader whether or not this wave reads the bundles)
      continue;
    }
    float* xt = p.x + r0 * 256;
    uint16_t* qt = QKV ? reinterpret_cast<uint16_t*>(p.qkv) + r0 * p.ldq : nullptr;
#pragma unroll
    for (int j = 0; j < 8; ++j) a0[j] = a1[j];
    bf16x8_t hf;
    static_for<PER_TILE>([&](auto sc_) __attribute__((always_inline)) {
      constexpr int s = decltype(sc_)::value;
      if constexpr (ST) step_begin(stg_); else if constexpr (s % PB == 0) CH_BARRIER();
      HMA_LDS(char)* wb;
      if constexpr (PER_TILE % NS == 0) {  // (every tile starts at slot 0: the slot of a step is a constant)
        wb = ring + (s % NS) * SLOT;
      } else {
        wb = ring + slot * SLOT;
        slot = slot + 1 == NS ? 0 : slot + 1;
      }
      if constexpr (s < SM) {
        // ---- x1 = x + o Wproj^T + b
        nb_mma(wb, a0, acc[2 * s], acc[2 * s + 1]);
        step_end(stg_);
        add4(acc[2 * s], lds_f4(bias + 128 * s));
        add4(acc[2 * s + 1], lds_f4(bias + 128 * s + 16));
        if constexpr (s == SM - 1) {
          ln_pack(a1, p.xhat2, p.rstd2, r0);  // xhat2 (norm2's affine sits in the packed fc1 weights / bias)
          if constexpr (DROP) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
              x1[t] = acc[t];
              acc[t] = f32x4v_t{0.f, 0.f, 0.f, 0.f};
            }
          }
#pragma unroll
          for (int pr = 0; pr < 8; ++pr) {  // + fc2 bias, once
            add4(acc[2 * pr], lds_f4(bias + 1024 + 128 * pr));
            add4(acc[2 * pr + 1], lds_f4(bias + 1024 + 128 * pr + 16));
          }
        }
      } else if constexpr (s < SQ) {
        constexpr int h = (s - SM) >> 1;
        if constexpr (((s - SM) & 1) == 0) {
          // ---- u = W1f[hidden block h] xhat2 + b1f; gelu; the lane's 8 hidden units are the next step's B operand
          f32x4v_t c0 = lds_f4v(bias + 2048 + 128 * h), c1 = lds_f4v(bias + 2048 + 128 * h + 16);
          nb_mma(wb, a1, c0, c1);
          step_end(stg_);
          // (the staged one-transcendental GELU of the fused MLP kernels, hma_common.h: ~10 VALU operations per value against the
          // ~19 of the rcp + exp form -- an fc1 step is VALU-bound: 16 MFMAs against 8 GELUs per lane)
          float uv[8], hv[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            uv[e] = c0[e];
            uv[4 + e] = c1[e];
          }
#ifdef CH_OLD_GELU  // (measurement builds: the rcp + exp form)
#pragma unroll
          for (int e = 0; e < 8; ++e) hv[e] = gelu_f(uv[e]);
#else
          gelu_n<8>(uv, hv);
#endif
          if constexpr (DROP) {
            const int64_t e0 = (r0 + tok) * 1024 + 32 * h + 8 * g;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              bool k0, k1;
              drop_keep2(dseed, p.drop_salt, e0 + e, dth, k0, k1);
              hv[e] = k0 ? hv[e] * dsc : 0.f;
              hv[e + 1] = k1 ? hv[e + 1] * dsc : 0.f;
            }
          }
          hf = as_frag(pack8(hv));
        } else {
          // ---- x2 += W2[:, hidden block h] gelu(u)
          ks_mma(wb, hf, acc);
          step_end(stg_);
          if constexpr (s == SQ - 1) {
            if constexpr (DROP) {
              const int64_t e0 = (r0 + tok) * 256 + 8 * g;
#pragma unroll
              for (int t = 0; t < 16; ++t) {
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                  bool k0, k1;
                  drop_keep2(dseed, p.drop_salt + 1, e0 + 32 * (t >> 1) + 4 * (t & 1) + r, dth, k0, k1);
                  acc[t][r] = x1[t][r] + (k0 ? acc[t][r] * dsc : 0.f);
                  acc[t][r + 1] = x1[t][r + 1] + (k1 ? acc[t][r + 1] * dsc : 0.f);
                }
              }
            }
            if constexpr (!BSP) {
#pragma unroll
              for (int pr = 0; pr < 8; ++pr) {
                if (pr % RG == 0) stage_reserve(stg_, RG);
                store_lines<false>(stg_, xt, Lf, 128 * pr, as_u4(acc[2 * pr]), as_u4(acc[2 * pr + 1]));
              }
            }
            if constexpr (QKV) ln_pack(a0, p.xhat1n, p.rstd1n, r0, BSP);  // the next block's norm1 (affine folded into its qkv weights / bias)
            if constexpr (!(QKV && CH_PF_SPREAD)) prefetch(tl + 1 < nt ? tl + 1 : tl, std::integral_constant<int, -1>{});
          }
        }
      } else {
        // ---- the next block's spatial qkv (the next tile's rows are requested a pair per step beside it)
        constexpr int pq = s - SQ;
        if constexpr (CH_PF_SPREAD && pq >= 2 && pq < 14) prefetch(tl + 1 < nt ? tl + 1 : tl, std::integral_constant<int, pq - 2>{});
        f32x4v_t c0 = lds_f4v(bias + 6144 + 128 * pq), c1 = lds_f4v(bias + 6144 + 128 * pq + 16);
        nb_mma(wb, a0, c0, c1);
        step_end(stg_);
        qb[pq & 7] = pack_pair(c0, c1);
        if constexpr (BSP) {
          // x of this tile one 128-row-byte piece per step (its registers are re-loaded with the next tile's rows from step pq + 6 on),
          // then the next block's xhat1, the qkv columns as they are finished
          if constexpr (pq < 8) store_lines<false>(stg_, xt, Lf, 128 * pq, as_u4(acc[2 * pq]), as_u4(acc[2 * pq + 1]));
          if constexpr (SAVE && pq >= 8 && pq < 12)
            store_lines<false>(stg_, reinterpret_cast<uint16_t*>(p.xhat1n) + r0 * 256, Lb, 128 * (pq - 8), __builtin_bit_cast(uint4, a0[2 * (pq - 8)]),
                               __builtin_bit_cast(uint4, a0[2 * (pq - 8) + 1]));
          if constexpr ((pq & 1) == 1) store_lines(stg_, qt, Lq, 64 * (pq - 1), qb[(pq - 1) & 7], qb[pq & 7]);
        } else if constexpr (ST) {
          if constexpr ((pq & 1) == 1) store_lines(stg_, qt, Lq, 64 * (pq - 1), qb[(pq - 1) & 7], qb[pq & 7]);
        } else if constexpr ((pq & 7) == 7) {
#pragma unroll
          for (int pp = 0; pp < 4; ++pp) store_lines(stg_, qt, Lq, 64 * (pq - 7) + 128 * pp, qb[2 * pp], qb[2 * pp + 1]);
        }
      }
    });
    CH_TOUCH_A(a1);
    CH_TOUCH_ACC(acc);
  }
  stage_finish(stg_);
}

// ------------------------------------------------------------------------------------------------ chain A + temporal attention + chain B
// The whole of a block between its spatial attention and the next block's (st_transformer.py:86 proj ... :111 temporal attention ...
// :112 MLP, :85-86 of the next block) in ONE launch, for training passes over windows of exactly T = 16 frames.  A compute wave owns a
// COLUMN: the 16 frames of one (sample, token position), lane tok = frame -- rows SA apart, so every row pointer / line offset of the
// chains is the same with the row pitch multiplied by SA, and each lane takes its OWN frame's shift / scale.  With a column in a wave the
// causal temporal attention (attention.py:37-61, 16 x 16 scores per head) is wave-local: the qkv blocks chain A's last 24 steps leave
// in the accumulators ARE the attention's MFMA operands (q / k of a head: one v_mfma_f32_16x16x32_bf16 gives S^T[key][query], lane =
// query, 4 keys in registers, soft-max = 4 registers + 2 shuffles; P^T packed is the B operand of the 16 x 16 x 16 product with V^T,
// which goes through a 1 KB per-wave LDS scratch to be read transposed), its output -- through the same scratch -- is chain B's first
// B operand, and the residual row x never leaves the accumulators between the two chains.  Against the three launches it replaces
// (hma_chain_a_fwd, hma_attn_temporal_fwd, hma_chain_b_fwd) a row moves 4 096 bytes less: x out and back (2 048), qkv_t back (1 536),
// o_t back (512); everything the backward needs is still written (xhat_m, xm, bf16(x2), qkv_t, o_t, xhat2, the next xhat1, rstd's).
// Steps per tile: 16 + 24 (chain A) + 8 + 64 (+ 24) (chain B); the next tile's rows are requested beside chain B's qkv steps.
struct col_map {
  int64_t base, end;  // this workgroup's columns [base, end)
  int nt;
};
__device__ __forceinline__ col_map make_col_map(int64_t cols, int nw) {
  col_map t;
  const int64_t per = (cols + gridDim.x - 1) / gridDim.x;
  t.base = (int64_t)blockIdx.x * per;
  t.end = t.base + per < cols ? t.base + per : cols;
  t.nt = t.end > t.base ? (int)((t.end - t.base + nw - 1) / nw) : 0;
  return t;
}
typedef __attribute__((ext_vector_type(4))) short s16x4v_t;
constexpr int AB_SMEM = NS * SLOT + 4 * (1280 + 2304) + NCW * 2048 + 16 * 2064;  // ring | biases | attention scratch | shift / scale rows
static_assert(AB_SMEM <= 163840, "one workgroup per CU");
// MOD = false: blocks without action tokens (no ModulateLayer: chain A is the spatial projection + residual, its bf16 copy is the temporal
// qkv's operand; nothing of xhat_m / xm / rstd_m is written).  DROP: mlp_drop > 0 -- the two nn.Dropout sites of Mlp.forward with the
// counter-based masks of chain B / hma_mlp_bwd (the residual row is kept apart from the branch output, which is masked before the add).
template <bool QKV, bool MOD = true, bool DROP = false>
__global__ __launch_bounds__(CH_THREADS, 2) void chain_ab_fwd_kernel(hma_chain_ab_fwd_t p) {
  static_assert(!ST, "the fused chain has no storer mode");
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  HMA_LDS(char)* lds = (HMA_LDS(char)*)smem;
  const uint32_t lds_b = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = NCW;
  const int SA = p.SA;
  const col_map cmap = make_col_map(p.B * (int64_t)SA, NW);
  const int nt = cmap.nt;
  // biases: chain A proj 0..255 | lin 256..511 | qkv_t 512..1279; chain B (at 1280) proj_t 0..255 | fc2 256..511 | fc1 512..1535 | qkv_s 1536..2303
  constexpr int BB = 1280;
  {
    HMA_LDS(float)* bl = (HMA_LDS(float)*)(lds + L_BIAS);
    for (int i = tid; i < BB + 2304; i += CH_THREADS) {
      float v = 0.f;
      if (i < 256) v = p.b_proj_s ? p.b_proj_s[i] : 0.f;
      else if (i < 512) v = p.b_lin ? p.b_lin[i - 256] : 0.f;
      else if (i < BB) v = p.b_qkv_t ? p.b_qkv_t[i - 512] : 0.f;
      else if (i < BB + 256) v = p.b_proj_t ? p.b_proj_t[i - BB] : 0.f;
      else if (i < BB + 512) v = p.b2 ? p.b2[i - BB - 256] : 0.f;
      else if (i < BB + 1536) v = p.b1[i - BB - 512];
      else v = (QKV && p.b_qkv_s) ? p.b_qkv_s[i - BB - 1536] : 0.f;
      bl[i] = v;
    }
  }
  __syncthreads();
  constexpr int SL = 8, SQT = MOD ? 16 : 8, SB = SQT + 24;  // first linear_out step, first temporal-qkv step, first chain B step
  constexpr int SM = SB + 8, SQ = SM + 64;            // first MLP step, first spatial-qkv step
  constexpr int PER_TILE = SQ + (QKV ? 24 : 0);
  static_assert(PER_TILE % PB == 0, "whole barrier groups per tile");
  constexpr int L_SCR = L_BIAS + 4 * (BB + 2304);     // per compute wave: 1 KB V^T scratch + 1 KB output scratch
  constexpr int L_SSC = L_SCR + NW * 2048;            // the tile's sample: 16 frames x (2 KB shift | scale + 16 bytes of skew)
  static_assert(L_SSC + 16 * 2064 <= AB_SMEM, "scratch and shift / scale rows fit beside the ring");
  if (wave > NCW) return;
  if (wave == NCW) {
    // (the ring walks its segments in order and wraps at the first empty one: without the modulation segment 1 is left out)
    constexpr int o = MOD ? 0 : 1;
    ring_src ws = {reinterpret_cast<const char*>(p.seg[0]), reinterpret_cast<const char*>(p.seg[1 + o]), reinterpret_cast<const char*>(p.seg[2 + o]),
                   reinterpret_cast<const char*>(p.seg[3 + o]), p.bundles[0], p.bundles[1 + o], p.bundles[2 + o], p.bundles[3 + o]};
    ws.s4 = reinterpret_cast<const char*>(p.seg[4 + o]);
    ws.n4 = p.bundles[4 + o];
    if constexpr (MOD) {
      ws.s5 = reinterpret_cast<const char*>(p.seg[5]);
      ws.n5 = p.bundles[5];
    }
    loader_run<0, NW>(ws, PER_TILE, nt, lds_b, lane, MOD ? p.ss : nullptr, 0, 1, lds, nullptr, cmap.base, SA, (uint32_t)L_SSC);
    return;
  }
  stage_t stg_ = make_stage(lds, wave, lane);
  const int tok = lane & 15, g = lane >> 4;
  // first row of the column of wave `wave` in tile tl: (b 16 + 0) SA + s; row of frame tok = that + tok SA
  auto col_of = [&](int tl) __attribute__((always_inline)) { return cmap.base + (int64_t)tl * NW + wave; };
  auto row_of = [&](int64_t c) __attribute__((always_inline)) {
    const int64_t b = c / SA;
    return b * 16 * SA + (c - b * SA);
  };
  bf16x8_t a0[8], a1[8];
  f32x4v_t acc[16];
  uint4 hqs[8];   // packed xhat_m of the tile in flight
  uint4 qv[24];   // the temporal q | k | v blocks (block pq = head pq & 7 of part pq >> 3)
  uint4 qb[8];    // the next block's spatial qkv blocks waiting for their burst
  const uint32_t lofs = (uint32_t)tok * (uint32_t)SA * 256u + 8u * (uint32_t)g, tok_sa = (uint32_t)tok * (uint32_t)SA;
  auto prefetch = [&](int tl, auto part_) __attribute__((always_inline)) {  // the next tile's o_s (8 loads) and x (16) rows
    constexpr int part = decltype(part_)::value;
    int64_t c = col_of(tl);
    c = c < cmap.end ? c : cmap.end - 1;
    // (a wave-uniform base + this lane's 32-bit element offset `lofs`: per-lane 64-bit row pointers are loop invariants the compiler
    // keeps in registers -- and spills: a scratch reload is a vector-memory load, and waiting for it drains every store in flight)
    const int64_t mu = row_of(c) * 256;
    uint32_t lo = lofs;
    asm volatile("" : "+v"(lo));  // (opaque: keeps `array + lofs` from being hoisted as a 64-bit pointer per array)
    const uint16_t* orow = reinterpret_cast<const uint16_t*>(p.o_s) + mu + lo;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (part < 0 || j / 2 == part) a1[j] = as_frag(*reinterpret_cast<const uint4*>(orow + 32 * j));
    const float* xrow = p.x + mu + lo;
#pragma unroll
    for (int pr = 0; pr < 8; ++pr) {
      if (part < 0 || pr + 4 == part) {
        acc[2 * pr] = ld4(xrow + 32 * pr);
        acc[2 * pr + 1] = ld4(xrow + 32 * pr + 4);
      }
    }
  };
  prefetch(0, std::integral_constant<int, -1>{});
  CH_TOUCH_A(a1);
  CH_TOUCH_ACC(acc);
  HMA_LDS(char)* ring = lds + lane * 16;
  HMA_LDS(char)* biasA = lds + L_BIAS + 32 * g;
  HMA_LDS(char)* biasB = lds + L_BIAS + 4 * BB + 32 * g;
  HMA_LDS(char)* scrV = lds + L_SCR + wave * 2048;
  const line_offs Lb = make_lines(512 * SA, tok, 16 * g, 64);        // bf16 [., 256] arrays, rows SA apart
  const line_offs Lf = make_lines(1024 * SA, tok, 32 * g, 16);       // fp32 [., 256]
  const line_offs Lq = make_lines(1536 * SA, tok, 16 * g, 64);       // qkv [., 768]
  const float c_log2 = p.attn_scale * 1.4426950408889634f;
  int slot = 0;
  uint32_t dseed = 0, dth = 0;
  float dsc = 1.f;
  f32x4v_t x1[DROP ? 16 : 1];  // (DROP: the residual row beside the branch output)
  if constexpr (DROP) {
    dseed = *p.drop_seed;
    dth = drop_thresh(p.drop_p);
    dsc = drop_scale(p.drop_p);
  }
  // a row statistic of this lane's frame: uniform base + an opaque 32-bit lane offset (a per-lane 64-bit pointer per array is a loop
  // invariant the compiler keeps -- three of them spilled in the dropout form, and a scratch reload drains the store queue)
  auto st_rstd = [&](float* arr, int64_t rc_, float v) __attribute__((always_inline)) {
    uint32_t ts = tok_sa;
    asm volatile("" : "+v"(ts));
    (arr + rc_)[ts] = v;
  };
  // LayerNorm (no affine) of the rows in acc -> packed bf16 B operand, saved with 1 / sigma for the backward
  auto ln_pack = [&](bf16x8_t (&dst)[8], void* xhat_out, float* rstd_out, int64_t rc, float eps) __attribute__((always_inline)) {
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) sum += (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / 256.0f);
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = acc[t][r] - mean;
        sq = __builtin_fmaf(d, d, sq);
      }
    }
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    const float rstd = rsqrtf(sq * (1.0f / 256.0f) + eps);
    const float nb = -mean * rstd;
#pragma unroll
    for (int pr = 0; pr < 8; ++pr) {
      float h[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = __builtin_fmaf(acc[2 * pr + (e >> 2)][e & 3], rstd, nb);
      dst[pr] = as_frag(pack8(h));
    }
    uint16_t* xo = reinterpret_cast<uint16_t*>(xhat_out) + rc * 256;
#pragma unroll
    for (int pp = 0; pp < 4; ++pp)
      store_lines<false>(stg_, xo, Lb, 128 * pp, __builtin_bit_cast(uint4, dst[2 * pp]), __builtin_bit_cast(uint4, dst[2 * pp + 1]));
    st_rstd(rstd_out, rc, rstd);
  };

#pragma unroll 1
  for (int tl = 0; tl < nt; ++tl) {
    const int64_t col = col_of(tl);
    if (col >= cmap.end) {  // (a wave past the workgroup's last column: possible in its last tile only)
#pragma unroll 1
      for (int s = 0; s < PER_TILE; s += PB) CH_BARRIER();
      slot = (slot + PER_TILE) % NS;
      continue;
    }
    const int64_t rc = row_of(col);                       // row of frame 0; frame tok is rc + tok SA
    const int64_t frame0 = (col / SA) * 16;               // (b, t = 0)
    float* xt = p.x + rc * 256;
    uint16_t* xb = reinterpret_cast<uint16_t*>(p.x2b) + rc * 256;
    uint16_t* xh = reinterpret_cast<uint16_t*>(p.xhat_m) + rc * 256;
    uint16_t* xm = reinterpret_cast<uint16_t*>(p.xm) + rc * 256;
    uint16_t* qt = reinterpret_cast<uint16_t*>(p.qkv_t) + rc * 768;
    uint16_t* ot = reinterpret_cast<uint16_t*>(p.o_t) + rc * 256;
    uint16_t* qs = QKV ? reinterpret_cast<uint16_t*>(p.qkv_s) + rc * 768 : nullptr;
#pragma unroll
    for (int j = 0; j < 8; ++j) a0[j] = a1[j];
    bf16x8_t hf;
    static_for<PER_TILE>([&](auto sc_) __attribute__((always_inline)) {
      constexpr int s = decltype(sc_)::value;
      if constexpr (s % PB == 0) CH_BARRIER();
      HMA_LDS(char)* wb = ring + slot * SLOT;
      slot = slot + 1 == NS ? 0 : slot + 1;
      if constexpr (s < SL) {
        // ---- chain A: x1 = x + o_s Wproj_s^T + b
        nb_mma(wb, a0, acc[2 * s], acc[2 * s + 1]);
        add4(acc[2 * s], lds_f4(biasA + 128 * s));
        add4(acc[2 * s + 1], lds_f4(biasA + 128 * s + 16));
        if constexpr (s == SL - 1 && !MOD) {
          // no action tokens: x1 is the temporal qkv's input; its bf16 copy is that GEMM's operand and the weight gradient's
#pragma unroll
          for (int pr = 0; pr < 8; ++pr) a0[pr] = as_frag(pack_pair(acc[2 * pr], acc[2 * pr + 1]));
#pragma unroll
          for (int pp = 0; pp < 4; ++pp)
            store_lines<false>(stg_, xb, Lb, 128 * pp, __builtin_bit_cast(uint4, a0[2 * pp]), __builtin_bit_cast(uint4, a0[2 * pp + 1]));
        }
        if constexpr (s == SL - 1 && MOD) {
          // LayerNorm (no affine, eps 1e-6) of the row, then the modulation with THIS lane's frame's shift / scale (L2-resident table)
          float sum = 0.f;
#pragma unroll
          for (int t = 0; t < 16; ++t) sum += (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
          sum += __shfl_xor(sum, 16, 64);
          sum += __shfl_xor(sum, 32, 64);
          const float mean = sum * (1.0f / 256.0f);
          float sq = 0.f;
#pragma unroll
          for (int t = 0; t < 16; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float d = acc[t][r] - mean;
              sq = __builtin_fmaf(d, d, sq);
            }
          }
          sq += __shfl_xor(sq, 16, 64);
          sq += __shfl_xor(sq, 32, 64);
          const float rstd = rsqrtf(sq * (1.0f / 256.0f) + 1e-6f);
          const float nb = -mean * rstd;
          // The loader staged the shift | scale rows of the sample of the tile's FIRST column (32 global loads per lane here -- the
          // L2-resident table read directly -- cost 25 us per launch: -DCH_AB_SSG); a wave whose column lies in the next sample
          // (a tile that straddles two samples) takes its rows from the table itself.
          uint32_t sso = (uint32_t)tok * 512u + 8u * (uint32_t)g;
          asm volatile("" : "+v"(sso));  // (opaque, as the row offsets: `p.ss + lane part` is not kept as a 64-bit pointer per lane)
          const float* ssr = p.ss + frame0 * 512 + sso;
          HMA_LDS(char)* ssl = lds + L_SSC + tok * 2064 + 32 * g;
#ifdef CH_AB_SSG
          const bool staged = false;
#else
          const bool staged = frame0 == ((cmap.base + (int64_t)tl * NW) / SA) * 16;
#endif
#pragma unroll
          for (int pr = 0; pr < 8; ++pr) {
            float h[8], mm[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = __builtin_fmaf(acc[2 * pr + (e >> 2)][e & 3], rstd, nb);
            float4 sh0, sh1, sc0, sc1;
            if (staged) {
              sh0 = lds_f4(ssl + 128 * pr); sh1 = lds_f4(ssl + 128 * pr + 16);
              sc0 = lds_f4(ssl + 1024 + 128 * pr); sc1 = lds_f4(ssl + 1024 + 128 * pr + 16);
            } else {
              sh0 = *reinterpret_cast<const float4*>(ssr + 32 * pr); sh1 = *reinterpret_cast<const float4*>(ssr + 32 * pr + 4);
              sc0 = *reinterpret_cast<const float4*>(ssr + 256 + 32 * pr); sc1 = *reinterpret_cast<const float4*>(ssr + 256 + 32 * pr + 4);
            }
            const float sh[8] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w};
            const float sc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
            hqs[pr] = pack8(h);
#pragma unroll
            for (int e = 0; e < 8; ++e) mm[e] = __builtin_fmaf(h[e], 1.0f + sc[e], sh[e]);
            a1[pr] = as_frag(pack8(mm));
          }
          st_rstd(p.rstd_m, rc, rstd);
#pragma unroll
          for (int pp = 0; pp < 4; ++pp) store_lines<false>(stg_, xh, Lb, 128 * pp, hqs[2 * pp], hqs[2 * pp + 1]);
#pragma unroll
          for (int pp = 0; pp < 4; ++pp)
            store_lines<false>(stg_, xm, Lb, 128 * pp, __builtin_bit_cast(uint4, a1[2 * pp]), __builtin_bit_cast(uint4, a1[2 * pp + 1]));
        }
      } else if constexpr (MOD && s < SQT) {
        // ---- x2 = x1 + xm Wlin^T + b: the residual row (it stays in acc until chain B's end); its bf16 copy is the qkv GEMM's operand
        constexpr int pr = s - SL;
        nb_mma(wb, a1, acc[2 * pr], acc[2 * pr + 1]);
        add4(acc[2 * pr], lds_f4(biasA + 1024 + 128 * pr));
        add4(acc[2 * pr + 1], lds_f4(biasA + 1024 + 128 * pr + 16));
        a0[pr] = as_frag(pack_pair(acc[2 * pr], acc[2 * pr + 1]));
        if constexpr (s == SQT - 1) {
#pragma unroll
          for (int pp = 0; pp < 4; ++pp)
            store_lines<false>(stg_, xb, Lb, 128 * pp, __builtin_bit_cast(uint4, a0[2 * pp]), __builtin_bit_cast(uint4, a0[2 * pp + 1]));
        }
      } else if constexpr (s < SB) {
        // ---- temporal qkv = bf16(x2) Wqkv^T + b: kept for the attention below, written out for the backward
        constexpr int pq = s - SQT;
        f32x4v_t c0 = lds_f4v(biasA + 2048 + 128 * pq), c1 = lds_f4v(biasA + 2048 + 128 * pq + 16);
        nb_mma(wb, a0, c0, c1);
        qv[pq] = pack_pair(c0, c1);
        if constexpr ((pq & 7) == 7) {
#pragma unroll
          for (int pp = 0; pp < 4; ++pp) store_lines(stg_, qt, Lq, 64 * (pq - 7) + 128 * pp, qv[pq - 7 + 2 * pp], qv[pq - 6 + 2 * pp]);
        }
        if constexpr (s == SB - 1) {
          // ---- causal temporal attention of the column, head by head (lane tok = query frame)
#pragma unroll
          for (int h = 0; h < 8; ++h) {
            const bf16x8_t aQ = as_frag(qv[h]), aK = as_frag(qv[8 + h]);
            *(HMA_LDS(u32x4_t)*)(scrV + tok * 64 + 16 * g) = __builtin_bit_cast(u32x4_t, qv[16 + h]);  // V of head h: [frame][32]
            const f32x4v_t z4 = {0.f, 0.f, 0.f, 0.f};
            f32x4v_t st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aK, aQ, z4, 0, 0, 0);  // [key = 4 g + r][query = tok]
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              st[r] = (4 * g + r <= tok) ? st[r] * c_log2 : -INFINITY;
              mx = fmaxf(mx, st[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float l = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              st[r] = __builtin_amdgcn_exp2f(st[r] - mx);
              l += st[r];
            }
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            const float inv = 1.0f / l;
            const uint2 pw = make_uint2(pack_bf16(st[0] * inv, st[1] * inv), pack_bf16(st[2] * inv, st[3] * inv));
            const s16x4v_t pT = __builtin_bit_cast(s16x4v_t, pw);  // B: k = key, n = query
            asm volatile("" ::: "memory");
            // A = V^T by ONE ds_read_b64_tr_b16 per product (see chain T backward): output row i of product dd = channel
            // 8 (i >> 2) + 4 dd + (i & 3), so the lane ends with channels 8 g .. 8 g + 7 of its frame -- o_t of head h in B-operand form
            // without a second trip through LDS
            typedef short ab_v4s16_t __attribute__((ext_vector_type(4)));
            f32x4v_t od[2];
#pragma unroll
            for (int dd = 0; dd < 2; ++dd) {
              const s16x4v_t vT = __builtin_bit_cast(
                  s16x4v_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((HMA_LDS(ab_v4s16_t)*)(scrV + (4 * g + (tok >> 2)) * 64 + 16 * (tok & 3) + 8 * dd)));
              od[dd] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vT, pT, z4, 0, 0, 0);  // [channel 8 g + 4 dd + r][query = tok]
            }
            asm volatile("" ::: "memory");
            a1[h] = as_frag(make_uint4(pack_bf16(od[0][0], od[0][1]), pack_bf16(od[0][2], od[0][3]), pack_bf16(od[1][0], od[1][1]),
                                       pack_bf16(od[1][2], od[1][3])));
          }
#pragma unroll
          for (int pp = 0; pp < 4; ++pp)
            store_lines<false>(stg_, ot, Lb, 128 * pp, __builtin_bit_cast(uint4, a1[2 * pp]), __builtin_bit_cast(uint4, a1[2 * pp + 1]));
#pragma unroll
          for (int j = 0; j < 8; ++j) a0[j] = a1[j];
        }
      } else if constexpr (s < SM) {
        // ---- chain B: x1 = x2 + o_t Wproj_t^T + b
        constexpr int pr = s - SB;
        nb_mma(wb, a0, acc[2 * pr], acc[2 * pr + 1]);
        add4(acc[2 * pr], lds_f4(biasB + 128 * pr));
        add4(acc[2 * pr + 1], lds_f4(biasB + 128 * pr + 16));
        if constexpr (s == SM - 1) {
          ln_pack(a1, p.xhat2, p.rstd2, rc, p.ln_eps);  // xhat2 (norm2's affine sits in the packed fc1 weights / bias)
          if constexpr (DROP) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
              x1[t] = acc[t];
              acc[t] = f32x4v_t{0.f, 0.f, 0.f, 0.f};
            }
          }
#pragma unroll
          for (int pr2 = 0; pr2 < 8; ++pr2) {  // + fc2 bias, once
            add4(acc[2 * pr2], lds_f4(biasB + 1024 + 128 * pr2));
            add4(acc[2 * pr2 + 1], lds_f4(biasB + 1024 + 128 * pr2 + 16));
          }
        }
      } else if constexpr (s < SQ) {
        constexpr int h = (s - SM) >> 1;
        if constexpr (((s - SM) & 1) == 0) {
          f32x4v_t c0 = lds_f4v(biasB + 2048 + 128 * h), c1 = lds_f4v(biasB + 2048 + 128 * h + 16);
          nb_mma(wb, a1, c0, c1);
          float uv[8], hv[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            uv[e] = c0[e];
            uv[4 + e] = c1[e];
          }
          gelu_n<8>(uv, hv);
          if constexpr (DROP) {  // (element index = row * 1024 + hidden unit, as chain B and hma_mlp_bwd count it)
            const int64_t e0 = (rc + (int64_t)tok_sa) * 1024 + 32 * h + 8 * g;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              bool k0, k1;
              drop_keep2(dseed, p.drop_salt, e0 + e, dth, k0, k1);
              hv[e] = k0 ? hv[e] * dsc : 0.f;
              hv[e + 1] = k1 ? hv[e + 1] * dsc : 0.f;
            }
          }
          hf = as_frag(pack8(hv));
        } else {
          ks_mma(wb, hf, acc);
          if constexpr (s == SQ - 1 && DROP) {
            const int64_t e0 = (rc + (int64_t)tok_sa) * 256 + 8 * g;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
#pragma unroll
              for (int r = 0; r < 4; r += 2) {
                bool k0, k1;
                drop_keep2(dseed, p.drop_salt + 1, e0 + 32 * (t >> 1) + 4 * (t & 1) + r, dth, k0, k1);
                acc[t][r] = x1[t][r] + (k0 ? acc[t][r] * dsc : 0.f);
                acc[t][r + 1] = x1[t][r + 1] + (k1 ? acc[t][r + 1] * dsc : 0.f);
              }
            }
          }
          if constexpr (s == SQ - 1) {
#pragma unroll
            for (int pr2 = 0; pr2 < 8; ++pr2) store_lines<false>(stg_, xt, Lf, 128 * pr2, as_u4(acc[2 * pr2]), as_u4(acc[2 * pr2 + 1]));
            if constexpr (QKV) ln_pack(a0, p.xhat1n, p.rstd1n, rc, p.ln_eps);
            if constexpr (!QKV) prefetch(tl + 1 < nt ? tl + 1 : tl, std::integral_constant<int, -1>{});
          }
        }
      } else {
        // ---- the next block's spatial qkv (the next tile's rows are requested a pair per step beside it)
        constexpr int pq = s - SQ;
        if constexpr (pq >= 2 && pq < 14) prefetch(tl + 1 < nt ? tl + 1 : tl, std::integral_constant<int, pq - 2>{});
        f32x4v_t c0 = lds_f4v(biasB + 6144 + 128 * pq), c1 = lds_f4v(biasB + 6144 + 128 * pq + 16);
        nb_mma(wb, a0, c0, c1);
        qb[pq & 7] = pack_pair(c0, c1);
        if constexpr ((pq & 7) == 7) {
#pragma unroll
          for (int pp = 0; pp < 4; ++pp) store_lines(stg_, qs, Lq, 64 * (pq - 7) + 128 * pp, qb[2 * pp], qb[2 * pp + 1]);
        }
      }
    });
    CH_TOUCH_A(a1);
    CH_TOUCH_ACC(acc);
  }
}

// ------------------------------------------------------------------------------------------------ chain T, backward
// The temporal attention's backward of a block (st_transformer.py:111, attention.py:37-61 causal, autograd mirror) in ONE launch, for
// training passes over windows of exactly 16 frames: d_o = bf16(dx) Wproj (the temporal projection's input gradient, hma_gemm_nt before)
// and the attention backward of the column (hma_attn_temporal_bwd before), without d_o's round trip through HBM.  A compute wave owns a
// COLUMN as in the fused forward chain: the N-block bundle pr of Wproj^T leaves head pr's 32 channels of d_o in the accumulators, packed
// they ARE the 16 x 16 x 32 MFMA operand (lane = frame, 8 channels); q / k / v of the head are read from the saved qkv rows in the same
// operand form.  Per head: S^T = K Q^T and dP^T = V dO^T (lane = query, 4 keys in registers), soft-max and delta = sum P dP with two
// shuffles each; dS and P in the other orientation (B operand with k = query) come back transposed from a 16 x 16 LDS slab by ONE
// ds_read_b64_tr_b16 each (attn_t_bwd_kernel recomputes both products and fetches the query's statistics by lane reads: 2 MFMAs, 12
// cross-lane reads and 4 exponentials more per head); the three time-contracting products dQ^T = K^T dS^T, dK^T = Q^T dS,
// dV^T = dO^T P run on the 16 x 16 x 16 MFMA with the [frame][channel] matrices read transposed from a per-wave LDS scratch, again one
// instruction per operand.  The read takes channel 8 (i >> 2) + 4 dd + (i & 3) for output row i of product dd, so a lane ends with
// channels 8 g .. 8 g + 7 of its frame -- the chains' chunk form, i.e. whole-line
// stores.  Eight steps per tile; the next tile's q / k / v of a head pair are requested into
// the pair's registers as soon as it is done (the two heads' 64 bytes are the halves of one cache line).
// Standalone at B = 32, SA = 320: 125 us (4.7 TB/s of 3 584 B per row) against 32 + 122 for the two launches; 56 us without its
// memory instructions.
constexpr int TB_PITCH = 80;                         // bytes per frame row of a scratch slab (32 channels + 16: the four frame groups of a gather hit four bank groups)
constexpr int TB_SLAB = 16 * TB_PITCH;
constexpr int TB_SP = 40;                            // bytes per query row of the dS / P slabs (16 keys + 8)
constexpr int TB_SCR = 3 * TB_SLAB + 2 * 16 * TB_SP; // Q | K | dO of the head in flight + dS | P [query][key], per compute wave
typedef short tb_v4s16_t __attribute__((ext_vector_type(4)));
constexpr int TB_SMEM = NS * SLOT + NCW * TB_SCR;
// TV = false: T == 16 (every lane of a column tile is a frame); TV = true: windows of T < 16 frames -- lane tok >= T has no row: its loads
// go to frame T - 1 (finite values), its gradient row enters as ZERO (an invalid query attends to valid keys: its dO must not reach
// their dK / dV), its stores are masked.  The causal mask keeps invalid KEYS away from valid queries by itself.
template <bool TV>
__global__ __launch_bounds__(CH_THREADS, 2) void chain_t_bwd_kernel(hma_chain_t_bwd_t p) {
  static_assert(!ST, "chain T has no storer mode");
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  HMA_LDS(char)* lds = (HMA_LDS(char)*)smem;
  const uint32_t lds_b = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = NCW;
  const int SA = p.SA;
  const col_map cmap = make_col_map(p.B * (int64_t)SA, NW);
  const int nt = cmap.nt;
  constexpr int PER_TILE = 8;
  static_assert(PER_TILE % PB == 0, "whole barrier groups per tile");
  if (wave > NCW) return;
  if (wave == NCW) {
    const ring_src ws = {reinterpret_cast<const char*>(p.w.seg[0]), reinterpret_cast<const char*>(p.w.seg[1]),
                         reinterpret_cast<const char*>(p.w.seg[2]), reinterpret_cast<const char*>(p.w.seg[3]),
                         p.w.bundles[0], p.w.bundles[1], p.w.bundles[2], p.w.bundles[3]};
    loader_run<1, NW>(ws, PER_TILE, nt, lds_b, lane, nullptr, 0, 1, lds);
    return;
  }
  stage_t stg_ = make_stage(lds, wave, lane);
  const int tok = lane & 15, g = lane >> 4;
  const int T = TV ? p.T : 16;
  const bool live = !TV || tok < T;           // this lane's frame exists
  const int tok_ld = TV ? (tok < T ? tok : T - 1) : tok;
  auto col_of = [&](int tl) __attribute__((always_inline)) { return cmap.base + (int64_t)tl * NW + wave; };
  auto row_of = [&](int64_t c) __attribute__((always_inline)) {
    const int64_t b = c / SA;
    return b * T * SA + (c - b * SA);
  };
  auto lane_row = [&](int tl) __attribute__((always_inline)) {  // this lane's row (frame tok) of the wave's column in tile tl
    int64_t c = col_of(tl);
    c = c < cmap.end ? c : cmap.end - 1;
    return row_of(c) + (int64_t)tok_ld * SA;
  };
  bf16x8_t a0[8], a1[8];  // bf16(dx) rows: this tile's / the next one's
  bf16x8_t qf[24];        // q | k | v of the column: fragment 8 part + head
  auto load_dy = [&](int64_t m, auto j_) __attribute__((always_inline)) {
    constexpr int j = decltype(j_)::value;
    if (CH_ABL & 2) return;
    a1[j] = as_frag(*reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(p.dy_bf16) + m * 256 + 8 * g + 32 * j));
  };
  auto load_head = [&](int64_t m, auto h_) __attribute__((always_inline)) {
    constexpr int h = decltype(h_)::value;
    if (CH_ABL & 2) return;
    const uint16_t* row = reinterpret_cast<const uint16_t*>(p.qkv) + m * 768 + 8 * g + 32 * h;
    qf[h] = as_frag(*reinterpret_cast<const uint4*>(row));
    qf[8 + h] = as_frag(*reinterpret_cast<const uint4*>(row + 256));
    qf[16 + h] = as_frag(*reinterpret_cast<const uint4*>(row + 512));
  };
  if (CH_ABL & 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) a1[j] = as_frag(make_uint4(lane, j, lane, j));
#pragma unroll
    for (int j = 0; j < 24; ++j) qf[j] = as_frag(make_uint4(0x3c003c00u + lane, 0x3c003c00u + j, 0x3c003c00u, 0x3c003c00u));
  }
  {
    const int64_t m = lane_row(0);
    static_for<8>([&](auto j_) __attribute__((always_inline)) {
      load_dy(m, j_);
      load_head(m, j_);
    });
  }
  CH_TOUCH_A(a1);
  HMA_LDS(char)* ring = lds + lane * 16;
  HMA_LDS(char)* sQ = lds + NS * SLOT + wave * TB_SCR;
  HMA_LDS(char)* sK = sQ + TB_SLAB;
  HMA_LDS(char)* sG = sK + TB_SLAB;
  HMA_LDS(char)* sD = sG + TB_SLAB;
  HMA_LDS(char)* sP = sD + 16 * TB_SP;
  const line_offs Lq = make_lines(1536 * SA, tok, 16 * g, 64);
  const float c_log2 = p.attn_scale * 1.4426950408889634f, scale = p.attn_scale;
  // ds_read_b64_tr_b16 (a 16-lane group reads a [4 row][16 column] block: lane i supplies the address of row i >> 2, columns
  // 4 (i & 3) .. + 3, and receives column i of the 4 rows): the A operand of a time-contracting product -- output row tok = channel
  // 8 (tok >> 2) + 4 dd + (tok & 3), frames 4 g .. 4 g + 3 -- is one such read of a [frame][channel] slab with the supplying lane at
  // channels 8 (tok & 3) + 4 dd; the transposed dS / P (B operand: key = tok, queries 4 g .. 4 g + 3) one of a [query][key] slab.
  const int tr_x = (4 * g + (tok >> 2)) * TB_PITCH + 16 * (tok & 3);
  const int tr_s = (4 * g + (tok >> 2)) * TB_SP + 8 * (tok & 3);
  int slot = 0;
  uint4 hold[3];  // dq | dk | dv of the even head, waiting for the odd one (a 128-byte line = the two heads' 64 bytes)
  const f32x4v_t z4 = {0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
  for (int tl = 0; tl < nt; ++tl) {
    const int64_t col = col_of(tl);
    if (col >= cmap.end) {  // (a wave past the workgroup's last column: possible in its last tile only)
#pragma unroll 1
      for (int s = 0; s < PER_TILE; s += PB) CH_BARRIER();
      slot = (slot + PER_TILE) % NS;
      continue;
    }
    const int64_t rc = row_of(col);
    uint16_t* dq_out = reinterpret_cast<uint16_t*>(p.dqkv) + rc * 768;
    const int64_t mn = lane_row(tl + 1 < nt ? tl + 1 : tl);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a0[j] = a1[j];
      if constexpr (TV) {
        if (!live) a0[j] = as_frag(make_uint4(0, 0, 0, 0));
      }
    }
    static_for<PER_TILE>([&](auto sc_) __attribute__((always_inline)) {
      constexpr int h = decltype(sc_)::value;
      if constexpr (h % PB == 0) CH_BARRIER();
      HMA_LDS(char)* wb = ring + slot * SLOT;
      slot = slot + 1 == NS ? 0 : slot + 1;
      // ---- d_o of head h = bf16(dx) Wproj[:, 32 h .. 32 h + 31]
      f32x4v_t c0 = z4, c1 = z4;
      nb_mma(wb, a0, c0, c1);
      if constexpr (h == 0) {
        // the next tile's bf16(dx) row, all of it in the first step: the opaque use of a1 at the end of the tile (where the compiler can
        // count the stores issued since) then waits for loads that are seven steps old and for none of this tile's stores
        static_for<8>([&](auto j_) __attribute__((always_inline)) { load_dy(mn, j_); });
      }
      const bf16x8_t aG = as_frag(pack_pair(c0, c1));
      const bf16x8_t aQ = qf[h], aK = qf[8 + h], aV = qf[16 + h];
      *(HMA_LDS(u32x4_t)*)(sQ + tok * TB_PITCH + 16 * g) = __builtin_bit_cast(u32x4_t, aQ);
      *(HMA_LDS(u32x4_t)*)(sK + tok * TB_PITCH + 16 * g) = __builtin_bit_cast(u32x4_t, aK);
      *(HMA_LDS(u32x4_t)*)(sG + tok * TB_PITCH + 16 * g) = __builtin_bit_cast(u32x4_t, aG);
      asm volatile("" ::: "memory");
      // orientation 1: lane = query (tok), registers = keys 4 g + r
      f32x4v_t st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aK, aQ, z4, 0, 0, 0);
      const f32x4v_t dpt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aV, aG, z4, 0, 0, 0);
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st[r] = (4 * g + r <= tok) ? st[r] * c_log2 : -INFINITY;
        mx = fmaxf(mx, st[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float l = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st[r] = __builtin_amdgcn_exp2f(st[r] - mx);
        l += st[r];
      }
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      const float inv = 1.0f / l;
      float delta = 0.f;  // sum_key P dP ( = dO . O)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st[r] *= inv;
        delta += st[r] * dpt[r];
      }
      delta += __shfl_xor(delta, 16, 64);
      delta += __shfl_xor(delta, 32, 64);
      const uint2 dsw = make_uint2(pack_bf16(st[0] * (dpt[0] - delta), st[1] * (dpt[1] - delta)),
                                   pack_bf16(st[2] * (dpt[2] - delta), st[3] * (dpt[3] - delta)));
      const uint2 pw = make_uint2(pack_bf16(st[0], st[1]), pack_bf16(st[2], st[3]));
      const s16x4v_t dsT = __builtin_bit_cast(s16x4v_t, dsw);  // B: k = key, n = query
      // the other orientation (B: k = query, n = key) through the [query][key] slabs
      typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
      *(HMA_LDS(u32x2_t)*)(sD + tok * TB_SP + 8 * g) = u32x2_t{dsw.x, dsw.y};
      *(HMA_LDS(u32x2_t)*)(sP + tok * TB_SP + 8 * g) = u32x2_t{pw.x, pw.y};
      asm volatile("" ::: "memory");
      const s16x4v_t dsB = __builtin_bit_cast(s16x4v_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((HMA_LDS(tb_v4s16_t)*)(sD + tr_s)));
      const s16x4v_t pB = __builtin_bit_cast(s16x4v_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((HMA_LDS(tb_v4s16_t)*)(sP + tr_s)));
      f32x4v_t dq[2], dk[2], dv[2];
#pragma unroll
      for (int dd = 0; dd < 2; ++dd) {
        const s16x4v_t qT = __builtin_bit_cast(s16x4v_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((HMA_LDS(tb_v4s16_t)*)(sQ + tr_x + 8 * dd)));
        const s16x4v_t kT = __builtin_bit_cast(s16x4v_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((HMA_LDS(tb_v4s16_t)*)(sK + tr_x + 8 * dd)));
        const s16x4v_t gT = __builtin_bit_cast(s16x4v_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((HMA_LDS(tb_v4s16_t)*)(sG + tr_x + 8 * dd)));
        dq[dd] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kT, dsT, z4, 0, 0, 0);  // dQ^T [channel 8 g + 4 dd + r][query = tok]
        dk[dd] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(qT, dsB, z4, 0, 0, 0);  // dK^T [.][key = tok]
        dv[dd] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(gT, pB, z4, 0, 0, 0);   // dV^T [.][key = tok]
      }
      asm volatile("" ::: "memory");
      const uint4 oq = make_uint4(pack_bf16(dq[0][0] * scale, dq[0][1] * scale), pack_bf16(dq[0][2] * scale, dq[0][3] * scale),
                                  pack_bf16(dq[1][0] * scale, dq[1][1] * scale), pack_bf16(dq[1][2] * scale, dq[1][3] * scale));
      const uint4 ok = make_uint4(pack_bf16(dk[0][0] * scale, dk[0][1] * scale), pack_bf16(dk[0][2] * scale, dk[0][3] * scale),
                                  pack_bf16(dk[1][0] * scale, dk[1][1] * scale), pack_bf16(dk[1][2] * scale, dk[1][3] * scale));
      const uint4 ov = make_uint4(pack_bf16(dv[0][0], dv[0][1]), pack_bf16(dv[0][2], dv[0][3]), pack_bf16(dv[1][0], dv[1][1]),
                                  pack_bf16(dv[1][2], dv[1][3]));
      if constexpr ((h & 1) == 0) {
        hold[0] = oq;
        hold[1] = ok;
        hold[2] = ov;
      } else {
        store_lines<true, TV>(stg_, dq_out, Lq, 64 * (h - 1), hold[0], oq, T);
        store_lines<true, TV>(stg_, dq_out, Lq, 64 * (h - 1) + 512, hold[1], ok, T);
        store_lines<true, TV>(stg_, dq_out, Lq, 64 * (h - 1) + 1024, hold[2], ov, T);
      }
      if constexpr ((h & 1) == 1) {  // the next tile's q / k / v of this head pair (the halves of a line), into the registers just read
        load_head(mn, std::integral_constant<int, h - 1>{});
        load_head(mn, sc_);
      }
    });
    CH_TOUCH_A(a1);
  }
}

// ------------------------------------------------------------------------------------------------ readout + cross-entropy
// out_x_proj (st_mask_git.py:681-683) + the factorised cross-entropy (compute_video_loss_and_acc, :603-630) of the image rows in one
// launch: the fp32 logits (2 x 512 per row) exist only as one factor's 128 accumulator registers per lane; what reaches HBM is the
// bf16 gradient of the logits (the readout's weight gradient and input gradient read it) and three sums.  32 N-block bundles per
// tile; after 16 of them a lane group holds a factor's 512 logits of its 16 rows (column 32 j + 8 g + 4 o + r in lg[2 j + o][r]).
template <int NW = NCW>
__global__ __launch_bounds__(CH_THREADS, 2) void readout_ce_kernel(hma_readout_ce_t p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  HMA_LDS(char)* lds = (HMA_LDS(char)*)smem;
  const uint32_t lds_b = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const tile_map tmap = make_tile_map(p.rows, NW);
  const int nt = tmap.nt;
  {
    HMA_LDS(float)* bl = (HMA_LDS(float)*)(lds + L_BIAS);
    for (int i = tid; i < 1024; i += CH_THREADS) bl[i] = p.bias ? p.bias[i] : 0.f;
  }
  sync_init(lds, tid);
  start_stagger();
  __syncthreads();
  if (ST && wave > NCW) {
    storer_run<NW>(lds, lane, wave - NCW - 1);
    return;
  }
  if (wave >= NW && wave != NCW) return;
  if (wave == NCW) {
    const ring_src ws = {reinterpret_cast<const char*>(p.w.seg[0]), reinterpret_cast<const char*>(p.w.seg[1]),
                         reinterpret_cast<const char*>(p.w.seg[2]), reinterpret_cast<const char*>(p.w.seg[3]),
                         p.w.bundles[0], p.w.bundles[1], p.w.bundles[2], p.w.bundles[3]};
    loader_run<0, NW>(ws, 32, nt, lds_b, lane, nullptr, p.rows, 1, lds);
    return;
  }
  stage_t stg_ = make_stage(lds, wave, lane);
  const int tok = lane & 15, g = lane >> 4;
  constexpr int V = 512;
  const float wgt = p.grad_scale * (p.grad_scale_dev ? *p.grad_scale_dev : 1.0f) / p.stats[2];
  const float eps = p.label_smoothing;
  float loss_acc = 0.f, acc_acc = 0.f;
  HMA_LDS(char)* ring = lds + lane * 16;
  HMA_LDS(char)* bias = lds + L_BIAS + 32 * g;
  const line_offs Ld = make_lines(2048, tok, 16 * g, 64);
  int slot = 0;
  bf16x8_t a[8];
  f32x4v_t lg[32];
  static_assert(32 % PB == 0, "whole barrier groups per tile");
#pragma unroll 1
  for (int tl = 0; tl < nt; ++tl) {
    const int64_t r0 = tile_row0(tmap, tl, NW, wave);
    if (r0 >= tmap.end) {
      if constexpr (ST) {
        steps_skip(stg_, 32);
      } else {
#pragma unroll 1
        for (int s = 0; s < 32; s += PB) CH_BARRIER();
      }
      slot = (slot + 32) % NS;
      continue;
    }
    const int64_t i = r0 + tok, frame = i / p.S;
    const float* xrow = p.x + (frame * p.SA + (i - frame * p.S)) * 256 + 8 * g;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 v0 = *reinterpret_cast<const float4*>(xrow + 32 * j), v1 = *reinterpret_cast<const float4*>(xrow + 32 * j + 4);
      const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      a[j] = as_frag(pack8(v));
    }
    const bool live = (int)(frame % p.T) >= 1 && p.input_ids[i] == p.mask_id;
    const int64_t lab = p.labels[i];
    float row_loss = 0.f;
    bool ok = true;
    char* dl = reinterpret_cast<char*>(p.dlogits) + r0 * 2048;
    static_for<32>([&](auto sc_) __attribute__((always_inline)) {
      constexpr int s = decltype(sc_)::value;
      constexpr int f = s >> 4, j = s & 15;
      if constexpr (ST) step_begin(stg_); else if constexpr (s % PB == 0) CH_BARRIER();
      HMA_LDS(char)* wb = ring + slot * SLOT;
      slot = slot + 1 == NS ? 0 : slot + 1;
      f32x4v_t c0 = lds_f4v(bias + 128 * s), c1 = lds_f4v(bias + 128 * s + 16);
      nb_mma(wb, a, c0, c1);
      step_end(stg_);
      lg[2 * j] = c0;
      lg[2 * j + 1] = c1;
      if constexpr (j == 15) {
        const int target = (int)(f == 0 ? lab % V : (lab / V) % V);
        float m = lg[0][0];
#pragma unroll
        for (int t = 0; t < 32; ++t) m = fmaxf(fmaxf(m, fmaxf(lg[t][0], lg[t][1])), fmaxf(lg[t][2], lg[t][3]));
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        int arg = 1 << 30;
        float se = 0.f, sx = 0.f, xt = 0.f;
        const int tg1 = target - 8 * g;
#pragma unroll
        for (int t = 0; t < 32; ++t) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int colg = 32 * (t >> 1) + 4 * (t & 1) + r;
            const float x = lg[t][r];
            if (x == m) arg = min(arg, colg);  // lowest index wins ties (torch.argmax on the reference path)
            se += __expf(x - m);
            sx += x;
            xt += colg == tg1 ? x : 0.f;
          }
          if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // (left alone the scheduler keeps all 128 exponentials in flight: spills)
        }
        arg += 8 * g;
        arg = min(arg, __shfl_xor(arg, 16, 64));
        arg = min(arg, __shfl_xor(arg, 32, 64));
        se += __shfl_xor(se, 16, 64); se += __shfl_xor(se, 32, 64);
        sx += __shfl_xor(sx, 16, 64); sx += __shfl_xor(sx, 32, 64);
        xt += __shfl_xor(xt, 16, 64); xt += __shfl_xor(xt, 32, 64);
        const float lse = m + __logf(se);
        row_loss += (1.f - eps) * (lse - xt) + eps * (lse - sx * (1.0f / V));
        ok = ok && (arg == target);
        if (p.dlogits) {
          const float w = live ? wgt : 0.f;
          int tg2 = target - 8 * g;
          asm volatile("" : "+v"(tg2));  // (opaque: re-using the 128 compare masks of the loop above costs 256 SGPRs -> spills)
          const float base = -eps * (1.0f / V) * w, hit = -(1.f - eps) * w;
#pragma unroll
          for (int jp = 0; jp < 8; ++jp) {
#pragma unroll
            for (int t = 4 * jp; t < 4 * jp + 4; ++t) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int colg = 32 * (t >> 1) + 4 * (t & 1) + r;
                lg[t][r] = __builtin_fmaf(w, __expf(lg[t][r] - lse), base + (colg == tg2 ? hit : 0.f));
              }
            }
            store_lines(stg_, dl, Ld, 1024 * f + 128 * jp, pack_pair(lg[4 * jp], lg[4 * jp + 1]), pack_pair(lg[4 * jp + 2], lg[4 * jp + 3]));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    });
    if (live && g == 0) {
      loss_acc += row_loss;
      acc_acc += ok ? 1.f : 0.f;
    }
  }
  stage_finish(stg_);
  // one atomic pair per wave
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    loss_acc += __shfl_xor(loss_acc, o, 64);
    acc_acc += __shfl_xor(acc_acc, o, 64);
  }
  if (lane == 0 && acc_acc != 0.f) atomicAdd(p.stats + 1, acc_acc);
  det_loss_add(p.stats, loss_acc, lane, gridDim.x * (unsigned)NW);  // (every compute wave gets here, also one whose rows lie past the matrix)
}

template <auto Kern>
int set_lds(int bytes) {
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return -(int)e;
    done = true;
  }
  return 0;
}

int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}

bool weights_ok(const hma_chain_weights_t& w, int expect) {
  int sum = 0;
  bool ended = false;
  for (int i = 0; i < 4; ++i) {
    if (w.bundles[i] < 0) return false;
    if (w.bundles[i] == 0) { ended = true; continue; }
    if (ended || !w.seg[i]) return false;
    sum += w.bundles[i];
  }
  return sum == expect;
}

// compute waves for a pass of M rows (passes that save nothing): 5 when 7-wave tiles would leave CUs without a tile and 5-wave tiles
// fill more of them; ONE or TWO when even 16- / 32-row tiles do not fill the chip -- a tile's time is its steps (40 / 96 bundles, ~1 100
// cycles each with five waves sharing the LDS, ~450 with one), so a pass of a few hundred rows (the interactive decode: one 320-row
// frame, sim/simulator.py:286-293) is as fast as its tiles are small: 20 workgroups of one compute wave + loader instead of 4 of five
int chain_waves(int64_t M) {
  if (NCW < 6) return NCW;
#ifdef CH_NO5  // (measurement: always full tiles -- fewer CUs, less weight traffic out of L2)
  return NCW;
#endif
  const int cus = num_cus();
#ifndef CH_NO_SMALL
  if (M <= (int64_t)16 * cus) return 1;
  if (M <= (int64_t)32 * cus) return 2;
#endif
  const int64_t tn = (M + 16 * NCW - 1) / (16 * NCW), t5 = (M + 79) / 80;
  return (tn < cus && t5 > tn) ? 5 : NCW;
}

int chain_grid(int64_t M, int nw = NCW) {
  const int64_t ntiles = (M + 16 * nw - 1) / (16 * nw);
#ifdef CH_GRID  // (debug builds: fewer workgroups than CUs)
  return (int)(ntiles < CH_GRID ? ntiles : CH_GRID);
#endif
  const int slots = num_cus() * WGS_PER_CU;
  return (int)(ntiles < slots ? ntiles : slots);
}

}  // namespace

#ifdef CH_PROF
extern "C" int hma_chain_debug_prof(unsigned long long* out128) {
  if (hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_ch_prof), sizeof(unsigned long long) * 128) != hipSuccess) return -1;
  return 0;
}
#endif

extern "C" int hma_chain_pack(void* stream, const float* src, int64_t row_stride, int64_t col_stride, const float* row_scale,
                              const float* col_scale, void* dst, int32_t kind, int32_t rows, int32_t cols, int32_t batch,
                              int64_t src_batch_stride, int64_t dst_batch_stride, int32_t bundle_stride) {
  if (!src || !dst || batch < 1 || bundle_stride < 1) return HMA_EINVAL;
  int nb;
  if (kind == 0) {
    if (cols != 256 || rows <= 0 || rows % 32) return HMA_EINVAL;
    nb = rows / 32;
  } else if (kind == 1) {
    if (rows != 256 || cols <= 0 || cols % 32) return HMA_EINVAL;
    nb = cols / 32;
  } else {
    return HMA_EINVAL;
  }
  hipLaunchKernelGGL(chain_pack_kernel, dim3(nb * 4, batch), dim3(256), 0, (hipStream_t)stream, src, row_stride, col_stride,
                     row_scale, col_scale, reinterpret_cast<uint16_t*>(dst), (int)kind, nb, src_batch_stride, dst_batch_stride,
                     (int)bundle_stride);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_chain_pack_multi(void* stream, const hma_pack_job_t* jobs, int32_t njobs) {
  if (njobs < 0 || (njobs > 0 && !jobs)) return HMA_EINVAL;
  for (int i = 0; i < njobs; ++i) {
    const hma_pack_job_t& j = jobs[i];
    if (!j.src || !j.dst || j.batch < 1 || j.bundle_stride < 1) return HMA_EINVAL;
    if (j.kind == 0 ? (j.cols != 256 || j.rows <= 0 || j.rows % 32) : (j.kind != 1 || j.rows != 256 || j.cols <= 0 || j.cols % 32))
      return HMA_EINVAL;
  }
  for (int i0 = 0; i0 < njobs; i0 += PACK_JOBS) {
    pack_jobs pj;
    pj.n = njobs - i0 < PACK_JOBS ? njobs - i0 : PACK_JOBS;
    int blocks = 0;
    for (int i = 0; i < pj.n; ++i) {
      pj.j[i] = jobs[i0 + i];
      pj.b0[i] = blocks;
      blocks += (pj.j[i].kind == 0 ? pj.j[i].rows : pj.j[i].cols) / 32 * 4 * pj.j[i].batch;
    }
    pj.b0[pj.n] = blocks;
    hipLaunchKernelGGL(chain_pack_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pj);
    HMA_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int hma_chain_a_fwd(void* stream, const hma_chain_a_fwd_t* p) {
  if (!p || !p->o || !p->x || !p->qkv || p->M <= 0 || p->M % 16 || p->ldq < 768) return HMA_EINVAL;
  const bool save = p->x_bf16 != nullptr;  // training: every saved activation, or none of them (inference / decode)
  if (save ? (p->use_mod && (!p->xhat || !p->xm || !p->rstd)) : (p->xhat || p->xm || p->rstd)) return HMA_EINVAL;
  if (!weights_ok(p->w, p->use_mod ? 40 : 32)) return HMA_EINVAL;
  if (p->use_mod && (!p->ss || p->rows_per_frame <= 0 || p->rows_per_frame % 16)) return HMA_EINVAL;
  const int nw = save ? NCW : chain_waves(p->M);
  const int grid = chain_grid(p->M, nw);
#define CH_LAUNCH_A(MOD_, SAVE_, NW_)                                                                                        \
  do {                                                                                                                      \
    if (int rc = set_lds<chain_a_fwd_kernel<MOD_, SAVE_, NW_>>(SMEM)) return rc;                                            \
    hipLaunchKernelGGL((chain_a_fwd_kernel<MOD_, SAVE_, NW_>), dim3(grid), dim3(CH_THREADS), SMEM, (hipStream_t)stream, *p);       \
  } while (0)
  if (p->use_mod) {
    if (save) CH_LAUNCH_A(true, true, NCW);
    else if (nw == 1) CH_LAUNCH_A(true, false, 1); else if (nw == 2) CH_LAUNCH_A(true, false, 2);
    else if (nw == 5) CH_LAUNCH_A(true, false, 5); else CH_LAUNCH_A(true, false, NCW);
  } else {
    if (save) CH_LAUNCH_A(false, true, NCW);
    else if (nw == 1) CH_LAUNCH_A(false, false, 1); else if (nw == 2) CH_LAUNCH_A(false, false, 2);
    else if (nw == 5) CH_LAUNCH_A(false, false, 5); else CH_LAUNCH_A(false, false, NCW);
  }
#undef CH_LAUNCH_A
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_chain_a_bwd(void* stream, const hma_chain_a_bwd_t* p) {
  if (!p || !p->dqkv || !p->dx || !p->dx1_bf16 || !p->d_o || p->M <= 0 || p->M % 16 || p->ldq < 768) return HMA_EINVAL;
  if (!weights_ok(p->w, p->use_mod ? 40 : 32)) return HMA_EINVAL;
  if (p->use_mod && (!p->ss || !p->xhat || !p->rstd || !p->dx2_bf16 || !p->dss || p->rows_per_frame <= 0 || p->rows_per_frame % 16))
    return HMA_EINVAL;
  const int grid = chain_grid(p->M);
  if (p->use_mod) {
    if (int rc = set_lds<chain_a_bwd_kernel<true>>(SMEM)) return rc;
    hipLaunchKernelGGL(chain_a_bwd_kernel<true>, dim3(grid), dim3(CH_THREADS), SMEM, (hipStream_t)stream, *p);
  } else {
    if (int rc = set_lds<chain_a_bwd_kernel<false>>(SMEM)) return rc;
    hipLaunchKernelGGL(chain_a_bwd_kernel<false>, dim3(grid), dim3(CH_THREADS), SMEM, (hipStream_t)stream, *p);
  }
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_chain_s_bwd(void* stream, const hma_chain_s_bwd_t* p) {
  if (!p || !p->dqkv || !p->dx || !p->xhat || !p->rstd || !p->dx_bf16 || p->M <= 0 || p->M % 16 || p->ldq < 768) return HMA_EINVAL;
  if (p->hb_rows < 0 || (p->hb_rows > 0 && (p->hb_rows % 16 || p->M % p->hb_rows || p->ldq != 768 || p->hb_rows > (1 << 20)))) return HMA_EINVAL;
  if (!weights_ok(p->w, 24)) return HMA_EINVAL;
  const int grid = chain_grid(p->M);
  if (int rc = set_lds<chain_s_bwd_kernel>(SMEM)) return rc;
  hipLaunchKernelGGL(chain_s_bwd_kernel, dim3(grid), dim3(CH_THREADS), SMEM, (hipStream_t)stream, *p);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_chain_t_bwd(void* stream, const hma_chain_t_bwd_t* p) {
  if (!p || !p->dy_bf16 || !p->qkv || !p->dqkv || p->B <= 0 || p->SA <= 0 || p->T < 1 || p->T > 16) return HMA_EINVAL;
  if ((int64_t)p->SA * 1536 * 15 >= (int64_t)1 << 31) return HMA_EINVAL;  // (line offsets are 32-bit)
  if (!weights_ok(p->w, 8)) return HMA_EINVAL;
  const int64_t cols = p->B * (int64_t)p->SA;
  const int64_t ntiles = (cols + NCW - 1) / NCW;
  const int slots = num_cus() * WGS_PER_CU;
  const int grid = (int)(ntiles < slots ? ntiles : slots);
  if (p->T == 16) {
    if (int rc = set_lds<chain_t_bwd_kernel<false>>(TB_SMEM)) return rc;
    hipLaunchKernelGGL(chain_t_bwd_kernel<false>, dim3(grid), dim3(CH_THREADS), TB_SMEM, (hipStream_t)stream, *p);
  } else {
    if (int rc = set_lds<chain_t_bwd_kernel<true>>(TB_SMEM)) return rc;
    hipLaunchKernelGGL(chain_t_bwd_kernel<true>, dim3(grid), dim3(CH_THREADS), TB_SMEM, (hipStream_t)stream, *p);
  }
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_chain_ab_fwd(void* stream, const hma_chain_ab_fwd_t* p) {
  if (!p || !p->o_s || !p->x || !p->b1 || p->B <= 0 || p->SA <= 0 || p->T != 16) return HMA_EINVAL;
  const bool mod = p->bundles[1] != 0;
  if (mod && (!p->ss || !p->xhat_m || !p->xm || !p->rstd_m)) return HMA_EINVAL;
  if (!p->x2b || !p->qkv_t || !p->o_t || !p->xhat2 || !p->rstd2) return HMA_EINVAL;
  const bool qkv = p->qkv_s != nullptr;
  if (qkv && (!p->xhat1n || !p->rstd1n)) return HMA_EINVAL;
  const bool drop = p->drop_p > 0.f;
  if (drop && (!p->drop_seed || p->drop_p >= 1.f || !mod)) return HMA_EINVAL;
  if ((int64_t)p->SA * 1536 * 15 >= (int64_t)1 << 31) return HMA_EINVAL;  // (line offsets are 32-bit)
  const int expect[6] = {8, mod ? 8 : 0, 24, 8, 64, qkv ? 24 : 0};
  for (int i = 0; i < 6; ++i)
    if (p->bundles[i] != expect[i] || (expect[i] > 0 && !p->seg[i])) return HMA_EINVAL;
  const int64_t cols = p->B * (int64_t)p->SA;
  const int64_t ntiles = (cols + NCW - 1) / NCW;
  const int slots = num_cus() * WGS_PER_CU;
  const int grid = (int)(ntiles < slots ? ntiles : slots);
#define CH_LAUNCH_AB(QKV_, MOD_, DROP_)                                                                                          \
  do {                                                                                                                          \
    if (int rc = set_lds<chain_ab_fwd_kernel<QKV_, MOD_, DROP_>>(AB_SMEM)) return rc;                                           \
    hipLaunchKernelGGL((chain_ab_fwd_kernel<QKV_, MOD_, DROP_>), dim3(grid), dim3(CH_THREADS), AB_SMEM, (hipStream_t)stream, *p); \
  } while (0)
  if (drop) {
    if (qkv) CH_LAUNCH_AB(true, true, true); else CH_LAUNCH_AB(false, true, true);
  } else if (mod) {
    if (qkv) CH_LAUNCH_AB(true, true, false); else CH_LAUNCH_AB(false, true, false);
  } else {
    if (qkv) CH_LAUNCH_AB(true, false, false); else CH_LAUNCH_AB(false, false, false);
  }
#undef CH_LAUNCH_AB
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_chain_b_fwd(void* stream, const hma_chain_b_fwd_t* p) {
  if (!p || !p->o || !p->x || !p->b1 || p->M <= 0 || p->M % 16) return HMA_EINVAL;
  const bool qkv = p->qkv != nullptr;
  if (qkv && p->ldq < 768) return HMA_EINVAL;
  if (!weights_ok(p->w, qkv ? 96 : 72)) return HMA_EINVAL;
  const bool save = p->xhat2 != nullptr;  // training: the two LayerNorm outputs are saved for the backward
  if (save && (!p->rstd2 || (qkv && (!p->xhat1n || !p->rstd1n)))) return HMA_EINVAL;
  const bool drop = p->drop_p != 0.f;
  if (drop && (!(p->drop_p > 0.f && p->drop_p < 1.f) || !p->drop_seed || !save)) return HMA_EINVAL;
  const int nw = save ? NCW : chain_waves(p->M);
  const int grid = chain_grid(p->M, nw);
  // (the bias area holds 2304 floats here: it runs into the shift / scale rows' space, which this chain does not use)
#define CH_LAUNCH_B(QKV_, NW_)                                                                                          \
  do {                                                                                                                 \
    if (int rc = set_lds<chain_b_fwd_kernel<QKV_, NW_>>(SMEM)) return rc;                                              \
    hipLaunchKernelGGL((chain_b_fwd_kernel<QKV_, NW_>), dim3(grid), dim3(CH_THREADS), SMEM, (hipStream_t)stream, *p);         \
  } while (0)
  if (drop) {
    if (qkv) {
      if (int rc = set_lds<chain_b_fwd_kernel<true, NCW, true, true>>(SMEM)) return rc;
      hipLaunchKernelGGL((chain_b_fwd_kernel<true, NCW, true, true>), dim3(grid), dim3(CH_THREADS), SMEM, (hipStream_t)stream, *p);
    } else {
      if (int rc = set_lds<chain_b_fwd_kernel<false, NCW, true, true>>(SMEM)) return rc;
      hipLaunchKernelGGL((chain_b_fwd_kernel<false, NCW, true, true>), dim3(grid), dim3(CH_THREADS), SMEM, (hipStream_t)stream, *p);
    }
  } else if (save) {
    if (qkv) {
      if (int rc = set_lds<chain_b_fwd_kernel<true, NCW, true>>(SMEM)) return rc;
      hipLaunchKernelGGL((chain_b_fwd_kernel<true, NCW, true>), dim3(grid), dim3(CH_THREADS), SMEM, (hipStream_t)stream, *p);
    } else {
      if (int rc = set_lds<chain_b_fwd_kernel<false, NCW, true>>(SMEM)) return rc;
      hipLaunchKernelGGL((chain_b_fwd_kernel<false, NCW, true>), dim3(grid), dim3(CH_THREADS), SMEM, (hipStream_t)stream, *p);
    }
  } else if (qkv) {
    if (nw == 1) CH_LAUNCH_B(true, 1); else if (nw == 2) CH_LAUNCH_B(true, 2); else if (nw == 5) CH_LAUNCH_B(true, 5); else CH_LAUNCH_B(true, NCW);
  } else {
    if (nw == 1) CH_LAUNCH_B(false, 1); else if (nw == 2) CH_LAUNCH_B(false, 2); else if (nw == 5) CH_LAUNCH_B(false, 5); else CH_LAUNCH_B(false, NCW);
  }
#undef CH_LAUNCH_B
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_readout_ce(void* stream, const hma_readout_ce_t* p) {
  if (!p || !p->x || !p->input_ids || !p->labels || !p->stats || p->rows <= 0 || p->rows % 16 || p->S <= 0 || p->SA < p->S || p->T <= 0)
    return HMA_EINVAL;
  if (!weights_ok(p->w, 32)) return HMA_EINVAL;
  const int grid = chain_grid(p->rows, NCW);
  if (int rc = set_lds<readout_ce_kernel<NCW>>(SMEM)) return rc;
  hipLaunchKernelGGL((readout_ce_kernel<NCW>), dim3(grid), dim3(CH_THREADS), SMEM, (hipStream_t)stream, *p);
  HMA_CHECK_LAUNCH();
  return 0;
}

extern "C" int hma_zero_f32(void* stream, float* p, int64_t n) {
  if (!p || n < 0) return HMA_EINVAL;
  if (n == 0) return 0;
  hipError_t e = hipMemsetAsync(p, 0, (size_t)n * 4, (hipStream_t)stream);
  return e == hipSuccess ? 0 : -(int)e;
}
