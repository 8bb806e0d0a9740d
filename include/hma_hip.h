/*
 * hma_hip.h -- C ABI of libhma_hip.so: the MI355X (gfx950) kernels of the HMA hot path.
 *
 * Drop-in boundary.  The reference (liruiw/HMA) is pure PyTorch: what it
 * binds for this path are ATen / xformers operators called from the hma/model Python files.  Each entry point
 * below names the reference call sites (file:line under /root/reference) whose arithmetic it
 * replaces.  Conventions:
 *   - plain pointers into device memory + sizes; no torch types; nothing is allocated or retained;
 *   - every function only ENQUEUES work on `stream` (a hipStream_t passed as void*) and returns
 *     0, or a negative hipError_t / HMA_E* code; it never synchronises;
 *   - activations are row-major "token grids": rows = (b, t, s) with s fastest, d_model = 256
 *     columns; bf16 means the 16 high bits of an IEEE float32 (round-to-nearest-even);
 *   - re-entrant and stateless: safe from the autograd thread and from hipGraph capture.
 */
#ifndef HMA_HIP_H
#define HMA_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMA_EINVAL (-10001) /* unsupported shape / null pointer */

/* A-operand element type / prologue and epilogue selectors of the two GEMM entry points */
enum { HMA_A_BF16 = 0, HMA_A_F32 = 1, HMA_A_BF16_AFFINE = 2,
       /* hma_gemm_tn / hma_gemm_tn_pair operands only (workspace path, ld == the matrix width, batch 1): bf16 [M, ld] in the order
        * hma_mlp_bwd stores gelu(u) / dL/du -- element (m, c) at ((((m / 128) (ld / 32) + c / 32) 4 + (m / 32) % 4) 1024 +
        * ((c / 8) % 2) 512 + (((c / 16) % 2) 32 + m % 32) 8 + c % 8: every (128-row tile, 32-column block, 32-row group) is 2 KB that
        * one wave writes with two contiguous 1 KB store instructions.  Allocate ceil(M / 128) 128 rows. */
       HMA_A_BF16_FRAG32 = 3,
       /* hma_gemm_tn's dY / A on its workspace (LDS-DMA) path only: a [M, 256 W] bf16 matrix (W = ld / 256) in the HEAD-BLOCKED order
        * of the spatial attention -- element (m, c) at (((frame 8 + head) W + c / 256) n + m % n) 32 + c % 32 with frame = m / n,
        * head = (c % 256) / 32, n = rows per frame passed in y_group_rows / a_group_rows (a multiple of 32 that divides M): every
        * (frame, head, q | k | v part) is one contiguous [n][32] block, so the attention backward writes whole 2 KB tiles
        * (hma_attn_spatial_bwd_blocked) and the consumers' 64-byte row pieces of one head are adjacent. */
       HMA_A_BF16_HEADBLK = 4 };
enum {
  HMA_EPI_BF16 = 0,      /* C(bf16)  = acc (+bias)                                            */
  HMA_EPI_F32 = 1,       /* C(f32)   = acc (+bias)                                            */
  HMA_EPI_RESID = 2,     /* C(f32)  += acc (+bias); optional C2(bf16) = new C                 */
  HMA_EPI_GELU2 = 3,     /* C(bf16)  = u = acc+bias ; C2(bf16) = gelu(u)                      */
  HMA_EPI_SILU2 = 4,     /* C(bf16)  = u = acc+bias ; C2(bf16) = silu(u)                      */
  HMA_EPI_DGELU = 5,     /* C(bf16)  = acc * gelu'(U)   (U bf16, may alias C)                 */
  HMA_EPI_DSILU = 6,     /* C(bf16)  = acc * silu'(U)                                         */
  HMA_EPI_ATOMIC_F32 = 7 /* atomicAdd(C(f32), acc)                                            */
};

/* C[m, n] = sum_k A'[m, k] * W[n, k]  -- nn.Linear forward / input-gradient.
 * Replaces: nn.Linear in attention.py:28,30,39,60; st_transformer.py:19-21,25-26;
 * st_mask_git.py:60-63,75,784-789 and their autograd mirrors.
 * A' = A (bf16), float->bf16 of A (f32), or bf16(xhat * gamma[k] + beta[k]) (LayerNorm affine,
 * st_transformer.py:50,75).  Row remap: logical row r reads/writes physical row
 * (r / group_rows) * group_stride + r % group_rows when group_rows > 0 (slices the image tokens
 * out of the (S + 64)-token frames, st_mask_git.py:681).  N % 128 == 0, K % 64 == 0. */
typedef struct {
  const void* A; int64_t lda; int32_t a_kind; int32_t _pad0;
  int64_t a_group_rows, a_group_stride;
  const float* gamma; const float* beta;
  const void* W; int64_t ldw;
  int64_t M, N, K;
  int32_t epi; int32_t _pad1;
  const float* bias;
  void* C; int64_t ldc; int64_t c_group_rows, c_group_stride;
  void* C2; int64_t ldc2;
  const void* U; int64_t ldu;
  /* batching over blockIdx.z (per-layer adaLN stacks): element strides, 0 = shared */
  int32_t batch; int32_t _pad2;
  int64_t sA, sW, sBias, sC, sC2, sU;
  /* Optional fused LayerNorm of the updated residual rows (HMA_EPI_RESID with N = K = 256, batch <= 1 only;
   * anything else with ln_xhat set is HMA_EINVAL): after C += A' W^T + bias the kernel also writes
   *   ln_xhat = LN(C row, ln_eps, no affine) (bf16, leading dimension 256, row index = C row) and ln_rstd,
   * i.e. what hma_ln_fwd would produce from the new residual (st_transformer.py:112 norm2), and, when ln_ss is
   * set, ln_xm = ln_xhat * (1 + scale[f]) + shift[f] with f = row / ln_rows_per_frame and ln_ss = [shift | scale]
   * (fp32, 512 per frame), i.e. hma_modln_fwd (st_mask_git.py:71-74). */
  void* ln_xhat; float* ln_rstd; const float* ln_ss; void* ln_xm;
  float ln_eps; int32_t ln_rows_per_frame;
  /* Optional inverted dropout (nn.Dropout in Mlp, st_transformer.py:24-27; training only), drop_p in (0, 1):
   *   HMA_EPI_GELU2: C2 = GELU(u) * keep / (1 - p)         (C, the saved pre-activation, is not dropped)
   *   HMA_EPI_DGELU: C  = acc * GELU'(U) * keep / (1 - p)   (the same mask: same seed, salt and element index)
   *   HMA_EPI_RESID: C += (acc + bias) * keep / (1 - p)
   * keep = 16 bits of hash(*drop_seed, drop_salt, (row * ldc + column) / 2) >= p * 2^16 (counter-based: one 32-bit hash decides an
   * element pair; hma_dropout_bf16, chain B and hma_mlp_bwd use the same function), so backward regenerates the mask instead of
   * storing it.  drop_seed is a DEVICE pointer (one uint32 the
   * host bumps per step: recorded launches and captured graphs stay valid).  Supported on the streaming K = 256 kernel
   * (GELU2 / DGELU) and the persistent kernel (RESID); otherwise HMA_EINVAL. */
  float drop_p; int32_t drop_salt; const uint32_t* drop_seed;
} hma_gemm_nt_t;
int hma_gemm_nt(void* stream, const hma_gemm_nt_t* p);

/* dW[n, k] += sum_m dY[m, n] * A'[m, k] ; dBias[n] += sum_m dY[m, n]  (fp32 atomics) -- the
 * weight/bias gradient of nn.Linear (autograd mirror of the call sites above).
 * N % 128 == 0, K % 128 == 0.  `splits` partitions M over blocks. */
typedef struct {
  const void* dY; int64_t ldy; int32_t y_kind; int32_t _pad0;   /* HMA_A_BF16 | HMA_A_F32 */
  int64_t y_group_rows, y_group_stride;
  const void* A; int64_t lda; int32_t a_kind; int32_t _pad1;
  int64_t a_group_rows, a_group_stride;
  const float* gamma; const float* beta;
  int64_t M, N, K;
  float* dW; int64_t lddw; float* dBias;
  int32_t splits; int32_t batch;
  int64_t sY, sA, sdW, sdBias;
  /* optional scratch for a two-stage (atomic-free) reduction of the M-splits: >= 256 * 65536 floats
   * covers every shape of this model; NULL or too small -> fp32 atomics */
  float* ws; int64_t ws_elems;
  /* optional, a_kind == HMA_A_BF16_AFFINE on the workspace (LDS-DMA) path only: the gradients of the LayerNorm affine that
   * was folded into this Linear's forward weights (hma_fold_ln_bf16), taken from the un-scaled product P = dY^T xhat
   * the reduction holds anyway: dgamma[k] += sum_n w_master[n][k] P[n][k], dbeta[k] += sum_n w_master[n][k] colsum(dY)[n]
   * (w_master = the Linear's fp32 weight, leading dimension lddw).  Replaces the dgamma / dbeta outputs of hma_ln_bwd
   * when the LayerNorm backward itself is fused into another kernel (hma_mlp_bwd). */
  const float* w_master; float* dgamma; float* dbeta;
} hma_gemm_tn_t;
int hma_gemm_tn(void* stream, const hma_gemm_tn_t* p);
/* Two independent weight gradients in ONE launch (e.g. a block's fc2 and fc1 after `loss.backward()` reaches them,
 * st_transformer.py:24-27): the M-splits of both share the 256 workgroups, which halves the partial-sum traffic of the
 * two-stage reduction per problem.  Both must name the same workspace (>= 256 * 65792 floats); any pair that is not
 * eligible for the LDS-DMA kernel is executed as two hma_gemm_tn calls.  Same results as two calls. */
int hma_gemm_tn_pair(void* stream, const hma_gemm_tn_t* a, const hma_gemm_tn_t* b);
/* Up to 16 independent weight gradients in ONE launch -- e.g. the seven of an STBlock (or the fourteen of two) at the end of its backward (the MLP's two, the two
 * attentions' projection + qkv pairs, the ModulateLayer's linear_out; st_transformer.py:85-112, st_mask_git.py:66-76): nothing
 * downstream reads a weight gradient before the optimizer, so the launches can wait until every operand of the block exists.  The
 * 256 workgroups are divided in proportion to the operand bytes: every problem is split over fewer M-slices (32 MB of bf16 partials
 * per LAUNCH, written and re-read, instead of per problem pair), one reduction launch instead of four per block.  All problems must
 * name the same workspace (>= 256 * 65792 floats); a set that is not eligible for the LDS-DMA kernel is executed as n hma_gemm_tn
 * calls.  Same results as n calls (the M-slices of a problem differ: fp32 summation order). */
int hma_gemm_tn_multi(void* stream, const hma_gemm_tn_t* const* probs, int32_t n);

/* LayerNorm over d_model = 256 without the affine (applied by the consumer GEMM's prologue):
 * xhat = (x - mean) * rstd (bf16), rstd saved.  st_transformer.py:50,75,86,112 (eps 1e-5). */
int hma_ln_fwd(void* stream, const float* x, void* xhat, float* rstd, int64_t rows, float eps);
/* dx += LN-backward(dxn * gamma); dgamma += sum dxn*xhat; dbeta += sum dxn  (gamma may be NULL:
 * no affine, then dgamma/dbeta are untouched).  dx_bf16 (optional, may be NULL): bf16 copy of the updated dx,
 * the operand the next linear's backward GEMMs read (autocast hands them grad_output in bf16). */
int hma_ln_bwd(void* stream, const void* dxn, const void* xhat, const float* rstd, const float* gamma,
               float* dx, float* dgamma, float* dbeta, int64_t rows, void* dx_bf16);
/* qk_norm=True (attention.py:31-35,44-48): per-head LayerNorm (32 values, shared affine gamma / beta [32]) of the q and k parts of a
 * packed qkv buffer [*, ld] (q | k | v), IN PLACE, one pass over `rows` token rows; raw (bf16 [rows, 512], may be NULL) receives the
 * pre-norm q | k for the backward.  Row r sits at (r / g_rows) * g_stride + r % g_rows (g_rows <= 0: r) -- the decode cache's frames. */
int hma_qknorm_fwd(void* stream, void* qkv, int64_t ld, void* raw, const float* gamma, const float* beta, float eps, int64_t rows,
                   int64_t g_rows, int64_t g_stride);
/* Backward: dqkv's q | k parts (gradient wrt the normalised values) become the gradient wrt the raw ones, in place; dgamma / dbeta [32]
 * are ADDED (fp32 atomics). */
int hma_qknorm_bwd(void* stream, void* dqkv, int64_t ld, const void* raw, const float* gamma, float eps, float* dgamma, float* dbeta,
                   int64_t rows);

/* ModulateLayer prologue, st_mask_git.py:71-74: xhat = LN(x, eps 1e-6, no affine);
 * xm = xhat * (1 + scale[bt]) + shift[bt], ss = [shift | scale] (fp32, 512 per (b,t)). */
int hma_modln_fwd(void* stream, const float* x, const float* ss, void* xhat, void* xm, float* rstd,
                  int64_t frames, int64_t rows_per_frame, float eps);
/* backward of the above: dss[bt] = [sum_s dxm | sum_s dxm*xhat]; dx += LN-backward(dxm*(1+scale)) */
int hma_modln_bwd(void* stream, const void* dxm, const void* xhat, const float* rstd, const float* ss,
                  float* dx, float* dss, int64_t frames, int64_t rows_per_frame, void* dx_bf16 /* optional, as hma_ln_bwd */);

/* Bidirectional spatial self-attention on packed qkv rows, attention.py:37-61 with causal=False:
 * qkv (bf16) [frames * n, 3 * 256] = [q | k | v], heads of 32; o (bf16) [frames * n, 256];
 * lse (fp32) [frames * n, 8] = log-sum-exp of the scaled scores (saved for backward). n % 32 == 0,
 * n <= 320. */
int hma_attn_spatial_fwd(void* stream, const void* qkv, void* o, float* lse, int64_t frames, int32_t n,
                         float scale);
/* dqkv (bf16, same layout as qkv) from do (bf16); delta ([frames*n, 8] fp32) receives rowsum(dO * O) per head.  n = 256 / 320: ONE
 * kernel per launch (q / k / v / dO of a (frame, head) staged in LDS once, dS shared through LDS); n = 64: the dq + dkv pair. */
int hma_attn_spatial_bwd(void* stream, const void* qkv, const void* o, const void* d_o, const float* lse,
                         float* delta, void* dqkv, int64_t frames, int32_t n, float scale);
/* The same backward with dqkv written in the HEAD-BLOCKED order (HMA_A_BF16_HEADBLK above, W = 3, rows per frame n): every gradient
 * tile leaves as 2 KB of contiguous memory.  Consumers: hma_gemm_tn (y_kind = HMA_A_BF16_HEADBLK) and hma_chain_s_bwd (hb_rows). */
int hma_attn_spatial_bwd_blocked(void* stream, const void* qkv, const void* o, const void* d_o, const float* lse,
                                 float* delta, void* dqkv, int64_t frames, int32_t n, float scale);
/* Causal temporal self-attention over the T frames of each (b, s) column, attention.py:37-61 with
 * causal=True as called from st_transformer.py:111; rows are (b, t, s): stride between frames is
 * n_s rows.  T <= 16. */
int hma_attn_temporal_fwd(void* stream, const void* qkv, void* o, int64_t batch, int32_t T, int32_t n_s,
                          float scale);
int hma_attn_temporal_bwd(void* stream, const void* qkv, const void* o, const void* d_o, void* dqkv,
                          int64_t batch, int32_t T, int32_t n_s, float scale);
/* Temporal attention against a per-layer qkv cache laid out [batch, T_cache, n_s, 768] (incremental
 * decode; exact because frame t depends on frames <= t only, st_transformer.py:83-111).
 * t_query < 0: prefill -- frames 0..T-1 attend causally, o rows (b, t, s) with T frames per sample.
 * t_query >= 0: only frame t_query is new -- o holds that frame, rows (b, s). */
int hma_attn_temporal_cached(void* stream, const void* qkv_cache, void* o, int64_t batch, int32_t T,
                             int32_t t_query, int32_t T_cache, int32_t n_s, float scale);

/* Fused token + action + positional embedding, factorization_utils.py:31-54 +
 * st_mask_git.py:640-672: x[b,t,s,:] = (id == mask_id ? mask_embed : E0[id % V] + E1[id / V]) +
 * pos[t,s,:] for s < S; = a_emb[b,t,:] + pos[t,s,:] for S <= s < S + A (A = 0 when a_emb is NULL).
 * ids int64 [B,T,S]; pos has row stride pos_frame_rows per frame. */
int hma_embed_fwd(void* stream, const int64_t* ids, const float* E0, const float* E1, const float* mask_embed,
                  const float* pos, const float* a_emb, float* x, int64_t B, int32_t T, int32_t S, int32_t A,
                  int32_t pos_frame_rows, int32_t V, int64_t mask_id);
int hma_embed_bwd(void* stream, const int64_t* ids, const float* dx, float* dE0, float* dE1, float* dmask_embed,
                  float* dpos, float* da_emb, int64_t B, int32_t T, int32_t S, int32_t A,
                  int32_t pos_frame_rows, int32_t V, int64_t mask_id);

/* ActionStat + BasicMLP, st_mask_git.py:134-138, 90-102 (fp32): rows = B*T.
 * Saves an (normalised input), xhat, rstd, h for backward. */
int hma_action_stem_fwd(void* stream, const float* a, const float* mean, const float* std, int32_t action_dim,
                        const float* W1, const float* b1, const float* ln_w, const float* ln_b,
                        const float* W2, const float* b2, float* an, float* xhat, float* rstd, float* h,
                        float* out, int64_t rows, int32_t d_a, int32_t skip_norm);
int hma_action_stem_bwd(void* stream, const float* dout, const float* an, const float* xhat, const float* rstd,
                        const float* h, const float* ln_w, const float* W2, float* dW1, float* db1,
                        float* dln_w, float* dln_b, float* dW2, float* db2, float* scratch,
                        int64_t rows, int32_t d_a);

/* Factorised-vocabulary cross-entropy + accuracy + logits gradient, st_mask_git.py:603-630, 714-716.
 * logits f32 [B*T*S, 2*512] (row = (b,t,s)); frames t >= 1 only; label smoothing 0.01; masked mean.
 * stats: fp32[HMA_CE_STATS_FLOATS], zeroed by the caller first.  stats[0] = sum loss*mask, stats[1] = sum acc*mask,
 * stats[2] = num masked; [3..5] are the kernels' own (a ticket counter and a 64-bit fixed-point accumulator: the loss
 * partials are added as integers and the last wave to arrive writes stats[0], so the sum does not depend on the order
 * the waves finish in -- two replays of one graph report the same loss bit for bit); [6..7] spare.
 * dlogits (bf16, may be NULL) = grad_scale * (*grad_scale_dev if not NULL) * mask / num_masked *
 * (softmax - smoothed one-hot): needs the count
 * first, so call hma_count_masked before it. */
#define HMA_CE_STATS_FLOATS 8
int hma_count_masked(void* stream, const int64_t* input_ids, float* stats, int64_t B, int32_t T, int32_t S,
                     int64_t mask_id);
int hma_ce_fwd_bwd(void* stream, const float* logits, const int64_t* input_ids, const int64_t* labels,
                   float* stats, void* dlogits, const float* grad_scale_dev, float grad_scale, int64_t B,
                   int32_t T, int32_t S, int64_t mask_id, float label_smoothing);

/* One MaskGIT sampling step on frame logits, st_mask_git.py:397-453 (temperature <= 1e-8):
 * logits f32 [B, T, S, 1024] frame out_t; writes samples into prompt[b, out_t, :] (int64 [B,T,S]),
 * updates unmasked (uint8 [B,S]).  conf_override (f32 [B,S]) replaces the confidences when not
 * NULL ("random" unmask mode, the torch.rand_like draw of :435); conf_out (f32 [B,S], may be NULL)
 * receives the model confidences.  n_mask = tokens to re-mask (ignored when last != 0). S <= 256.
 * logits_T / logits_t: frame count / index of the logits buffer when it differs from the prompt's
 * (incremental decode keeps one frame of logits); logits_T <= 0 means (T, out_t). */
int hma_maskgit_step(void* stream, const float* logits, int64_t* prompt, uint8_t* unmasked,
                     const float* conf_override, float* conf_out, int64_t B, int32_t T, int32_t S,
                     int32_t out_t, int32_t n_mask, int32_t last, int64_t mask_id, int32_t logits_T,
                     int32_t logits_t);
/* The same step with categorical sampling (temperature > 1e-8, st_mask_git.py:411-416:
 * Categorical(probs = softmax / temperature).sample(); the temperature cancels in Categorical's normalisation).
 * sample_noise f32 [B, S, 2, 512] holds the Exp(1) draws q of torch.multinomial's single-sample path
 * (sample = argmax_k p_k / q_k) for factor v at [.., v, :]; injecting them makes a run replayable.  The confidence of a
 * token is the product of the sampled entries' probabilities (:420). */
int hma_maskgit_step_sampled(void* stream, const float* logits, int64_t* prompt, uint8_t* unmasked,
                             const float* conf_override, float* conf_out, const float* sample_noise,
                             int64_t B, int32_t T, int32_t S, int32_t out_t, int32_t n_mask, int32_t last,
                             int64_t mask_id, int32_t logits_T, int32_t logits_t);
/* Either of the two steps above as TWO launches when the caller owns scratch: one wave per token over the whole chip writes the
 * sampled ids (samp_scratch, int32 [B, S]) and confidences (conf_out, f32 [B, S], required), then one workgroup per sample ranks
 * and updates.  Same results; for small B (a decode step has 64 samples) the one-launch form leaves most CUs idle.
 * sample_noise NULL = greedy. */
int hma_maskgit_step_wide(void* stream, const float* logits, int64_t* prompt, uint8_t* unmasked,
                          const float* conf_override, float* conf_out, int32_t* samp_scratch, const float* sample_noise,
                          int64_t B, int32_t T, int32_t S, int32_t out_t, int32_t n_mask, int32_t last, int64_t mask_id,
                          int32_t logits_T, int32_t logits_t);

/* sum of squares of g[0:n) accumulated into *out (fp32 atomic; zero it first) -- clip_grad_norm_,
 * train_multi.py:594 */
int hma_sqnorm(void* stream, const float* g, int64_t n, float* out);
/* AdamW on a flat range with the clip coefficient min(1, max_norm / (sqrt(*sqnorm) + 1e-6)) read
 * on device (sqnorm NULL or max_norm <= 0: no clip); also emits the bf16 copy of the new weights.
 * A non-finite *sqnorm skips the update (weights, moments untouched).
 * torch.optim.AdamW semantics (decoupled decay, bias correction), train_multi.py:900-922.
 * flags (may be NULL) holds one byte per 64 elements of the range (which must start on a multiple
 * of 64): 0 = frozen, 1 = update without weight decay ("bias" parameters, :907-918), 2 = decay. */
int hma_adamw(void* stream, float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n,
              float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
              const float* sqnorm, float max_norm, const uint8_t* flags);
/* The same update with the range's update count kept ON THE DEVICE: step_pair[parity] holds the number of updates the
 * range has received, the launch uses step = that + 1 and writes the new count to step_pair[parity ^ 1] (the caller
 * alternates `parity` per call on a range).  When *sqnorm is not finite -- a NaN / Inf loss on ANY rank reaches every
 * rank's norm through the gradient all-reduce -- nothing is written and the count does not advance: the
 * all-rank-consistent form of the reference's non-finite-loss skip (train_multi.py:572-583), with no host sync. */
int hma_adamw_counted(void* stream, float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n,
                      float lr, float beta1, float beta2, float eps, float weight_decay, int32_t* step_pair,
                      int32_t parity, const float* sqnorm, float max_norm, const uint8_t* flags);
/* dst(bf16) = src(f32) for n elements */
int hma_cast_bf16(void* stream, const float* src, void* dst, int64_t n);

/* ---- Diffusion head (DiffLoss / SimpleMLPAdaLN, hma/model/diffloss.py; hma/diffusion/gaussian_diffusion.py) --------
 * The head's Linear layers run through hma_gemm_nt / hma_gemm_tn; these are the row kernels around them.  Hidden
 * width W: multiple of 256, <= 2048.  `mod` tensors are the bf16 outputs of the adaLN_modulation Linears
 * ([rows, ldm], shift | scale | gate at the given column offsets).  Schedule tables are fp32 device arrays.
 *
 * hma_diff_prepare: x_t = sqrt_ac[t] x0 + sqrt_1mac[t] noise (gaussian_diffusion.py:203-219; noise NULL: x_t = x0, the
 *   sampler's input), also as bf16 zero-padded to `pad` columns (the input_proj GEMM operand), and the sinusoidal
 *   timestep embedding of tmap[t] (tmap NULL: t) as bf16 [rows, 256] (diffloss.py:79-90). */
int hma_diff_prepare(void* stream, const float* x0, const float* noise, const int64_t* t, const float* sqrt_ac,
                     const float* sqrt_1mac, const int32_t* tmap, float* xt, void* xt_pad, void* tfreq, int64_t n,
                     int32_t C, int32_t pad);
/* sy = bf16(SiLU(y)) -- the nn.SiLU in front of every adaLN_modulation (diffloss.py:120,142); dy = dsy * SiLU'(y) */
int hma_silu_cast(void* stream, const float* y, void* sy, int64_t n);
int hma_silu_bwd(void* stream, const float* y, const float* dsy, float* dy, int64_t n);
/* out = bf16(LN(x; gamma, beta | no affine, eps) * (1 + scale) + shift)   ResBlock / FinalLayer, diffloss.py:116-124,140-149.
 * backward: dx += LN-backward, dmod[shift] = dout, dmod[scale] = dout * LN(x), dgamma / dbeta += (atomics). */
int hma_adaln_fwd(void* stream, const float* x, const void* mod, int64_t ldm, int32_t off_shift, int32_t off_scale,
                  const float* gamma, const float* beta, float eps, void* out, int64_t n, int32_t W);
int hma_adaln_bwd(void* stream, const void* dout, const float* x, const void* mod, int64_t ldm, int32_t off_shift,
                  int32_t off_scale, const float* gamma, const float* beta, float eps, float* dx, void* dmod,
                  float* dgamma, float* dbeta, int64_t n, int32_t W);
/* the same with `accumulate` = 0: dx is WRITTEN (the first producer of a gradient buffer: no zero-fill and no read of it); 1 = as above */
int hma_adaln_bwd_acc(void* stream, const void* dout, const float* x, const void* mod, int64_t ldm, int32_t off_shift,
                  int32_t off_scale, const float* gamma, const float* beta, float eps, float* dx, void* dmod,
                  float* dgamma, float* dbeta, int64_t n, int32_t W, int32_t accumulate);
/* x += gate * h (diffloss.py:124); backward: dh = dx * gate, dmod[gate] = dx * h (dx is also the identity branch's grad) */
int hma_gate_fwd(void* stream, float* x, const void* mod, int64_t ldm, int32_t off_gate, const void* h, int64_t n, int32_t W);
int hma_gate_bwd(void* stream, const float* dx, const void* mod, int64_t ldm, int32_t off_gate, const void* h, void* dh,
                 void* dmod, int64_t n, int32_t W);
/* Per-row training loss mean_c (noise - eps)^2 + vb (LossType.MSE + ModelVarType.LEARNED_RANGE with the mean frozen:
 * KL to the true posterior, or the discretised-Gaussian decoder NLL at t == 0; gaussian_diffusion.py:650-745) from the
 * network output out [rows, ldo] = [eps | v | padding]; stats (fp32[HMA_CE_STATS_FLOATS], zeroed first; layout and the
 * order-independent sum as for hma_ce_fwd_bwd): stats[0] = sum_rows loss * mask; rows_out (optional) = per-row
 * loss; dout (optional, [rows, ldo], only the 2C used columns are written) = d loss / d out * grad_scale * mask / *denom.
 * tables6 = [sqrt_recip_ac | sqrt_recipm1_ac | posterior_mean_coef1 | coef2 | posterior_log_variance_clipped |
 * log(betas)], n_steps each. */
int hma_diff_loss(void* stream, const float* out, int64_t ldo, const float* x0, const float* xt, const float* noise,
                  const int64_t* t, const float* tables6, int32_t n_steps, const float* mask, const float* denom,
                  float grad_scale, float* stats, float* rows_out, float* dout, int64_t n, int32_t C);
/* One reverse step x <- mean(x, out) + [step != 0] exp(logvar / 2) noise temperature  (p_sample, :358-394) */
int hma_diff_p_sample(void* stream, const float* out, int64_t ldo, float* x, const float* noise, const float* tables6,
                      int32_t n_steps, int32_t step, float temperature, int32_t clip_denoised, int64_t n, int32_t C);
/* The same step under classifier-free guidance (DiffLoss.sample's cfg != 1 branch, diffloss.py:39-43, with
 * SimpleMLPAdaLN.forward_with_cfg :235-243): n is even, `out` is the network on [x[:n/2] | x[:n/2]] with conditions
 * [cond | uncond]; every row's eps becomes uncond + cfg_scale (cond - uncond) of its half-batch partner pair, the
 * variance channels stay the row's own. */
int hma_diff_p_sample_cfg(void* stream, const float* out, int64_t ldo, float* x, const float* noise, const float* tables6,
                          int32_t n_steps, int32_t step, float temperature, int32_t clip_denoised, int64_t n, int32_t C,
                          float cfg_scale);

/* ---- STMAR (continuous latents, hma/model/st_mar.py) input / output stages around the ST-transformer trunk --------------
 * hma_mar_patchify: latents [frames, H, W, C] -> patches [frames * H/p * W/p, p*p*C] in (p, q, c) channel order
 *   (st_mar.py:199-207); pixels flagged in `masked` (uint8, optional) are replaced by mask_token[c] first (:245);
 *   out_bf16 (zero-padded to `pad` columns: the token_embed GEMM operand) and / or out_f32; patch_mask (optional) = 1
 *   where any pixel of the patch is masked (:260). */
int hma_mar_patchify(void* stream, const float* latents, const uint8_t* masked, const float* mask_token, void* out_bf16,
                     int32_t pad, float* out_f32, float* patch_mask, int64_t frames, int32_t H, int32_t W, int32_t C,
                     int32_t patch);
/* dmask_token[c] += sum over masked pixels of d patches[row, (p, q, c)]  (backward of the mask-latent fill) */
int hma_mar_mask_token_bwd(void* stream, const float* dpatches, int64_t ld, const uint8_t* masked, float* dmask_token,
                           int64_t frames, int32_t H, int32_t W, int32_t C, int32_t patch);
/* x = z_proj_ln(concat(xtok rows, a_emb repeated A times) + pos_embed_TSC[t, s])  (st_mar.py:155-178; eps 1e-6, affine);
 * pos rows of frame t start at pos + t * pos_frame_stride.  backward: dxtok = d(image rows), da_emb[f] += sum over
 * the action rows (atomics), dpos += the batch sum of every position (whole samples: frames % T == 0), dgamma / dbeta += (atomics). */
int hma_mar_embed_fwd(void* stream, const float* xtok, const float* a_emb, const float* pos, int64_t pos_frame_stride,
                      const float* gamma, const float* beta, float eps, float* x, void* xhat, float* rstd, int64_t frames,
                      int32_t T, int32_t S, int32_t A);
int hma_mar_embed_bwd(void* stream, const float* dx, const void* xhat, const float* rstd, const float* gamma, float* dxtok,
                      float* da_emb, float* dpos, int64_t pos_frame_stride, float* dgamma, float* dbeta, int64_t frames,
                      int32_t T, int32_t S, int32_t A);
/* z = decoder_norm(y) + diffusion_pos_embed_learned[t * S + s]  (st_mar.py:192-194; rows (b, t, s)); backward:
 * dy = LN-backward(dz * gamma), dpos2 += the batch sum of every position (rows % (T * S) == 0), dgamma / dbeta += (atomics). */
int hma_mar_readout_fwd(void* stream, const float* y, const float* gamma, const float* beta, float eps, const float* pos2,
                        float* z, void* yhat, float* rstd, int64_t rows, int32_t T, int32_t S);
int hma_mar_readout_bwd(void* stream, const float* dz, const void* yhat, const float* rstd, const float* gamma, float* dy,
                        float* dpos2, float* dgamma, float* dbeta, int64_t rows, int32_t T, int32_t S);

/* MaskGIT training collator on device, hma/data.py:28-98 (get_maskgit_collator.collate_fn) on ids [B, T, HW]:
 * factorise (num_factored sub-vocabularies of V: 2 x 512 for the shipped models), corruption where r_corrupt[.., k] < corrupt_thresh (data.py:42-49), non-MLM corruption of
 * frames >= first_masked_frame where r_nonmlm[.., k] > correct_rate[t - fmf] (:51-64), cosine masking where
 * r_mask < mask_prob[b, t - fmf] -> mask_id (:68-83).  Every random draw is an input tensor (device pointers:
 * r_corrupt / random_values [B, T, HW, num_factored]; r_nonmlm [B, T - fmf, HW, num_factored]; correct_rate [T - fmf]; mask_prob
 * [B, T - fmf]; r_mask [B, T - fmf, HW]; NULL = that stage is off; mask_prob NULL = ids pass through unchanged, as
 * the reference does without dataloader_apply_mask).  any_masked (optional int32 flag) is OR-ed with 1 if any token
 * was masked (the reference redraws until that is the case, :72).  Pure function: bit-exact given the same draws. */
int hma_maskgit_collate(void* stream, const int64_t* ids, int64_t* out_ids, const float* r_corrupt, float corrupt_thresh,
                        const int64_t* random_values, const float* r_nonmlm, const float* correct_rate,
                        const float* mask_prob, const float* r_mask, int64_t B, int32_t T, int32_t HW,
                        int32_t first_masked_frame, int32_t V, int32_t num_factored /* 1 | 2: last dim of the draws */,
                        int64_t mask_id, int32_t* any_masked);
/* dst (bf16, [rows, cols]) = src (fp32) * keep / (1 - p) with the mask of hma_gemm_nt_t's dropout (element index row * cols
 * + col): the gradient that flows back through the Dropout after fc2 (st_transformer.py:26). */
int hma_dropout_bf16(void* stream, const float* src, void* dst, int64_t rows, int32_t cols, float p, const uint32_t* seed_dev,
                     int32_t salt);
/* dst[b][c][r] (bf16) = src[b][r][c] (f32): transposed bf16 copies of weights for the dgrad GEMMs */
int hma_transpose_cast_bf16(void* stream, const float* src, void* dst, int32_t rows, int32_t cols,
                            int32_t batch, int64_t src_stride, int64_t dst_stride);
/* A LayerNorm's affine folded into the Linear that consumes it (st_transformer.py:86 norm1 -> spatial qkv, :112 norm2 ->
 * mlp.fc1): Linear(xhat * gamma + beta) = xhat @ (W * gamma)^T + (bias + W @ beta).  Wf[b][n][k] (bf16) = W[b][n][k] *
 * gamma[b][k], bf[b][n] (f32) = (bias ? bias[b][n] : 0) + sum_k W[b][n][k] * beta[b][k]; W / gamma / beta / bias of batch b
 * sit `in_stride` floats after those of batch b - 1 (the per-layer blocks of the flat parameter buffer). */
int hma_fold_ln_bf16(void* stream, const float* W, const float* gamma, const float* beta, const float* bias, void* Wf, float* bf,
                     int32_t rows, int32_t cols, int32_t batch, int64_t in_stride, int64_t wf_stride, int64_t bf_stride);

/* ---- Fused MLP block (st_transformer.py:24-27 Mlp.forward as called from STBlock.forward :112; forward: mlp_drop == 0) ----------
 * hma_mlp_pack: a weight matrix rearranged into the MFMA-fragment order the fused kernels stream (512 fragments of 1 KB):
 *   kind 0: logical A[1024][256], kind 1: logical A[256][1024]; A[r][c] = src[r * row_stride + c * col_stride] *
 *   (row_scale ? row_scale[r] : 1) * (col_scale ? col_scale[c] : 1), rounded to bf16.  `batch` matrices, `src_batch_stride`
 *   floats apart (row_scale / col_scale move by the same stride), outputs `dst_batch_stride` bf16 elements apart.
 *   fc1 forward / recompute:  kind 0 of fc1.weight [1024][256] with col_scale = norm2.weight   (w1p)
 *   fc2 forward:              kind 1 of fc2.weight [256][1024]                                 (w2p)
 *   d gelu input:             kind 0 of fc2.weight^T (row_stride 1, col_stride 1024)           (w2tp)
 *   d xhat:                   kind 1 of fc1.weight^T (row_stride 1, col_stride 256) with row_scale = norm2.weight (w1tp) */
int hma_mlp_pack(void* stream, const float* src, int64_t row_stride, int64_t col_stride, const float* row_scale,
                 const float* col_scale, void* dst, int32_t kind, int32_t batch, int64_t src_batch_stride,
                 int64_t dst_batch_stride);
/* x += fc2(gelu(fc1(norm2(x)))) with xhat = LN(x) (no affine; bf16, saved by the kernel that last wrote x), b1 = fc1.bias
 * + fc1.weight @ norm2.bias (hma_fold_ln_bf16's bf), b2 = fc2.bias or NULL.  The hidden activation stays on chip.
 * ln_xhat != NULL: also ln_xhat = LN(new x, ln_eps, no affine) (bf16) and ln_rstd -- the next block's norm1
 * (st_transformer.py:86), what hma_ln_fwd would produce. */
typedef struct {
  const void* xhat; float* x;
  const void* w1p; const void* w2p; const float* b1; const float* b2;
  void* ln_xhat; float* ln_rstd; float ln_eps; int32_t _pad;
  int64_t M;
} hma_mlp_fwd_t;
int hma_mlp_fwd(void* stream, const hma_mlp_fwd_t* p);
/* Backward of the block above from dy = bf16 gradient wrt its output (the bf16 copy of the residual gradient dx):
 * u is recomputed from xhat; hg = gelu(u) and du = (dy fc2.weight) * gelu'(u) are written (bf16, ceil(M / 128) 128 rows of
 * 1024, in the HMA_A_BF16_FRAG32 order) for the two weight gradients (hma_gemm_tn_pair: fc2 from dy / hg as a_kind =
 * HMA_A_BF16_FRAG32, fc1 from du as y_kind = HMA_A_BF16_FRAG32 / xhat with the affine + dgamma / dbeta);
 * dx += LayerNorm-backward(du fc1.weight * gamma) (rstd = the saved 1 / sigma of norm2) and dx_bf16 = bf16(new dx),
 * which must not alias dy (the fc2 weight gradient still reads dy). */
typedef struct {
  const void* xhat; const float* rstd; const void* dy;
  float* dx; void* dx_bf16;
  const void* w1p; const void* w2tp; const void* w1tp; const float* b1;
  void* hg; void* du;
  int64_t M;
  /* mlp_drop > 0 (else 0 / NULL): the forward's two masks re-created (salts drop_salt: activation, drop_salt + 1: branch output,
   * see hma_chain_b_fwd_t): dy is masked as it is loaded and the masked rows are written to dy_drop (bf16 [M,256]: what the fc2
   * weight / bias gradient must read instead of dy), hg and du carry the activation mask. */
  float drop_p; int32_t drop_salt; const uint32_t* drop_seed; void* dy_drop;
} hma_mlp_bwd_t;
int hma_mlp_bwd(void* stream, const hma_mlp_bwd_t* p);

/* ---- Row-local chains of an STBlock (st_transformer.py:85-112): everything between two attentions in ONE launch ----------
 * A layer is two attentions plus two row-local chains; a chain keeps a token row on chip from the attention output to the
 * next attention's qkv input.  Kernel structure (csrc/chain.hip): 7 compute waves own 16 token rows each (lane = a token's
 * quarter row, v_mfma_f32_16x16x32_bf16 with the weights as the A operand: what a GEMM leaves in a lane's accumulators is,
 * packed to bf16, the next GEMM's B operand), an 8th wave streams the chain's packed weights through a 4-slot LDS-DMA
 * ring (16 KB bundles = 32 output columns x 256 k), one s_barrier per bundle.
 *
 * hma_chain_pack: weight bundles in streaming order.  Logical matrix A[r][c] = src[r * row_stride + c * col_stride] *
 * (row_scale ? row_scale[r] : 1) * (col_scale ? col_scale[c] : 1), rounded to bf16.
 *   kind 0 ("N-block"): A is [rows][256]; bundle b = rows 32 b .. 32 b + 31: 16 fragments (j, o) at 2 j + o, lane (i, g) holds
 *                       A[32 b + 8 (i >> 2) + (i & 3) + 4 o][8 (4 j + g) .. + 7]          (rows / 32 bundles)
 *   kind 1 ("K-slice"): A is [256][cols]; bundle b = columns 32 b .. 32 b + 31: 16 fragments t, lane (i, g) holds
 *                       A[32 (t >> 1) + 8 (i >> 2) + (i & 3) + 4 (t & 1)][32 b + 8 g .. + 7]  (cols / 32 bundles)
 * `batch` matrices `src_batch_stride` floats apart (the scales move with them), outputs `dst_batch_stride` bf16 apart; bundle b
 * is written at bundle slot b * bundle_stride of the output (2: the fc1 / fc2 bundles of chain B interleave). */
int hma_chain_pack(void* stream, const float* src, int64_t row_stride, int64_t col_stride, const float* row_scale,
                   const float* col_scale, void* dst, int32_t kind, int32_t rows, int32_t cols, int32_t batch,
                   int64_t src_batch_stride, int64_t dst_batch_stride, int32_t bundle_stride);
/* Several packing jobs in ONE launch each (the weight copies are rebuilt after every optimizer step -- 31 launches of ~10 us at the
 * headline model, most of it launch and drain): a job is the argument list of hma_chain_pack / hma_mlp_pack (rows, cols and
 * bundle_stride unused) / hma_transpose_cast_bf16 (src, dst, rows, cols, batch and the two batch strides used).  The jobs of one
 * call must not write overlapping outputs; any number of jobs (24 per launch).  Results are identical to the single calls. */
typedef struct {
  const float* src; int64_t row_stride, col_stride; const float* row_scale; const float* col_scale; void* dst;
  int32_t kind, rows, cols, batch; int64_t src_batch_stride, dst_batch_stride; int32_t bundle_stride, reserved;
} hma_pack_job_t;
int hma_chain_pack_multi(void* stream, const hma_pack_job_t* jobs, int32_t njobs);
int hma_mlp_pack_multi(void* stream, const hma_pack_job_t* jobs, int32_t njobs);
int hma_transpose_cast_bf16_multi(void* stream, const hma_pack_job_t* jobs, int32_t njobs);
/* The packed weights of one chain: up to 4 segments of bundles, consumed in order once per 112-row tile. */
typedef struct { const void* seg[4]; int32_t bundles[4]; } hma_chain_weights_t;

/* Chain A forward (st_transformer.py:86 proj, :102-104 ModulateLayer = st_mask_git.py:66-76, :111 qkv of attention.py:39):
 *   x1 = x + o Wproj^T + b_proj;  xhat = LN(x1, 1e-6, no affine);  xm = xhat (1 + scale[f]) + shift[f];
 *   x2 = x1 + xm Wlin^T + b_lin;  qkv = bf16(x2) Wqkv^T + b_qkv
 * in: o [M,256] bf16 (spatial attention output), x [M,256] fp32 (updated in place to x2), ss [frames,512] = shift | scale;
 * out: xhat, xm, x_bf16 = bf16(x2) [M,256] bf16 and rstd [M] (saved for backward; each may be NULL), qkv [.,768] bf16 at
 * row remap(m) (q_group_* as in hma_gemm_nt: the decode K/V cache).  weights: N-block bundles of proj (8), lin (8), qkv (24).
 * rows_per_frame % 16 == 0.  use_mod == 0 skips the modulate stage (no action tokens): x2 = x1, weights = proj (8), qkv (24). */
typedef struct {
  hma_chain_weights_t w;
  const void* o; float* x; const float* ss;
  const float* b_proj; const float* b_lin; const float* b_qkv;
  void* xhat; void* xm; float* rstd; void* x_bf16;
  void* qkv; int64_t ldq; int64_t q_group_rows, q_group_stride;
  int64_t M; int32_t rows_per_frame; int32_t use_mod;
} hma_chain_a_fwd_t;
int hma_chain_a_fwd(void* stream, const hma_chain_a_fwd_t* p);

/* Chain A backward (autograd mirror of the above):
 *   dx2 = dx + dqkv Wqkv;  dxm = bf16(dx2) Wlin;  g = dxm (1 + scale[f]);
 *   dx1 = dx2 + rstd (g - mean(g) - xhat mean(g xhat));  d_o = bf16(dx1) Wproj
 *   dss[f] += [sum_rows dxm | sum_rows dxm xhat]   (fp32 atomics: zero dss first)
 * in: dqkv [M,768] bf16, dx [M,256] fp32 (updated in place to dx1), xhat / rstd saved by the forward, ss;
 * out: dx2_bf16 (dY of the linear_out weight gradient), dx1_bf16 (dY of the projection's), d_o [M,256] bf16.
 * weights: N-block bundles of Wqkv^T in three k-chunks (3 x 8: chunk c = rows 256 c .. of qkv.weight, A[n][k] = W[256 c + k][n]),
 * Wlin^T (8), Wproj^T (8). */
typedef struct {
  hma_chain_weights_t w;
  const void* dqkv; int64_t ldq; float* dx;
  const void* xhat; const float* rstd; const float* ss;
  void* dx2_bf16; void* dx1_bf16; void* d_o; float* dss;
  int64_t M; int32_t rows_per_frame; int32_t use_mod;
} hma_chain_a_bwd_t;
int hma_chain_a_bwd(void* stream, const hma_chain_a_bwd_t* p);

/* Chain S backward -- the spatial side of a block's backward behind the attention backward (st_transformer.py:85-86: norm1 and the
 * qkv Linear of attention.py:39, autograd mirror), what hma_gemm_nt (dqkv -> bf16 g) + hma_ln_bwd did in two launches:
 *   g = dqkv Wqkv diag(gamma);  dx <- dx + rstd (g - mean(g) - xhat mean(g xhat));  dx_bf16 = bf16(dx)
 * in: dqkv [M, ldq >= 768] bf16 (spatial attention backward), xhat [M,256] bf16 / rstd [M] saved by the forward (norm1's output
 * without its affine), dx [M,256] fp32 (updated in place); out: dx_bf16 [M,256].  M % 16 == 0.
 * weights: 24 N-block bundles of Wqkv^T in three k-chunks as in chain A backward, with norm1's gamma folded into the OUTPUT rows
 * (hma_chain_pack row_scale).  norm1's dgamma / dbeta are not produced here: they come out of the qkv weight gradient's reduction
 * (hma_gemm_tn_t w_master / dgamma / dbeta). */
typedef struct {
  hma_chain_weights_t w;
  const void* dqkv; int64_t ldq; float* dx;
  const void* xhat; const float* rstd;
  void* dx_bf16;
  int64_t M;
  /* > 0: dqkv is in the head-blocked order of hma_attn_spatial_bwd_blocked (rows per frame, a multiple of 16 that divides M; ldq = 768);
   * 0: row-major [M, ldq] */
  int64_t hb_rows;
} hma_chain_s_bwd_t;
int hma_chain_s_bwd(void* stream, const hma_chain_s_bwd_t* p);

/* Chain T backward -- the temporal attention's backward of a block in one launch (st_transformer.py:111; attention.py:37-61 causal and
 * :60 proj, autograd mirror), training passes over windows of 1 <= T <= 16 frames: what hma_gemm_nt (d_o = bf16(dx) Wproj) and
 * hma_attn_temporal_bwd did in two launches with d_o's round trip through HBM.  A compute wave owns the T frames of a (sample, token
 * position) column (a tile has 16 frame lanes: with T < 16 the last 16 - T are masked -- loads clamped, gradient rows zeroed, stores
 * switched off).  Rows are (b, t, s), s fastest, SA rows per frame, B samples, M = T B SA.
 * in: dy_bf16 [M,256] bf16 (the gradient of the block's x behind the temporal attention's residual add), qkv [M,768] bf16 (saved by the
 * forward); out: dqkv [M,768] bf16 (dq, dk scaled by attn_scale as hma_attn_temporal_bwd writes them).
 * weights: the 8 N-block bundles of Wproj^T (hma_chain_pack of the transposed weight, kind 0). */
typedef struct {
  hma_chain_weights_t w;
  const void* dy_bf16; const void* qkv; void* dqkv;
  int64_t B; int32_t T; int32_t SA;
  float attn_scale; int32_t _pad;
} hma_chain_t_bwd_t;
int hma_chain_t_bwd(void* stream, const hma_chain_t_bwd_t* p);

/* Chain B forward (inference / decode passes, and training passes with the fields at the end of the struct): st_transformer.py:111
 * proj, :112 norm2 + Mlp (:24-27), and the NEXT block's :85-86 norm1 + qkv (attention.py:39):
 *   x1 = x + o Wproj^T + b_proj;  x2 = x1 + fc2(gelu(fc1(LN(x1)))) ;  qkv = LN(x2) Wqkv'^T + b_qkv'
 * in: o [M,256] bf16 (temporal attention output), x [M,256] fp32 (updated in place to x2); out: qkv [M, ldq] bf16 of the next
 * block (NULL: last block, no qkv stage).  Both LayerNorms are affine-free here: their gamma / beta are folded into the
 * weights / biases that follow (hma_chain_pack col_scale; hma_fold_ln_bf16's bias).  weights: N-block bundles of proj (8), then
 * 64 bundles alternating fc1 (N-block of gamma-folded fc1.weight, hidden block h) and fc2 (K-slice of fc2.weight, the same h),
 * then N-block bundles of the next block's gamma-folded qkv (24).  b1 = folded fc1 bias (1024), b2 = fc2.bias or NULL. */
typedef struct {
  hma_chain_weights_t w;
  const void* o; float* x;
  const float* b_proj; const float* b1; const float* b2; const float* b_qkv;
  void* qkv; int64_t ldq;
  int64_t M; float ln_eps; int32_t _pad;
  /* training (all NULL in inference): xhat2 = LN(x1) bf16 [M,256] and rstd2 [M] (norm2, what hma_mlp_bwd re-reads), xhat1n / rstd1n
   * = LN(x2) and its 1 / sigma (the next block's norm1; required with qkv) */
  void* xhat2; float* rstd2; void* xhat1n; float* rstd1n;
  /* training with mlp_drop > 0 (drop_p in (0, 1), else 0 / NULL): the two nn.Dropout sites of Mlp.forward (st_transformer.py:25-26)
   * with hma_gemm_nt_t's counter-based mask: gelu(u) * keep(drop_salt, row * 1024 + hidden unit) / (1 - p), then
   * (fc2 output + b2) * keep(drop_salt + 1, row * 256 + column) / (1 - p) before the residual add.  Requires xhat2. */
  float drop_p; int32_t drop_salt; const uint32_t* drop_seed;
} hma_chain_b_fwd_t;
int hma_chain_b_fwd(void* stream, const hma_chain_b_fwd_t* p);

/* Chain A + causal temporal attention + chain B in ONE launch (training passes over windows of exactly T = 16 frames): everything of
 * an STBlock between its spatial attention and the next block's -- st_transformer.py:86 proj, :102-104 ModulateLayer, :111 temporal
 * attention (attention.py:37-61, causal), :112 norm2 + Mlp, and :85-86 norm1 + qkv of the next block.  What hma_chain_a_fwd (use_mod = 1,
 * every saved activation), hma_attn_temporal_fwd and hma_chain_b_fwd (training form, no dropout) compute in three launches; a compute
 * wave owns the 16 frames of one (sample, token position) column, so the attention is wave-local and the residual row never leaves the
 * registers between the chains (4 096 fewer bytes per row).  Rows are (b, t, s), s fastest, SA rows per frame, B samples.
 * in: o_s [M,256] bf16, x [M,256] fp32 (updated in place to the block's output), ss [B*16, 512] = shift | scale.
 * out (all saved for the backward): xhat_m, xm, x2b [M,256] bf16 + rstd_m [M]; qkv_t [M,768] bf16; o_t [M,256] bf16; xhat2 [M,256] +
 * rstd2 [M]; with qkv_s != NULL also the next block's xhat1n + rstd1n and qkv_s [M,768] bf16 (NULL: last block).
 * weights (hma_chain_pack bundles): seg 0 proj_s (8), 1 linear_out (8), 2 temporal qkv (24), 3 proj_t (8), 4 the 64 alternating fc1 / fc2
 * bundles, 5 the next block's folded spatial qkv (24, or 0).  Biases as in the two chains (b1 = folded fc1 bias, required).
 * bundles[1] == 0: a block WITHOUT action tokens (no ModulateLayer, st_transformer.py:102-104 skipped): ss, b_lin, xhat_m, xm, rstd_m are
 * not used and x2b = bf16(x + proj_s(o_s)) is the temporal qkv's operand (what hma_chain_a_fwd writes with use_mod = 0). */
typedef struct {
  const void* seg[6]; int32_t bundles[6];
  const void* o_s; float* x; const float* ss;
  const float* b_proj_s; const float* b_lin; const float* b_qkv_t; const float* b_proj_t; const float* b1; const float* b2; const float* b_qkv_s;
  void* xhat_m; void* xm; float* rstd_m; void* x2b;
  void* qkv_t; void* o_t;
  void* xhat2; float* rstd2; void* xhat1n; float* rstd1n; void* qkv_s;
  int64_t B; int32_t T; int32_t SA;
  float attn_scale; float ln_eps;
  /* training with mlp_drop > 0 (drop_p in (0, 1), else 0 / NULL): the two nn.Dropout sites of Mlp.forward, as in hma_chain_b_fwd_t
   * (the same element counters and salts: hma_mlp_bwd re-creates the masks).  Needs the modulated form (bundles[1] == 8). */
  float drop_p; int32_t drop_salt; const uint32_t* drop_seed;
} hma_chain_ab_fwd_t;
int hma_chain_ab_fwd(void* stream, const hma_chain_ab_fwd_t* p);

/* Readout + factorised cross-entropy of the image rows in one launch (st_mask_git.py:681-683 out_x_proj; :603-630
 * compute_video_loss_and_acc with label_smoothing): what hma_gemm_nt (x -> fp32 logits) + hma_ce_fwd_bwd do, without the logits in
 * HBM.  rows = B * T * S image rows (a multiple of 16); row i reads x row (i / S) * SA + i % S of the [*, 256] fp32 residual stream.
 * w: 32 N-block bundles (hma_chain_pack kind 0) of out_x_proj.weight [1024][256]; bias [1024] or NULL.  A row counts when its frame
 * index (i / S) % T >= 1 and input_ids[i] == mask_id.  stats (fp32[HMA_CE_STATS_FLOATS], as for hma_ce_fwd_bwd): stats[0] = sum of
 * row losses (order-independent), stats[1] += rows whose two factor arg-maxes both hit; stats[2] (the masked-row count,
 * hma_count_masked) is read.  dlogits (bf16 [rows, 1024], or NULL) = grad_scale *
 * *grad_scale_dev / stats[2] * (softmax - smoothed one-hot) per factor, zero for rows that do not count. */
typedef struct {
  hma_chain_weights_t w;
  const float* x; const float* bias;
  const int64_t* input_ids; const int64_t* labels;
  float* stats; void* dlogits; const float* grad_scale_dev;
  int64_t rows; int64_t mask_id;
  int32_t S; int32_t SA; int32_t T; float grad_scale; float label_smoothing; int32_t _pad;
} hma_readout_ce_t;
int hma_readout_ce(void* stream, const hma_readout_ce_t* p);

/* n floats at p = 0 (captured in graphs in front of kernels that accumulate with atomics) */
int hma_zero_f32(void* stream, float* p, int64_t n);

/* library identity, for the loader: returns 0x484d4104 */
int hma_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HMA_HIP_H */
