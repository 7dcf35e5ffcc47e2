/*
 * butd_attention.h -- C ABI of the fused cross-modal attention / FFN kernels (gfx950, fp32 MFMA).
 *
 * These replace the stock-torch op chains behind nn.MultiheadAttention / nn.Linear / nn.LayerNorm at
 * the reference call sites models/encoder_decoder_layers.py:47-73,87-122,133-155,297-330,356-404
 * (arithmetic: torch/nn/functional.py multi_head_attention_forward).  All tensors are dense fp32,
 * row-major, batch-first; pointers are device pointers; `stream` is a hipStream_t; every call is an
 * asynchronous launch that neither allocates nor synchronises (hipGraph-capturable); return value 0
 * or a hipError_t.
 */
#ifndef BUTD_ATTENTION_H
#define BUTD_ATTENTION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *butd_stream_t; /* hipStream_t */

/* One dense problem   C[M,N] (op)= epilogue( sum_k A(m,k) * B(n,k) ).
 * A(m,k) = a[m*lda_m + k*lda_k], combined with a2[...] (same strides) when a2 != NULL:
 *          a2_mode 0: a + a2 (e.g. src + pos);  a2_mode 1: a * (a2 > 0 ? a2_scale : 0)  (ReLU/dropout gate),
 * B(n,k) = b[n*ldb_n + k*ldb_k];  exactly one of each stride pair is 1 (the contiguous one).
 * epilogue(v) = relu?( (v + bias[n]) * scale ) , optionally * dropout keep-mask / (1-p);
 * accumulate != 0 adds into C with atomics (used with split_k > 1 for weight gradients).
 * ones_col != 0: B gets a virtual extra column n == N of ones whose results are atomically added to
 * bias_grad[m] instead of C (column sums -> bias gradient in the same pass). */
typedef struct {
  const float *a, *a2, *b, *bias;
  float *c, *bias_grad;
  int M, N, K;
  long lda_m, lda_k, ldb_n, ldb_k, ldc;
  float scale;
  int a2_mode;
  float a2_scale;
  int relu, accumulate, ones_col, split_k;
  float dropout_p;        /* 0 = off */
  uint32_t dropout_site;  /* stream id of the counter-based RNG */
  /* Optional per-channel affine + ReLU applied to an operand while it is staged (folds
   * BatchNorm + ReLU of the producing layer into the consumer GEMM):
   *   A(m,k) <- relu(A(m,k) * a_chan_scale[k] + a_chan_shift[k])   (channel = contraction index)
   *   B(n,k) <- relu(B(n,k) * b_chan_scale[n] + b_chan_shift[n])   (channel = output column)   */
  const float *a_chan_scale, *a_chan_shift, *b_chan_scale, *b_chan_shift;
  /* Optional operand dropout, applied after the affine+ReLU (folds ``Dropout(ReLU(BatchNorm(z)))`` of
   * the producing layer, models/modules.py:92-108, into the consumer):
   *   X <- keep(step counter, site, i) ? X / (1-p) : 0,   i = offset (in floats) of the element from the
   * operand's base pointer `a` / `b` -- so a forward product, the weight-gradient product that re-reads
   * the same activation through different strides, and butd_mlp_mask_stats (include/butd_mlp.h) all
   * regenerate the same mask when they are handed the same base pointer and site. */
  float a_drop_p;
  uint32_t a_drop_site;
  float b_drop_p;
  uint32_t b_drop_site;
  /* Optional column statistics of the stored result (plain-store problems only: accumulate == 0,
   * split_k == 1): col_sum[n] += sum_m C[m,n], col_sumsq[n] += sum_m C[m,n]^2 in double (atomics, the
   * caller zero-fills) -- the BatchNorm batch statistics of a 1x1 convolution, in the same pass. */
  double *col_sum, *col_sumsq;
  /* Optional accumulation into existing data by the plain-store epilogue (accumulate == 0,
   * split_k == 1; every element has exactly one writer, so no atomics):
   *   c_add != 0:  C <- C + epilogue(v)            (e.g. the residual-path gradient already in C)
   *   c2 != NULL:  C2 <- C2 + epilogue(v) as well   (same ldc; a second consumer of the same product)
   * They let a block return  d_res + dq*Wq  and  dq*Wq  from ONE product instead of autograd adding
   * tensors afterwards.  No other problem of the same launch may write C (c_add) or C2. */
  int c_add;
  float *c2;
  /* Column statistics spread over col_slots (a power of two, 0/1 = one) copies of the sum arrays,
   * slot s at col_sum + s * col_slot_stride: a workgroup adds into slot (its linear index mod
   * col_slots), so 10^4..10^5 row tiles do not serialise on the same 2*N addresses; the consumer sums
   * the slots (butd_sa_bn_finalize). */
  int col_slots;
  long col_slot_stride;
  /* != 0: round the operands to bf16 (nearest even) while they are staged and multiply on the bf16 matrix
   * cores (v_mfma_f32_16x16x32_bf16), fp32 accumulation and epilogue; every tensor stays fp32 in memory.  The
   * "bf16 attention / FFN" operating point of BASELINE configs[3]; honoured when every float4-aligned
   * problem of a launch asks for it, ignored (fp32) for the element-wise staged ones (a2, unaligned). */
  int compute_bf16;
  /* Optional gate on the stored result (plain-store problems only: accumulate == 0, ones_col == 0):
   *   C[m,n] <- c_gate[m*ldc + n] > 0 ? epilogue(v) * c_gate_scale : 0
   * -- the backward of ReLU (+ dropout: the saved activation h = relu(z) * keep / (1-p) is > 0 exactly where the
   * gradient passes, and c_gate_scale = 1 / (1-p)) applied by the product that CREATES the gradient d_h, so the
   * products that read it need no companion operand (a2_mode 1 does the same on the reading side and keeps those
   * products on the element-wise staging path).  c_gate has the layout of C (same ldc). */
  const float *c_gate;
  float c_gate_scale;
  /* Optional: the A operand's per-channel affine computed HERE from the BatchNorm column sums the producing product
   * left behind (training mode; replaces a_chan_scale / a_chan_shift and the bookkeeping launch between two products of
   * a Conv1d+BatchNorm chain, include/butd_mlp.h):  for channel k of the contraction
   *   mean = a_bn_sum[k] / count,  var = max(a_bn_sumsq[k] / count - mean^2, 0),  rstd = 1 / sqrt(var + eps),
   *   scale = gamma[k] * rstd,  shift = beta[k] - mean * scale,   A(m,k) <- relu(A(m,k) * scale + shift)
   * -- the arithmetic of butd_mlp_bn_finalize.  The problem's FIRST workgroup also writes mean, rstd, scale, shift to
   * a_bn_out[0..3][k] (rows a_bn_ld floats apart; saved for the backward pass), updates the running statistics
   * (momentum, unbiased variance) and increments *a_bn_nbt.  Float4-aligned problems only (K <= 320 per slice). */
  const double *a_bn_sum, *a_bn_sumsq;
  const float *a_bn_gamma, *a_bn_beta;
  float *a_bn_running_mean, *a_bn_running_var;
  long long *a_bn_nbt;
  float *a_bn_out;
  long a_bn_ld, a_bn_count;
  float a_bn_eps, a_bn_momentum;
  /* Optional: the stored result is the gradient that arrives at a BatchNorm + ReLU (a Conv1d + BatchNorm + ReLU chain's
   * backward, pytorch_utils.py:39-58 / models/modules.py:19-42), and this product applies the ReLU gate and leaves the two
   * column sums of the BatchNorm backward behind -- what butd_mlp_mask_stats (include/butd_mlp.h) did in a pass of its own:
   *   z = c_bn_z[m*ldc + n] (the layer's saved pre-activation, C's layout),  mean, rstd, scale, shift = c_bn_aff[0..3][n]
   *   (rows c_bn_ld floats apart: the a_bn_out table of the forward pass),
   *   g = scale * z + shift > 0 ? epilogue(v) : 0;   C[m,n] <- g;
   *   col_sum[n] += sum_m g,   col_sumsq[n] += sum_m g * (z - mean) * rstd       (double, atomics; caller zero-fills)
   * With c_bn_drop_p > 0 the activation was followed by a Dropout (ThreeLayerMLP, models/modules.py:64-72):
   *   g <- keep(step counter, c_bn_drop_site, m*ldc + n) ? g / (1 - p) : 0     before it is stored and summed
   * -- the mask the forward drew for the operand with base pointer c_bn_z's twin (same site, same element offsets).
   * Plain-store problems without ReLU / output dropout / c2; col_sum and col_sumsq are required. */
  const float *c_bn_z, *c_bn_aff;
  long c_bn_ld;
  float c_bn_drop_p;
  uint32_t c_bn_drop_site;
  /* Deterministic split-K (round 5): with c_partial != NULL an accumulate != 0 problem issues NO atomics.  Slice s of its
   * split_k slices stores its share with plain (float4) stores at c_partial + s * c_partial_stride, laid out dense
   * [M][N] with the ones-column's results (the bias gradient, M floats) behind it at offset M * N; c and bias_grad are
   * not touched.  Every slice must own at least one 32-deep slab of the contraction (the caller passes the effective
   * slice count); no bias / relu / output dropout.  c_partial 16-byte aligned, c_partial_stride a multiple of 4.
   * Device-scope float atomics are served by the memory side on this part (eight L2s, one per XCD): ~25 ns per
   * thousand and a 32-byte request each -- a 288 x 288 weight gradient in 8 slices was 0.67 M of them. */
  float *c_partial;
  long c_partial_stride;
  /* A FOLD problem (fold_src != NULL; M, N, K, a, b are ignored): the sum, in slice order (bit-reproducible), of
   * fold_count partial slabs that an EARLIER launch on the same stream wrote as above:
   *   c[i]         (=|+=, c_add)  sum_s fold_src[s * fold_stride + i],              0 <= i < fold_len   (fold_len % 4 == 0)
   *   bias_grad[j] (=|+=, c_add)  sum_s fold_src[s * fold_stride + fold_len + j],   0 <= j < fold_len2  (bias_grad may be NULL)
   * As one more problem of a grouped launch it costs no launch of its own ("ride along", as the LayerNorm backward's
   * column sums): an element-wise pass of ceil((fold_len + fold_len2) / 2048) workgroups. */
  const float *fold_src;
  int fold_count;
  long fold_stride, fold_len, fold_len2;
} butd_gemm_problem;

/* Launches up to 32 independent problems in ONE 1-D grid (every problem owns a range of workgroups).
 * rng_counter: device pointer to a uint64 step counter (may be NULL when no problem uses dropout). */
int butd_gemm_grouped(const butd_gemm_problem *problems, int count, const uint64_t *rng_counter,
                      butd_stream_t stream);

/* Tuning hook: force the workgroup tile (tile_m x tile_n, an entry of the kernel's menu: 32x32, 64x64,
 * 32x96, 64x96, 96x32, 128x64, 128x96; (64, -64) = 64x64 on the one-slab-ahead K loop) of every following
 * butd_gemm_grouped launch; (0, 0) returns to the built-in choice.  Returns 0 or hipErrorInvalidValue.
 * Not thread-safe. */
int butd_gemm_set_tile(int tile_m, int tile_n);

/* Scaled-dot-product attention core for head_dim <= 48 (BUTD-DETR: 8 heads x 36).
 * q (B,Lq,H*D) already scaled by 1/sqrt(D) (the projection's epilogue does it), k, v (B,Lk,H*D);
 * key_padding_mask (B,Lk) uint8, nonzero = masked (may be NULL); out (B,Lq,H*D);
 * lse (B,H,Lq) = log-sum-exp of each score row (saved for backward).
 * Dropout with probability dropout_p on the softmax probabilities (0 = off). */
int butd_attention_fwd(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                       const float *v, const uint8_t *key_padding_mask, float *out, float *lse,
                       float dropout_p, uint32_t dropout_site, const uint64_t *rng_counter,
                       butd_stream_t stream);

/* Round 6 microbenchmark (profiles/r06_split_bf16.txt): butd_attention_fwd with every operand split into two bf16 halves
 * (x = hi + lo) and every product as hi.hi + lo.hi + hi.lo on v_mfma_f32_16x16x32_bf16, fp32 accumulation -- ~16 bits of
 * mantissa per operand.  Same arguments and results as butd_attention_fwd; head dimension 36 only.  Not on the product
 * path: a NAMED extra precision (neither the reference's fp32 arithmetic nor configs[3]'s bf16). */
int butd_attention_fwd_split_bf16(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                                  const float *v, const uint8_t *key_padding_mask, float *out, float *lse,
                                  float dropout_p, uint32_t dropout_site, const uint64_t *rng_counter,
                                  butd_stream_t stream);

/* Backward of the above.  delta (B,H,Lq) scratch (written here); dq (B,Lq,.), dk, dv (B,Lk,.) are
 * overwritten.  The gradient rows may be wider than H*D: ld_dq / ld_dkv are their row strides in floats
 * (0 = H*D), so dq|dk|dv (or dk|dv) can sit side by side in one matrix and the input-projection
 * gradients become ONE product over the concatenated contraction.  dq is multiplied by dq_scale on the
 * way out (the 1/sqrt(D) the forward projection applied to q). */
int butd_attention_bwd(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                       const float *v, const uint8_t *key_padding_mask, const float *out,
                       const float *dout, const float *lse, float *delta, float *dq, float *dk,
                       float *dv, long ld_dq, long ld_dkv, float dq_scale, float dropout_p,
                       uint32_t dropout_site, const uint64_t *rng_counter, butd_stream_t stream);

/* Backward of butd_attention_fwd for SHORT KEY SETS (Lk <= butd_attention_bwd_short_keys_max() = 144: the <= 80 text
 * tokens and the 132 detected boxes of the cross-attention sites, models/encoder_decoder_layers.py:87-122,376-395)
 * as ONE kernel that computes every score tile once: same arguments and results as butd_attention_bwd, except that
 * dk and dv are ACCUMULATED into (one atomic add per element and workgroup): the caller zero-fills them.  fp32. */
int butd_attention_bwd_short_keys_max(void);
int butd_attention_bwd_short_keys(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                                  const float *v, const uint8_t *key_padding_mask, const float *out,
                                  const float *dout, const float *lse, float *delta, float *dq, float *dk,
                                  float *dv, long ld_dq, long ld_dkv, float dq_scale, float dropout_p,
                                  uint32_t dropout_site, const uint64_t *rng_counter, butd_stream_t stream);

/* Backward of butd_attention_fwd as ONE pass that computes every score tile once (54 matrix instructions per 16 x 16
 * tile instead of 72, one softmax / dropout pass instead of two).  A workgroup owns a chunk of keys -- 256 for LONG KEY
 * SETS (the 1024 seed points every encoder layer attends over and the decoder / text streams cross-attend to,
 * models/encoder_decoder_layers.py:60-85,356-375), 64 otherwise -- keeps their fragments and dK / dV accumulators in
 * registers and walks its share of the queries; what is summed ACROSS workgroups (dQ over the key chunks; dK / dV over
 * the query splits of short key sets) goes to slabs of `ws` and one small launch adds the slabs in order: no atomics,
 * bit-reproducible.  Same arguments and results as butd_attention_bwd (delta is formed inside).  fp32, head dimension 36.
 *   butd_attention_bwd_long_keys_scratch: floats of `ws` this call needs (>= 0), or -1 when the shape is not served
 *   (the caller then uses butd_attention_bwd).
 *   butd_attention_bwd_long_keys_set_chunk: tuning hook -- (keys per workgroup: 256 / 128 / 64, query splits) of every
 *   following call; (0, 0) returns to the built-in rule.  Not thread-safe. */
int butd_attention_bwd_long_keys_set_chunk(int keys, int q_splits);
long butd_attention_bwd_long_keys_scratch(int B, int H, int Lq, int Lk, int D, long ld_dq);
int butd_attention_bwd_long_keys(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                                 const float *v, const uint8_t *key_padding_mask, const float *out,
                                 const float *dout, const float *lse, float *dq, float *dk, float *dv, long ld_dq,
                                 long ld_dkv, float dq_scale, float dropout_p, uint32_t dropout_site,
                                 const uint64_t *rng_counter, float *ws, long ws_floats, butd_stream_t stream);

/* butd_attention_bwd_long_keys with the matrix steps on the bf16 matrix cores (bf16 LDS images, v_mfma_f32_16x16x32_bf16,
 * fp32 accumulation; BASELINE configs[3]): the same plan (256- and 64-key chunks, query splits), slabs and fold; `_scratch`
 * returns -1 where the fp32 entry point does (the caller then uses butd_attention_bwd_bf16). */
long butd_attention_bwd_long_keys_bf16_scratch(int B, int H, int Lq, int Lk, int D, long ld_dq);
int butd_attention_bwd_long_keys_bf16(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                                      const float *v, const uint8_t *key_padding_mask, const float *out,
                                      const float *dout, const float *lse, float *dq, float *dk, float *dv, long ld_dq,
                                      long ld_dkv, float dq_scale, float dropout_p, uint32_t dropout_site,
                                      const uint64_t *rng_counter, float *ws, long ws_floats, butd_stream_t stream);

/* The same two entry points with the matrix steps on the bf16 matrix cores (BASELINE configs[3]: "bf16 attention"):
 * operands rounded to bf16 (nearest even) in registers, v_mfma_f32_16x16x16_bf16, fp32 accumulation; scores'
 * statistics, exponentials, the (o, m, l) state, dropout masks and every tensor in memory are fp32 as above. */
int butd_attention_fwd_bf16(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                            const float *v, const uint8_t *key_padding_mask, float *out, float *lse,
                            float dropout_p, uint32_t dropout_site, const uint64_t *rng_counter,
                            butd_stream_t stream);
int butd_attention_bwd_bf16(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                            const float *v, const uint8_t *key_padding_mask, const float *out,
                            const float *dout, const float *lse, float *delta, float *dq, float *dk,
                            float *dv, long ld_dq, long ld_dkv, float dq_scale, float dropout_p,
                            uint32_t dropout_site, const uint64_t *rng_counter, butd_stream_t stream);

/* y = LayerNorm(residual + dropout(x)) over the last dim (cols <= 1024), eps as nn.LayerNorm.
 * Saves mean/rstd (rows) for backward. */
int butd_add_dropout_layernorm_fwd(int rows, int cols, const float *x, const float *residual,
                                   const float *gamma, const float *beta, float eps, float *y,
                                   float *mean, float *rstd, float dropout_p, uint32_t dropout_site,
                                   const uint64_t *rng_counter, butd_stream_t stream);

/* The same, and additionally y_pos = y + pos (both rows x cols; pos == NULL <=> y_pos == NULL): the query input
 * `tgt + query_pos` of the NEXT attention block (encoder_decoder_layers.py:356-404) leaves the kernel that produced
 * `tgt` instead of being a separate elementwise pass. */
int butd_add_dropout_layernorm_fwd_pos(int rows, int cols, const float *x, const float *residual,
                                       const float *gamma, const float *beta, float eps, float *y,
                                       float *mean, float *rstd, float dropout_p, uint32_t dropout_site,
                                       const uint64_t *rng_counter, const float *pos, float *y_pos,
                                       butd_stream_t stream);

/* Backward: given dy and the saved statistics, produces d_residual (= d of the pre-norm sum) and
 * dx (= d_residual * dropout mask / (1-p)); accumulates dgamma/dbeta (cols) with atomics
 * (caller zero-fills them).  dx may alias d_residual when dropout_p == 0. */
int butd_add_dropout_layernorm_bwd(int rows, int cols, const float *dy, const float *x,
                                   const float *residual, const float *gamma, const float *mean,
                                   const float *rstd, float *dx, float *d_residual, float *dgamma,
                                   float *dbeta, float dropout_p, uint32_t dropout_site,
                                   const uint64_t *rng_counter, butd_stream_t stream);

/* The same without atomics: workgroup w of the launch (butd_layernorm_bwd_blocks(rows) of them) writes the column sums
 * of ITS rows to partials[w][0 .. cols) (dgamma) and partials[w][cols .. 2 cols) (dbeta); partials is
 * (blocks, 2 * cols), fully overwritten.  The caller folds it -- dgamma | dbeta = ones(1, blocks) . partials is one more
 * problem (M = 1, N = 2 cols, K = blocks) of the grouped product that follows a LayerNorm backward in every block of
 * encoder_decoder_layers.py:75-124, 166-186, 340-406, so the 128 .. 512 same-address float atomics per column (2.9 us of
 * the kernel's 8.5 us at 2048 rows, 5.0 of 18.4 at 8192) disappear and the two gradients become bit-reproducible. */
int butd_layernorm_bwd_blocks(int rows);
int butd_add_dropout_layernorm_bwd_partial(int rows, int cols, const float *dy, const float *x,
                                           const float *residual, const float *gamma, const float *mean,
                                           const float *rstd, float *dx, float *d_residual, float *partials,
                                           float dropout_p, uint32_t dropout_site,
                                           const uint64_t *rng_counter, butd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BUTD_ATTENTION_H */
