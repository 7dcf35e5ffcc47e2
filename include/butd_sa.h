/*
 * butd_sa.h -- C ABI of the set-abstraction shared-MLP pipeline (gfx950).
 *
 * Replaces, for PointnetSAModuleVotes (pointnet2/pointnet2_modules.py:210-272), the op chain
 *   QueryAndGroup (pointnet2_utils.py:317-376) -> SharedMLP = 3 x [Conv2d 1x1 -> BatchNorm2d -> ReLU]
 *   (pytorch_utils.py:11-36) -> max_pool2d over nsample (pointnet2_modules.py:251-257)
 * and its backward.  Activations live position-major: row p = (b*npoint + j)*nsample + k, columns =
 * channels, so every 1x1 convolution is a plain row-major GEMM (butd_gemm_grouped, with the previous
 * layer's BatchNorm+ReLU folded into its operand load) and nothing is ever transposed to NCHW.
 * All pointers are device pointers, fp32 unless stated; calls are asynchronous launches on `stream`
 * (hipGraph-capturable); return 0 or a hipError_t.
 */
#ifndef BUTD_SA_H
#define BUTD_SA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *butd_stream_t;

/* X[p, 0:3] = (xyz[b, idx[p]] - new_xyz[b, j]) (/ radius if normalize), X[p, 3:3+C] = feats[b, idx[p], :]
 * xyz (B,N,3); new_xyz (B,np,3); feats point-major with row stride feat_stride floats (may be NULL,
 * C = 0); idx (B,np,ns) int32; X (B*np*ns rows of ldx >= 3+C floats; columns 3+C..ldx-1 are zero-filled:
 * ldx = 3+C rounded up to a multiple of 4 keeps every row 16-byte aligned for the GEMM's float4 path). */
int butd_sa_group(int B, int N, int np, int ns, int C, const float *xyz, const float *new_xyz,
                  const float *feats, long feat_stride, const int *idx, float radius, int normalize,
                  float *X, int ldx, butd_stream_t stream);

/* First shared-MLP layer on a THIN grouped input (K <= 8 channels: SA1's xyz + colour): Z[p,c] =
 * sum_k X[p,k]*W[c,k] (X rows of exactly 8 floats, columns K..7 zero; W (C,K) row-major; no bias) and,
 * when sum/sumsq != NULL, the BatchNorm column sums of Z (double atomics, caller zero-fills) in the same
 * HBM pass.  Replaces a one-slab GEMM + butd_sa_colstats for this layer. */
int butd_sa_thin_conv(long P, int C, int K, const float *X, int ldx, const float *W, float *Z,
                      double *sum, double *sumsq, butd_stream_t stream);

/* Column statistics of Z (P x C): sum[c] += sum_p z, sumsq[c] += sum_p z^2 (double, atomically
 * accumulated: caller zero-fills).  If pool_ns > 0 also the per-group (pool_ns consecutive rows)
 * extrema of every column: zmax/zmin (P/pool_ns x C) and the row offset inside the group of their
 * FIRST occurrence amax/amin (uint8).  C <= 256. */
int butd_sa_colstats(long P, int C, const float *Z, double *sum, double *sumsq, int pool_ns,
                     float *zmax, float *zmin, uint8_t *amax, uint8_t *amin, butd_stream_t stream);

/* BatchNorm bookkeeping of one layer (training): from sum/sumsq over `count` rows (given as `slots`
 * partial copies `slot_stride` doubles apart, slots <= 1: one copy) ->
 * mean[c], rstd[c] = 1/sqrt(var_biased+eps), scale[c] = gamma*rstd, shift[c] = beta - mean*scale;
 * running_mean/var <- (1-momentum)*running + momentum*(mean / var_unbiased); num_batches_tracked += 1.
 * training == 0: scale/shift from the running statistics, nothing updated. */
int butd_sa_bn_finalize(int C, long count, const double *sum, const double *sumsq, int slots,
                        long slot_stride, const float *gamma,
                        const float *beta, float eps, float momentum, int training, float *running_mean,
                        float *running_var, int64_t *num_batches_tracked, float *mean, float *rstd,
                        float *scale, float *shift, butd_stream_t stream);

/* Pooled output of the last layer: for group g=(b,j) and channel c the max over the group of
 * relu(scale*z+shift) is relu(scale*zsel+shift) with zsel = zmax if scale >= 0 else zmin.
 * Writes out_cm (B,C,np) [the module's (B,C,npoint) output], out_pm (B,np,C) [point-major copy for the
 * next level], zsel (G x C) and asel (G x C, uint8). */
int butd_sa_pool_finalize(int B, int np, int C, const float *zmax, const float *zmin,
                          const uint8_t *amax, const uint8_t *amin, const float *scale,
                          const float *shift, float *out_cm, float *out_pm, float *zsel, uint8_t *asel,
                          butd_stream_t stream);

/* Backward through max-pool + ReLU + BatchNorm(train) of the last layer, part 1: per channel
 * S1[c] = sum_g dy, S2[c] = sum_g dy * zhat_sel over the groups whose pooled activation is > 0
 * (dy = d_out_pm[b,j,c], the gradient of the pooled output POSITION-major (B,np,C): coalesced for the
 * channel-per-thread kernels); S1 = dbeta, S2 = dgamma.  Caller zero-fills S1/S2 (double). */
int butd_sa_pool_bwd_stats(int B, int np, int C, const float *d_out_pm, const float *zsel,
                           const float *scale, const float *shift, const float *mean,
                           const float *rstd, double *S1, double *S2, butd_stream_t stream);

/* part 2, in place on Z (P x C) -> dZ:  dZ[p,c] = gamma*rstd * (dy3[p,c] - S1/P - zhat[p,c]*S2/P),
 * dy3[p,c] = d_out_pm[b,j,c] if k(p) == asel[g,c] and the pooled activation is > 0, else 0.
 * training == 0 (BN uses running statistics): dZ = scale * dy3. */
int butd_sa_dz_last(int B, int np, int ns, int C, float *Z, const float *d_out_pm, const float *zsel,
                    const uint8_t *asel, const float *gamma, const float *scale, const float *shift,
                    const float *mean, const float *rstd, const double *S1, const double *S2,
                    int training, butd_stream_t stream);

/* Backward of the LAST shared-MLP layer + max-pool of a level in TRAINING mode, without forming its dense gradient
 * (csrc/sa_last_bwd.hip; replaces butd_sa_dz_last + the weight- / input-gradient products of that layer +
 * butd_sa_mask_stats of the layer below; pointnet2_modules.py:243-257, pytorch_utils.py:11-36).  The BatchNorm backward
 *   dZ3 = s (g - m1 - zhat m2),  s = gamma rstd,  m1 = S1_3 / P,  m2 = S2_3 / P          (butd_sa_pool_bwd_stats)
 * is dense only through m1 and m2; g has ONE non-zero per (group, channel), at the arg-max row.  By linearity
 *   dH2 = dZ3 W3   = [row r: sum over the channels whose arg-max is r of g s W3[c,:]] - H2 A + d,  A = W3^T diag(s m2 rstd) W3
 *   dW3 = dZ3^T H2 = s (T - m1 S^T - m2 rstd (W3 Gram - mu S^T)),  T[c,:] = sum_groups g H2[arg-max row,:],
 *                                                                   S = column sums of H2,  Gram = H2^T H2
 * with H2 = relu(scale2 * Z2 + shift2) recomputed from Z2 (P x C2).  Outputs:
 *   dH2 (P x C2)  the gradient of layer 2's activation ALREADY gated by its ReLU (what butd_sa_dz_mid re-derives),
 *   dW3 (C3 x C2) written (not accumulated),
 *   S1_2, S2_2    the two sums of layer 2's BatchNorm backward (what butd_sa_mask_stats produced), written.
 * Every reduction goes through per-workgroup partials summed in double: no atomics, bit-reproducible.
 * Supported: ns in {16, 32, 64}, (C2, C3) in {(64, 128), (128, 256)} -- butd_sa_last_bwd_supported; scratch sizes
 * (floats / doubles, uninitialised) from butd_sa_last_bwd_scratch. */
int butd_sa_last_bwd_supported(int ns, int C2, int C3);
/* The matching forward: the last layer's product Z3 = relu(scale2 * Z2 + shift2) W3^T, its BatchNorm column sums
 * (sum, sumsq: double, ADDED -- the caller zero-fills) and the per-group extrema of butd_sa_colstats (zmax, zmin: G x C3;
 * amax, amin: G x C3 uint8, first position of the extremum) WITHOUT writing Z3: with butd_sa_last_bwd nothing reads it
 * again.  Same support set.  sched: two uint32 of device memory, ZERO on entry and left zero (the kernel hands its
 * 64-row blocks out through them so that workgroups placed late -- other queues may hold CUs -- take no fixed share);
 * one pair per stream that may run this call concurrently. */
int butd_sa_last_fwd(int B, int np, int ns, int C2, int C3, const float *Z2, const float *scale2,
                     const float *shift2, const float *W3, double *sum, double *sumsq, float *zmax, float *zmin,
                     uint8_t *amax, uint8_t *amin, unsigned int *sched, butd_stream_t stream);
int butd_sa_last_bwd_scratch(long P, int C2, int C3, long *ws_floats, long *ws_doubles);
int butd_sa_last_bwd(int B, int np, int ns, int C2, int C3, const float *Z2, const float *scale2,
                     const float *shift2, const float *mean2, const float *rstd2, const float *W3,
                     const float *d_out_pm, const float *zsel, const uint8_t *asel, const float *scale3,
                     const float *shift3, const float *mean3, const float *rstd3, const double *S1_3,
                     const double *S2_3, float *dH2, float *dW3, double *S1_2, double *S2_2, float *ws_f,
                     double *ws_d, butd_stream_t stream);

/* The FIRST layer's backward when its input needs no gradient (SA1: xyz + colour) and the grouped input has 8 columns,
 * training mode: one pass over (dH1, Z1, X) gives the BatchNorm sums S1, S2 (what butd_sa_mask_stats gave; WRITTEN) and
 * dW1 (C1 x 8, row stride 8, written) = dZ1^T X by linearity of the BatchNorm backward -- neither butd_sa_dz_mid nor the
 * thin weight-gradient product run.  Scratch sizes from butd_sa_first_bwd_scratch.  Per-workgroup partials summed in
 * double: no atomics. */
int butd_sa_first_bwd_scratch(long P, int C1, int Kp, long *ws_floats, long *ws_doubles);
int butd_sa_first_bwd(long P, int C1, int Kp, const float *dH1, const float *Z1, const float *X, const float *scale1,
                      const float *shift1, const float *mean1, const float *rstd1, const float *W1, float *dW1,
                      double *S1, double *S2, float *ws_f, double *ws_d, butd_stream_t stream);

/* SA1 (no input gradient, 64-wide layers, 8 grouped input columns, training mode): layer 2's AND layer 1's backward from
 * layer 2's gated gradient G2 (what butd_sa_last_bwd wrote) in ONE pass over (G2, Z2, Z1, X), with no per-row output:
 *   dZ2 = gamma2 rstd2 (g2 - S1_2/P - zhat2 S2_2/P);  dW2 (C x C) = dZ2^T H1;  dH1 = dZ2 W2;  g1 = dH1 gated by layer 1;
 *   S1_1, S2_1 (written) and dW1 (C x 8) as butd_sa_first_bwd gives them.
 * Replaces butd_sa_dz_mid + the layer's weight- / input-gradient products + butd_sa_first_bwd (2.45 GB -> 0.84 GB at the
 * bench size).  Partials per workgroup, summed in double: no atomics. */
/* Z1 may be NULL: the forward then never wrote it (butd_sa_first_two_fwd) and the kernel recomputes z1 = X W1^T per tile. */
int butd_sa_mid_first_bwd_scratch(long P, int C, int Kp, long *ws_floats, long *ws_doubles);
int butd_sa_mid_first_bwd(long P, int C, int Kp, const float *G2, const float *Z2, const float *Z1, const float *X,
                          const float *gamma2, const float *scale2, const float *shift2, const float *mean2,
                          const float *rstd2, const double *S1_2, const double *S2_2, const float *scale1,
                          const float *shift1, const float *mean1, const float *rstd1, const float *W2, const float *W1,
                          float *dW2, float *dW1, double *S1_1, double *S2_1, float *ws_f, double *ws_d,
                          butd_stream_t stream);

/* SA2-4 (128-wide layers, training mode): layer 2's backward from layer 2's gated gradient G2 (what butd_sa_last_bwd wrote)
 * in ONE pass over (G2, Z2, Z1) per 32-row block (round 5):
 *   dZ2 = gamma2 rstd2 (g2 - S1_2/P - zhat2 S2_2/P);  dW2 (C x C, written) = dZ2^T relu(scale1 Z1 + shift1);
 *   G1 (P x C, written) = (dZ2 W2) gated by layer 1's ReLU;  S1_1[c] = sum G1, S2_1[c] = sum G1 zhat1 (written, double).
 * Replaces butd_sa_dz_mid + the layer's weight- / input-gradient product pair with its statistics epilogue (no dense dZ2,
 * no ungated dH1: 958 MB -> 537 MB at SA2, B = 8); layer 1's backward (butd_sa_dz_mid on G1 + its product pair) follows
 * unchanged.  Partials per workgroup, summed in double: no atomics.  C = 128. */
int butd_sa_mid_wide_bwd_scratch(long P, int C, long *ws_floats, long *ws_doubles);
int butd_sa_mid_wide_bwd(long P, int C, const float *G2, const float *Z2, const float *Z1, const float *gamma2,
                         const float *scale2, const float *shift2, const float *mean2, const float *rstd2,
                         const double *S1_2, const double *S2_2, const float *scale1, const float *shift1,
                         const float *mean1, const float *rstd1, const float *W2, float *G1, float *dW2, double *S1_1,
                         double *S2_1, float *ws_f, double *ws_d, butd_stream_t stream);

/* SA1's first two layers FORWARD without writing Z1 (C = 64, 8 grouped input columns).  Z1 = X W1^T is linear in X, so
 * layer 1's BatchNorm sums follow from the moments of X:  phase 0 -- mom (72 doubles, zero on entry) <- [column sums of X |
 * X^T X], sum1[c] = W1[c] . SX, sumsq1[c] = W1[c]^T XX W1[c] (written; the caller then runs butd_sa_bn_finalize as usual).
 * phase 1 -- with layer 1's scale / shift: Z2 = relu(scale1 * (X W1^T) + shift1) W2^T (z1 formed per tile, 8 multiply-adds
 * per element), written, and its column sums ADDED to sum2 / sumsq2 (zero on entry).  Replaces butd_sa_thin_conv, the
 * layer-2 product launch and butd_sa_colstats; the backward of these layers is butd_sa_mid_first_bwd with Z1 = NULL. */
int butd_sa_first_two_fwd(long P, int C, int Kp, const float *X, const float *W1, const float *W2, double *mom,
                          double *sum1, double *sumsq1, const float *scale1, const float *shift1, float *Z2, double *sum2,
                          double *sumsq2, int phase, butd_stream_t stream);

/* The FIRST layer of a level WITH input features (SA2-SA4) without the grouped input X (round 6; csrc/sa_first_linear.hip).
 * A grouped row copies a point's features (pointnet2_utils.py:317-376), so with W1 = [Wx | Wf] (3 | C columns)
 *   Z1[p, :] = Y[b, idx[p], :] + dxyz[p] Wx^T,   Y = feats Wf^T  (B*N x C1: the caller's product over the level's N points),
 *   dxyz[p]  = (xyz[b, idx[p]] - new_xyz[b, j]) (/ radius if normalize)
 * butd_sa_first_linear_fwd writes Z1 (P x C1) and, when sum / sumsq != NULL, adds its BatchNorm column sums (double atomics
 * into `slots` private copies `slot_stride` doubles apart, caller zero-fills, as butd_gemm_problem.col_slots).  W1 (C1 rows of
 * ldw floats): only its first three columns are read.  C1 in {64, 128, 256}.
 * butd_sa_first_linear_bwd: with dZ1 = gamma rstd (g - S1/P - zhat S2/P) of the GATED gradient G1 (training; scale * g
 * otherwise: the arithmetic of butd_sa_dz_mid), over the inverted neighbour lists (butd_sa_inverse_index):
 *   T[b, n, :] = sum of dZ1 over the grouped rows that copy point n   (B*N x C1, written)
 *   dWx[c, i]  = sum_p dZ1[p, c] dxyz[p, i]                             (C1 x 3, row stride ld_dwx, written; partials in ws,
 *                                                                        summed in double in a fixed order)
 * The caller's products over B*N rows finish the layer: d_feats = T Wf, dWf = T^T feats.  Neither dZ1 nor the grouped
 * input gradient (P x (3 + C)) exists in memory.  ws: butd_sa_first_linear_bwd_scratch floats. */
int butd_sa_first_linear_supported(int C1);
int butd_sa_first_linear_fwd(int B, int N, int np, int ns, int C1, const float *xyz, const float *new_xyz, const int *idx,
                             float radius, int normalize, const float *Y, const float *W1, long ldw, float *Z1, double *sum,
                             double *sumsq, int slots, long slot_stride, butd_stream_t stream);
int butd_sa_first_linear_bwd_scratch(int B, int N, int C1, long *ws_floats);
int butd_sa_first_linear_bwd(int B, int N, int np, int ns, int C1, const float *xyz, const float *new_xyz,
                             const int *start, const int *list, float radius, int normalize, const float *G1,
                             const float *Z1, const float *gamma1, const float *scale1, const float *shift1,
                             const float *mean1, const float *rstd1, const double *S1, const double *S2, int training,
                             float *T, float *dWx, long ld_dwx, float *ws, butd_stream_t stream);

/* Hidden layers, part 1 (read-only pass over dH, Z (P x C)): with g = dH * [scale*z+shift > 0],
 * S1[c] += sum_p g, S2[c] += sum_p g*zhat (double, caller zero-fills). */
int butd_sa_mask_stats(long P, int C, const float *dH, const float *Z, const float *scale,
                       const float *shift, const float *mean, const float *rstd, double *S1,
                       double *S2, butd_stream_t stream);

/* part 2, in place on dH -> dZ = gamma*rstd*(g - S1/P - zhat*S2/P) with g re-derived from dH and Z as in
 * part 1 (training == 0: dZ = scale*g). */
int butd_sa_dz_mid(long P, int C, float *g, const float *Z, const float *gamma, const float *scale,
                   const float *shift, const float *mean, const float *rstd, const double *S1, const double *S2,
                   int training, butd_stream_t stream);

/* d_feats_pm[b, idx[p], c] += dX[p, 3 + c]  (dX: P rows of ldx >= 3+C floats, d_feats_pm (B,N,C)
 * point-major, caller zero-fills). */
int butd_sa_scatter_rows(int B, int N, int np, int ns, int C, const float *dX, int ldx, const int *idx,
                         float *d_feats_pm, butd_stream_t stream);

/* The same sum as a GATHER.  butd_sa_inverse_index inverts the neighbour lists idx (B,np,ns) over the level's N input points:
 * start (B*N + 1 ints; start[b*N + i] .. start[b*N + i + 1] = the slice of `list` for point i of batch b) and list (B*np*ns
 * ints: grouped row indices p, ascending inside a slice: the result is unique); count (B*N ints) is scratch, ZERO on entry,
 * the slice lengths on return.  It depends on the coordinates alone: Pointnet2Backbone.plan builds it for the next batch off the critical
 * path.  butd_sa_gather_rows then writes (not adds) d_feats_pm[b, i, c] = sum over the slice of dX[p, 3 + c]: no float
 * atomics (butd_sa_scatter_rows issues B*np*ns*C of them: 3.3e7 at SA2, B = 8).  C <= 256 and 256 % C == 0. */
int butd_sa_inverse_index(int B, int N, int np, int ns, const int *idx, int *count, int *start, int *list,
                          butd_stream_t stream);
int butd_sa_gather_rows(int B, int N, int C, const float *dX, int ldx, const int *start, const int *list,
                        float *d_feats_pm, butd_stream_t stream);

/* The whole set-abstraction level for INFERENCE in one kernel (csrc/sa_fused.hip): ball-query neighbourhoods
 * gathered into LDS tiles, three 1x1 convolutions with folded BatchNorm + ReLU run LDS -> MFMA -> LDS, max-pool
 * over nsample; only the pooled features are written.  Replaces QueryAndGroup + SharedMLP + F.max_pool2d
 * (pointnet2_utils.py:317-376, pytorch_utils.py:11-36, pointnet2_modules.py:243-257) in eval mode.
 *   xyz (B,N,3), new_xyz (B,np,3), feats point-major with row stride feat_stride (NULL when C = 0),
 *   idx (B,np,ns) int32, ns in {16, 32, 64};  layer l: w[l] (c_out[l] x k_l, row stride ldw[l]; k_0 = 3+C,
 *   k_l = c_out[l-1]), y = relu(scale[l] * (x . w^T) + shift[l])  (scale = gamma / sqrt(var + eps),
 *   shift = beta - mean * scale);  c_out[0], c_out[1] in {32, 64, 96, 128}, c_out[2] a multiple of 32 <= 256;
 *   out_pm (B,np,c_out[2]) and / or out_cm (B,c_out[2],np) (either may be NULL). */
int butd_sa_fused_eval(int B, int N, int np, int ns, int C, const float *xyz, const float *new_xyz,
                       const float *feats, long feat_stride, const int *idx, float radius, int normalize,
                       const int *c_out, const float *const *w, const long *ldw,
                       const float *const *scale, const float *const *shift, float *out_pm, float *out_cm,
                       butd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BUTD_SA_H */
