/*
 * butd_mlp.h -- C ABI of the 1x1-convolution + BatchNorm1d + ReLU (+ Dropout) chains around the decoder
 * (gfx950).
 *
 * Replaces, for the reference's
 *   ThreeLayerMLP            (models/modules.py:89-108;  three per ClsAgnosticPredictHead, :111-180),
 *   PointsObjClsModule       (models/modules.py:19-49),
 *   PositionEmbeddingLearned (models/modules.py:52-67),
 * the stock chain  Conv1d(k=1) -> BatchNorm1d -> ReLU -> Dropout -> ... -> Conv1d(k=1)  and its backward.
 * Activations are position-major (P = B*L rows, channels contiguous), so every Conv1d is one
 * butd_gemm_grouped problem (include/butd_attention.h): its epilogue accumulates the BatchNorm batch
 * statistics (col_sum / col_sumsq), and the NEXT product applies BatchNorm + ReLU + Dropout while it
 * stages the operand (a_chan_scale/shift, a_drop_*), so normalised / activated / dropped tensors never
 * exist in HBM.  G parallel chains that share their input (the three heads of a predict head) keep their
 * hidden activations side by side in one (P, G*H) matrix and run as grouped launches.
 * What lives here is the glue between the products.  All pointers are device pointers, fp32 unless
 * stated; calls are asynchronous launches on `stream` (hipGraph-capturable); return 0 or a hipError_t.
 */
#ifndef BUTD_MLP_H
#define BUTD_MLP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *butd_stream_t;

#define BUTD_MLP_MAX_SEGMENTS 8

/* BatchNorm parameters of one chain (one column segment of the concatenated hidden matrix). */
typedef struct {
  const float *gamma, *beta;
  float *running_mean, *running_var;
  int64_t *num_batches_tracked; /* may be NULL */
} butd_bn_segment;

/* BatchNorm bookkeeping of one hidden layer of nseg <= 8 chains, Cseg channels each (training: batch
 * statistics from sum/sumsq over `count` rows, biased variance for the normalisation, unbiased for
 * running_var, running <- (1-momentum)*running + momentum*batch, num_batches_tracked += 1; eval:
 * running statistics, nothing updated).  Writes mean, rstd, scale = gamma*rstd, shift = beta-mean*scale
 * for all nseg*Cseg concatenated columns. */
int butd_mlp_bn_finalize(int nseg, int Cseg, long count, const double *sum, const double *sumsq,
                         const butd_bn_segment *segs, float eps, float momentum, int training,
                         float *mean, float *rstd, float *scale, float *shift, butd_stream_t stream);

/* Backward through Dropout + ReLU of a hidden layer, in place on dH (P x C, row stride ld):
 *   g = dH * keep/(1-p) * [scale*z+shift > 0];   S1[c] += sum_p g,  S2[c] += sum_p g*zhat
 * (double atomics, caller zero-fills; S1 = dbeta, S2 = dgamma).  The keep mask is the one the forward
 * product drew: element (p, c) of column segment i = c / seg_cols is keyed by site0 + i and index
 * p*ld + (c - i*seg_cols), i.e. its offset from the segment's base pointer.  C % 4 == 0. */
int butd_mlp_mask_stats(long P, int C, long ld, float *dH, const float *Z, const float *scale,
                        const float *shift, const float *mean, const float *rstd, float drop_p,
                        uint32_t site0, int seg_cols, const uint64_t *rng_counter, double *S1,
                        double *S2, butd_stream_t stream);

/* Backward through BatchNorm, in place on g -> dZ = scale*(g - S1/P - zhat*S2/P)  (training == 0:
 * running statistics, dZ = scale*g).  S1f / S2f (may be NULL): fp32 copies of the two sums, i.e. the BatchNorm
 * bias / weight gradients in the parameters' dtype, written by the same launch. */
int butd_mlp_dz(long P, int C, long ld, float *g, const float *Z, const float *scale,
                const float *mean, const float *rstd, const double *S1, const double *S2,
                int training, float *S1f, float *S2f, butd_stream_t stream);

/* out[p, c] = relu(scale[c] * Z[p, c] + shift[c])  (P x C, row stride ld for both): the materialised
 * output of a chain that ENDS in BatchNorm + ReLU (the SharedMLP of PointnetFPModule,
 * pointnet2_modules.py:371-416).  C % 4 == 0. */
int butd_mlp_bn_relu_apply(long P, int C, long ld, const float *Z, const float *scale,
                           const float *shift, float *out, butd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BUTD_MLP_H */
