/*
 * butd_panel.h -- C ABI of the row-panel chain kernels (gfx950, fp32 MFMA): several dependent row-wise operators of
 * the attention / FFN stack in ONE launch.
 *
 * The reference's layers (models/encoder_decoder_layers.py:75-124, 166-186, 340-406) are chains of operators that
 * act on every row of a (rows, 288) activation independently:
 *     out-projection -> dropout -> + residual -> LayerNorm -> (+ query_pos) -> the NEXT block's query projection
 *     Linear -> ReLU -> Dropout -> Linear -> Dropout -> + residual -> LayerNorm            (the FFN)
 * A 2048-row decoder block makes each of them a 10 us launch whose floor is the launch itself.  A workgroup of the
 * panel kernel owns R = 16 or 32 rows, keeps them in LDS across the whole chain (the product of one stage is the A
 * operand of the next), streams the weights of each stage from L2 straight into MFMA operand registers, and writes
 * only what the backward pass or another kernel reads.  Every stage is
 *     z   = (IN . W^T + bias) * scale                      IN = an LDS panel (rows x K), W (N x K) row-major
 *     z   = relu(z)                        [relu]
 *     pre <- z                             [pre != NULL: the tensor butd_add_dropout_layernorm_bwd calls x]
 *     z   = dropout(z; p, site)            [drop_p > 0: counter-hash of element row * N + col, as the GEMM epilogue
 *                                           and butd_add_dropout_layernorm_fwd draw it -- backward regenerates it]
 *     y   = LayerNorm(res + z)             [ln: res from global memory or from an LDS panel; saves mean / rstd]
 *     out <- y;  out_pos <- y + pos        [optional global stores; both may also stay in LDS for later stages]
 * All pointers are device pointers, fp32, dense row-major; asynchronous, hipGraph-capturable; returns 0 or a hipError_t.
 */
#ifndef BUTD_PANEL_H
#define BUTD_PANEL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *butd_stream_t;

#define BUTD_PANEL_MAX_STAGES 6
#define BUTD_PANEL_MAX_BUFFERS 4
#define BUTD_PANEL_MAX_COLS 288 /* widest panel (N and K of every stage, and in_cols) */

typedef struct {
  const float *w;    /* (N, K) row-major: nn.Linear.weight */
  const float *bias; /* N, or NULL */
  int N, K;          /* N a multiple of 16, <= BUTD_PANEL_MAX_COLS; K in {128, 256, 288} (the K loop is straight-line
                        code per depth: d_model 288, dim_feedforward 256 of the reference's configuration) */
  float scale;
  int in_buf; /* LDS panel holding the A operand (width K) */
  int relu;
  float drop_p;
  uint32_t drop_site;
  float *pre; /* rows x N or NULL */
  int ln;
  const float *res; /* rows x N (used when res_buf < 0) */
  int res_buf;      /* LDS panel holding the residual, or -1 */
  const float *gamma, *beta;
  float eps;
  float *mean, *rstd; /* rows each (ln != 0) */
  float *out;         /* rows x N or NULL */
  int out_buf;        /* LDS panel that receives the stage result (!= in_buf, != res_buf); -1 = none: allowed for a
                         stage without LayerNorm whose result no later stage reads (it then ends without a barrier) */
  const float *pos;   /* rows x N or NULL (LayerNorm stages only) */
  float *out_pos;     /* rows x N or NULL (needs pos) */
  int pos_buf;        /* LDS panel that receives result + pos, or -1 (needs pos; != out_buf) */
} butd_panel_stage;

/* Runs `nstages` stages over the rows of `in` (rows x in_cols, dense; in_cols a multiple of 4), loaded into LDS panel
 * `in_buf`.  in_pos != NULL: in + in_pos is loaded into panel `in_sum_buf` as well and, if in_sum != NULL, stored to it
 * (the `src + pos` a block without a predecessor kernel forms itself).  nbuf = number of LDS panels used (<= 4).
 * rng_counter: device step counter of the dropout hash (may be NULL when no stage drops). */
int butd_panel_chain(int rows, const float *in, int in_cols, int in_buf, const float *in_pos, float *in_sum,
                     int in_sum_buf, const butd_panel_stage *stages, int nstages, int nbuf,
                     const uint64_t *rng_counter, butd_stream_t stream);

/* Tuning hook: force the panel height (16 or 32; 0 = built-in choice: 16 rows below 4096 rows, 32 from there). */
int butd_panel_set_rows(int rows_per_workgroup);

#ifdef __cplusplus
}
#endif
#endif /* BUTD_PANEL_H */
