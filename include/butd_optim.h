/*
 * butd_optim.h -- C ABI of the flat AdamW update (gfx950).
 *
 * The reference optimises with torch.optim.AdamW over three parameter groups (main_utils.py:258-283)
 * after clip_grad_norm_(0.1) (main_utils.py:432-436).  With all parameters, gradients and moments
 * packed into contiguous fp32 buffers the whole update is ONE streaming kernel (28 B/parameter) instead
 * of the multi-tensor path's 18 launches at a tenth of the HBM rate.
 */
#ifndef BUTD_OPTIM_H
#define BUTD_OPTIM_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef void *butd_stream_t;

/* AdamW (decoupled weight decay, torch semantics) on p[begin:end) of the packed buffers:
 *   g' = g * *grad_scale (device scalar: the clip coefficient, NULL = 1)
 *   p *= 1 - lr*wd;  m = b1*m + (1-b1)*g';  v = b2*v + (1-b2)*g'^2;
 *   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps),   t = *step (device scalar, float, >= 1).
 * hyper: NULL, or a device pointer to {lr, weight_decay} that REPLACE the by-value arguments -- the reference
 * steps an LR scheduler every iteration (main_utils.py:438); a launch captured in a hipGraph bakes by-value
 * arguments in, a device-resident pair is re-read by every replay. */
int butd_adamw_flat(float *p, const float *g, float *m, float *v, long begin, long end, float lr,
                    float beta1, float beta2, float eps, float weight_decay, const float *step,
                    const float *grad_scale, const float *hyper, butd_stream_t stream);

/* clip_grad_norm_ (main_utils.py:432-436) on the packed gradient buffer g[0:n): writes the coefficient the AdamW
 * kernel applies,
 *   norm = ||g||_2 / grad_div;   *grad_scale = min(max_norm / (norm + 1e-6), 1) / grad_div   (max_norm <= 0: 1 / grad_div)
 * (grad_div: the buffer holds the SUM over that many ranks), and the norm itself to *norm_out (may be NULL).
 * Two launches: per-workgroup fp64 sums of squares into `workspace` (butd_clip_workspace_bytes() bytes, no
 * initialisation needed), then one workgroup folds them -- no zero-initialised semaphore, no atomics, so the
 * result does not depend on a memset having run (torch's multi-block reduction inside a
 * replayed hipGraph). */
size_t butd_clip_workspace_bytes(void);
int butd_clip_coefficient(const float *g, long n, float max_norm, float grad_div, void *workspace,
                          float *grad_scale, float *norm_out, butd_stream_t stream);

/* Gradient packing: dst[dst_off[i] : dst_off[i] + numel[i]) = src[i][0 : numel[i]) for n segments in ONE
 * launch (the step gathers ~330 freshly produced parameter gradients into the flat all-reduce / AdamW
 * buffer; torch's multi-tensor copy takes 10 launches and 0.28 ms for the 85.7 MB).  table: device int64
 * array [src pointers (n) | dst offsets in floats (n) | numel (n) | first workgroup of every segment
 * (n + 1)], a workgroup copies BUTD_GATHER_CHUNK consecutive floats of one segment. */
#define BUTD_GATHER_CHUNK 4096
int butd_gather_segments(int n, const int64_t *table, float *dst, butd_stream_t stream, long total_blocks);
/* out[0:numel) = srcs[0] + ... + srcs[n-1] (1 <= n <= 8 device pointers in a HOST array, 16-byte aligned): the summed
 * gradient of a tensor that feeds several blocks -- what torch's autograd engine does with n-1 pairwise adds
 * (torch/csrc/autograd/input_buffer.cpp) in one pass.  Used by butd_detr_amd/fan_out.py for the tensors the
 * reference hands to several layers (query_pos: encoder_decoder_layers.py:356-404; the encoder outputs:
 * bdetr.py:277-299). */
int butd_sum_tensors(int n, const float *const *srcs, long numel, float *out, butd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
