/*
 * butd_rowwise.h -- C ABI of small row-wise operators that the reference writes as chains of stock elementwise /
 * reduction ops (gfx950).  Each chain is 3-13 launches of a few microseconds in the captured step; here one each.
 *
 *   butd_l2_normalize_fwd / _bwd   F.normalize(x, p=2, dim=-1) of the contrastive-alignment projections
 *                                  (models/bdetr.py:263-268, 300-305, 289-293): y = x / max(||x||_2, eps)
 *   butd_three_nn_weights          the inverse-distance weights of PointnetFPModule
 *                                  (pointnet2/pointnet2_modules.py:392-396; ThreeNN returns sqrt of the squared
 *                                  distances, pointnet2_utils.py:142)
 *
 * Device pointers, fp32, dense row-major; asynchronous launches on `stream` (hipGraph-capturable); 0 or a hipError_t.
 */
#ifndef BUTD_ROWWISE_H
#define BUTD_ROWWISE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *butd_stream_t;

/* y[r] = x[r] / max(||x[r]||_2, eps) for rows of `cols` floats (cols a multiple of 4, <= 1024). */
int butd_l2_normalize_fwd(long rows, int cols, const float *x, float eps, float *y, butd_stream_t stream);

/* dx of the above given g = dL/dy (the derivative torch.autograd composes from norm -> clamp_min -> div):
 *   n = ||x||, c = max(n, eps):   dx = g / c - [n >= eps] * x * (x . g) / (c * c * n).   */
int butd_l2_normalize_bwd(long rows, int cols, const float *x, const float *g, float eps, float *dx,
                          butd_stream_t stream);

/* dist2 (rows, 3) squared distances of the three nearest neighbours -> dist (rows, 3) = sqrt(dist2) and
 * weight (rows, 3) = r / (r0 + r1 + r2), r = 1 / (dist + 1e-8); the operations and their order are the reference's
 * (fp32, no contraction), so the values equal the stock op chain's to the last ulp of the device's sqrt / divide.  dist may be NULL. */
int butd_three_nn_weights(long rows, const float *dist2, float *dist, float *weight, butd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BUTD_ROWWISE_H */
