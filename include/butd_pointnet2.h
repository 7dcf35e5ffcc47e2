/*
 * butd_pointnet2.h -- C ABI of the MI355X (gfx950) PointNet++ operator library.
 *
 * Drop-in boundary for the reference's native extension `pointnet2._ext`
 * (nickgkan/butd_detr, pointnet2/_ext_src).  Each entry point replaces one
 * `*_kernel_wrapper` the reference's C++ shims call; argument order, meaning and
 * tensor layouts are the reference's, plus an explicit stream and an error code:
 *
 *   - all pointers are DEVICE pointers to dense, contiguous buffers (fp32 data, int32 indices);
 *   - `stream` is a hipStream_t (NULL = the legacy default stream); launches are asynchronous,
 *     never synchronise the host and never allocate, so they can be captured into a hipGraph;
 *   - the return value is 0 on success or the hipError_t of the failed launch (the reference
 *     prints and exit(-1)s instead, include/cuda_utils.h:35-44); `butd_error_string` names it;
 *   - inputs are never written; outputs are fully overwritten unless stated otherwise.
 *
 * No torch types appear here: the Python side (butd_detr_amd/pointnet2_ext.py) binds these
 * with ctypes and passes `tensor.data_ptr()` / the current torch stream handle.
 */
#ifndef BUTD_POINTNET2_H
#define BUTD_POINTNET2_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *butd_stream_t; /* hipStream_t */

#define BUTD_POINTNET2_ABI_VERSION 1

int butd_pointnet2_abi_version(void);
const char *butd_error_string(int err);

/* include/cuda_utils.h:18-24 opt_n_threads(): 2^floor(log2(work_size)) clamped to [1,512].  It is the
 * reference's FPS block size and therefore fixes the FPS tie-break rule. */
int butd_opt_n_threads(int work_size);

/*
 * Replaces furthest_point_sampling_kernel_wrapper (src/sampling.cpp:16-18, src/sampling_gpu.cu:180-234).
 * dataset (b,n,3) f32; temp (b,n) f32 scratch (the reference pre-fills it with 1e10,
 * src/sampling.cpp:78-80; this implementation initialises whatever part of it it uses, so the
 * caller may pass uninitialised memory); idxs (b,m) i32, idxs[:,0] = 0.
 * Bit-exact with the reference algorithm incl. the |p|^2 <= 1e-3 skip and the 512-slot
 * tree-reduction tie-break.  Requires 1 <= m, n < 2^31 / 3.
 */
int butd_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp,
                                 int *idxs, butd_stream_t stream);

/* The parallel half of butd_furthest_point_sampling for clouds that are already in furthest-point order (the
 * backbone's levels 2-4, models/backbone_module.py:131-140: "inds == arange(m)").  For 2 <= m <= 2048, m <= n <= 8192
 * it leaves a verdict per scene in ((int *)temp)[scene * n]: 0 <=> the reference algorithm (sampling_gpu.cu:100-117,
 * incl. the |p|^2 <= 1e-3 skip and the tree reduction's tie order) selects 0, 1, ..., m-1; non-zero otherwise.  The
 * test is exact: every sample must win its iteration against every other competing point, ties by the tie order.
 * n*m + m*m/2 independent distance evaluations instead of m-1 dependent iterations.  butd_furthest_point_sampling
 * runs it first whenever temp is given; the serial kernel of a scene with verdict 0 writes 0..m-1 and returns.
 * Other shapes: no-op. */
int butd_fps_prefix_check(int b, int n, int m, const float *dataset, float *temp, butd_stream_t stream);

/* Bytes of extra device workspace the pruned large-cloud FPS path wants for (b, n); 0 = none needed.
 * Pass it to butd_furthest_point_sampling_ws; without it the streaming kernel is used. */
size_t butd_fps_workspace_bytes(int b, int n);
int butd_furthest_point_sampling_ws(int b, int n, int m, const float *dataset, float *temp,
                                    int *idxs, void *workspace, size_t workspace_bytes,
                                    butd_stream_t stream);

/* Replaces gather_points_kernel_wrapper (src/sampling.cpp:9-11, sampling_gpu.cu:27-35).
 * points (b,c,n), idx (b,npoints) -> out (b,c,npoints). */
int butd_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx,
                       float *out, butd_stream_t stream);

/* Replaces gather_points_grad_kernel_wrapper (src/sampling.cpp:12-14, sampling_gpu.cu:54-62).
 * grad_out (b,c,npoints), idx (b,npoints) -> grad_points (b,c,n): ACCUMULATES with atomic adds into
 * grad_points, which the caller zero-fills first (as src/sampling.cpp:52-54 does). */
int butd_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                            const int *idx, float *grad_points, butd_stream_t stream);

/* Replaces query_ball_point_kernel_wrapper (src/ball_query.cpp:9-11, ball_query_gpu.cu:51-59).
 * new_xyz (b,m,3) centres, xyz (b,n,3) -> idx (b,m,nsample) i32: the first `nsample` point indices
 * (ascending) with d2 < radius*radius (fp32, strict), padded with the first hit; all-zero rows for
 * centres without a hit.  idx needs no pre-zeroing.  Bit-exact. */
int butd_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                    const float *xyz, int *idx, butd_stream_t stream);

/* Spatially pruned variant for large clouds (uniform grid + per-centre hit bitmap; same results, bit for
 * bit).  butd_ball_query_workspace_bytes: bytes of device workspace it wants for (b, n, m), 0 = the
 * streaming kernel is the better one for this shape.  butd_ball_query_ws takes the pruned path whenever
 * the workspace is 16-byte aligned and holds >= 24*b*n + 264192*b bytes and n <= 2^18; otherwise it is
 * butd_ball_query. */
size_t butd_ball_query_workspace_bytes(int b, int n, int m);
int butd_ball_query_ws(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                       const float *xyz, int *idx, void *workspace, size_t workspace_bytes,
                       butd_stream_t stream);

/* Replaces group_points_kernel_wrapper (src/group_points.cpp:9-11, group_points_gpu.cu:35-44).
 * points (b,c,n), idx (b,npoints,nsample) -> out (b,c,npoints,nsample). */
int butd_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                      const int *idx, float *out, butd_stream_t stream);

/* Replaces group_points_grad_kernel_wrapper (src/group_points.cpp:13-15, group_points_gpu.cu:71-80).
 * grad_out (b,c,npoints,nsample), idx -> grad_points (b,c,n): atomic accumulation into a
 * caller-zeroed buffer. */
int butd_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                           const int *idx, float *grad_points, butd_stream_t stream);

/* Replaces three_nn_kernel_wrapper (src/interpolate.cpp:9-10, interpolate_gpu.cu:66-73).
 * unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3) f32 squared distances, idx (b,n,3) i32, ascending
 * by distance, ties to the lower index; slots beyond m stay (+inf, 0).  Bit-exact. */
int butd_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                  int *idx, butd_stream_t stream);

/* Replaces three_interpolate_kernel_wrapper (src/interpolate.cpp:11-13, interpolate_gpu.cu:108-117).
 * points (b,c,m), idx (b,n,3), weight (b,n,3) -> out (b,c,n) = (p1*w1 + p2*w2) + p3*w3 in fp32. */
int butd_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                           const float *weight, float *out, butd_stream_t stream);

/* Replaces three_interpolate_grad_kernel_wrapper (src/interpolate.cpp:14-17, interpolate_gpu.cu:150-159).
 * grad_out (b,c,n), idx, weight -> grad_points (b,c,m): atomic accumulation into a caller-zeroed buffer. */
int butd_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                const float *weight, float *grad_points, butd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BUTD_POINTNET2_H */
