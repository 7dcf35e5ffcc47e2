// augment_ops.hip -- device-side scene augmentation (include/butd_augment.h), gfx950.
// Bandwidth-trivial kernels (one pass over B x N x 6 floats); the point is that the cloud never leaves HBM.
// -ffp-contract=off: every numpy step is one rounding.
#include <hip/hip_runtime.h>

#include "zero_fill.h"
#include <math.h>
#include <stdint.h>

#include "../../include/butd_augment.h"

namespace {

__device__ inline uint32_t mix32(uint64_t x) {  // splitmix64 finaliser
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebull;
  x ^= x >> 31;
  return (uint32_t)(x >> 16);
}
__device__ inline double uniform01(uint64_t seed, uint64_t index) {
  return (double)mix32(seed + 0x9e3779b97f4a7c15ull * (index + 1)) * (1.0 / 4294967296.0);
}

// (3x3 double) @ (float xyz), rounded to float: pc[:, :3] = rot(pc[:, :3], theta)
__device__ inline void rotate_f32(const double *m, float &x, float &y, float &z) {
  const double px = x, py = y, pz = z;
  const float nx = (float)(m[0] * px + m[1] * py + m[2] * pz);
  const float ny = (float)(m[3] * px + m[4] * py + m[5] * pz);
  const float nz = (float)(m[6] * px + m[7] * py + m[8] * pz);
  x = nx;
  y = ny;
  z = nz;
}

__global__ __launch_bounds__(256) void augment_points_kernel(
    int N, int ld, int has_color, const float *__restrict__ pc_in, const butd_scene_augment *__restrict__ params,
    const double *__restrict__ noise, const double *__restrict__ gain, double mr, double mg, double mb,
    uint64_t seed, float *__restrict__ pc_out) {
  const int b = blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const butd_scene_augment &A = params[b];
  const size_t row = (size_t)b * N + n;
  const float *src = pc_in + row * ld;
  float *dst = pc_out + row * ld;
  float x = src[0], y = src[1], z = src[2];
  if (A.flip_yz) x = -x;
  if (A.flip_xz) y = -y;
  rotate_f32(A.rz, x, y, z);
  rotate_f32(A.rx, x, y, z);
  rotate_f32(A.ry, x, y, z);
  float v[3] = {x, y, z};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double nz = noise ? noise[row * 3 + a] : uniform01(seed, row * 6 + a) * 5e-3;
    v[a] = (float)((double)v[a] + nz);
    v[a] = (float)((double)v[a] + A.shift[a]);
    v[a] = v[a] * (float)A.scale;   // `pc[:, :3] *= scale` with a PYTHON float (:396): numpy multiplies in float32
    dst[a] = v[a];
  }
  const double mean[3] = {mr, mg, mb};
  for (int c = 3; c < ld; ++c) {
    float col = src[c];
    if (has_color && c < 6) {
      const int a = c - 3;
      const double g = gain ? gain[row * 3 + a] : 0.98 + 0.04 * uniform01(seed, row * 6 + 3 + a);
      col = (float)((double)col + mean[a]);
      col = (float)((double)col * g);
      col = (float)((double)col - mean[a]);
    }
    dst[c] = col;
  }
}

__device__ inline void rotate_f64(const double *m, double &x, double &y, double &z) {
  const double nx = m[0] * x + m[1] * y + m[2] * z;
  const double ny = m[3] * x + m[4] * y + m[5] * z;
  const double nz = m[6] * x + m[7] * y + m[8] * z;
  x = nx;
  y = ny;
  z = nz;
}

__global__ __launch_bounds__(256) void augment_boxes_kernel(int total, int D, const float *__restrict__ in,
                                                            const butd_scene_augment *__restrict__ params,
                                                            float *__restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;  // b * D + d
  if (i >= total) return;
  const butd_scene_augment &A = params[i / D];
  double c[3], h[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    c[a] = in[(size_t)i * 6 + a];
    h[a] = (double)in[(size_t)i * 6 + 3 + a] / 2;
  }
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = 0; k < 8; ++k) {  // box2points: corner k = (x: k&2, y: k&1, z: k&4) -- the hull is order-free
    double x = (k & 2) ? c[0] + h[0] : c[0] - h[0];
    double y = (k & 1) ? c[1] + h[1] : c[1] - h[1];
    double z = (k & 4) ? c[2] + h[2] : c[2] - h[2];
    rotate_f64(A.rz, x, y, z);
    rotate_f64(A.rx, x, y, z);
    rotate_f64(A.ry, x, y, z);
    if (A.flip_yz) x = -x;
    if (A.flip_xz) y = -y;
    double p[3] = {x, y, z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      p[a] = (p[a] + A.shift[a]) * A.scale;
      lo[a] = fmin(lo[a], p[a]);
      hi[a] = fmax(hi[a], p[a]);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    out[(size_t)i * 6 + a] = (float)((lo[a] + hi[a]) / 2);
    out[(size_t)i * 6 + 3 + a] = (float)(hi[a] - lo[a]);
  }
}

__device__ inline unsigned ordered_key(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float ordered_value(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// scratch[b][t][0..2] = max key(-coord), [3..5] = max key(coord); zero = "no point"
__global__ __launch_bounds__(256) void instance_hull_kernel(int N, int ldp, int G, const float *__restrict__ pc,
                                                            const int64_t *__restrict__ instance,
                                                            uint32_t *__restrict__ scratch) {
  const int b = blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const long long t = instance[(size_t)b * N + n];
  if (t < 0 || t >= G) return;
  const float *p = pc + ((size_t)b * N + n) * ldp;
  uint32_t *s = scratch + ((size_t)b * G + t) * 6;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    atomicMax(s + a, ordered_key(-p[a]));
    atomicMax(s + 3 + a, ordered_key(p[a]));
  }
}

// Target objects given as POINT LISTS (the resident scene store: all objects of all scenes as one CSR array): for target
// slot t of sample b, object target_ids[b][t] of scene scene[b]: (i) point_instance_label[b][p] = t for its points --
// the reference assigns in slot order, so the LAST slot that contains a point wins (joint_det_dataset.py:507-508) =
// a max over slots; (ii) the hull of ALL its points (Scan.get_object_bbox, visual_data_handlers.py:217-219), also of
// points a later slot claims.
__global__ __launch_bounds__(256) void object_scan_kernel(int N, int ldp, int G, const int *__restrict__ scene,
                                                          const long long *__restrict__ obj_ptr, long long ptr_stride,
                                                          const int *__restrict__ obj_points,
                                                          const int *__restrict__ target_ids,
                                                          const float *__restrict__ pc, long long *__restrict__ label,
                                                          uint32_t *__restrict__ scratch) {
  const int b = blockIdx.z, t = blockIdx.y;
  const int tid = target_ids[(size_t)b * G + t];
  if (tid < 0) return;
  const long long *ptr = obj_ptr + (size_t)scene[b] * ptr_stride;
  const long long begin = ptr[tid], end = ptr[tid + 1];
  uint32_t *s = scratch + ((size_t)b * G + t) * 6;
  for (long long i = begin + (long long)blockIdx.x * 256 + threadIdx.x; i < end; i += (long long)gridDim.x * 256) {
    const int n = obj_points[i];
    if (n < 0 || n >= N) continue;
    if (label) atomicMax(label + (size_t)b * N + n, (long long)t);
    const float *p = pc + ((size_t)b * N + n) * ldp;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      atomicMax(s + a, ordered_key(-p[a]));
      atomicMax(s + 3 + a, ordered_key(p[a]));
    }
  }
}

__global__ __launch_bounds__(256) void instance_box_kernel(int total, const uint32_t *__restrict__ scratch,
                                                           const double *__restrict__ jitter,
                                                           float *__restrict__ out, float *__restrict__ mask) {
  const int i = blockIdx.x * 256 + threadIdx.x;  // b * G + t
  if (i >= total) return;
  const uint32_t *s = scratch + (size_t)i * 6;
  if (s[3] == 0u) {  // no point carries this id: padding slot (joint_det_dataset.py:518-520)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      out[(size_t)i * 6 + a] = 1000.0f;
      out[(size_t)i * 6 + 3 + a] = 0.0f;
    }
    mask[i] = 0.0f;
    return;
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float mn = -ordered_value(s[a]), mx = ordered_value(s[3 + a]);
    // visual_data_handlers.py:248-258 in float32, then joint_det_dataset.py:511-514 in double
    const float ctr = (mx + mn) / 2.0f;
    const float len = mx - mn;
    const float lo = ctr - len / 2.0f, hi = ctr + len / 2.0f;
    double c = ((double)lo + (double)hi) * 0.5;
    double l = (double)hi - (double)lo;
    if (jitter) {
      c *= jitter[(size_t)i * 6 + a];
      l *= jitter[(size_t)i * 6 + 3 + a];
    }
    out[(size_t)i * 6 + a] = (float)c;
    out[(size_t)i * 6 + 3 + a] = (float)l;
  }
  mask[i] = 1.0f;
}

}  // namespace

extern "C" {

int butd_augment_points(int B, int N, int C, int has_color, const float *pc_in,
                        const butd_scene_augment *params, const double *noise, const double *color_gain,
                        double mean_r, double mean_g, double mean_b, uint64_t seed, float *pc_out,
                        butd_stream_t stream) {
  if (B <= 0 || N <= 0) return 0;
  if (C < 0 || (has_color && C < 3)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(augment_points_kernel, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, N, 3 + C,
                     has_color, pc_in, params, noise, color_gain, mean_r, mean_g, mean_b, seed, pc_out);
  return (int)hipGetLastError();
}

int butd_augment_boxes(int B, int D, const float *boxes_in, const butd_scene_augment *params,
                       float *boxes_out, butd_stream_t stream) {
  const int total = B * D;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(augment_boxes_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, total, D,
                     boxes_in, params, boxes_out);
  return (int)hipGetLastError();
}

int butd_instance_boxes(int B, int N, int ldp, int G, const float *pc, const int64_t *instance,
                        const double *jitter, uint32_t *scratch, float *center_size, float *mask,
                        butd_stream_t stream) {
  if (B <= 0 || G <= 0) return 0;
  if (ldp < 3) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = butd_zero_async(scratch, sizeof(uint32_t) * 6 * (size_t)B * G, s);
  if (e != hipSuccess) return (int)e;
  if (N > 0)
    hipLaunchKernelGGL(instance_hull_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, N, ldp, G, pc, instance,
                       scratch);
  const int total = B * G;
  hipLaunchKernelGGL(instance_box_kernel, dim3((total + 255) / 256), dim3(256), 0, s, total, scratch, jitter,
                     center_size, mask);
  return (int)hipGetLastError();
}

int butd_object_boxes(int B, int N, int ldp, int G, const int *scene, const long long *obj_ptr, long long ptr_stride,
                      const int *obj_points, const int *target_ids, const float *pc, const double *jitter,
                      long long *point_instance_label, uint32_t *scratch, float *center_size, float *mask,
                      butd_stream_t stream) {
  if (B <= 0 || G <= 0) return 0;
  if (ldp < 3 || N <= 0) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = butd_zero_async(scratch, sizeof(uint32_t) * 6 * (size_t)B * G, s);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(object_scan_kernel, dim3(8, G, B), dim3(256), 0, s, N, ldp, G, scene, obj_ptr, ptr_stride,
                     obj_points, target_ids, pc, point_instance_label, scratch);
  const int total = B * G;
  hipLaunchKernelGGL(instance_box_kernel, dim3((total + 255) / 256), dim3(256), 0, s, total, scratch, jitter,
                     center_size, mask);
  return (int)hipGetLastError();
}

}  // extern "C"
