// rowwise_ops.hip -- small row-wise operators (include/butd_rowwise.h): L2 normalisation of the contrastive
// projections (forward / backward) and the inverse-distance weights of the feature-propagation modules.  gfx950.
// Compiled with -ffp-contract=off: the three-nn weights must round exactly like the stock op chain.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/butd_rowwise.h"

namespace {

// sum over the 16 lanes of a DPP row (one matrix row per 16 lanes); every lane of the row gets the total
__device__ inline float row16_sum(float v) {
  v += __shfl_xor(v, 1, 16);
  v += __shfl_xor(v, 2, 16);
  v += __shfl_xor(v, 4, 16);
  v += __shfl_xor(v, 8, 16);
  return v;
}

constexpr int kMaxVec = 16;   // float4 per lane: cols <= 16 lanes * 16 * 4 = 1024

// 16 lanes per row, 4 rows per wave, 16 rows per 256-thread workgroup
template <bool BWD>
__global__ __launch_bounds__(256) void l2_normalize_kernel(long rows, int cols, const float *__restrict__ x,
                                                           const float *__restrict__ g, float eps,
                                                           float *__restrict__ out) {
  const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (row >= rows) return;
  const int l = threadIdx.x & 15, n4 = cols >> 2;
  const float4 *xr = reinterpret_cast<const float4 *>(x + row * cols);
  float4 xv[kMaxVec], gv[kMaxVec];
  float ss = 0.f, xg = 0.f;
#pragma unroll
  for (int j = 0; j < kMaxVec; ++j) {
    const int c4 = l + 16 * j;
    if (c4 < n4) {
      xv[j] = xr[c4];
      ss += xv[j].x * xv[j].x + xv[j].y * xv[j].y + xv[j].z * xv[j].z + xv[j].w * xv[j].w;
      if (BWD) {
        gv[j] = reinterpret_cast<const float4 *>(g + row * cols)[c4];
        xg += xv[j].x * gv[j].x + xv[j].y * gv[j].y + xv[j].z * gv[j].z + xv[j].w * gv[j].w;
      }
    }
  }
  const float n = sqrtf(row16_sum(ss));
  const float c = fmaxf(n, eps);
  float4 *orow = reinterpret_cast<float4 *>(out + row * cols);
  if (!BWD) {
    const float inv = 1.f / c;
#pragma unroll
    for (int j = 0; j < kMaxVec; ++j) {
      const int c4 = l + 16 * j;
      if (c4 < n4) orow[c4] = make_float4(xv[j].x * inv, xv[j].y * inv, xv[j].z * inv, xv[j].w * inv);
    }
  } else {
    xg = row16_sum(xg);
    const float inv = 1.f / c;
    const float k = n >= eps ? xg / (c * c * n) : 0.f;
#pragma unroll
    for (int j = 0; j < kMaxVec; ++j) {
      const int c4 = l + 16 * j;
      if (c4 < n4)
        orow[c4] = make_float4(gv[j].x * inv - xv[j].x * k, gv[j].y * inv - xv[j].y * k, gv[j].z * inv - xv[j].z * k,
                               gv[j].w * inv - xv[j].w * k);
    }
  }
}

__global__ __launch_bounds__(256) void three_nn_weights_kernel(long rows, const float *__restrict__ dist2,
                                                               float *__restrict__ dist, float *__restrict__ weight) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const float d0 = sqrtf(dist2[r * 3 + 0]), d1 = sqrtf(dist2[r * 3 + 1]), d2 = sqrtf(dist2[r * 3 + 2]);
  const float r0 = 1.0f / (d0 + 1e-8f), r1 = 1.0f / (d1 + 1e-8f), r2 = 1.0f / (d2 + 1e-8f);
  const float s = (r0 + r1) + r2;          // torch.sum over three elements: left to right
  if (dist) {
    dist[r * 3 + 0] = d0; dist[r * 3 + 1] = d1; dist[r * 3 + 2] = d2;
  }
  weight[r * 3 + 0] = r0 / s; weight[r * 3 + 1] = r1 / s; weight[r * 3 + 2] = r2 / s;
}

}  // namespace

extern "C" {

int butd_l2_normalize_fwd(long rows, int cols, const float *x, float eps, float *y, butd_stream_t stream) {
  if (rows <= 0) return 0;
  if (cols <= 0 || (cols & 3) || cols > 64 * kMaxVec || ((((uintptr_t)x) | ((uintptr_t)y)) & 15)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((l2_normalize_kernel<false>), dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, (hipStream_t)stream,
                     rows, cols, x, (const float *)nullptr, eps, y);
  return (int)hipGetLastError();
}

int butd_l2_normalize_bwd(long rows, int cols, const float *x, const float *g, float eps, float *dx,
                          butd_stream_t stream) {
  if (rows <= 0) return 0;
  if (cols <= 0 || (cols & 3) || cols > 64 * kMaxVec || ((((uintptr_t)x) | ((uintptr_t)g) | ((uintptr_t)dx)) & 15))
    return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((l2_normalize_kernel<true>), dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, (hipStream_t)stream,
                     rows, cols, x, g, eps, dx);
  return (int)hipGetLastError();
}

int butd_three_nn_weights(long rows, const float *dist2, float *dist, float *weight, butd_stream_t stream) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(three_nn_weights_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     rows, dist2, dist, weight);
  return (int)hipGetLastError();
}

}  // extern "C"
