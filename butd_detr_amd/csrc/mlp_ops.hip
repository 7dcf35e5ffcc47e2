// mlp_ops.hip -- glue kernels of the Conv1d+BatchNorm1d+ReLU(+Dropout) chains (include/butd_mlp.h).
//
// The 1x1 convolutions are butd_gemm_grouped problems (attention_ops.hip) whose epilogue leaves the
// BatchNorm column sums behind and whose operand staging applies BatchNorm+ReLU+Dropout of the previous
// layer; here: the per-channel bookkeeping between two products and the two element-wise halves of the
// BatchNorm backward.  P is a few thousand rows (B x queries), C a few hundred channels: everything is
// launch-latency bound, so each stage is ONE launch over all concatenated chains.
#include <hip/hip_runtime.h>
// (ablation hook, scratch/r6_prio.sh: -DBUTD_MAIN_PRIO=n raises the wave priority of this unit's kernels -- the captured
// step's prefetch branches share CUs with them; profiles/r06_side_branches.txt)
#ifdef BUTD_MAIN_PRIO
#define BUTD_MAIN_PRIO_SET() __builtin_amdgcn_s_setprio(BUTD_MAIN_PRIO)
#else
#define BUTD_MAIN_PRIO_SET()
#endif

#include <math.h>
#include <stdint.h>

#include "../../include/butd_mlp.h"
#include "rng.h"

namespace {

struct Segments {
  butd_bn_segment s[BUTD_MLP_MAX_SEGMENTS];
};

__global__ void mlp_bn_finalize_kernel(Segments segs, int Cseg, long count,
                                       const double *__restrict__ sum, const double *__restrict__ sumsq,
                                       float eps, float momentum, int training,
                                       float *__restrict__ mean, float *__restrict__ rstd,
                                       float *__restrict__ scale, float *__restrict__ shift) {
  BUTD_MAIN_PRIO_SET();
  const butd_bn_segment &S = segs.s[blockIdx.y];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && training && S.num_batches_tracked) *S.num_batches_tracked += 1;
  if (c >= Cseg) return;
  const int o = blockIdx.y * Cseg + c;
  float mu, var;
  if (training) {
    const double m = sum[o] / (double)count;
    double v = sumsq[o] / (double)count - m * m;
    if (v < 0.0) v = 0.0;
    mu = (float)m;
    var = (float)v;
    const double unbiased = count > 1 ? v * (double)count / (double)(count - 1) : v;
    S.running_mean[c] = (1.f - momentum) * S.running_mean[c] + momentum * mu;
    S.running_var[c] = (1.f - momentum) * S.running_var[c] + momentum * (float)unbiased;
  } else {
    mu = S.running_mean[c];
    var = S.running_var[c];
  }
  const float rs = 1.0f / sqrtf(var + eps);
  const float g = S.gamma[c];
  mean[o] = mu;
  rstd[o] = rs;
  scale[o] = g * rs;
  shift[o] = S.beta[c] - mu * g * rs;
}

// Tiling of the element-wise kernels: a workgroup covers kRows rows x 256 columns; a thread owns one
// float4 column quad and every 4th row.
constexpr int kRows = 32;

__global__ __launch_bounds__(256) void mlp_mask_stats_kernel(
    long P, int C, long ld, float *__restrict__ dH, const float *__restrict__ Z,
    const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ mean,
    const float *__restrict__ rstd, float drop_p, uint32_t site0, int seg_cols,
    const uint64_t *__restrict__ rng_counter, double *__restrict__ S1, double *__restrict__ S2) {
  BUTD_MAIN_PRIO_SET();
  __shared__ float red[2][4][256];
  const int cq = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int c = blockIdx.y * 256 + cq * 4;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const float4 sc = *reinterpret_cast<const float4 *>(scale + c);
    const float4 sh = *reinterpret_cast<const float4 *>(shift + c);
    const float4 mu = *reinterpret_cast<const float4 *>(mean + c);
    const float4 rs = *reinterpret_cast<const float4 *>(rstd + c);
    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
    const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w};
    const bool drop = drop_p > 0.f;
    const int seg = c / seg_cols, cl = c - seg * seg_cols;
    const uint32_t key = rng::site_key(drop && rng_counter ? *rng_counter : 0ull, site0 + (uint32_t)seg);
    const float inv = drop ? 1.f / (1.f - drop_p) : 1.f;
    const long r0 = (long)blockIdx.x * kRows, r1 = min(P, r0 + kRows);
    for (long r = r0 + ph; r < r1; r += 4) {
      const float4 d4 = *reinterpret_cast<const float4 *>(dH + r * ld + c);
      const float4 z4 = *reinterpret_cast<const float4 *>(Z + r * ld + c);
      const float dv[4] = {d4.x, d4.y, d4.z, d4.w}, zv[4] = {z4.x, z4.y, z4.z, z4.w};
      float g[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        g[e] = (scv[e] * zv[e] + shv[e] > 0.f) ? dv[e] : 0.f;
        if (drop) g[e] = rng::keep_keyed(key, (uint32_t)(r * ld + cl + e), drop_p) ? g[e] * inv : 0.f;
        s1[e] += g[e];
        s2[e] += g[e] * ((zv[e] - muv[e]) * rsv[e]);
      }
      *reinterpret_cast<float4 *>(dH + r * ld + c) = make_float4(g[0], g[1], g[2], g[3]);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[0][ph][cq * 4 + e] = s1[e];
    red[1][ph][cq * 4 + e] = s2[e];
  }
  __syncthreads();
  const int col = blockIdx.y * 256 + threadIdx.x;
  if (col < C) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      a += (double)red[0][t][threadIdx.x];
      b += (double)red[1][t][threadIdx.x];
    }
    atomicAdd(S1 + col, a);
    atomicAdd(S2 + col, b);
  }
}

__global__ __launch_bounds__(256) void mlp_dz_kernel(long P, int C, long ld, float *__restrict__ g,
                                                     const float *__restrict__ Z,
                                                     const float *__restrict__ scale,
                                                     const float *__restrict__ mean,
                                                     const float *__restrict__ rstd,
                                                     const double *__restrict__ S1,
                                                     const double *__restrict__ S2, int training,
                                                     float *__restrict__ S1f, float *__restrict__ S2f) {
  BUTD_MAIN_PRIO_SET();
  const int cq = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int c = blockIdx.y * 256 + cq * 4;
  if (c >= C) return;
  if (S1f && blockIdx.x == 0 && ph == 0) {   // the BatchNorm bias / weight gradients as fp32 (they ARE the two sums)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      S1f[c + e] = (float)S1[c + e];
      S2f[c + e] = (float)S2[c + e];
    }
  }
  const float4 sc = *reinterpret_cast<const float4 *>(scale + c);
  const float4 mu = *reinterpret_cast<const float4 *>(mean + c);
  const float4 rs = *reinterpret_cast<const float4 *>(rstd + c);
  const float scv[4] = {sc.x, sc.y, sc.z, sc.w};
  const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w};
  float m1[4], m2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    m1[e] = training ? (float)(S1[c + e] / (double)P) : 0.f;
    m2[e] = training ? (float)(S2[c + e] / (double)P) : 0.f;
  }
  const long r0 = (long)blockIdx.x * kRows, r1 = min(P, r0 + kRows);
  for (long r = r0 + ph; r < r1; r += 4) {
    const float4 g4 = *reinterpret_cast<const float4 *>(g + r * ld + c);
    const float4 z4 = *reinterpret_cast<const float4 *>(Z + r * ld + c);
    const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, zv[4] = {z4.x, z4.y, z4.z, z4.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = scv[e] * (gv[e] - m1[e] - ((zv[e] - muv[e]) * rsv[e]) * m2[e]);
    *reinterpret_cast<float4 *>(g + r * ld + c) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

__global__ __launch_bounds__(256) void mlp_bn_relu_apply_kernel(long P, int C, long ld,
                                                                const float *__restrict__ Z,
                                                                const float *__restrict__ scale,
                                                                const float *__restrict__ shift,
                                                                float *__restrict__ out) {
  BUTD_MAIN_PRIO_SET();
  const int cq = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int c = blockIdx.y * 256 + cq * 4;
  if (c >= C) return;
  const float4 sc = *reinterpret_cast<const float4 *>(scale + c);
  const float4 sh = *reinterpret_cast<const float4 *>(shift + c);
  const long r0 = (long)blockIdx.x * kRows, r1 = min(P, r0 + kRows);
  for (long r = r0 + ph; r < r1; r += 4) {
    const float4 z = *reinterpret_cast<const float4 *>(Z + r * ld + c);
    *reinterpret_cast<float4 *>(out + r * ld + c) =
        make_float4(fmaxf(sc.x * z.x + sh.x, 0.f), fmaxf(sc.y * z.y + sh.y, 0.f),
                    fmaxf(sc.z * z.z + sh.z, 0.f), fmaxf(sc.w * z.w + sh.w, 0.f));
  }
}

inline int launch_status() { return (int)hipGetLastError(); }

}  // namespace

extern "C" {

int butd_mlp_bn_finalize(int nseg, int Cseg, long count, const double *sum, const double *sumsq,
                         const butd_bn_segment *segs, float eps, float momentum, int training,
                         float *mean, float *rstd, float *scale, float *shift, butd_stream_t stream) {
  if (nseg < 1 || nseg > BUTD_MLP_MAX_SEGMENTS || Cseg < 1) return (int)hipErrorInvalidValue;
  Segments s;
  for (int i = 0; i < BUTD_MLP_MAX_SEGMENTS; ++i) s.s[i] = segs[i < nseg ? i : 0];
  hipLaunchKernelGGL(mlp_bn_finalize_kernel, dim3((Cseg + 255) / 256, nseg), dim3(256), 0,
                     (hipStream_t)stream, s, Cseg, count, sum, sumsq, eps, momentum, training, mean, rstd,
                     scale, shift);
  return launch_status();
}

int butd_mlp_mask_stats(long P, int C, long ld, float *dH, const float *Z, const float *scale,
                        const float *shift, const float *mean, const float *rstd, float drop_p,
                        uint32_t site0, int seg_cols, const uint64_t *rng_counter, double *S1,
                        double *S2, butd_stream_t stream) {
  if (P < 1 || C < 4 || (C & 3) || (ld & 3) || seg_cols < 4 || (seg_cols & 3))
    return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mlp_mask_stats_kernel, dim3((unsigned)((P + kRows - 1) / kRows), (C + 255) / 256),
                     dim3(256), 0, (hipStream_t)stream, P, C, ld, dH, Z, scale, shift, mean, rstd, drop_p,
                     site0, seg_cols, rng_counter, S1, S2);
  return launch_status();
}

int butd_mlp_dz(long P, int C, long ld, float *g, const float *Z, const float *scale,
                const float *mean, const float *rstd, const double *S1, const double *S2,
                int training, float *S1f, float *S2f, butd_stream_t stream) {
  if (P < 1 || C < 4 || (C & 3) || (ld & 3)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mlp_dz_kernel, dim3((unsigned)((P + kRows - 1) / kRows), (C + 255) / 256),
                     dim3(256), 0, (hipStream_t)stream, P, C, ld, g, Z, scale, mean, rstd, S1, S2,
                     training, S1f, S2f);
  return launch_status();
}

int butd_mlp_bn_relu_apply(long P, int C, long ld, const float *Z, const float *scale,
                           const float *shift, float *out, butd_stream_t stream) {
  if (P < 1 || C < 4 || (C & 3) || (ld & 3)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mlp_bn_relu_apply_kernel, dim3((unsigned)((P + kRows - 1) / kRows), (C + 255) / 256),
                     dim3(256), 0, (hipStream_t)stream, P, C, ld, Z, scale, shift, out);
  return launch_status();
}

}  // extern "C"
