// sa_ops.hip -- streaming kernels of the set-abstraction shared-MLP pipeline (include/butd_sa.h).
//
// The 1x1 convolutions themselves are butd_gemm_grouped launches (attention_ops.hip) over
// position-major activations, with the producer's BatchNorm+ReLU folded into the operand load; what
// lives here is everything around them: the grouping gather, the BatchNorm statistics (double
// accumulators: E[z^2]-E[z]^2 over 10^6 rows in fp32 would lose the variance), the max-pool with
// first-occurrence arg-max, and the element-wise halves of the backward.  All are HBM-streaming
// kernels: coalesced float4 rows, one pass per tensor.
#include <hip/hip_runtime.h>
// (-DBUTD_MAIN_PRIO=n raises the wave priority of this unit's kernels: the captured step's prefetch branches share CUs
// with them; profiles/r06_side_branches.txt)
#ifdef BUTD_MAIN_PRIO
#define BUTD_MAIN_PRIO_SET() __builtin_amdgcn_s_setprio(BUTD_MAIN_PRIO)
#else
#define BUTD_MAIN_PRIO_SET()
#endif

#include <stdlib.h>
#include <math.h>
#include <stdint.h>

#include "../../include/butd_sa.h"

namespace {

constexpr int kThreads = 256;

// ---------------------------------------------------------------------------------------- grouping
__global__ __launch_bounds__(kThreads) void sa_group_kernel(int N, int np, int ns, int C,
                                                            const float *__restrict__ xyz,
                                                            const float *__restrict__ new_xyz,
                                                            const float *__restrict__ feats,
                                                            long feat_stride,
                                                            const int *__restrict__ idx, float radius,
                                                            int normalize, float *__restrict__ X,
                                                            int ldx, long total) {
  BUTD_MAIN_PRIO_SET();
  const int Cin = 3 + C;
  for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < total; e += (long)gridDim.x * kThreads) {
    const long p = e / ldx;
    const int c = (int)(e - p * ldx);
    if (c >= Cin) {   // padding columns (row stride rounded up for 16-byte rows)
      X[e] = 0.f;
      continue;
    }
    const long g = p / ns;           // (b, j)
    const long b = g / np;
    const int src = idx[p];
    float v;
    if (c < 3) {
      v = xyz[(b * N + src) * 3 + c] - new_xyz[g * 3 + c];
      if (normalize) v = v / radius;
    } else {
      v = feats[(b * N + src) * feat_stride + (c - 3)];
    }
    X[e] = v;
  }
}

// ------------------------------------------------------------------------- column stats (+ pooling)
// Workgroup = 256 threads over a chunk of kChunkRows rows x all C columns.  A thread owns FOUR adjacent
// columns (one float4 per row) and every TPG-th GROUP of the chunk, TPG = 256 / (C/4) (groups =
// pool_ns rows, or single rows when not pooling), so the pooling needs no cross-thread step and every
// load is a coalesced 16-byte access; sums are reduced across the TPG row-phases in LDS and leave the
// block as ONE double atomic per column.
constexpr int kChunkRows = 1024;     // tall inputs (>= 2^19 rows): fewest same-address atomics (measured 256 /
                                     // 512 / 1024 / 2048: colstats 84 / 77 / 75 / 73 us, mask_stats 82 / 79 / 75 / 76,
                                     // thin conv 128 / 112 / 93 / 116)
constexpr int kChunkRowsSmall = 64;  // otherwise: four times the workgroups (a 65 536-row level is 256
                                     // workgroups of 256 rows, and every thread then walks 16-64 rows of
                                     // dependent load -> store: 100 us where the data takes 25)
inline int chunk_rows(long P) {
  return P >= (1L << 19) ? kChunkRows : kChunkRowsSmall;
}

__global__ __launch_bounds__(kThreads) void sa_colstats_kernel(
    long P, int C, const float *__restrict__ Z, double *__restrict__ sum, double *__restrict__ sumsq,
    int pool_ns, float *__restrict__ zmax, float *__restrict__ zmin, uint8_t *__restrict__ amax,
    uint8_t *__restrict__ amin, int chunk) {
  BUTD_MAIN_PRIO_SET();
  __shared__ float red[2][kThreads][4];
  const int c4n = C >> 2;                 // float4 columns (16, 32 or 64)
  const int tpg = kThreads / c4n;         // row phases (16, 8 or 4)
  const int cq = threadIdx.x % c4n, sub = threadIdx.x / c4n;
  const int gs = pool_ns > 0 ? pool_ns : 1;
  const long row0 = (long)blockIdx.x * chunk;
  const long rows = min((long)chunk, P - row0);
  const long ngroups = rows / gs;
  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
  if (pool_ns <= 0) {
    // plain sums: four independent row loads in flight per thread (one per trip left the read-only
    // stream at 2 TB/s)
    long r = sub;
    for (; r + 3 * tpg < rows; r += 4 * tpg) {
      float4 v4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        v4[u] = *reinterpret_cast<const float4 *>(Z + (row0 + r + (long)u * tpg) * C + cq * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float v[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[e] += v[e];
          q[e] += v[e] * v[e];
        }
      }
    }
    for (; r < rows; r += tpg) {
      const float4 v4 = *reinterpret_cast<const float4 *>(Z + (row0 + r) * C + cq * 4);
      const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[e] += v[e];
        q[e] += v[e] * v[e];
      }
    }
  }
  for (long g = sub; pool_ns > 0 && g < ngroups; g += tpg) {
    const float *z = Z + (row0 + g * gs) * C + cq * 4;
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    float mn[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    int ax[4] = {0, 0, 0, 0}, an[4] = {0, 0, 0, 0};
    for (int k0 = 0; k0 < gs; k0 += 4) {   // gs is 16, 32 or 64: four row loads in flight
      float4 v4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v4[u] = *reinterpret_cast<const float4 *>(z + (long)(k0 + u) * C);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u;
        const float v[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[e] += v[e];
          q[e] += v[e] * v[e];
          if (v[e] > mx[e]) { mx[e] = v[e]; ax[e] = k; }
          if (v[e] < mn[e]) { mn[e] = v[e]; an[e] = k; }
        }
      }
    }
    if (pool_ns > 0) {
      const long o = ((row0 / gs) + g) * C + cq * 4;
      *reinterpret_cast<float4 *>(zmax + o) = make_float4(mx[0], mx[1], mx[2], mx[3]);
      *reinterpret_cast<float4 *>(zmin + o) = make_float4(mn[0], mn[1], mn[2], mn[3]);
      *reinterpret_cast<uchar4 *>(amax + o) = make_uchar4(ax[0], ax[1], ax[2], ax[3]);
      *reinterpret_cast<uchar4 *>(amin + o) = make_uchar4(an[0], an[1], an[2], an[3]);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[0][threadIdx.x][e] = s[e];
    red[1][threadIdx.x][e] = q[e];
  }
  __syncthreads();
  if (threadIdx.x < C) {
    const int cq2 = threadIdx.x >> 2, e = threadIdx.x & 3;
    double a = 0.0, b = 0.0;
    for (int t = 0; t < tpg; ++t) {
      a += (double)red[0][cq2 + t * c4n][e];
      b += (double)red[1][cq2 + t * c4n][e];
    }
    atomicAdd(sum + threadIdx.x, a);
    atomicAdd(sumsq + threadIdx.x, b);
  }
}

// ------------------------------------------------------------------ first layer of SA1 (thin input)
// Z[p, c] = sum_k X[p, k] * W[c, k] for K <= 8 input channels (xyz + colour + padding) and the column
// sums of Z in the same pass.  As a 64x64-tile GEMM this product is one mostly-empty K slab per tile
// (140 us for 10^6 rows, plus 128 us for the separate statistics pass); it is a pure HBM stream:
// 32 B in, 4*C B out per row.
__global__ __launch_bounds__(kThreads) void sa_thin_conv_kernel(
    long P, int C, int K, const float *__restrict__ X, int ldx, const float *__restrict__ W,
    float *__restrict__ Z, double *__restrict__ sum, double *__restrict__ sumsq, int chunk) {
  BUTD_MAIN_PRIO_SET();
  __shared__ float red[2][kThreads][4];
  const int c4n = C >> 2, tpg = kThreads / c4n;
  const int cq = threadIdx.x % c4n, sub = threadIdx.x / c4n;
  const long row0 = (long)blockIdx.x * chunk;
  const long rows = min((long)chunk, P - row0);
  float w[4][8];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int k = 0; k < 8; ++k) w[e][k] = k < K ? W[(long)(cq * 4 + e) * K + k] : 0.f;
  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
  for (long r = sub; r < rows; r += tpg) {
    const float *xr = X + (row0 + r) * ldx;
    const float4 x0 = *reinterpret_cast<const float4 *>(xr);
    const float4 x1 = *reinterpret_cast<const float4 *>(xr + 4);
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    float z[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) a += x[k] * w[e][k];
      z[e] = a;
      s[e] += a;
      q[e] += a * a;
    }
    *reinterpret_cast<float4 *>(Z + (row0 + r) * C + cq * 4) = make_float4(z[0], z[1], z[2], z[3]);
  }
  if (sum == nullptr) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[0][threadIdx.x][e] = s[e];
    red[1][threadIdx.x][e] = q[e];
  }
  __syncthreads();
  if (threadIdx.x < C) {
    const int cq2 = threadIdx.x >> 2, e = threadIdx.x & 3;
    double a = 0.0, b = 0.0;
    for (int t = 0; t < tpg; ++t) {
      a += (double)red[0][cq2 + t * c4n][e];
      b += (double)red[1][cq2 + t * c4n][e];
    }
    atomicAdd(sum + threadIdx.x, a);
    atomicAdd(sumsq + threadIdx.x, b);
  }
}

__global__ void sa_bn_finalize_kernel(int C, long count, const double *__restrict__ sum,
                                      const double *__restrict__ sumsq, int slots, long slot_stride,
                                      const float *__restrict__ gamma,
                                      const float *__restrict__ beta, float eps, float momentum,
                                      int training, float *__restrict__ running_mean,
                                      float *__restrict__ running_var, int64_t *__restrict__ nbt,
                                      float *__restrict__ mean, float *__restrict__ rstd,
                                      float *__restrict__ scale, float *__restrict__ shift) {
  BUTD_MAIN_PRIO_SET();
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && training && nbt) *nbt += 1;
  if (c >= C) return;
  float mu, var;
  if (training) {
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < (slots > 1 ? slots : 1); ++k) {
      s1 += sum[c + k * slot_stride];
      s2 += sumsq[c + k * slot_stride];
    }
    const double m = s1 / (double)count;
    double v = s2 / (double)count - m * m;
    if (v < 0.0) v = 0.0;
    mu = (float)m;
    var = (float)v;
    const double unbiased = count > 1 ? v * (double)count / (double)(count - 1) : v;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  } else {
    mu = running_mean[c];
    var = running_var[c];
  }
  const float rs = 1.0f / sqrtf(var + eps);
  mean[c] = mu;
  rstd[c] = rs;
  scale[c] = gamma[c] * rs;
  shift[c] = beta[c] - mu * gamma[c] * rs;
}

__global__ __launch_bounds__(kThreads) void sa_pool_finalize_kernel(
    int np, int C, long total, const float *__restrict__ zmax, const float *__restrict__ zmin,
    const uint8_t *__restrict__ amax, const uint8_t *__restrict__ amin, const float *__restrict__ scale,
    const float *__restrict__ shift, float *__restrict__ out_cm, float *__restrict__ out_pm,
    float *__restrict__ zsel, uint8_t *__restrict__ asel) {
  BUTD_MAIN_PRIO_SET();
  const long e = (long)blockIdx.x * kThreads + threadIdx.x;  // over (g, c), c fastest
  if (e >= total) return;
  const int c = (int)(e % C);
  const long g = e / C;
  const long b = g / np;
  const int j = (int)(g - b * np);
  const float sc = scale[c];
  const bool up = sc >= 0.f;
  const float z = up ? zmax[e] : zmin[e];
  const float y = fmaxf(sc * z + shift[c], 0.f);
  zsel[e] = z;
  asel[e] = up ? amax[e] : amin[e];
  out_pm[e] = y;
  out_cm[(b * C + c) * np + j] = y;
}

// ---------------------------------------------------------------------------------------- backward
__global__ __launch_bounds__(kThreads) void sa_pool_bwd_stats_kernel(
    int np, int C, long G, int groups_per_block, const float *__restrict__ d_out_pm, const float *__restrict__ zsel,
    const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ mean,
    const float *__restrict__ rstd, double *__restrict__ S1, double *__restrict__ S2) {
  BUTD_MAIN_PRIO_SET();
  // block = groups_per_block groups x all channels: thread t owns column (t % C), groups sub, sub+tpc, ...
  __shared__ float red[2][kThreads];
  const int tpc = kThreads / C;
  const int col = threadIdx.x % C, sub = threadIdx.x / C;
  const long g0 = (long)blockIdx.x * groups_per_block;
  const long ng = min((long)groups_per_block, G - g0);
  float s1 = 0.f, s2 = 0.f;
  if (sub < tpc) {
    const float sc = scale[col], sh = shift[col], mu = mean[col], rs = rstd[col];
    for (long gi = sub; gi < ng; gi += tpc) {
      const long g = g0 + gi;
      const float z = zsel[g * C + col];
      if (sc * z + sh > 0.f) {
        const float dy = d_out_pm[g * C + col];
        s1 += dy;
        s2 += dy * (z - mu) * rs;
      }
    }
  }
  red[0][threadIdx.x] = s1;
  red[1][threadIdx.x] = s2;
  __syncthreads();
  if (threadIdx.x < C) {
    double a = 0.0, b2 = 0.0;
    for (int t = 0; t < tpc; ++t) {
      a += (double)red[0][threadIdx.x + t * C];
      b2 += (double)red[1][threadIdx.x + t * C];
    }
    atomicAdd(S1 + threadIdx.x, a);
    atomicAdd(S2 + threadIdx.x, b2);
  }
}

// dZ of the last layer, in place on Z: workgroup = kChunkRows rows x all columns, a thread owns one
// float4 column group and every TPG-th row (same decomposition as the statistics kernels).
__global__ __launch_bounds__(kThreads) void sa_dz_last_kernel(
    int np, int ns, int C, long P, float *__restrict__ Z, const float *__restrict__ d_out_pm,
    const float *__restrict__ zsel, const uint8_t *__restrict__ asel, const float *__restrict__ gamma,
    const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ mean,
    const float *__restrict__ rstd, const double *__restrict__ S1, const double *__restrict__ S2,
    int training, int chunk) {
  BUTD_MAIN_PRIO_SET();
  const int c4n = C >> 2, tpg = kThreads / c4n;
  const int cq = threadIdx.x % c4n, sub = threadIdx.x / c4n;
  const long row0 = (long)blockIdx.x * chunk;
  const long rows = min((long)chunk, P - row0);
  const double invP = 1.0 / (double)P;
  float sc[4], sh[4], mu[4], rs[4], ga[4], a1[4], a2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = cq * 4 + e;
    sc[e] = scale[c]; sh[e] = shift[c]; mu[e] = mean[c]; rs[e] = rstd[c]; ga[e] = gamma[c];
    a1[e] = (float)(S1[c] * invP);
    a2[e] = (float)(S2[c] * invP);
  }
  for (long r = sub; r < rows; r += tpg) {
    const long p = row0 + r;
    const long g = p / ns;
    const int k = (int)(p - g * ns);
    const long o = p * C + cq * 4;
    const float4 z4 = *reinterpret_cast<const float4 *>(Z + o);
    const float4 dy4 = *reinterpret_cast<const float4 *>(d_out_pm + g * C + cq * 4);
    const float dyv[4] = {dy4.x, dy4.y, dy4.z, dy4.w};
    const float4 zs4 = *reinterpret_cast<const float4 *>(zsel + g * C + cq * 4);
    const uchar4 as4 = *reinterpret_cast<const uchar4 *>(asel + g * C + cq * 4);
    const float z[4] = {z4.x, z4.y, z4.z, z4.w}, zs[4] = {zs4.x, zs4.y, zs4.z, zs4.w};
    const int as[4] = {as4.x, as4.y, as4.z, as4.w};
    float out[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float dy = 0.f;
      if (as[e] == k && sc[e] * zs[e] + sh[e] > 0.f) dy = dyv[e];
      out[e] = training ? ga[e] * rs[e] * (dy - a1[e] - (z[e] - mu[e]) * rs[e] * a2[e]) : sc[e] * dy;
    }
    *reinterpret_cast<float4 *>(Z + o) = make_float4(out[0], out[1], out[2], out[3]);
  }
}

__global__ __launch_bounds__(kThreads) void sa_mask_stats_kernel(
    long P, int C, const float *__restrict__ dH, const float *__restrict__ Z,
    const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ mean,
    const float *__restrict__ rstd, double *__restrict__ S1, double *__restrict__ S2, int chunk) {
  BUTD_MAIN_PRIO_SET();
  __shared__ float red[2][kThreads][4];
  const int c4n = C >> 2, tpg = kThreads / c4n;
  const int cq = threadIdx.x % c4n, sub = threadIdx.x / c4n;
  const long row0 = (long)blockIdx.x * chunk;
  const long rows = min((long)chunk, P - row0);
  const float4 sc4 = *reinterpret_cast<const float4 *>(scale + cq * 4);
  const float4 sh4 = *reinterpret_cast<const float4 *>(shift + cq * 4);
  const float4 mu4 = *reinterpret_cast<const float4 *>(mean + cq * 4);
  const float4 rs4 = *reinterpret_cast<const float4 *>(rstd + cq * 4);
  const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
  const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, rs[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  for (long r = sub; r < rows; r += tpg) {
    const long o = (row0 + r) * C + cq * 4;
    const float4 z4 = *reinterpret_cast<const float4 *>(Z + o);
    const float4 g4 = *reinterpret_cast<const float4 *>(dH + o);
    const float z[4] = {z4.x, z4.y, z4.z, z4.w};
    float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (!(sc[e] * z[e] + sh[e] > 0.f)) g[e] = 0.f;
      s1[e] += g[e];
      s2[e] += g[e] * (z[e] - mu[e]) * rs[e];
    }   // read-only pass: butd_sa_dz_mid re-derives the mask from Z instead of reading a stored g
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[0][threadIdx.x][e] = s1[e];
    red[1][threadIdx.x][e] = s2[e];
  }
  __syncthreads();
  if (threadIdx.x < C) {
    const int cq2 = threadIdx.x >> 2, e = threadIdx.x & 3;
    double a = 0.0, b = 0.0;
    for (int t = 0; t < tpg; ++t) {
      a += (double)red[0][cq2 + t * c4n][e];
      b += (double)red[1][cq2 + t * c4n][e];
    }
    atomicAdd(S1 + threadIdx.x, a);
    atomicAdd(S2 + threadIdx.x, b);
  }
}

__global__ __launch_bounds__(kThreads) void sa_dz_mid_kernel(
    long P, int C, float *__restrict__ g, const float *__restrict__ Z, const float *__restrict__ gamma,
    const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ mean,
    const float *__restrict__ rstd, const double *__restrict__ S1, const double *__restrict__ S2,
    int training, int chunk) {
  BUTD_MAIN_PRIO_SET();
  const int c4n = C >> 2, tpg = kThreads / c4n;
  const int cq = threadIdx.x % c4n, sub = threadIdx.x / c4n;
  const long row0 = (long)blockIdx.x * chunk;
  const long rows = min((long)chunk, P - row0);
  const double invP = 1.0 / (double)P;
  float sc[4], sh[4], mu[4], rs[4], ga[4], a1[4], a2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = cq * 4 + e;
    sc[e] = scale[c]; sh[e] = shift[c]; mu[e] = mean[c]; rs[e] = rstd[c]; ga[e] = gamma[c];
    a1[e] = (float)(S1[c] * invP);
    a2[e] = (float)(S2[c] * invP);
  }
  for (long r = sub; r < rows; r += tpg) {
    const long o = (row0 + r) * C + cq * 4;
    const float4 z4 = *reinterpret_cast<const float4 *>(Z + o);
    const float4 g4 = *reinterpret_cast<const float4 *>(g + o);
    const float z[4] = {z4.x, z4.y, z4.z, z4.w};
    float gv[4] = {g4.x, g4.y, g4.z, g4.w};
    float out[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (!(sc[e] * z[e] + sh[e] > 0.f)) gv[e] = 0.f;   // ReLU mask, as butd_sa_mask_stats applied it
      out[e] = training ? ga[e] * rs[e] * (gv[e] - a1[e] - (z[e] - mu[e]) * rs[e] * a2[e]) : sc[e] * gv[e];
    }
    *reinterpret_cast<float4 *>(g + o) = make_float4(out[0], out[1], out[2], out[3]);
  }
}

__global__ __launch_bounds__(kThreads) void sa_scatter_rows_kernel(int N, int np, int ns, int C,
                                                                   const float *__restrict__ dX,
                                                                   const int *__restrict__ idx,
                                                                   float *__restrict__ d_feats,
                                                                   int ldx, long total) {
  BUTD_MAIN_PRIO_SET();
  for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < total; e += (long)gridDim.x * kThreads) {
    const long p = e / C;
    const int c = (int)(e - p * C);
    const long b = p / ((long)np * ns);
    atomicAdd(d_feats + (b * N + idx[p]) * C + c, dX[p * ldx + 3 + c]);
  }
}

inline unsigned blocks_for(long total, int cap = 16384) {
  long b = (total + kThreads - 1) / kThreads;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}
inline bool cols_ok(int C) { return C >= 16 && C <= kThreads && kThreads % C == 0; }

// ----------------------------------------------------------------- feature gradient as a gather (no float atomics)
// d_feats[b, i, :] = sum over the grouped rows p with idx[p] == i of dX[p, 3:].  The inverse of the neighbour lists (for
// every point the grouped rows that copy it) is a function of the coordinates alone: built once per batch (count, scan,
// fill: integer atomics on B*N counters), it turns butd_sa_scatter_rows' 3.3e7 float atomics (SA2, B = 8) into plain
// row reads.
__global__ __launch_bounds__(kThreads) void sa_inv_count_kernel(int N, long npns, long P, const int *__restrict__ idx,
                                                                int *__restrict__ count) {
  BUTD_MAIN_PRIO_SET();
  for (long p = (long)blockIdx.x * kThreads + threadIdx.x; p < P; p += (long)gridDim.x * kThreads)
    atomicAdd(count + (p / npns) * N + idx[p], 1);
}

// one workgroup per batch element: start[b*N + i] = b*npns + exclusive prefix of count; count is zeroed (the fill's cursor)
__global__ __launch_bounds__(1024) void sa_inv_scan_kernel(int N, long npns, int *__restrict__ count, int *__restrict__ start,
                                                           int last_batch) {
  BUTD_MAIN_PRIO_SET();
  __shared__ int part[1024];
  const int b = blockIdx.x, t = threadIdx.x;
  const int per = (N + 1023) / 1024, i0 = t * per, i1 = min(N, i0 + per);
  int s = 0;
  for (int i = i0; i < i1; ++i) s += count[(long)b * N + i];
  part[t] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {           // inclusive scan of the 1024 partial sums
    const int v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = (int)((long)b * npns) + (t ? part[t - 1] : 0);
  for (int i = i0; i < i1; ++i) {
    const int c = count[(long)b * N + i];
    start[(long)b * N + i] = run;
    count[(long)b * N + i] = 0;
    run += c;
  }
  if (b == last_batch && t == 0) start[(long)(b + 1) * N] = (int)((long)(b + 1) * npns);
}

__global__ __launch_bounds__(kThreads) void sa_inv_fill_kernel(int N, long npns, long P, const int *__restrict__ idx,
                                                               int *__restrict__ cursor, const int *__restrict__ start,
                                                               int *__restrict__ list) {
  BUTD_MAIN_PRIO_SET();
  for (long p = (long)blockIdx.x * kThreads + threadIdx.x; p < P; p += (long)gridDim.x * kThreads) {
    const long i = (p / npns) * N + idx[p];
    list[start[i] + atomicAdd(cursor + i, 1)] = (int)p;
  }
}

// one WAVE per point: its slice sorted ascending -- bitonic network in LDS for slices of <= 1024 entries (typical: 16; a
// point in a dense cluster: hundreds), one lane's Shell sort in place beyond that.  The lists, and with them the order
// of the gather's sums, are then the same on every run.  (First version: one THREAD per slice -- 0.3 ms per level on the
// prefetch stream from the imbalance of a few long slices.)
__global__ __launch_bounds__(64) void sa_inv_sort_kernel(long R, const int *__restrict__ start, int *__restrict__ list) {
  BUTD_MAIN_PRIO_SET();
  __shared__ int buf[1024];
  const long r = blockIdx.x;
  if (r >= R) return;
  int *a = list + start[r];
  const int n = start[r + 1] - start[r];
  if (n <= 1) return;
  const int lane = threadIdx.x;
  if (n > 1024) {
    if (lane == 0) {
      const int gaps[9] = {1750, 701, 301, 132, 57, 23, 10, 4, 1};
      for (int gi = 0; gi < 9; ++gi) {
        const int gap = gaps[gi];
        for (int i = gap; i < n; ++i) {
          const int v = a[i];
          int j = i;
          for (; j >= gap && a[j - gap] > v; j -= gap) a[j] = a[j - gap];
          a[j] = v;
        }
      }
    }
    return;
  }
  int m = 2;
  while (m < n) m <<= 1;
  for (int i = lane; i < m; i += 64) buf[i] = i < n ? a[i] : 0x7fffffff;
  __syncthreads();
  for (int k = 2; k <= m; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = lane; i < m; i += 64) {
        const int l = i ^ j;
        if (l > i) {
          const int x = buf[i], y = buf[l];
          const bool up = (i & k) == 0;
          if ((x > y) == up) { buf[i] = y; buf[l] = x; }
        }
      }
      __syncthreads();
    }
  for (int i = lane; i < n; i += 64) a[i] = buf[i];
}

// thread = (point row, column); rows_per_block = kThreads / C
__global__ __launch_bounds__(kThreads) void sa_gather_rows_kernel(long R, int C, const float *__restrict__ dX, int ldx,
                                                                  const int *__restrict__ start,
                                                                  const int *__restrict__ list,
                                                                  float *__restrict__ d_feats) {
  BUTD_MAIN_PRIO_SET();
  const int c = threadIdx.x % C;
  const long r = (long)blockIdx.x * (kThreads / C) + threadIdx.x / C;
  if (r >= R) return;
  const int a = start[r], b = start[r + 1];
  float acc = 0.f;
  int j = a;
  for (; j + 3 < b; j += 4) {
    const int p0 = list[j], p1 = list[j + 1], p2 = list[j + 2], p3 = list[j + 3];
    const float v0 = dX[(long)p0 * ldx + 3 + c], v1 = dX[(long)p1 * ldx + 3 + c], v2 = dX[(long)p2 * ldx + 3 + c],
                v3 = dX[(long)p3 * ldx + 3 + c];
    acc += (v0 + v1) + (v2 + v3);
  }
  for (; j < b; ++j) acc += dX[(long)list[j] * ldx + 3 + c];
  d_feats[r * C + c] = acc;
}

}  // namespace

extern "C" {

int butd_sa_group(int B, int N, int np, int ns, int C, const float *xyz, const float *new_xyz,
                  const float *feats, long feat_stride, const int *idx, float radius, int normalize,
                  float *X, int ldx, butd_stream_t stream) {
  if (ldx < 3 + C) return (int)hipErrorInvalidValue;
  const long total = (long)B * np * ns * ldx;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(sa_group_kernel, dim3(blocks_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                     N, np, ns, C, xyz, new_xyz, feats, feat_stride, idx, radius, normalize, X, ldx, total);
  return (int)hipGetLastError();
}

int butd_sa_thin_conv(long P, int C, int K, const float *X, int ldx, const float *W, float *Z,
                      double *sum, double *sumsq, butd_stream_t stream) {
  if (P <= 0) return 0;
  if (!cols_ok(C) || K < 1 || K > 8 || ldx != 8 || ((sum == nullptr) != (sumsq == nullptr)))
    return (int)hipErrorInvalidValue;
  const int chunk = chunk_rows(P);
  hipLaunchKernelGGL(sa_thin_conv_kernel, dim3((unsigned)((P + chunk - 1) / chunk)), dim3(kThreads), 0,
                     (hipStream_t)stream, P, C, K, X, ldx, W, Z, sum, sumsq, chunk);
  return (int)hipGetLastError();
}

int butd_sa_colstats(long P, int C, const float *Z, double *sum, double *sumsq, int pool_ns,
                     float *zmax, float *zmin, uint8_t *amax, uint8_t *amin, butd_stream_t stream) {
  if (P <= 0) return 0;
  const int chunk = chunk_rows(P);
  if (!cols_ok(C) || (pool_ns > 0 && (chunk % pool_ns || P % pool_ns || pool_ns % 4)))
    return (int)hipErrorInvalidValue;
  const unsigned blocks = (unsigned)((P + chunk - 1) / chunk);
  hipLaunchKernelGGL(sa_colstats_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, P, C, Z,
                     sum, sumsq, pool_ns, zmax, zmin, amax, amin, chunk);
  return (int)hipGetLastError();
}

int butd_sa_bn_finalize(int C, long count, const double *sum, const double *sumsq, int slots,
                        long slot_stride, const float *gamma, const float *beta, float eps, float momentum, int training, float *running_mean,
                        float *running_var, int64_t *num_batches_tracked, float *mean, float *rstd,
                        float *scale, float *shift, butd_stream_t stream) {
  if (C <= 0) return 0;
  hipLaunchKernelGGL(sa_bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C,
                     count, sum, sumsq, slots, slot_stride, gamma, beta, eps, momentum, training, running_mean,
                     running_var,
                     num_batches_tracked, mean, rstd, scale, shift);
  return (int)hipGetLastError();
}

int butd_sa_pool_finalize(int B, int np, int C, const float *zmax, const float *zmin,
                          const uint8_t *amax, const uint8_t *amin, const float *scale,
                          const float *shift, float *out_cm, float *out_pm, float *zsel, uint8_t *asel,
                          butd_stream_t stream) {
  const long total = (long)B * np * C;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(sa_pool_finalize_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)),
                     dim3(kThreads), 0, (hipStream_t)stream, np, C, total, zmax, zmin, amax, amin, scale,
                     shift, out_cm, out_pm, zsel, asel);
  return (int)hipGetLastError();
}

int butd_sa_pool_bwd_stats(int B, int np, int C, const float *d_out_pm, const float *zsel,
                           const float *scale, const float *shift, const float *mean,
                           const float *rstd, double *S1, double *S2, butd_stream_t stream) {
  const long G = (long)B * np;
  if (G <= 0) return 0;
  if (!cols_ok(C)) return (int)hipErrorInvalidValue;
  // few groups per workgroup while that keeps the grid small enough for the per-column atomics
  const int gpb = G >= 8192 ? 32 : 8;
  hipLaunchKernelGGL(sa_pool_bwd_stats_kernel, dim3((unsigned)((G + gpb - 1) / gpb)),
                     dim3(kThreads), 0, (hipStream_t)stream, np, C, G, gpb, d_out_pm, zsel, scale, shift, mean,
                     rstd, S1, S2);
  return (int)hipGetLastError();
}

int butd_sa_dz_last(int B, int np, int ns, int C, float *Z, const float *d_out_pm, const float *zsel,
                    const uint8_t *asel, const float *gamma, const float *scale, const float *shift,
                    const float *mean, const float *rstd, const double *S1, const double *S2,
                    int training, butd_stream_t stream) {
  const long P = (long)B * np * ns;
  if (P <= 0) return 0;
  if (!cols_ok(C)) return (int)hipErrorInvalidValue;
  const int chunk = chunk_rows(P);
  hipLaunchKernelGGL(sa_dz_last_kernel, dim3((unsigned)((P + chunk - 1) / chunk)),
                     dim3(kThreads), 0, (hipStream_t)stream, np, ns, C, P, Z, d_out_pm, zsel, asel, gamma,
                     scale, shift, mean, rstd, S1, S2, training, chunk);
  return (int)hipGetLastError();
}

int butd_sa_mask_stats(long P, int C, const float *dH, const float *Z, const float *scale,
                       const float *shift, const float *mean, const float *rstd, double *S1,
                       double *S2, butd_stream_t stream) {
  if (P <= 0) return 0;
  if (!cols_ok(C)) return (int)hipErrorInvalidValue;
  const int chunk = chunk_rows(P);
  const unsigned blocks = (unsigned)((P + chunk - 1) / chunk);
  hipLaunchKernelGGL(sa_mask_stats_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, P, C,
                     dH, Z, scale, shift, mean, rstd, S1, S2, chunk);
  return (int)hipGetLastError();
}

int butd_sa_dz_mid(long P, int C, float *g, const float *Z, const float *gamma, const float *scale,
                   const float *shift, const float *mean, const float *rstd, const double *S1, const double *S2,
                   int training, butd_stream_t stream) {
  if (P <= 0) return 0;
  if (!cols_ok(C)) return (int)hipErrorInvalidValue;
  const int chunk = chunk_rows(P);
  hipLaunchKernelGGL(sa_dz_mid_kernel, dim3((unsigned)((P + chunk - 1) / chunk)),
                     dim3(kThreads), 0, (hipStream_t)stream, P, C, g, Z, gamma, scale, shift, mean, rstd,
                     S1, S2, training, chunk);
  return (int)hipGetLastError();
}

int butd_sa_scatter_rows(int B, int N, int np, int ns, int C, const float *dX, int ldx, const int *idx,
                         float *d_feats_pm, butd_stream_t stream) {
  if (ldx < 3 + C) return (int)hipErrorInvalidValue;
  const long total = (long)B * np * ns * C;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(sa_scatter_rows_kernel, dim3(blocks_for(total, 65536)), dim3(kThreads), 0,
                     (hipStream_t)stream, N, np, ns, C, dX, idx, d_feats_pm, ldx, total);
  return (int)hipGetLastError();
}

int butd_sa_inverse_index(int B, int N, int np, int ns, const int *idx, int *count, int *start, int *list,
                          butd_stream_t stream) {
  const long npns = (long)np * ns, P = (long)B * npns;
  if (P <= 0) return 0;
  if (P >= (1L << 31) || N <= 0) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sa_inv_count_kernel, dim3(blocks_for(P, 65536)), dim3(kThreads), 0, st, N, npns, P, idx, count);
  hipLaunchKernelGGL(sa_inv_scan_kernel, dim3(B), dim3(1024), 0, st, N, npns, count, start, B - 1);
  hipLaunchKernelGGL(sa_inv_fill_kernel, dim3(blocks_for(P, 65536)), dim3(kThreads), 0, st, N, npns, P, idx, count, start, list);
  const long R = (long)B * N;
  hipLaunchKernelGGL(sa_inv_sort_kernel, dim3((unsigned)R), dim3(64), 0, st, R, start, list);
  return (int)hipGetLastError();
}

int butd_sa_gather_rows(int B, int N, int C, const float *dX, int ldx, const int *start, const int *list,
                        float *d_feats_pm, butd_stream_t stream) {
  const long R = (long)B * N;
  if (R <= 0 || C <= 0) return 0;
  if (ldx < 3 + C || C > kThreads || kThreads % C) return (int)hipErrorInvalidValue;
  const int rpb = kThreads / C;
  hipLaunchKernelGGL(sa_gather_rows_kernel, dim3((unsigned)((R + rpb - 1) / rpb)), dim3(kThreads), 0, (hipStream_t)stream, R, C,
                     dX, ldx, start, list, d_feats_pm);
  return (int)hipGetLastError();
}

}  // extern "C"
