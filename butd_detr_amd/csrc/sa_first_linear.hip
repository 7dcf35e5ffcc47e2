// sa_first_linear.hip -- the FIRST shared-MLP layer of a set-abstraction level that has input features (SA2-SA4), forward
// and backward, by LINEARITY of the grouping: the grouped input X (P = B*npoint*nsample rows) is never formed
// (include/butd_sa.h, butd_sa_first_linear_fwd / _bwd; round 6).
//
// Reference arithmetic: QueryAndGroup (pointnet2_utils.py:317-376) builds, for grouped row p = (b, j, k) with source point
// n = idx[b, j, k],
//     X[p, :] = [ (xyz[b, n] - new_xyz[b, j]) (/ radius),  feats[b, n, :] ]                       (P x (3 + C))
// and the first SharedMLP layer (pytorch_utils.py:11-36) is the 1x1 convolution Z1 = X W1^T followed by BatchNorm + ReLU.
// A grouped row is a COPY of a point's features, so with W1 = [Wx | Wf] (3 | C columns)
//     Z1[p, :] = Y[b, n, :] + dxyz[p] Wx^T,          Y = feats Wf^T                                  (B*N rows, not P)
// -- the C-wide product runs over the N points of the level (P / N = 16 times fewer rows at SA2) and the P x C1 result is
// a GATHER of its rows plus three multiply-adds; X (P x 132: 138 MB at SA2, B = 8) is neither written nor read.
// Backward, with dZ1 the BatchNorm backward of the gated gradient G1 (what butd_sa_dz_mid forms in place),
//     T[b, n, :]  = sum over the grouped rows p that copy point n of dZ1[p, :]                     (the inverted lists)
//     d_feats     = T Wf              dWf = T^T feats                 (products over B*N rows: the caller's grouped GEMM)
//     dWx[c, i]   = sum_p dZ1[p, c] dxyz[p, i]                        (taken in the same pass as T)
// so neither dZ1 nor dX (P x 132) exists in memory: one read of (G1, Z1) instead of butd_sa_dz_mid + the weight- /
// input-gradient product pair + butd_sa_gather_rows (1.15 GB + 0.15 GB -> 0.27 GB at SA2, B = 8).
// Summation order: the C feature terms of a Z1 element are summed by the matrix cores, the three coordinate terms are added
// after them (the dense path sums the coordinate terms first) -- fp32 reassociation, covered by the parity tests.
//
//   sa_first_linear_fwd_kernel     Z1 rows + the BatchNorm column sums (double atomics into `slots` private copies)
//   sa_first_linear_bwd_kernel     T and per-workgroup partials of dWx  (a 32-lane group walks one point's list)
//   sa_first_linear_dwx_kernel     partials -> dWx (summed in double, fixed order: bit-reproducible)
#include <hip/hip_runtime.h>
// (-DBUTD_MAIN_PRIO=n raises the wave priority of this unit's kernels: the captured step's prefetch branches share CUs
// with them; profiles/r06_side_branches.txt)
#ifdef BUTD_MAIN_PRIO
#define BUTD_MAIN_PRIO_SET() __builtin_amdgcn_s_setprio(BUTD_MAIN_PRIO)
#else
#define BUTD_MAIN_PRIO_SET()
#endif

#include <stdint.h>

#include "../../include/butd_sa.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kFwdRows = 128;      // grouped rows per workgroup of the forward gather (a thread: 16 rows x 4 channels at C1 = 128)
constexpr int kBwdGrid = 4096;     // most workgroups of the backward pass (a lane group owns one point at a time: the lists are uneven)

template <int C1>
__global__ __launch_bounds__(kThreads) void sa_first_linear_fwd_kernel(
    long P, int N, int np, int ns, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const int *__restrict__ idx, float radius, int normalize, const float *__restrict__ Y, const float *__restrict__ W1,
    long ldw, float *__restrict__ Z1, double *__restrict__ sum, double *__restrict__ sumsq, int slots, long slot_stride) {
  BUTD_MAIN_PRIO_SET();
  constexpr int QN = C1 / 4, RP = kThreads / QN;
  __shared__ float red[RP][2][C1];
  const int tid = threadIdx.x, q = tid % QN, rsub = tid / QN;
  f4 wx[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) wx[j][e] = W1[(long)(4 * q + e) * ldw + j];
  const long row0 = (long)blockIdx.x * kFwdRows;
  const long npns = (long)np * ns;
  f4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  constexpr int U = 4;                 // rows in flight per thread: index -> (coordinates, Y row) is a dependent chain
  for (int r0 = rsub; r0 < kFwdRows; r0 += U * RP) {
    long pp[U], src[U];
    bool in[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      pp[u] = row0 + r0 + u * RP;
      in[u] = r0 + u * RP < kFwdRows && pp[u] < P;
      src[u] = in[u] ? (pp[u] / npns) * N + idx[pp[u]] : 0;
    }
    f4 y[U];
    float d[U][3];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      y[u] = in[u] ? *reinterpret_cast<const f4 *>(Y + src[u] * C1 + 4 * q) : f4{0.f, 0.f, 0.f, 0.f};
      const long g = in[u] ? pp[u] / ns : 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) d[u][j] = xyz[src[u] * 3 + j] - new_xyz[g * 3 + j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!in[u]) continue;
      if (normalize) {
#pragma unroll
        for (int j = 0; j < 3; ++j) d[u][j] = d[u][j] / radius;
      }
      f4 z;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        z[e] = ((y[u][e] + d[u][0] * wx[0][e]) + d[u][1] * wx[1][e]) + d[u][2] * wx[2][e];
        s1[e] += z[e];
        s2[e] += z[e] * z[e];
      }
      __builtin_nontemporal_store(z, reinterpret_cast<f4 *>(Z1 + pp[u] * C1 + 4 * q));
    }
  }
  if (sum == nullptr) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[rsub][0][4 * q + e] = s1[e];
    red[rsub][1][4 * q + e] = s2[e];
  }
  __syncthreads();
  const int slot = slots > 1 ? (int)(blockIdx.x & (unsigned)(slots - 1)) : 0;
  for (int i = tid; i < 2 * C1; i += kThreads) {
    const int which = i / C1, c = i - which * C1;
    double a = 0.0;
#pragma unroll
    for (int t = 0; t < RP; ++t) a += (double)red[t][which][c];
    atomicAdd((which ? sumsq : sum) + (long)slot * slot_stride + c, a);
  }
}

template <int C1>
__global__ __launch_bounds__(kThreads) void sa_first_linear_bwd_kernel(
    long R, long P, int ns, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const int *__restrict__ start, const int *__restrict__ list, float radius, int normalize,
    const float *__restrict__ G1, const float *__restrict__ Z1, const float *__restrict__ gamma,
    const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ mean,
    const float *__restrict__ rstd, const double *__restrict__ S1, const double *__restrict__ S2, int training,
    float *__restrict__ T, float *__restrict__ ws) {
  BUTD_MAIN_PRIO_SET();
  constexpr int QN = C1 / 4, RP = kThreads / QN;
  __shared__ float red[RP][3][C1];
  const int tid = threadIdx.x, q = tid % QN, rsub = tid / QN;
  const double invP = 1.0 / (double)P;
  f4 k0, k1, k2, sc, sh;            // dZ = k0 g - k1 - k2 (z - mu)   (training);   dZ = scale g   (running statistics)
  f4 mu;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * q + e;
    sc[e] = scale[c]; sh[e] = shift[c]; mu[e] = mean[c];
    const float ga = gamma[c], rs = rstd[c];
    const float a1 = (float)(S1[c] * invP), a2 = (float)(S2[c] * invP);
    k0[e] = training ? ga * rs : sc[e];
    k1[e] = training ? a1 : 0.f;
    k2[e] = training ? rs * a2 : 0.f;
  }
  f4 wacc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) wacc[j] = f4{0.f, 0.f, 0.f, 0.f};
  constexpr int U = 4;                 // list entries in flight per lane group
  for (long pt = (long)blockIdx.x * RP + rsub; pt < R; pt += (long)gridDim.x * RP) {
    const int a = start[pt], b = start[pt + 1];
    const float px[3] = {xyz[pt * 3], xyz[pt * 3 + 1], xyz[pt * 3 + 2]};
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = a; j0 < b; j0 += U) {
      long p[U];
      bool in[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        in[u] = j0 + u < b;
        p[u] = in[u] ? (long)list[j0 + u] : 0;
      }
      f4 gv[U], zv[U];
      float d[U][3];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        gv[u] = in[u] ? __builtin_nontemporal_load(reinterpret_cast<const f4 *>(G1 + p[u] * C1 + 4 * q)) : f4{0.f, 0.f, 0.f, 0.f};
        zv[u] = in[u] ? __builtin_nontemporal_load(reinterpret_cast<const f4 *>(Z1 + p[u] * C1 + 4 * q)) : f4{0.f, 0.f, 0.f, 0.f};
        const long g = p[u] / ns;
#pragma unroll
        for (int i = 0; i < 3; ++i) d[u][i] = px[i] - new_xyz[g * 3 + i];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!in[u]) continue;
        if (normalize) {
#pragma unroll
          for (int i = 0; i < 3; ++i) d[u][i] = d[u][i] / radius;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float gg = sc[e] * zv[u][e] + sh[e] > 0.f ? gv[u][e] : 0.f;      // (idempotent: G1 arrives gated)
          // the arithmetic of butd_sa_dz_mid: gamma rstd (g - a1 - (z - mu) rstd a2)
          const float dz = training ? k0[e] * (gg - k1[e] - (zv[u][e] - mu[e]) * k2[e]) : k0[e] * gg;
          acc[e] += dz;
#pragma unroll
          for (int i = 0; i < 3; ++i) wacc[i][e] += dz * d[u][i];
        }
      }
    }
    *reinterpret_cast<f4 *>(T + pt * C1 + 4 * q) = acc;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rsub][j][4 * q + e] = wacc[j][e];
  __syncthreads();
  float *out = ws + (long)blockIdx.x * 3 * C1;
  for (int i = tid; i < 3 * C1; i += kThreads) {
    const int j = i / C1, c = i - j * C1;
    float a = 0.f;
#pragma unroll
    for (int t = 0; t < RP; ++t) a += red[t][j][c];
    out[i] = a;
  }
}

// dWx[c, j] = sum over the workgroups' partials: one workgroup per element, a fixed tree in double (bit-reproducible)
__global__ __launch_bounds__(256) void sa_first_linear_dwx_kernel(int C1, int parts, const float *__restrict__ ws,
                                                                  float *__restrict__ dWx, long ld) {
  BUTD_MAIN_PRIO_SET();
  __shared__ double red[256];
  const int i = blockIdx.x;            // j * C1 + c
  const int j = i / C1, c = i - j * C1;
  double a = 0.0;
  for (int t = threadIdx.x; t < parts; t += 256) a += (double)ws[(long)t * 3 * C1 + i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) dWx[(long)c * ld + j] = (float)red[0];
}

inline int bwd_grid(long R) {
  const long want = (R + 7) / 8;
  return (int)(want < kBwdGrid ? (want < 1 ? 1 : want) : kBwdGrid);
}

}  // namespace

extern "C" {

int butd_sa_first_linear_supported(int C1) { return C1 == 128 || C1 == 64 || C1 == 256; }

int butd_sa_first_linear_fwd(int B, int N, int np, int ns, int C1, const float *xyz, const float *new_xyz, const int *idx,
                             float radius, int normalize, const float *Y, const float *W1, long ldw, float *Z1, double *sum,
                             double *sumsq, int slots, long slot_stride, butd_stream_t stream) {
  const long P = (long)B * np * ns;
  if (P <= 0) return 0;
  if (!butd_sa_first_linear_supported(C1) || !xyz || !new_xyz || !idx || !Y || !W1 || !Z1 || ldw < 3 ||
      ((sum == nullptr) != (sumsq == nullptr)) || (slots > 1 && (slots & (slots - 1))) ||
      ((((uintptr_t)Y) | ((uintptr_t)Z1)) & 15))
    return (int)hipErrorInvalidValue;
  const dim3 grid((unsigned)((P + kFwdRows - 1) / kFwdRows)), blk(kThreads);
  hipStream_t st = (hipStream_t)stream;
#define BUTD_FWD(C)                                                                                                   \
  hipLaunchKernelGGL((sa_first_linear_fwd_kernel<C>), grid, blk, 0, st, P, N, np, ns, xyz, new_xyz, idx, radius,      \
                     normalize, Y, W1, ldw, Z1, sum, sumsq, slots, slot_stride)
  if (C1 == 128) BUTD_FWD(128);
  else if (C1 == 64) BUTD_FWD(64);
  else BUTD_FWD(256);
#undef BUTD_FWD
  return (int)hipGetLastError();
}

int butd_sa_first_linear_bwd_scratch(int B, int N, int C1, long *ws_floats) {
  if (!ws_floats || !butd_sa_first_linear_supported(C1) || (long)B * N <= 0) return (int)hipErrorInvalidValue;
  *ws_floats = (long)bwd_grid((long)B * N) * 3 * C1;
  return 0;
}

int butd_sa_first_linear_bwd(int B, int N, int np, int ns, int C1, const float *xyz, const float *new_xyz,
                             const int *start, const int *list, float radius, int normalize, const float *G1,
                             const float *Z1, const float *gamma1, const float *scale1, const float *shift1,
                             const float *mean1, const float *rstd1, const double *S1, const double *S2, int training,
                             float *T, float *dWx, long ld_dwx, float *ws, butd_stream_t stream) {
  const long R = (long)B * N, P = (long)B * np * ns;
  if (R <= 0 || P <= 0) return 0;
  if (!butd_sa_first_linear_supported(C1) || !xyz || !new_xyz || !start || !list || !G1 || !Z1 || !gamma1 || !scale1 ||
      !shift1 || !mean1 || !rstd1 || !S1 || !S2 || !T || !dWx || !ws || ld_dwx < 3 ||
      ((((uintptr_t)G1) | ((uintptr_t)Z1) | ((uintptr_t)T)) & 15))
    return (int)hipErrorInvalidValue;
  const int grid = bwd_grid(R);
  hipStream_t st = (hipStream_t)stream;
#define BUTD_BWD(C)                                                                                                   \
  hipLaunchKernelGGL((sa_first_linear_bwd_kernel<C>), dim3(grid), dim3(kThreads), 0, st, R, P, ns, xyz, new_xyz, start, \
                     list, radius, normalize, G1, Z1, gamma1, scale1, shift1, mean1, rstd1, S1, S2, training, T, ws)
  if (C1 == 128) BUTD_BWD(128);
  else if (C1 == 64) BUTD_BWD(64);
  else BUTD_BWD(256);
#undef BUTD_BWD
  hipLaunchKernelGGL(sa_first_linear_dwx_kernel, dim3(3 * C1), dim3(256), 0, st, C1, grid, ws, dWx, ld_dwx);
  return (int)hipGetLastError();
}

}  // extern "C"
