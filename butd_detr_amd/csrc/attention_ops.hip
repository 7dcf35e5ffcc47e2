// attention_ops.hip -- fp32 MFMA building blocks of the fused cross-modal attention / FFN path
// (include/butd_attention.h).  gfx950 only.
//
//   gemm_kernel          grouped dense products on v_mfma_f32_16x16x4_f32 (exact fp32, 157 TF peak):
//                        64x64 output tile per 4-wave workgroup, 32x32 per wave (2x2 MFMA tiles), BK=16,
//                        both operands staged K-contiguous in LDS so a lane's four k-steps are ONE
//                        ds_read_b128.  One launch serves up to 4 problems (Q/K/V projections, or the
//                        three input-gradient products), with bias / scale / ReLU / dropout epilogues,
//                        operand-add on load (src + pos) and a virtual ones-column for bias gradients.
//   ln_fwd / ln_bwd      y = LayerNorm(residual + dropout(x)), one wave per row, column partial sums
//                        for dgamma/dbeta reduced per workgroup before touching global atomics.
//
// The k index of a contraction may be permuted freely as long as A and B use the same permutation;
// MFMA step s of lane-group g = lane>>4 consumes k = 4*g + s of the current 16-wide slab, which makes
// every operand fragment 16 contiguous bytes.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/butd_attention.h"
#include "rng.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBM = 64, kBN = 64, kBK = 16, kLd = kBK + 4;  // LDS row stride 20 floats = 80 B
constexpr int kGemmThreads = 256;
constexpr int kMaxProblems = 4;

struct GemmBatch {
  butd_gemm_problem p[kMaxProblems];
  int z_begin[kMaxProblems + 1];  // blockIdx.z range of each problem (split_k slices)
  int count;
};

// Stage a (rows x 16) slab of an operand into LDS as tile[row][k].
//   element(row, k) = src[row*ld_row + k*ld_k] (+ src2[...]);  rows >= nrows (and k >= kend) read 0,
//   except the virtual ones-row (row == nrows && ones): 1.0 wherever k is in range.
__device__ inline float combine(float a, float a2, int mode, float gate_scale) {
  return mode == 0 ? a + a2 : a * (a2 > 0.f ? gate_scale : 0.f);
}

__device__ inline void stage_tile(float (*tile)[kLd], const float *__restrict__ src,
                                  const float *__restrict__ src2, int mode2, float scale2,
                                  long ld_row, long ld_k, int row0, int nrows, int k0, int kend,
                                  bool ones, int tid) {
  if (ld_k == 1) {  // contraction-contiguous: each thread moves 4 consecutive k of one row
    const int r = tid >> 2, kq = (tid & 3) * 4;
    const int gr = row0 + r, gk = k0 + kq;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (gr < nrows) {
      const float *p = src + (long)gr * ld_row + gk;
      const bool vec = (gk + 3 < kend) && ((ld_row & 3) == 0) && ((((uintptr_t)src) & 15) == 0);
      if (vec) {
        const float4 q = *reinterpret_cast<const float4 *>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        if (src2) {
          const float4 q2 = *reinterpret_cast<const float4 *>(src2 + (long)gr * ld_row + gk);
          v[0] = combine(v[0], q2.x, mode2, scale2); v[1] = combine(v[1], q2.y, mode2, scale2);
          v[2] = combine(v[2], q2.z, mode2, scale2); v[3] = combine(v[3], q2.w, mode2, scale2);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (gk + i < kend)
            v[i] = src2 ? combine(p[i], src2[(long)gr * ld_row + gk + i], mode2, scale2) : p[i];
      }
    } else if (ones && gr == nrows) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = (gk + i < kend) ? 1.f : 0.f;
    }
    *reinterpret_cast<float4 *>(&tile[r][kq]) = make_float4(v[0], v[1], v[2], v[3]);
  } else {  // row-contiguous: each thread moves 4 consecutive rows of one k and transposes into LDS
    const int k = tid >> 4, r4 = (tid & 15) * 4;
    const int gk = k0 + k, gr = row0 + r4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (gk < kend) {
      const float *p = src + (long)gk * ld_k + gr;
      const bool vec = (gr + 3 < nrows) && ((ld_k & 3) == 0) && ((((uintptr_t)src) & 15) == 0);
      if (vec) {
        const float4 q = *reinterpret_cast<const float4 *>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        if (src2) {
          const float4 q2 = *reinterpret_cast<const float4 *>(src2 + (long)gk * ld_k + gr);
          v[0] = combine(v[0], q2.x, mode2, scale2); v[1] = combine(v[1], q2.y, mode2, scale2);
          v[2] = combine(v[2], q2.z, mode2, scale2); v[3] = combine(v[3], q2.w, mode2, scale2);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (gr + i < nrows)
            v[i] = src2 ? combine(p[i], src2[(long)gk * ld_k + gr + i], mode2, scale2) : p[i];
          else if (ones && gr + i == nrows) v[i] = 1.f;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[r4 + i][k] = v[i];
  }
}

__global__ __launch_bounds__(kGemmThreads) void gemm_kernel(GemmBatch batch,
                                                            const uint64_t *__restrict__ rng_counter) {
  __shared__ __attribute__((aligned(16))) float As[kBM][kLd];
  __shared__ __attribute__((aligned(16))) float Bs[kBN][kLd];

  int pi = 0;
  while (pi + 1 < batch.count && (int)blockIdx.z >= batch.z_begin[pi + 1]) ++pi;
  const butd_gemm_problem &P = batch.p[pi];
  const int slice = blockIdx.z - batch.z_begin[pi];
  const int m0 = blockIdx.y * kBM, n0 = blockIdx.x * kBN;
  const int ncols = P.N + (P.ones_col ? 1 : 0);
  if (m0 >= P.M || n0 >= ncols) return;

  // contraction range of this split-K slice (multiples of kBK)
  const int kslab = (P.K + kBK - 1) / kBK;
  const int per = (kslab + P.split_k - 1) / P.split_k;
  const int kbeg = slice * per * kBK;
  const int kend = min(P.K, (slice + 1) * per * kBK);
  if (kbeg >= kend && slice > 0) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int k0 = kbeg; k0 < kend; k0 += kBK) {
    stage_tile(As, P.a, P.a2, P.a2_mode, P.a2_scale, P.lda_m, P.lda_k, m0, P.M, k0, kend, false, tid);
    stage_tile(Bs, P.b, nullptr, 0, 0.f, P.ldb_n, P.ldb_k, n0, P.N, k0, kend, P.ones_col != 0, tid);
    __syncthreads();
    f32x4 af[2], bf[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
      af[i] = *reinterpret_cast<const f32x4 *>(&As[wr * 32 + i * 16 + fr][fg * 4]);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      bf[j] = *reinterpret_cast<const f32x4 *>(&Bs[wc * 32 + j * 16 + fr][fg * 4]);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    __syncthreads();
  }

  // epilogue: lane holds C[row = fg*4 + r][col = fr] of each 16x16 tile
  const bool drop = P.dropout_p > 0.f;
  const float inv_keep = drop ? 1.f / (1.f - P.dropout_p) : 1.f;
  const uint64_t ctr = (drop && rng_counter) ? *rng_counter : 0ull;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wc * 32 + j * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wr * 32 + i * 16 + fg * 4 + r;
        if (m >= P.M) continue;
        float v = acc[i][j][r];
        if (n < P.N) {
          if (P.bias && slice == 0) v += P.bias[n];
          v *= P.scale;
          if (P.relu) v = fmaxf(v, 0.f);
          if (drop)
            v = rng::keep(ctr, P.dropout_site, (uint32_t)((long)m * P.N + n), P.dropout_p)
                    ? v * inv_keep : 0.f;
          float *dst = P.c + (long)m * P.ldc + n;
          if (P.accumulate) atomicAdd(dst, v);
          else *dst = v;
        } else if (P.ones_col && n == P.N) {
          atomicAdd(P.bias_grad + m, v * P.scale);
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// y = LayerNorm(residual + dropout(x))
// ------------------------------------------------------------------------------------------------
constexpr int kLnThreads = 256;
constexpr int kLnMaxPerLane = 16;  // cols <= 1024

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

template <int PER>
__global__ __launch_bounds__(kLnThreads) void ln_fwd_kernel(
    int rows, int cols, const float *__restrict__ x, const float *__restrict__ residual,
    const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
    float *__restrict__ y, float *__restrict__ mean, float *__restrict__ rstd, float dropout_p,
    uint32_t site, const uint64_t *__restrict__ rng_counter) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (kLnThreads / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bool drop = dropout_p > 0.f;
  const float inv_keep = drop ? 1.f / (1.f - dropout_p) : 1.f;
  const uint64_t ctr = (drop && rng_counter) ? *rng_counter : 0ull;
  float s[PER];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = lane + i * 64;
    float v = 0.f;
    if (c < cols) {
      float xv = x[(long)row * cols + c];
      if (drop) xv = rng::keep(ctr, site, (uint32_t)((long)row * cols + c), dropout_p) ? xv * inv_keep : 0.f;
      v = xv + (residual ? residual[(long)row * cols + c] : 0.f);
    }
    s[i] = v;
    sum += v;
  }
  const float mu = wave_sum(sum) / cols;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = lane + i * 64;
    const float d = c < cols ? s[i] - mu : 0.f;
    sq += d * d;
  }
  const float rs = rsqrtf(wave_sum(sq) / cols + eps);
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = lane + i * 64;
    if (c < cols) y[(long)row * cols + c] = (s[i] - mu) * rs * gamma[c] + beta[c];
  }
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
}

// Each workgroup walks kLnRowsPerBlock rows (4 waves x 16 rows), keeps dgamma/dbeta partials of its
// columns in registers, reduces them across its waves through LDS and issues ONE atomic per column.
constexpr int kLnRowsPerWave = 16;
template <int PER>
__global__ __launch_bounds__(kLnThreads) void ln_bwd_kernel(
    int rows, int cols, const float *__restrict__ dy, const float *__restrict__ x,
    const float *__restrict__ residual, const float *__restrict__ gamma,
    const float *__restrict__ mean, const float *__restrict__ rstd, float *__restrict__ dx,
    float *__restrict__ d_residual, float *__restrict__ dgamma, float *__restrict__ dbeta,
    float dropout_p, uint32_t site, const uint64_t *__restrict__ rng_counter) {
  __shared__ float red[2][kLnThreads / 64][64 * PER];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool drop = dropout_p > 0.f;
  const float inv_keep = drop ? 1.f / (1.f - dropout_p) : 1.f;
  const uint64_t ctr = (drop && rng_counter) ? *rng_counter : 0ull;
  float g[PER], pg[PER], pb[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = lane + i * 64;
    g[i] = c < cols ? gamma[c] : 0.f;
    pg[i] = 0.f;
    pb[i] = 0.f;
  }
  const int row0 = (blockIdx.x * (kLnThreads / 64) + wave) * kLnRowsPerWave;
  for (int rr = 0; rr < kLnRowsPerWave; ++rr) {
    const int row = row0 + rr;
    if (row >= rows) break;
    const float mu = mean[row], rs = rstd[row];
    float xh[PER], gy[PER];
    bool kp[PER];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = lane + i * 64;
      xh[i] = 0.f; gy[i] = 0.f; kp[i] = true;
      if (c < cols) {
        const long o = (long)row * cols + c;
        float xv = x[o];
        if (drop) {
          kp[i] = rng::keep(ctr, site, (uint32_t)o, dropout_p);
          xv = kp[i] ? xv * inv_keep : 0.f;
        }
        const float sv = xv + (residual ? residual[o] : 0.f);
        const float d = dy[o];
        xh[i] = (sv - mu) * rs;
        gy[i] = d * g[i];
        pg[i] += d * xh[i];
        pb[i] += d;
        c1 += gy[i];
        c2 += gy[i] * xh[i];
      }
    }
    c1 = wave_sum(c1) / cols;
    c2 = wave_sum(c2) / cols;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = lane + i * 64;
      if (c < cols) {
        const long o = (long)row * cols + c;
        const float ds = (gy[i] - c1 - xh[i] * c2) * rs;
        if (d_residual) d_residual[o] = ds;
        if (dx) dx[o] = kp[i] ? ds * inv_keep : 0.f;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    red[0][wave][lane + i * 64] = pg[i];
    red[1][wave][lane + i * 64] = pb[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < cols; c += kLnThreads) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < kLnThreads / 64; ++w) {
      a += red[0][w][c];
      b += red[1][w][c];
    }
    atomicAdd(dgamma + c, a);
    atomicAdd(dbeta + c, b);
  }
}

}  // namespace

extern "C" {

int butd_gemm_grouped(const butd_gemm_problem *problems, int count, const uint64_t *rng_counter,
                      butd_stream_t stream) {
  if (count <= 0) return 0;
  if (count > kMaxProblems) return (int)hipErrorInvalidValue;
  GemmBatch batch;
  int gx = 0, gy = 0, z = 0;
  batch.count = 0;
  for (int i = 0; i < count; ++i) {
    butd_gemm_problem p = problems[i];
    if (p.M <= 0 || p.N <= 0) continue;
    if (p.split_k < 1) p.split_k = 1;
    if (p.split_k > 1 && !p.accumulate) return (int)hipErrorInvalidValue;
    const int ncols = p.N + (p.ones_col ? 1 : 0);
    gx = max(gx, (ncols + kBN - 1) / kBN);
    gy = max(gy, (p.M + kBM - 1) / kBM);
    batch.z_begin[batch.count] = z;
    batch.p[batch.count++] = p;
    z += p.split_k;
  }
  if (batch.count == 0) return 0;
  for (int i = batch.count; i <= kMaxProblems; ++i) batch.z_begin[i] = z;
  hipLaunchKernelGGL(gemm_kernel, dim3(gx, gy, z), dim3(kGemmThreads), 0, (hipStream_t)stream, batch,
                     rng_counter);
  return (int)hipGetLastError();
}

#define LN_DISPATCH(KERNEL, ...)                                                                   \
  do {                                                                                             \
    const int per = (cols + 63) / 64;                                                              \
    if (per <= 4) hipLaunchKernelGGL((KERNEL<4>), grid, dim3(kLnThreads), 0, s, __VA_ARGS__);      \
    else if (per <= 5) hipLaunchKernelGGL((KERNEL<5>), grid, dim3(kLnThreads), 0, s, __VA_ARGS__); \
    else if (per <= 8) hipLaunchKernelGGL((KERNEL<8>), grid, dim3(kLnThreads), 0, s, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<16>), grid, dim3(kLnThreads), 0, s, __VA_ARGS__);              \
  } while (0)

int butd_add_dropout_layernorm_fwd(int rows, int cols, const float *x, const float *residual,
                                   const float *gamma, const float *beta, float eps, float *y,
                                   float *mean, float *rstd, float dropout_p, uint32_t dropout_site,
                                   const uint64_t *rng_counter, butd_stream_t stream) {
  if (rows <= 0) return 0;
  if (cols <= 0 || cols > 64 * kLnMaxPerLane) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((rows + kLnThreads / 64 - 1) / (kLnThreads / 64));
  LN_DISPATCH(ln_fwd_kernel, rows, cols, x, residual, gamma, beta, eps, y, mean, rstd, dropout_p,
              dropout_site, rng_counter);
  return (int)hipGetLastError();
}

int butd_add_dropout_layernorm_bwd(int rows, int cols, const float *dy, const float *x,
                                   const float *residual, const float *gamma, const float *mean,
                                   const float *rstd, float *dx, float *d_residual, float *dgamma,
                                   float *dbeta, float dropout_p, uint32_t dropout_site,
                                   const uint64_t *rng_counter, butd_stream_t stream) {
  if (rows <= 0) return 0;
  if (cols <= 0 || cols > 64 * kLnMaxPerLane) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const int rows_per_block = (kLnThreads / 64) * kLnRowsPerWave;
  const dim3 grid((rows + rows_per_block - 1) / rows_per_block);
  LN_DISPATCH(ln_bwd_kernel, rows, cols, dy, x, residual, gamma, mean, rstd, dx, d_residual, dgamma,
              dbeta, dropout_p, dropout_site, rng_counter);
  return (int)hipGetLastError();
}

}  // extern "C"

// ================================================================================================
// Attention core (flash-style, fp32 MFMA 16x16x4, head_dim <= 48)
// ================================================================================================
// All score tiles are computed TRANSPOSED so that every per-query quantity (running max, running sum,
// rescale factor, delta) is lane-local:   S^T[key][q] = sum_d K[key][d] Q[q][d]   has C-layout
// lane (c = lane&15, g = lane>>4) -> S^T[key = 4g+i][q = c], i = 0..3, and that register quartet is
// exactly the B-operand fragment (k = 4g+s, col = q) of the next product  O^T[n][q] += V^T[n][key] P^T.
// A wave owns 16 queries and walks the keys 64 at a time; the four lanes {c, c+16, c+32, c+48} that
// share a query combine their partial max / sum with v_permlane16_swap / v_permlane32_swap (VALU, no
// LDS).  K/V fragments come straight from global memory (the four waves of a workgroup read the same
// tile, so all but the first hit L1).
namespace {

constexpr int kAttnThreads = 256;
constexpr float kNegInf = -INFINITY;

__device__ inline float quad_max(float v) {  // over lanes c, c^16, c^32, c^48 ; result in all four
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ inline float quad_sum(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// additive score bias of key `kk`: 0 when it takes part, -inf when padded (mask byte != 0) or beyond Lk.
// Branch-free on purpose: the mask byte is always loaded (index clamped).
__device__ inline float key_bias(const uint8_t *__restrict__ mb, int kk, int Lk) {
  const int kc = kk < Lk ? kk : Lk - 1;
  const unsigned mv = mb ? (unsigned)mb[kc] : 0u;
  return (kk < Lk && mv == 0u) ? 0.f : kNegInf;
}

// row `r` of a (rows x D) head slice, elements d = g*NS + s, s = 0..NS-1 (zero outside)
template <int NS>
__device__ inline void load_row_frag(float (&f)[NS], const float *__restrict__ base, long stride, int r,
                                     int nrows, int g, int D) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int d = g * NS + s;
    f[s] = (r < nrows && d < D) ? base[(long)r * stride + d] : 0.f;
  }
}

template <int NS, int NT>
__global__ __launch_bounds__(kAttnThreads) void attn_fwd_kernel(
    int H, int Lq, int Lk, int D, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const uint8_t *__restrict__ mask, float *__restrict__ out,
    float *__restrict__ lse, float p_drop, uint32_t site, const uint64_t *__restrict__ rng_counter) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const long E = (long)H * D;
  const int q0 = blockIdx.x * 64 + wave * 16;
  if (q0 >= Lq) return;
  const float *qb = q + (long)b * Lq * E + h * D;
  const float *kb = k + (long)b * Lk * E + h * D;
  const float *vb = v + (long)b * Lk * E + h * D;
  const uint8_t *mb = mask ? mask + (long)b * Lk : nullptr;
  const bool drop = p_drop > 0.f;
  const float inv_keep = drop ? 1.f / (1.f - p_drop) : 1.f;
  const uint64_t ctr = (drop && rng_counter) ? *rng_counter : 0ull;
  const int qi = q0 + fr;  // this lane's query (column of every transposed tile)

  float qf[NS];
  load_row_frag<NS>(qf, qb, E, qi, Lq, fg, D);
  float m = kNegInf, l = 0.f;
  f32x4 o[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int key0 = 0; key0 < Lk; key0 += 64) {
    f32x4 st[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float kf[NS];
      load_row_frag<NS>(kf, kb, E, key0 + t * 16 + fr, Lk, fg, D);
      st[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NS; ++s)
        st[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s], qf[s], st[t], 0, 0, 0);
    }
    float tmax = kNegInf;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kk = key0 + t * 16 + fg * 4 + i;
        const float sc = st[t][i] + key_bias(mb, kk, Lk);  // -inf for masked / out-of-range keys
        st[t][i] = sc;
        tmax = fmaxf(tmax, sc);
      }
    tmax = quad_max(tmax);
    const float m_new = fmaxf(m, tmax);
    const bool dead = m_new == kNegInf;  // nothing but masked keys so far
    const float alpha = dead ? 1.f : __expf(m - m_new);
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float p = dead ? 0.f : __expf(st[t][i] - m_new);
        psum += p;
        float pd = p;
        if (drop) {
          const int kk = key0 + t * 16 + fg * 4 + i;
          const uint32_t idx = (uint32_t)((((long)b * H + h) * Lq + qi) * Lk + kk);
          pd = rng::keep(ctr, site, idx, p_drop) ? p * inv_keep : 0.f;
        }
        st[t][i] = pd;
      }
    l = l * alpha + psum;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o[nt] *= alpha;
    // O^T[n][q] += V^T[n][key] P^T[key][q]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int kk = key0 + t * 16 + fg * 4 + s;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int n = nt * 16 + fr;
          const float a = (kk < Lk && n < D) ? vb[(long)kk * E + n] : 0.f;
          o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, st[t][s], o[nt], 0, 0, 0);
        }
      }
    m = m_new;
  }
  l = quad_sum(l);
  if (qi < Lq) {
    const float inv_l = 1.f / l;  // l == 0 (every key masked) -> inf * 0 = NaN like torch's softmax
    float *ob = out + ((long)b * Lq + qi) * E + h * D;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = nt * 16 + fg * 4 + i;
        if (n < D) ob[n] = o[nt][i] * inv_l;
      }
    if (fg == 0) lse[((long)b * H + h) * Lq + qi] = m + __logf(l);
  }
}

// delta[b,h,q] = sum_n dO[q][n] * O[q][n]
__global__ __launch_bounds__(256) void attn_delta_kernel(int H, int Lq, int D, long total,
                                                         const float *__restrict__ out,
                                                         const float *__restrict__ dout,
                                                         float *__restrict__ delta) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over (b, q, h)
  if (i >= total) return;
  const int h = (int)(i % H);
  const long bq = i / H;
  const int qq = (int)(bq % Lq);
  const long b = bq / Lq;
  const float *o = out + bq * (long)H * D + (long)h * D;
  const float *g = dout + bq * (long)H * D + (long)h * D;
  float s = 0.f;
  for (int n = 0; n < D; ++n) s += o[n] * g[n];
  delta[(b * H + h) * Lq + qq] = s;
}

// dQ: same walk as the forward with K and V swapping roles.
template <int NS, int NT>
__global__ __launch_bounds__(kAttnThreads) void attn_bwd_dq_kernel(
    int H, int Lq, int Lk, int D, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const uint8_t *__restrict__ mask, const float *__restrict__ dout,
    const float *__restrict__ lse, const float *__restrict__ delta, float *__restrict__ dq,
    float p_drop, uint32_t site, const uint64_t *__restrict__ rng_counter) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const long E = (long)H * D;
  const int q0 = blockIdx.x * 64 + wave * 16;
  if (q0 >= Lq) return;
  const float *qb = q + (long)b * Lq * E + h * D;
  const float *gb = dout + (long)b * Lq * E + h * D;
  const float *kb = k + (long)b * Lk * E + h * D;
  const float *vb = v + (long)b * Lk * E + h * D;
  const uint8_t *mb = mask ? mask + (long)b * Lk : nullptr;
  const bool drop = p_drop > 0.f;
  const float inv_keep = drop ? 1.f / (1.f - p_drop) : 1.f;
  const uint64_t ctr = (drop && rng_counter) ? *rng_counter : 0ull;
  const int qi = q0 + fr;

  float qf[NS], gf[NS];
  load_row_frag<NS>(qf, qb, E, qi, Lq, fg, D);
  load_row_frag<NS>(gf, gb, E, qi, Lq, fg, D);
  const float my_lse = qi < Lq ? lse[((long)b * H + h) * Lq + qi] : 0.f;
  const float my_delta = qi < Lq ? delta[((long)b * H + h) * Lq + qi] : 0.f;
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int key0 = 0; key0 < Lk; key0 += 64) {
    f32x4 ds[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float kf[NS], vf[NS];
      load_row_frag<NS>(kf, kb, E, key0 + t * 16 + fr, Lk, fg, D);
      load_row_frag<NS>(vf, vb, E, key0 + t * 16 + fr, Lk, fg, D);
      f32x4 st = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s], qf[s], st, 0, 0, 0);  // S^T
        dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[s], gf[s], dp, 0, 0, 0);  // dP^T
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kk = key0 + t * 16 + fg * 4 + i;
        const float p = qi < Lq ? __expf(st[i] + key_bias(mb, kk, Lk) - my_lse) : 0.f;
        float dpe = dp[i];
        if (drop) {
          const uint32_t idx = (uint32_t)((((long)b * H + h) * Lq + qi) * Lk + kk);
          dpe = rng::keep(ctr, site, idx, p_drop) ? dpe * inv_keep : 0.f;
        }
        ds[t][i] = p * (dpe - my_delta);
      }
    }
    // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int kk = key0 + t * 16 + fg * 4 + s;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int d = nt * 16 + fr;
          const float a = (kk < Lk && d < D) ? kb[(long)kk * E + d] : 0.f;
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ds[t][s], acc[nt], 0, 0, 0);
        }
      }
  }
  if (qi < Lq) {
    float *ob = dq + ((long)b * Lq + qi) * E + h * D;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int d = nt * 16 + fg * 4 + i;
        if (d < D) ob[d] = acc[nt][i];
      }
  }
}

// dK, dV: a wave owns 16 keys (columns) and walks the queries; tiles are NOT transposed here:
// S[q][key] has C-layout lane (c = key, g) -> q = 4g+i, which is the B fragment of
// dV^T[n][key] += dO^T[n][q] P[q][key]  and  dK^T[d][key] += Q^T[d][q] dS[q][key].
template <int NS, int NT>
__global__ __launch_bounds__(kAttnThreads) void attn_bwd_dkv_kernel(
    int H, int Lq, int Lk, int D, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const uint8_t *__restrict__ mask, const float *__restrict__ dout,
    const float *__restrict__ lse, const float *__restrict__ delta, float *__restrict__ dk,
    float *__restrict__ dv, float p_drop, uint32_t site, const uint64_t *__restrict__ rng_counter) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const long E = (long)H * D;
  const int k0 = blockIdx.x * 64 + wave * 16;
  if (k0 >= Lk) return;
  const float *qb = q + (long)b * Lq * E + h * D;
  const float *gb = dout + (long)b * Lq * E + h * D;
  const float *kb = k + (long)b * Lk * E + h * D;
  const float *vb = v + (long)b * Lk * E + h * D;
  const float *lb = lse + ((long)b * H + h) * Lq;
  const float *db = delta + ((long)b * H + h) * Lq;
  const bool drop = p_drop > 0.f;
  const float inv_keep = drop ? 1.f / (1.f - p_drop) : 1.f;
  const uint64_t ctr = (drop && rng_counter) ? *rng_counter : 0ull;
  const int ki = k0 + fr;  // this lane's key (column)
  const float my_bias = key_bias(mask ? mask + (long)b * Lk : nullptr, ki, Lk);
  const bool key_ok = my_bias == 0.f;

  float kf[NS], vf[NS];
  load_row_frag<NS>(kf, kb, E, ki, Lk, fg, D);
  load_row_frag<NS>(vf, vb, E, ki, Lk, fg, D);
  f32x4 ak[NT], av[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    ak[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    av[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int qs = 0; qs < Lq; qs += 16) {
    float qf[NS], gf[NS];
    load_row_frag<NS>(qf, qb, E, qs + fr, Lq, fg, D);
    load_row_frag<NS>(gf, gb, E, qs + fr, Lq, fg, D);
    f32x4 st = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      st = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[s], kf[s], st, 0, 0, 0);  // S[q][key]
      dp = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[s], vf[s], dp, 0, 0, 0);  // dP[q][key]
    }
    f32x4 pd, ds;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int qq = qs + fg * 4 + i;
      const bool valid = key_ok && qq < Lq;
      const float p = valid ? __expf(st[i] - lb[qq]) : 0.f;
      float keepf = 1.f;
      if (drop) {
        const uint32_t idx = (uint32_t)((((long)b * H + h) * Lq + qq) * Lk + ki);
        keepf = rng::keep(ctr, site, idx, p_drop) ? inv_keep : 0.f;
      }
      pd[i] = p * keepf;
      ds[i] = valid ? p * (dp[i] * keepf - db[qq]) : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int qq = qs + fg * 4 + s;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + fr;
        const bool ok = qq < Lq && n < D;
        const float ag = ok ? gb[(long)qq * E + n] : 0.f;
        const float aq = ok ? qb[(long)qq * E + n] : 0.f;
        av[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag, pd[s], av[nt], 0, 0, 0);
        ak[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, ds[s], ak[nt], 0, 0, 0);
      }
    }
  }
  if (ki < Lk) {
    float *okp = dk + ((long)b * Lk + ki) * E + h * D;
    float *ovp = dv + ((long)b * Lk + ki) * E + h * D;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = nt * 16 + fg * 4 + i;
        if (n < D) {
          okp[n] = ak[nt][i];
          ovp[n] = av[nt][i];
        }
      }
  }
}

}  // namespace

extern "C" {

#define ATTN_DISPATCH(KERNEL, grid, ...)                                                            \
  do {                                                                                              \
    if (D <= 16) hipLaunchKernelGGL((KERNEL<4, 1>), grid, dim3(kAttnThreads), 0, s, __VA_ARGS__);   \
    else if (D <= 32) hipLaunchKernelGGL((KERNEL<8, 2>), grid, dim3(kAttnThreads), 0, s, __VA_ARGS__); \
    else if (D <= 36) hipLaunchKernelGGL((KERNEL<9, 3>), grid, dim3(kAttnThreads), 0, s, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<12, 3>), grid, dim3(kAttnThreads), 0, s, __VA_ARGS__);          \
  } while (0)

int butd_attention_fwd(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                       const float *v, const uint8_t *key_padding_mask, float *out, float *lse,
                       float dropout_p, uint32_t dropout_site, const uint64_t *rng_counter,
                       butd_stream_t stream) {
  if (B <= 0 || H <= 0 || Lq <= 0) return 0;
  if (D <= 0 || D > 48 || Lk <= 0) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((Lq + 63) / 64, H, B);
  ATTN_DISPATCH(attn_fwd_kernel, grid, H, Lq, Lk, D, q, k, v, key_padding_mask, out, lse, dropout_p,
                dropout_site, rng_counter);
  return (int)hipGetLastError();
}

int butd_attention_bwd(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                       const float *v, const uint8_t *key_padding_mask, const float *out,
                       const float *dout, const float *lse, float *delta, float *dq, float *dk,
                       float *dv, float dropout_p, uint32_t dropout_site,
                       const uint64_t *rng_counter, butd_stream_t stream) {
  if (B <= 0 || H <= 0 || Lq <= 0) return 0;
  if (D <= 0 || D > 48 || Lk <= 0) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)B * Lq * H;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, H, Lq,
                     D, total, out, dout, delta);
  const dim3 gq((Lq + 63) / 64, H, B), gk((Lk + 63) / 64, H, B);
  ATTN_DISPATCH(attn_bwd_dq_kernel, gq, H, Lq, Lk, D, q, k, v, key_padding_mask, dout, lse, delta, dq,
                dropout_p, dropout_site, rng_counter);
  ATTN_DISPATCH(attn_bwd_dkv_kernel, gk, H, Lq, Lk, D, q, k, v, key_padding_mask, dout, lse, delta,
                dk, dv, dropout_p, dropout_site, rng_counter);
  return (int)hipGetLastError();
}

}  // extern "C"
