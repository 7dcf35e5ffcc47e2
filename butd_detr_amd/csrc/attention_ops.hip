// attention_ops.hip -- fp32 MFMA building blocks of the fused cross-modal attention / FFN path
// (include/butd_attention.h).  gfx950 only.
//
//   (the grouped GEMM lives in gemm_ops.hip)
//   ln_fwd / ln_bwd      y = LayerNorm(residual + dropout(x)), one wave per row, column partial sums
//                        for dgamma/dbeta reduced per workgroup before touching global atomics.
//
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
#include <stdlib.h>

#include <type_traits>

#include "../../include/butd_attention.h"
#include "rng.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// y = LayerNorm(residual + dropout(x))
// ------------------------------------------------------------------------------------------------
constexpr int kLnThreads = 256;
constexpr int kLnMaxPerLane = 16;  // cols <= 1024

// wave-wide sum on the DPP crossbar: running sums inside each 16-lane row (row_shr 1/2/4/8), then
// row_bcast:15 / row_bcast:31 carry the row totals; lane 63 holds the total, v_readlane broadcasts it.
// (__shfl_xor would be six ds_bpermute round trips, ~100 cycles each.)
__device__ inline float wave_sum(float v) {
#define BUTD_ADD_DPP(CTRL, RMASK)                                                                  \
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 " CTRL " row_mask:" RMASK " bank_mask:0xf" : "+v"(v))
  BUTD_ADD_DPP("row_shr:1", "0xf");
  BUTD_ADD_DPP("row_shr:2", "0xf");
  BUTD_ADD_DPP("row_shr:4", "0xf");
  BUTD_ADD_DPP("row_shr:8", "0xf");
  BUTD_ADD_DPP("row_bcast:15", "0xa");
  BUTD_ADD_DPP("row_bcast:31", "0xc");
#undef BUTD_ADD_DPP
  asm volatile("s_nop 1" ::: "memory");
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}

template <int PER>
__global__ __launch_bounds__(kLnThreads) void ln_fwd_kernel(
    int rows, int cols, const float *__restrict__ x, const float *__restrict__ residual,
    const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
    float *__restrict__ y, float *__restrict__ mean, float *__restrict__ rstd, float dropout_p,
    uint32_t site, const uint64_t *__restrict__ rng_counter, const float *__restrict__ pos,
    float *__restrict__ y_pos) {
  BUTD_MAIN_PRIO_SET();
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
    if (c < cols) {
      const float out = (s[i] - mu) * rs * gamma[c] + beta[c];
      y[(long)row * cols + c] = out;
      // the next block's query input y + pos, written while y is in registers (it used to be a separate add)
      if (y_pos) y_pos[(long)row * cols + c] = out + pos[(long)row * cols + c];
    }
  }
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
}

// Each workgroup covers 64 rows (16 waves x 4 rows: short serial chains), keeps dgamma/dbeta partials
// of its columns in registers, reduces them across its waves through LDS and issues ONE atomic per
// column.
constexpr int kLnRowsPerWave = 4;
constexpr int kLnBwdThreads = 1024;
template <int PER>
__global__ __launch_bounds__(kLnBwdThreads) void ln_bwd_kernel(
    int rows, int cols, const float *__restrict__ dy, const float *__restrict__ x,
    const float *__restrict__ residual, const float *__restrict__ gamma,
    const float *__restrict__ mean, const float *__restrict__ rstd, float *__restrict__ dx,
    float *__restrict__ d_residual, float *__restrict__ dgamma, float *__restrict__ dbeta,
    float dropout_p, uint32_t site, const uint64_t *__restrict__ rng_counter, int rows_per_wave,
    float *__restrict__ partials) {
  BUTD_MAIN_PRIO_SET();
  __shared__ float red[2][kLnBwdThreads / 64][64 * PER];
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
  const int row0 = (blockIdx.x * (kLnBwdThreads / 64) + wave) * rows_per_wave;
  for (int rr = 0; rr < rows_per_wave; ++rr) {
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
  for (int c = threadIdx.x; c < cols; c += kLnBwdThreads) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < kLnBwdThreads / 64; ++w) {
      a += red[0][w][c];
      b += red[1][w][c];
    }
    if (partials) {   // this workgroup's column sums, folded by whoever runs next (no same-address atomics)
      partials[(long)blockIdx.x * 2 * cols + c] = a;
      partials[(long)blockIdx.x * 2 * cols + cols + c] = b;
    } else {
      atomicAdd(dgamma + c, a);
      atomicAdd(dbeta + c, b);
    }
  }
}

}  // namespace

extern "C" {

#define LN_DISPATCH_T(THREADS, KERNEL, ...)                                                     \
  do {                                                                                          \
    const int per = (cols + 63) / 64;                                                           \
    if (per <= 4) hipLaunchKernelGGL((KERNEL<4>), grid, dim3(THREADS), 0, s, __VA_ARGS__);      \
    else if (per <= 5) hipLaunchKernelGGL((KERNEL<5>), grid, dim3(THREADS), 0, s, __VA_ARGS__); \
    else if (per <= 8) hipLaunchKernelGGL((KERNEL<8>), grid, dim3(THREADS), 0, s, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<16>), grid, dim3(THREADS), 0, s, __VA_ARGS__);              \
  } while (0)
#define LN_DISPATCH(KERNEL, ...) LN_DISPATCH_T(kLnThreads, KERNEL, __VA_ARGS__)

int butd_add_dropout_layernorm_fwd(int rows, int cols, const float *x, const float *residual,
                                   const float *gamma, const float *beta, float eps, float *y,
                                   float *mean, float *rstd, float dropout_p, uint32_t dropout_site,
                                   const uint64_t *rng_counter, butd_stream_t stream) {
  if (rows <= 0) return 0;
  if (cols <= 0 || cols > 64 * kLnMaxPerLane) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((rows + kLnThreads / 64 - 1) / (kLnThreads / 64));
  LN_DISPATCH(ln_fwd_kernel, rows, cols, x, residual, gamma, beta, eps, y, mean, rstd, dropout_p,
              dropout_site, rng_counter, (const float *)nullptr, (float *)nullptr);
  return (int)hipGetLastError();
}

int butd_add_dropout_layernorm_fwd_pos(int rows, int cols, const float *x, const float *residual,
                                       const float *gamma, const float *beta, float eps, float *y,
                                       float *mean, float *rstd, float dropout_p, uint32_t dropout_site,
                                       const uint64_t *rng_counter, const float *pos, float *y_pos,
                                       butd_stream_t stream) {
  if (rows <= 0) return 0;
  if (cols <= 0 || cols > 64 * kLnMaxPerLane || (pos == nullptr) != (y_pos == nullptr)) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((rows + kLnThreads / 64 - 1) / (kLnThreads / 64));
  LN_DISPATCH(ln_fwd_kernel, rows, cols, x, residual, gamma, beta, eps, y, mean, rstd, dropout_p,
              dropout_site, rng_counter, pos, y_pos);
  return (int)hipGetLastError();
}

namespace {
// one row per wave while that still leaves workgroups for every CU to spare (a 2048-row call is 128
// workgroups); several rows per wave only for very tall inputs, where it trims the dgamma/dbeta atomics
// (measured: the column-sum atomics are half of the kernel's time at 8192 rows -- 512 workgroups on the same
// 576 addresses; two rows per wave halve them there: 21.2 -> 18.4 us.  Folding private copies with a
// last-workgroup ticket needs an agent-scope release per workgroup = an L2 write-back each: 159 us.)
inline int ln_bwd_rows_per_wave(int rows) {
  return rows >= 65536 ? kLnRowsPerWave : rows >= 8192 ? 2 : 1;      // (measured: profiles/r04_small_kernels.txt)
}
inline int ln_bwd_blocks(int rows) {
  const int rows_per_block = (kLnBwdThreads / 64) * ln_bwd_rows_per_wave(rows);
  return (rows + rows_per_block - 1) / rows_per_block;
}
int ln_bwd_launch(int rows, int cols, const float *dy, const float *x, const float *residual, const float *gamma,
                  const float *mean, const float *rstd, float *dx, float *d_residual, float *dgamma, float *dbeta,
                  float *partials, float dropout_p, uint32_t dropout_site, const uint64_t *rng_counter,
                  butd_stream_t stream) {
  if (rows <= 0) return 0;
  if (cols <= 0 || cols > 64 * kLnMaxPerLane) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const int rpw = ln_bwd_rows_per_wave(rows);
  const dim3 grid(ln_bwd_blocks(rows));
  LN_DISPATCH_T(kLnBwdThreads, ln_bwd_kernel, rows, cols, dy, x, residual, gamma, mean, rstd, dx, d_residual, dgamma,
                dbeta, dropout_p, dropout_site, rng_counter, rpw, partials);
  return (int)hipGetLastError();
}
}  // namespace

int butd_add_dropout_layernorm_bwd(int rows, int cols, const float *dy, const float *x,
                                   const float *residual, const float *gamma, const float *mean,
                                   const float *rstd, float *dx, float *d_residual, float *dgamma,
                                   float *dbeta, float dropout_p, uint32_t dropout_site,
                                   const uint64_t *rng_counter, butd_stream_t stream) {
  return ln_bwd_launch(rows, cols, dy, x, residual, gamma, mean, rstd, dx, d_residual, dgamma, dbeta, nullptr,
                       dropout_p, dropout_site, rng_counter, stream);
}

int butd_layernorm_bwd_blocks(int rows) { return rows > 0 ? ln_bwd_blocks(rows) : 0; }

int butd_add_dropout_layernorm_bwd_partial(int rows, int cols, const float *dy, const float *x,
                                           const float *residual, const float *gamma, const float *mean,
                                           const float *rstd, float *dx, float *d_residual, float *partials,
                                           float dropout_p, uint32_t dropout_site,
                                           const uint64_t *rng_counter, butd_stream_t stream) {
  if (partials == nullptr) return (int)hipErrorInvalidValue;
  return ln_bwd_launch(rows, cols, dy, x, residual, gamma, mean, rstd, dx, d_residual, nullptr, nullptr, partials,
                       dropout_p, dropout_site, rng_counter, stream);
}

}  // extern "C"

// ================================================================================================
// Attention core (flash-style, fp32 MFMA 16x16x4, head_dim <= 48, head_dim % 4 == 0)
// ================================================================================================
// All score tiles are computed TRANSPOSED so that every per-query quantity (running max, running sum,
// rescale factor, delta) is lane-local:   S^T[key][q] = sum_d K[key][d] Q[q][d]   has C-layout
// lane (c = lane&15, g = lane>>4) -> S^T[key = 4g+i][q = c], i = 0..3, and that register quartet is
// exactly the B-operand fragment (k = 4g+s, col = q) of the next product  O^T[n][q] += V^T[n][key] P^T.
// The four lanes {c, c+16, c+32, c+48} that share a query combine their partial max / sum with
// v_permlane16_swap / v_permlane32_swap (VALU, no LDS).
//
// A workgroup = 4 waves x 16 queries.  The 64-key K and V tiles are staged once per workgroup into
// LDS (coalesced float4 global loads, double-buffered, one barrier per tile: tile i+1 is in flight in
// registers while tile i is multiplied) in a "fragment" image: row = key, element d stored at
// [d / NS][d % NS] with each of the four d-groups padded to 16 bytes, so an MFMA operand fragment
// (NS consecutive d of one group) is two ds_read_b128 + one ds_read_b32, and the row stride of 52
// floats keeps both access patterns (whole fragments for S^T, single elements for V^T / K^T operands)
// essentially bank-conflict free.
namespace {

constexpr int kAttnThreads = 256;
constexpr float kNegInf = -INFINITY;

// ---- fp32 / bf16 matrix steps ---------------------------------------------------------------------------
// BF = false: v_mfma_f32_16x16x4_f32 (exact fp32).  BF = true (BASELINE configs[3], "bf16 attention"): operands
// rounded to bf16 (nearest even) in registers, v_mfma_f32_16x16x16_bf16, fp32 accumulation; softmax statistics,
// the exponentials and everything in memory stay fp32.  A contraction index may be permuted freely as long as
// both operands use the same permutation, so the head-dimension fragment of a lane (NS consecutive d) is simply
// cut into quartets (zero padded: 36 = 9 quartets of the four lane groups) -- the LDS images are the fp32 ones.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int NS, bool BF>
struct DFrag {                      // one lane's share of a head-dimension operand
  float f[NS];
};
template <int NS>
struct DFrag<NS, true> {
  bf16x4 v[(NS + 3) / 4];
};
template <int NS, bool BF>
__device__ inline DFrag<NS, BF> make_frag(const float (&f)[NS]) {
  DFrag<NS, BF> r;
  if constexpr (!BF) {
#pragma unroll
    for (int s = 0; s < NS; ++s) r.f[s] = f[s];
  } else {
#pragma unroll
    for (int m = 0; m < (NS + 3) / 4; ++m) {
      float e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) e[i] = (4 * m + i < NS) ? f[4 * m + i] : 0.f;
      r.v[m] = (bf16x4){(__bf16)e[0], (__bf16)e[1], (__bf16)e[2], (__bf16)e[3]};
    }
  }
  return r;
}
// acc += A . B over the head dimension
template <int NS, bool BF>
__device__ inline f32x4 mma_d(const DFrag<NS, BF> &a, const DFrag<NS, BF> &b, f32x4 acc) {
  if constexpr (!BF) {
#pragma unroll
    for (int s = 0; s < NS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[s], b.f[s], acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int m = 0; m < (NS + 3) / 4; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.v[m], b.v[m], acc, 0, 0, 0);
  }
  return acc;
}
// acc += A . B over the 16 rows of a score tile (this lane: rows 4g .. 4g+3), bf16 only: one instruction
__device__ inline f32x4 mma_k16(const float (&a)[4], const f32x4 &b, f32x4 acc) {
  const bf16x4 pa = {(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3]};
  const bf16x4 pb = {(__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, pb, acc, 0, 0, 0);
}

// Grid (tiles, H, B) -> (tile, h, b) of this workgroup.  Hardware workgroup i runs on XCD i % 8 and every XCD
// has its own L2: in launch order the 16 query tiles of one (b, h) would be sprayed over all eight XCDs and
// each L2 would fetch the same K / V slices (counter traffic 63 MB per forward launch for 20-38 MB of
// operands).  The logical index gives every XCD one contiguous range of the linear grid, i.e. whole
// (b, h) pairs; bijective for any grid size.
struct TileId { int t, h, b; };
__device__ inline TileId tile_id() {
  const int gx = (int)gridDim.x, gy = (int)gridDim.y;
  const int nb = gx * gy * (int)gridDim.z;
  const int i = (int)blockIdx.x + gx * ((int)blockIdx.y + gy * (int)blockIdx.z);
  const int xcd = i & 7, q = nb >> 3, r = nb & 7;
  const int l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (i >> 3);
  TileId id;
  id.t = l % gx;
  id.h = (l / gx) % gy;
  id.b = l / (gx * gy);
  return id;
}

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
// Branch-free on purpose: selects on MFMA accumulators behind short-circuit branches were miscompiled
// into wrong results by hipcc 7.2; the mask byte is always loaded (index clamped).
__device__ inline float key_bias(const uint8_t *__restrict__ mb, int kk, int Lk) {
  const int kc = kk < Lk ? kk : Lk - 1;
  const unsigned mv = mb ? (unsigned)mb[kc] : 0u;
  return (kk < Lk && mv == 0u) ? 0.f : kNegInf;
}

// row `r` of a (rows x D) head slice in global memory, elements d = g*NS + s (zero outside)
template <int NS>
__device__ inline void load_row_frag(float (&f)[NS], const float *__restrict__ base, long stride, int r,
                                     int nrows, int g, int D) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int d = g * NS + s;
    f[s] = (r < nrows && d < D) ? base[(long)r * stride + d] : 0.f;
  }
}

// ---- LDS fragment image of a 64-row head slice --------------------------------------------------
template <int NS>
struct Img {
  static constexpr int NSP = (NS + 3) / 4 * 4;  // slots per d-group, 16-byte multiple
  static constexpr int LD = 4 * NSP + 4;        // row stride in floats (52 for head_dim 36)
  static constexpr int kVecPerThread = (64 * NS + kAttnThreads - 1) / kAttnThreads;  // float4 per thread
  struct Regs { float4 v[kVecPerThread]; };

  // global -> registers: rows [row0, row0+64) of a (L x D) slice with row stride E; D = 4*NS' <= 4*NS
  static __device__ inline void fetch(Regs &r, const float *__restrict__ base, long E, int D, int row0,
                                      int L, int tid) {
    const int vpr = D >> 2;  // float4 per row
#pragma unroll
    for (int j = 0; j < kVecPerThread; ++j) {
      const int f = tid + j * kAttnThreads;
      const int row = f / vpr, c4 = f - row * vpr;
      const int gr = row0 + row;
      r.v[j] = (row < 64 && gr < L) ? *reinterpret_cast<const float4 *>(base + (long)gr * E + c4 * 4)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // registers -> LDS image
  static __device__ inline void commit(float (*img)[LD], const Regs &r, int D, int tid) {
    const int vpr = D >> 2;
#pragma unroll
    for (int j = 0; j < kVecPerThread; ++j) {
      const int f = tid + j * kAttnThreads;
      const int row = f / vpr, c4 = f - row * vpr;
      if (row < 64) {
        const float e[4] = {r.v[j].x, r.v[j].y, r.v[j].z, r.v[j].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int d = c4 * 4 + i;
          const int g = d / NS, sl = d - g * NS;
          img[row][g * NSP + sl] = e[i];
        }
      }
    }
  }
  // operand fragment of row `row`, group g: d = g*NS + s, s = 0..NS-1
  static __device__ inline void frag(float (&f)[NS], const float (*img)[LD], int row, int g) {
    const float *p = &img[row][g * NSP];
#pragma unroll
    for (int q = 0; q + 4 <= NS; q += 4) {
      const float4 v = *reinterpret_cast<const float4 *>(p + q);
      f[q] = v.x; f[q + 1] = v.y; f[q + 2] = v.z; f[q + 3] = v.w;
    }
#pragma unroll
    for (int q = NS / 4 * 4; q < NS; ++q) f[q] = p[q];
  }
  // column offset of element d inside a row of the image
  static __device__ inline int col(int d) {
    const int g = d / NS;
    return g * NSP + (d - g * NS);
  }
};

// ---- transposed LDS image of a 64-row head slice: row = head-dim element n, column = key ---------------------
// The P.V product  O^T[n][q] += V^T[n][key] P^T[key][q]  takes, per lane, V at FOUR CONSECUTIVE KEYS of one n: in the
// transposed image that is one ds_read_b128 (the row-major image needed four scalar reads and, because rows >= D do
// not exist there, a select per value).  Rows D .. 16*NT-1 are zeroed once and never written again.  LD = 68: rows
// 4 banks apart, so the 16 rows x 4 key groups of a wave's read cover all banks evenly.
template <int NT>
struct TImg {
  static constexpr int ROWS = NT * 16;
  static constexpr int LD = 64 + 4;
};

// One thread's share of staging 64-row tiles of (L x D) head slices: which float4s it moves and where they land in the
// fragment image (Img<NS>) resp. the transposed image (TImg).  Everything here depends on the thread id only and is
// computed once, outside the key loop (the per-tile cost used to be an integer division per element).
template <int NS, int NT>
struct Stage {
  using I = Img<NS>;
  static constexpr int kVec = I::kVecPerThread;
  int goff[kVec];       // element offset of the float4 inside the tile: row * E + 4 * c4
  int row[kVec];        // its row (for the ragged last tile); 64 = this thread has no j-th float4
  int koff[kVec][4];    // float offsets inside Img (row * LD + col(d))
  int toff[kVec];       // float offset of element 0 inside TImg (d * LD + row); elements 1..3 are + LD each
  __device__ inline void init(int tid, int D, long E) {
    const int vpr = D >> 2;
#pragma unroll
    for (int j = 0; j < kVec; ++j) {
      const int f = tid + j * kAttnThreads;
      const int r = vpr == NS ? f / NS : f / vpr, c4 = f - r * vpr;   // (a constant divisor in the usual case)
      const bool ok = r < 64;
      row[j] = ok ? r : 64;
      goff[j] = ok ? (int)(r * E) + c4 * 4 : 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) koff[j][i] = ok ? r * I::LD + I::col(c4 * 4 + i) : 0;
      toff[j] = ok ? (c4 * 4) * TImg<NT>::LD + r : 0;
    }
  }
  // rows [0, nrows) of the tile at `tile` exist (nrows >= 64: the common, unchecked case)
  __device__ inline void fetch(float4 (&v)[kVec], const float *__restrict__ tile, int nrows) const {
    if (nrows >= 64) {
#pragma unroll
      for (int j = 0; j < kVec; ++j)
        if (row[j] < 64) v[j] = *reinterpret_cast<const float4 *>(tile + goff[j]);
    } else {
#pragma unroll
      for (int j = 0; j < kVec; ++j)
        v[j] = row[j] < nrows ? *reinterpret_cast<const float4 *>(tile + goff[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ inline void commit_frag(float *img, const float4 (&v)[kVec]) const {
#pragma unroll
    for (int j = 0; j < kVec; ++j)
      if (row[j] < 64) {
        img[koff[j][0]] = v[j].x; img[koff[j][1]] = v[j].y; img[koff[j][2]] = v[j].z; img[koff[j][3]] = v[j].w;
      }
  }
  __device__ inline void commit_t(float *timg, const float4 (&v)[kVec]) const {
    constexpr int LD = TImg<NT>::LD;
#pragma unroll
    for (int j = 0; j < kVec; ++j)
      if (row[j] < 64) {
        float *p = timg + toff[j];
        p[0] = v[j].x; p[LD] = v[j].y; p[2 * LD] = v[j].z; p[3 * LD] = v[j].w;
      }
  }
};

// ---- dropout of attention probabilities ----------------------------------------------------------------------
// One 32-bit hash decides TWO neighbouring keys of a query (16 bits each: drop when the field is below
// round(p * 65536)), so the kernels that hold four consecutive keys per lane (forward, dQ) hash twice per quartet
// instead of four times.  Element (row, key): pair index row * ceil(Lk / 2) + key / 2, field key & 1.
constexpr float kLog2e = 1.4426950408889634f;
__device__ inline uint32_t drop_threshold(float p) { return (uint32_t)(p * 65536.f + 0.5f); }
__device__ inline uint32_t pair_hash(uint32_t key, uint32_t pair) { return rng::mix32(pair ^ key); }

// NG = 2 (small grids, e.g. the decoder's 256 queries: 256 workgroups = ONE wave per SIMD, nothing to
// hide a stall behind): two wave groups of a 512-thread workgroup walk the even / odd key tiles of the
// same 64 queries with their own LDS images and merge their (o, m, l) states through LDS at the end.
//
// On this part the fp32 matrix instructions and ordinary VALU instructions do not overlap, not even between waves of
// one SIMD (scratch/ubench/mfma_valu.hip: one MFMA wave + one VALU wave per SIMD take 0.88 x the SUM of their times),
// so the time of a key tile is its 84 matrix instructions PLUS every VALU instruction around them.  The loop is
// therefore written for instruction count: exp2 with the log2(e) factor folded into one fma, bias / all-keys-masked
// handling only in tiles that have masked keys, dropout hashed per key pair and its 1/(1-p) applied once at the end,
// V operands as ds_read_b128 from the transposed image (no per-value select), staging offsets precomputed.
// -DATTN_PROF (scratch/attn_phase_prof.sh; never in the product build): wave 0 of workgroup 0 accumulates the shader clock
// (s_memtime) it spends in each phase of a key tile -- S = K.Q^T, softmax + dropout, P.V, commit + barrier -- into
// g_attn_prof[0..4] (+ [5] = tiles), read back by butd_attention_prof_read.
#ifdef ATTN_PROF
__device__ unsigned long long g_attn_prof[8];
#define PROF_T(x) const unsigned long long x = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0)
#else
#define PROF_T(x)
#endif

#ifndef ATTN_FWD_WAVES       // (ablation hook: minimum waves per SIMD the forward kernels are compiled for)
#define ATTN_FWD_WAVES 1
#endif
template <int NS, int NT, int NG, bool BF = false>
__global__ __launch_bounds__(kAttnThreads * NG, ATTN_FWD_WAVES) void attn_fwd_kernel(
    int H, int Lq, int Lk, int D, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const uint8_t *__restrict__ mask, float *__restrict__ out,
    float *__restrict__ lse, float p_drop, uint32_t site, const uint64_t *__restrict__ rng_counter) {
  BUTD_MAIN_PRIO_SET();
  using I = Img<NS>;
  using T = TImg<NT>;
  // (head dimension 36: the third, 4-row tile of the transposed products as v_mfma_f32_4x4x1 -- see attn_bwd_longk_kernel)
  constexpr bool THIN = !BF && NT == 3 && NS == 9;
  __shared__ __attribute__((aligned(16))) float KimgG[NG][2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float VtG[NG][2][T::ROWS][T::LD];
  __shared__ __attribute__((aligned(16))) float BiasG[NG][2][64];
  const int grp = threadIdx.x / kAttnThreads;
  float(*Kimg)[64][I::LD] = KimgG[grp];
  float(*Vt)[T::ROWS][T::LD] = VtG[grp];
  float(*Bias)[64] = BiasG[grp];
  const int tid = threadIdx.x % kAttnThreads, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const TileId wg = tile_id();
  const int b = wg.b, h = wg.h;
  const long E = (long)H * D;
  const int q0 = wg.t * 64 + wave * 16;
  const bool live = q0 < Lq;  // whole wave beyond Lq: still stages tiles and hits the barriers
  const float *qb = q + (long)b * Lq * E + h * D;
  const float *kb = k + (long)b * Lk * E + h * D;
  const float *vb = v + (long)b * Lk * E + h * D;
  const uint8_t *mb = mask ? mask + (long)b * Lk : nullptr;
  const bool drop = p_drop > 0.f;
  const uint32_t thr = drop_threshold(p_drop);
  const uint32_t hkey = (drop && rng_counter) ? rng::site_key(*rng_counter, site) : rng::site_key(0ull, site);
  const int qi = q0 + fr;  // this lane's query (column of every transposed tile)
  const uint32_t LkP = (uint32_t)(Lk + 1) >> 1;
  const uint32_t pair_row = (uint32_t)(((long)b * H + h) * Lq + qi) * LkP + (uint32_t)fg * 2u;

  float qf[NS];
  load_row_frag<NS>(qf, qb, E, qi, Lq, fg, D);
  const DFrag<NS, BF> qF = make_frag<NS, BF>(qf);
  float m = kNegInf, l = 0.f;
  f32x4 o[NT];
  int vrow[NT];     // this lane's row of the transposed V image per output tile
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) vrow[nt] = (THIN && nt == 2) ? 32 + (fr & 3) : nt * 16 + fr;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  Stage<NS, NT> sg;
  sg.init(tid, D, E);
  // rows D .. ROWS-1 of the transposed images: zero, once
  for (int r = D + (tid >> 6); r < T::ROWS; r += kAttnThreads / 64) {
    Vt[0][r][tid & 63] = 0.f;
    Vt[1][r][tid & 63] = 0.f;
  }

  // key tiles of this wave group: grp, grp + NG, ... (a tile past Lk stages zeros with -inf bias and
  // contributes nothing, so both groups run the same number of iterations and barriers)
  const int iters = ((Lk + 63) / 64 + NG - 1) / NG;
  float4 kr[Stage<NS, NT>::kVec], vr[Stage<NS, NT>::kVec];
  float br = 0.f;
  {
    const int key0 = grp * 64;
    sg.fetch(kr, kb + (long)key0 * E, Lk - key0);
    sg.fetch(vr, vb + (long)key0 * E, Lk - key0);
    if (tid < 64) br = key_bias(mb, key0 + tid, Lk);
    sg.commit_frag(&Kimg[0][0][0], kr);
    sg.commit_t(&Vt[0][0][0], vr);
    if (tid < 64) Bias[0][tid] = br;
  }
  __syncthreads();
  int cur = 0;
#ifdef ATTN_PROF
  unsigned long long pf[5] = {0, 0, 0, 0, 0};
#endif
  for (int it = 0; it < iters; ++it) {
    const int key0 = (it * NG + grp) * 64;
    const bool more = it + 1 < iters;
    PROF_T(t0);
    if (more) {
      const int nk = key0 + NG * 64;
      sg.fetch(kr, kb + (long)nk * E, Lk - nk);
      sg.fetch(vr, vb + (long)nk * E, Lk - nk);
      if (tid < 64) br = key_bias(mb, nk + tid, Lk);
    }
    PROF_T(t1);
#ifdef ATTN_PROF
    unsigned long long t2 = t1, t3 = t1, t4 = t1;
#endif
    if (live) {
      // 16-key sub-tiles of this tile that hold a key at all (uniform; 4 except in the last tile: 80 text tokens = 4 + 1,
      // 132 boxes = 4 + 4 + 1): the others keep a zero score, get the -inf bias below and are skipped by both products
      const int live_t = min(4, (Lk - key0 + 15) >> 4);
      f32x4 st[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        st[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (t < live_t) {
          float kf[NS];
          I::frag(kf, Kimg[cur], t * 16 + fr, fg);
          st[t] = mma_d<NS, BF>(make_frag<NS, BF>(kf), qF, st[t]);
        }
      }
      // the first V operands travel while the softmax runs
      f32x4 va[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) va[nt] = *reinterpret_cast<const f32x4 *>(&Vt[cur][vrow[nt]][fg * 4]);
#ifdef ATTN_PROF
      asm volatile("s_nop 0" :: "v"(st[3][3]));      // (the last score accumulator has to be there)
      __builtin_amdgcn_sched_barrier(0);
      t2 = __builtin_readcyclecounter();
      __builtin_amdgcn_sched_barrier(0);
#endif

      const bool masked_tile = mb != nullptr || key0 + 64 > Lk;   // uniform: bias and dead-row handling needed
      float m_new, alpha, m2;
      if (masked_tile) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float4 bb = *reinterpret_cast<const float4 *>(&Bias[cur][t * 16 + fg * 4]);
          st[t][0] += bb.x; st[t][1] += bb.y; st[t][2] += bb.z; st[t][3] += bb.w;
        }
      }
      float tmax = fmaxf(fmaxf(st[0][0], st[0][1]), fmaxf(st[0][2], st[0][3]));
#pragma unroll
      for (int t = 1; t < 4; ++t) tmax = fmaxf(tmax, fmaxf(fmaxf(st[t][0], st[t][1]), fmaxf(st[t][2], st[t][3])));
      tmax = quad_max(tmax);
      m_new = fmaxf(m, tmax);
      m2 = m_new * kLog2e;
      alpha = __builtin_amdgcn_exp2f(__builtin_fmaf(m, kLog2e, -m2));
      if (masked_tile) {
        const bool dead = m_new == kNegInf;  // nothing but masked keys so far: exp2(-inf - 0) = 0 everywhere
        m2 = dead ? 0.f : m2;
        alpha = dead ? 1.f : alpha;
      }
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[t][i], kLog2e, -m2));
          psum += p;
          st[t][i] = p;
        }
      if (drop) {
        const uint32_t pair0 = pair_row + (uint32_t)(key0 >> 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const uint32_t h0 = pair_hash(hkey, pair0 + t * 8), h1 = pair_hash(hkey, pair0 + t * 8 + 1);
          st[t][0] = (h0 & 0xffffu) >= thr ? st[t][0] : 0.f;
          st[t][1] = (h0 >> 16) >= thr ? st[t][1] : 0.f;
          st[t][2] = (h1 & 0xffffu) >= thr ? st[t][2] : 0.f;
          st[t][3] = (h1 >> 16) >= thr ? st[t][3] : 0.f;
        }
      }
      l = l * alpha + psum;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) o[nt] *= alpha;
#ifdef ATTN_PROF
      asm volatile("s_nop 0" :: "v"(st[3][3]), "v"(o[NT - 1][3]));
      __builtin_amdgcn_sched_barrier(0);
      t3 = __builtin_readcyclecounter();
      __builtin_amdgcn_sched_barrier(0);
#endif
      // O^T[n][q] += V^T[n][key] P^T[key][q]
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 vn[NT];
        if (t < 3) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            vn[nt] = *reinterpret_cast<const f32x4 *>(&Vt[cur][vrow[nt]][(t + 1) * 16 + fg * 4]);
        }
        if constexpr (BF) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float a4[4] = {va[nt][0], va[nt][1], va[nt][2], va[nt][3]};
            o[nt] = mma_k16(a4, st[t], o[nt]);
          }
        } else if (t < live_t) {
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              if (THIN && nt == 2) o[nt] = __builtin_amdgcn_mfma_f32_4x4x1f32(va[nt][s], st[t][s], o[nt], 0, 0, 0);
              else o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[nt][s], st[t][s], o[nt], 0, 0, 0);
            }
        }
        if (t < 3) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) va[nt] = vn[nt];
        }
      }
      m = m_new;
#ifdef ATTN_PROF
      asm volatile("s_nop 0" :: "v"(o[NT - 1][3]));
      __builtin_amdgcn_sched_barrier(0);
      t4 = __builtin_readcyclecounter();
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    if (more) {
      sg.commit_frag(&Kimg[cur ^ 1][0][0], kr);
      sg.commit_t(&Vt[cur ^ 1][0][0], vr);
      if (tid < 64) Bias[cur ^ 1][tid] = br;
    }
    __syncthreads();
#ifdef ATTN_PROF
    {
      PROF_T(t5);
      pf[0] += t1 - t0; pf[1] += t2 - t1; pf[2] += t3 - t2; pf[3] += t4 - t3; pf[4] += t5 - t4;
    }
#endif
    cur ^= 1;
  }
#ifdef ATTN_PROF
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int i = 0; i < 5; ++i) g_attn_prof[i] = pf[i];
    g_attn_prof[5] = (unsigned long long)iters;
  }
#endif
  if constexpr (NG == 2) {
    constexpr int kX = 4 * NT + 2;
    float *xch = &KimgG[0][0][0][0];   // free after the loop's last barrier
    static_assert(sizeof(float) * 256 * kX <= sizeof(KimgG[0]), "exchange area");
    if (grp == 1 && live) {
      float *px = xch + (wave * 64 + lane) * kX;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) px[nt * 4 + i] = o[nt][i];
      px[4 * NT] = m;
      px[4 * NT + 1] = l;
    }
    __syncthreads();
    if (grp == 1) return;
    if (live) {
      const float *px = xch + (wave * 64 + lane) * kX;
      const float m1 = px[4 * NT], l1 = px[4 * NT + 1];
      const float m_new = fmaxf(m, m1);
      const bool dead = m_new == kNegInf;
      const float a0 = dead ? 1.f : __expf(m - m_new), a1 = dead ? 1.f : __expf(m1 - m_new);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) o[nt][i] = o[nt][i] * a0 + px[nt * 4 + i] * a1;
      l = l * a0 + l1 * a1;
      m = m_new;
    }
  }
  if (live) {
    l = quad_sum(l);
    if constexpr (THIN) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[2][i] = quad_sum(o[2][i]);
    }
    if (qi < Lq) {
      // l == 0 (every key masked) -> inf * 0 = NaN like torch's softmax; the dropout scale 1/(1-p) is applied here
      const float inv_l = (drop ? 1.f / (1.f - p_drop) : 1.f) / l;
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
}

// dQ: same walk as the forward with K and V swapping roles.  Also produces
// delta[b,h,q] = sum_n dO[q][n] * O[q][n]  (each query row belongs to exactly one wave here) for the
// dK/dV kernel that is launched next -- it used to be a launch of its own.
template <int NS, int NT, int NG, bool BF = false>
__global__ __launch_bounds__(kAttnThreads * NG) void attn_bwd_dq_kernel(
    int H, int Lq, int Lk, int D, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const uint8_t *__restrict__ mask, const float *__restrict__ out,
    const float *__restrict__ dout, const float *__restrict__ lse, float *__restrict__ delta,
    float *__restrict__ dq, long ldo, float dq_scale,
    float p_drop, uint32_t site, const uint64_t *__restrict__ rng_counter) {
  BUTD_MAIN_PRIO_SET();
  using I = Img<NS>;
  using T = TImg<NT>;
  // (head dimension 36: the third, 4-row tile of the transposed products as v_mfma_f32_4x4x1 -- see attn_bwd_longk_kernel)
  constexpr bool THIN = !BF && NT == 3 && NS == 9;
  __shared__ __attribute__((aligned(16))) float KimgG[NG][2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float VimgG[NG][2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float KtG[NG][2][T::ROWS][T::LD];   // K transposed: the dQ product's operand
  __shared__ __attribute__((aligned(16))) float BiasG[NG][2][64];
  const int grp = threadIdx.x / kAttnThreads;   // key-tile group, see attn_fwd_kernel
  float(*Kimg)[64][I::LD] = KimgG[grp];
  float(*Vimg)[64][I::LD] = VimgG[grp];
  float(*Kt)[T::ROWS][T::LD] = KtG[grp];
  float(*Bias)[64] = BiasG[grp];
  const int tid = threadIdx.x % kAttnThreads, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const TileId wg = tile_id();
  const int b = wg.b, h = wg.h;
  const long E = (long)H * D;
  const int q0 = wg.t * 64 + wave * 16;
  const bool live = q0 < Lq;
  const float *qb = q + (long)b * Lq * E + h * D;
  const float *gb = dout + (long)b * Lq * E + h * D;
  const float *kb = k + (long)b * Lk * E + h * D;
  const float *vb = v + (long)b * Lk * E + h * D;
  const uint8_t *mb = mask ? mask + (long)b * Lk : nullptr;
  const bool drop = p_drop > 0.f;
  const float inv_keep = drop ? 1.f / (1.f - p_drop) : 1.f;
  const uint32_t thr = drop_threshold(p_drop);
  const uint32_t hkey = (drop && rng_counter) ? rng::site_key(*rng_counter, site) : rng::site_key(0ull, site);
  const int qi = q0 + fr;
  const uint32_t LkP = (uint32_t)(Lk + 1) >> 1;
  const uint32_t pair_row = (uint32_t)(((long)b * H + h) * Lq + qi) * LkP + (uint32_t)fg * 2u;

  float qf[NS], gf[NS];
  load_row_frag<NS>(qf, qb, E, qi, Lq, fg, D);
  load_row_frag<NS>(gf, gb, E, qi, Lq, fg, D);
  // rows beyond Lq: lse = +inf makes every probability exp(s - inf) = 0
  const float my_lse = qi < Lq ? lse[((long)b * H + h) * Lq + qi] : INFINITY;
  const float lse2 = my_lse * kLog2e;          // p = exp2(s * log2(e) - lse2): one fma + v_exp_f32 per score
  float my_delta;
  {
    float of[NS];
    load_row_frag<NS>(of, out + (long)b * Lq * E + h * D, E, qi, Lq, fg, D);
    float part = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) part += gf[s] * of[s];
    my_delta = quad_sum(part);   // the four lanes of a query hold disjoint d-groups
    if (grp == 0 && fg == 0 && qi < Lq) delta[((long)b * H + h) * Lq + qi] = my_delta;
  }
  // dP is only ever used as dropout(dP) = keep * dP / (1 - p): the scale rides on this lane's dO fragment
#pragma unroll
  for (int s = 0; s < NS; ++s) gf[s] *= inv_keep;
  const DFrag<NS, BF> qF = make_frag<NS, BF>(qf), gF = make_frag<NS, BF>(gf);
  f32x4 acc[NT];
  int krow[NT];     // this lane's row of the transposed K image per output tile
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) krow[nt] = (THIN && nt == 2) ? 32 + (fr & 3) : nt * 16 + fr;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  Stage<NS, NT> sg;
  sg.init(tid, D, E);
  for (int r = D + (tid >> 6); r < T::ROWS; r += kAttnThreads / 64) {   // rows D.. of the transposed images: zero, once
    Kt[0][r][tid & 63] = 0.f;
    Kt[1][r][tid & 63] = 0.f;
  }

  // key tiles of this wave group: grp, grp + NG, ... (a tile past Lk stages zeros with -inf bias and
  // contributes nothing, so both groups run the same number of iterations and barriers)
  const int iters = ((Lk + 63) / 64 + NG - 1) / NG;
  float4 kr[Stage<NS, NT>::kVec], vr[Stage<NS, NT>::kVec];
  float br = 0.f;
  {
    const int key0 = grp * 64;
    sg.fetch(kr, kb + (long)key0 * E, Lk - key0);
    sg.fetch(vr, vb + (long)key0 * E, Lk - key0);
    if (tid < 64) br = key_bias(mb, key0 + tid, Lk);
    sg.commit_frag(&Kimg[0][0][0], kr);
    sg.commit_t(&Kt[0][0][0], kr);
    sg.commit_frag(&Vimg[0][0][0], vr);
    if (tid < 64) Bias[0][tid] = br;
  }
  __syncthreads();
  int cur = 0;
  for (int it = 0; it < iters; ++it) {
    const int key0 = (it * NG + grp) * 64;
    const bool more = it + 1 < iters;
    if (more) {
      const int nk = key0 + NG * 64;
      sg.fetch(kr, kb + (long)nk * E, Lk - nk);
      sg.fetch(vr, vb + (long)nk * E, Lk - nk);
      if (tid < 64) br = key_bias(mb, nk + tid, Lk);
    }
    if (live) {
      const bool masked_tile = mb != nullptr || key0 + 64 > Lk;   // uniform
      f32x4 ds[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float kf[NS], vf[NS];
        I::frag(kf, Kimg[cur], t * 16 + fr, fg);
        I::frag(vf, Vimg[cur], t * 16 + fr, fg);
        f32x4 st = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (BF) {
          st = mma_d<NS, BF>(make_frag<NS, BF>(kf), qF, st);   // S^T
          dp = mma_d<NS, BF>(make_frag<NS, BF>(vf), gF, dp);   // dP^T
        } else {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s], qf[s], st, 0, 0, 0);  // S^T
          dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[s], gf[s], dp, 0, 0, 0);  // dP^T
        }
        }
        if (masked_tile) {
          const float4 bb = *reinterpret_cast<const float4 *>(&Bias[cur][t * 16 + fg * 4]);
          st[0] += bb.x; st[1] += bb.y; st[2] += bb.z; st[3] += bb.w;
        }
        if (drop) {   // the forward's mask: one hash per key pair (pair_hash)
          const uint32_t pair0 = pair_row + (uint32_t)(key0 >> 1) + t * 8;
          const uint32_t h0 = pair_hash(hkey, pair0), h1 = pair_hash(hkey, pair0 + 1);
          dp[0] = (h0 & 0xffffu) >= thr ? dp[0] : 0.f;
          dp[1] = (h0 >> 16) >= thr ? dp[1] : 0.f;
          dp[2] = (h1 & 0xffffu) >= thr ? dp[2] : 0.f;
          dp[3] = (h1 >> 16) >= thr ? dp[3] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[i], kLog2e, -lse2));
          ds[t][i] = p * (dp[i] - my_delta);
        }
      }
      // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
      f32x4 ka[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) ka[nt] = *reinterpret_cast<const f32x4 *>(&Kt[cur][krow[nt]][fg * 4]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 kn[NT];
        if (t < 3) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            kn[nt] = *reinterpret_cast<const f32x4 *>(&Kt[cur][krow[nt]][(t + 1) * 16 + fg * 4]);
        }
        if constexpr (BF) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float a4[4] = {ka[nt][0], ka[nt][1], ka[nt][2], ka[nt][3]};
            acc[nt] = mma_k16(a4, ds[t], acc[nt]);
          }
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              if (THIN && nt == 2) acc[nt] = __builtin_amdgcn_mfma_f32_4x4x1f32(ka[nt][s], ds[t][s], acc[nt], 0, 0, 0);
              else acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[nt][s], ds[t][s], acc[nt], 0, 0, 0);
            }
        }
        if (t < 3) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) ka[nt] = kn[nt];
        }
      }
    }
    if (more) {
      sg.commit_frag(&Kimg[cur ^ 1][0][0], kr);
      sg.commit_t(&Kt[cur ^ 1][0][0], kr);
      sg.commit_frag(&Vimg[cur ^ 1][0][0], vr);
      if (tid < 64) Bias[cur ^ 1][tid] = br;
    }
    __syncthreads();
    cur ^= 1;
  }
  if constexpr (NG == 2) {   // dQ is a plain sum over the key tiles: add the second group's share
    float *xch = &KimgG[0][0][0][0];
    static_assert(sizeof(float) * 256 * 4 * NT <= sizeof(KimgG[0]), "exchange area");
    if (grp == 1 && live) {
      float *px = xch + (wave * 64 + lane) * (4 * NT);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) px[nt * 4 + i] = acc[nt][i];
    }
    __syncthreads();
    if (grp == 1) return;
    if (live) {
      const float *px = xch + (wave * 64 + lane) * (4 * NT);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[nt][i] += px[nt * 4 + i];
    }
  }
  if constexpr (THIN) {
    if (live) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[2][i] = quad_sum(acc[2][i]);
    }
  }
  if (live && qi < Lq) {
    float *ob = dq + ((long)b * Lq + qi) * ldo + h * D;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int d = nt * 16 + fg * 4 + i;
        if (d < D) ob[d] = acc[nt][i] * dq_scale;
      }
  }
}

// dK, dV: a workgroup owns 64 keys (a wave 16 of them, as columns) and walks the queries 64 at a time
// through LDS images of Q and dO; tiles are NOT transposed here:  S[q][key] has C-layout
// lane (c = key, g) -> q = 4g+i, which is the B fragment of
// dV^T[n][key] += dO^T[n][q] P[q][key]  and  dK^T[d][key] += Q^T[d][q] dS[q][key].
// NG = 2 (few key tiles: cross-attention to 80 tokens / 132 boxes is 128-192 workgroups): two wave
// groups walk the even / odd QUERY tiles and add their dK / dV shares through LDS at the end.
template <int NS, int NT, int NG, bool BF = false>
__global__ __launch_bounds__(kAttnThreads * NG) void attn_bwd_dkv_kernel(
    int H, int Lq, int Lk, int D, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const uint8_t *__restrict__ mask, const float *__restrict__ dout,
    const float *__restrict__ lse, const float *__restrict__ delta, float *__restrict__ dk,
    float *__restrict__ dv, long ldo, float p_drop, uint32_t site,
    const uint64_t *__restrict__ rng_counter) {
  BUTD_MAIN_PRIO_SET();
  using I = Img<NS>;
  // (head dimension 36: the third, 4-row tile of the transposed products as v_mfma_f32_4x4x1 -- see attn_bwd_longk_kernel)
  constexpr bool THIN = !BF && NT == 3 && NS == 9;
  __shared__ __attribute__((aligned(16))) float QimgG[NG][2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float GimgG[NG][2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float LseG[NG][2][64];   // lse * log2(e)
  __shared__ __attribute__((aligned(16))) float DelG[NG][2][64];
  const int grp = threadIdx.x / kAttnThreads;
  float(*Qimg)[64][I::LD] = QimgG[grp];
  float(*Gimg)[64][I::LD] = GimgG[grp];
  float(*Lse)[64] = LseG[grp];
  float(*Del)[64] = DelG[grp];
  const int tid = threadIdx.x % kAttnThreads, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const TileId wg = tile_id();
  const int b = wg.b, h = wg.h;
  const long E = (long)H * D;
  const int k0 = wg.t * 64 + wave * 16;
  const bool live = k0 < Lk;
  const float *qb = q + (long)b * Lq * E + h * D;
  const float *gb = dout + (long)b * Lq * E + h * D;
  const float *kb = k + (long)b * Lk * E + h * D;
  const float *vb = v + (long)b * Lk * E + h * D;
  const float *lb = lse + ((long)b * H + h) * Lq;
  const float *db = delta + ((long)b * H + h) * Lq;
  const bool drop = p_drop > 0.f;
  const float inv_keep = drop ? 1.f / (1.f - p_drop) : 1.f;
  const uint32_t thr = drop_threshold(p_drop);
  const uint32_t hkey = (drop && rng_counter) ? rng::site_key(*rng_counter, site) : rng::site_key(0ull, site);
  const uint32_t LkP = (uint32_t)(Lk + 1) >> 1;
  const int ki = k0 + fr;  // this lane's key (column)
  const float my_bias = key_bias(mask ? mask + (long)b * Lk : nullptr, ki, Lk);
  const bool wave_masked = __any(my_bias != 0.f);    // uniform: only then the bias is added at all
  // the forward's dropout mask (pair_hash): this lane's key selects the pair column and the 16-bit field
  const uint32_t pair_col = (uint32_t)(((long)b * H + h) * Lq) * LkP + (uint32_t)(ki >> 1);
  const uint32_t field_shift = (uint32_t)(ki & 1) * 16u;

  float kf[NS], vf[NS];
  load_row_frag<NS>(kf, kb, E, ki, Lk, fg, D);
  load_row_frag<NS>(vf, vb, E, ki, Lk, fg, D);
  // dP is only ever used as dropout(dP) = keep * dP / (1 - p): the scale rides on this lane's V fragment
#pragma unroll
  for (int s = 0; s < NS; ++s) vf[s] *= inv_keep;
  const DFrag<NS, BF> kF = make_frag<NS, BF>(kf), vF = make_frag<NS, BF>(vf);
  // operands of the two transposed products: element n of rows q .. q+3.  n >= D reads a slot of the row padding that
  // is zeroed once below and never written by a commit (no select per value)
  constexpr int kZeroSlot = 4 * I::NSP;
  int ncol[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = (THIN && nt == 2) ? 32 + (fr & 3) : nt * 16 + fr;
    ncol[nt] = n < D ? I::col(n) : kZeroSlot;
  }
  f32x4 ak[NT], av[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    ak[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    av[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int e = tid; e < 2 * 64; e += kAttnThreads) {
    Qimg[e >> 6][e & 63][kZeroSlot] = 0.f;
    Gimg[e >> 6][e & 63][kZeroSlot] = 0.f;
  }

  Stage<NS, NT> sg;
  sg.init(tid, D, E);
  float4 qr[Stage<NS, NT>::kVec], gr[Stage<NS, NT>::kVec];
  float sr = 0.f;  // threads 0..63: lse * log2(e) of row tid ; threads 64..127: delta of row tid-64
  auto fetch_stats = [&](int qs) {
    if (tid < 64) sr = (qs + tid < Lq) ? lb[qs + tid] * kLog2e : INFINITY;       // +inf -> probability 0
    else if (tid < 128) sr = (qs + tid - 64 < Lq) ? db[qs + tid - 64] : 0.f;
  };
  auto commit_stats = [&](int buf) {
    if (tid < 64) Lse[buf][tid] = sr;
    else if (tid < 128) Del[buf][tid - 64] = sr;
  };
  // query tiles of this wave group: grp, grp + NG, ... (a tile past Lq stages zeros with lse = +inf:
  // every probability is 0, so both groups run the same number of iterations and barriers)
  const int iters = ((Lq + 63) / 64 + NG - 1) / NG;
  {
    const int qs = grp * 64;
    sg.fetch(qr, qb + (long)qs * E, Lq - qs);
    sg.fetch(gr, gb + (long)qs * E, Lq - qs);
    fetch_stats(qs);
    sg.commit_frag(&Qimg[0][0][0], qr);
    sg.commit_frag(&Gimg[0][0][0], gr);
    commit_stats(0);
  }
  __syncthreads();
  int cur = 0;
  for (int it = 0; it < iters; ++it) {
    const int qs = (it * NG + grp) * 64;
    const bool more = it + 1 < iters;
    if (more) {
      const int nq = qs + NG * 64;
      sg.fetch(qr, qb + (long)nq * E, Lq - nq);
      sg.fetch(gr, gb + (long)nq * E, Lq - nq);
      fetch_stats(nq);
    }
    if (live) {
      const uint32_t pair_tile = pair_col + (uint32_t)(qs + fg * 4) * LkP;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float qf[NS], gf[NS];
        I::frag(qf, Qimg[cur], t * 16 + fr, fg);
        I::frag(gf, Gimg[cur], t * 16 + fr, fg);
        f32x4 st = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (BF) {
          st = mma_d<NS, BF>(make_frag<NS, BF>(qf), kF, st);   // S[q][key]
          dp = mma_d<NS, BF>(make_frag<NS, BF>(gf), vF, dp);   // dP[q][key] / (1 - p)
        } else {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          st = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[s], kf[s], st, 0, 0, 0);  // S[q][key]
          dp = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[s], vf[s], dp, 0, 0, 0);  // dP[q][key] / (1 - p)
        }
        }
        // operands of the transposed products for these 16 queries: in flight under the softmax arithmetic
        float ga[4][NT], qa[4][NT];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float *grow = Gimg[cur][t * 16 + fg * 4 + s];
          const float *qrow = Qimg[cur][t * 16 + fg * 4 + s];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            ga[s][nt] = grow[ncol[nt]];
            qa[s][nt] = qrow[ncol[nt]];
          }
        }
        const float4 l4 = *reinterpret_cast<const float4 *>(&Lse[cur][t * 16 + fg * 4]);
        const float4 d4 = *reinterpret_cast<const float4 *>(&Del[cur][t * 16 + fg * 4]);
        const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq4[4] = {d4.x, d4.y, d4.z, d4.w};
        if (wave_masked) {
          st[0] += my_bias; st[1] += my_bias; st[2] += my_bias; st[3] += my_bias;
        }
        f32x4 pd, ds;
#pragma unroll
        for (int i = 0; i < 4; ++i) pd[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[i], kLog2e, -lq[i]));
        if (drop) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const uint32_t hh = pair_hash(hkey, pair_tile + (uint32_t)(t * 16 + i) * LkP);
            const bool keep = ((hh >> field_shift) & 0xffffu) >= thr;
            ds[i] = pd[i] * ((keep ? dp[i] : 0.f) - dq4[i]);
            pd[i] = keep ? pd[i] : 0.f;                      // the 1/(1-p) of dV is applied once, at the end
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) ds[i] = pd[i] * (dp[i] - dq4[i]);
        }
        if constexpr (BF) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float g4[4] = {ga[0][nt], ga[1][nt], ga[2][nt], ga[3][nt]};
            const float q4[4] = {qa[0][nt], qa[1][nt], qa[2][nt], qa[3][nt]};
            av[nt] = mma_k16(g4, pd, av[nt]);
            ak[nt] = mma_k16(q4, ds, ak[nt]);
          }
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              if (THIN && nt == 2) {
                av[nt] = __builtin_amdgcn_mfma_f32_4x4x1f32(ga[s][nt], pd[s], av[nt], 0, 0, 0);
                ak[nt] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[s][nt], ds[s], ak[nt], 0, 0, 0);
              } else {
                av[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s][nt], pd[s], av[nt], 0, 0, 0);
                ak[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s][nt], ds[s], ak[nt], 0, 0, 0);
              }
            }
        }
      }
    }
    if (more) {
      sg.commit_frag(&Qimg[cur ^ 1][0][0], qr);
      sg.commit_frag(&Gimg[cur ^ 1][0][0], gr);
      commit_stats(cur ^ 1);
    }
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) av[nt] *= inv_keep;
  if constexpr (NG == 2) {
    float *xch = &QimgG[0][0][0][0];
    static_assert(sizeof(float) * 256 * 8 * NT <= sizeof(QimgG[0]), "exchange area");
    if (grp == 1 && live) {
      float *px = xch + (wave * 64 + lane) * (8 * NT);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          px[nt * 8 + i] = ak[nt][i];
          px[nt * 8 + 4 + i] = av[nt][i];
        }
    }
    __syncthreads();
    if (grp == 1) return;
    if (live) {
      const float *px = xch + (wave * 64 + lane) * (8 * NT);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ak[nt][i] += px[nt * 8 + i];
          av[nt][i] += px[nt * 8 + 4 + i];
        }
    }
  }
  if constexpr (THIN) {
    if (live) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ak[2][i] = quad_sum(ak[2][i]);
        av[2][i] = quad_sum(av[2][i]);
      }
    }
  }
  if (live && ki < Lk) {
    float *okp = dk + ((long)b * Lk + ki) * ldo + h * D;
    float *ovp = dv + ((long)b * Lk + ki) * ldo + h * D;
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

// ---- single-pass backward for SHORT KEY SETS (Lk <= 16 * NKT <= 144: the 80 text tokens, the 132 boxes) -----------
// The two-kernel backward above computes S and dP twice (once per kernel: 72 matrix instructions per 16 x 16 score
// tile) and is two launches of 20 us for work worth 3 us at these sizes.  Here a workgroup keeps ALL keys of its
// (b, h) in LDS (K, V as fragment images, K also transposed), a wave owns 16 queries per query tile and walks the key
// sub-tiles once: S[q][key] and dP[q][key] (non-transposed: C layout lane (c = key, g) -> q = 4g + i, the B operand
// of the dK^T / dV^T products as in attn_bwd_dkv_kernel), P and dS once; dV^T += dO^T P, dK^T += Q^T dS into
// accumulators that stay in registers for every key sub-tile across the workgroup's query tiles; dS is transposed
// through a wave-private 16 x 16 LDS patch (4 ds_write_b32 + 1 ds_read_b128 per lane, no barrier) for
// dQ^T += K^T dS^T.  54 matrix instructions per score tile.  dQ rows are written once (every key is in this
// workgroup); dK / dV are summed over the four waves through LDS and added to global memory with one atomic per
// element and workgroup -- the caller zero-fills them (they sit in the step's zero arena).
template <int NS, int NT, int NKT>
__global__ __launch_bounds__(kAttnThreads) void attn_bwd_smallk_kernel(
    int H, int Lq, int Lk, int D, int q_tiles_per_wg, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const uint8_t *__restrict__ mask, const float *__restrict__ out,
    const float *__restrict__ dout, const float *__restrict__ lse, float *__restrict__ delta,
    float *__restrict__ dq, float *__restrict__ dk, float *__restrict__ dv, long ld_dq, long ld_dkv, float dq_scale,
    float p_drop, uint32_t site, const uint64_t *__restrict__ rng_counter) {
  BUTD_MAIN_PRIO_SET();
  using I = Img<NS>;
  // (head dimension 36: the third, 4-row tile of the transposed products as v_mfma_f32_4x4x1 -- see attn_bwd_longk_kernel)
  constexpr bool THIN = NT == 3 && NS == 9;
  constexpr int KR = NKT * 16;    // staged key rows
  constexpr int LDT = KR + 4;     // row stride of the transposed K image
  constexpr int LDX = 20;         // row stride of a wave's transpose patch (conflict-free for both access patterns)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float(*Kimg)[I::LD] = reinterpret_cast<float(*)[I::LD]>(smem);
  float(*Vimg)[I::LD] = Kimg + KR;
  float *Kt = reinterpret_cast<float *>(Vimg + KR);   // [NT * 16][LDT]
  float *X = Kt + NT * 16 * LDT;                       // [4 waves][16][LDX]
  float *Del = X + 4 * 16 * LDX;                       // [4 waves][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const TileId wg = tile_id();
  const int b = wg.b, h = wg.h;
  const long E = (long)H * D;
  const float *qb = q + (long)b * Lq * E + h * D;
  const float *gb = dout + (long)b * Lq * E + h * D;
  const float *ob = out + (long)b * Lq * E + h * D;
  const float *kb = k + (long)b * Lk * E + h * D;
  const float *vb = v + (long)b * Lk * E + h * D;
  const uint8_t *mb = mask ? mask + (long)b * Lk : nullptr;
  const bool drop = p_drop > 0.f;
  const float inv_keep = drop ? 1.f / (1.f - p_drop) : 1.f;
  const uint32_t thr = drop_threshold(p_drop);
  const uint32_t hkey = (drop && rng_counter) ? rng::site_key(*rng_counter, site) : rng::site_key(0ull, site);
  const uint32_t LkP = (uint32_t)(Lk + 1) >> 1;

  // ---- stage every key once: K, V as fragment images, K transposed as well
  {
    const int vpr = D >> 2;
    for (int f = tid; f < KR * vpr; f += kAttnThreads) {
      const int row = f / vpr, c4 = f - row * vpr;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (row < Lk) {
        kv = *reinterpret_cast<const float4 *>(kb + (long)row * E + c4 * 4);
        vv = *reinterpret_cast<const float4 *>(vb + (long)row * E + c4 * 4);
      }
      const float ke[4] = {kv.x, kv.y, kv.z, kv.w}, ve[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int d = c4 * 4 + i;
        Kimg[row][I::col(d)] = ke[i];
        Vimg[row][I::col(d)] = ve[i];
        Kt[d * LDT + row] = ke[i];
      }
    }
    for (int e = tid; e < (NT * 16 - D) * KR; e += kAttnThreads) {   // rows D .. of the transposed image: zero
      const int r = D + e / KR, c = e - (e / KR) * KR;
      Kt[r * LDT + c] = 0.f;
    }
  }
  // per lane and key sub-tile: this lane's key is t * 16 + fr
  float bias_t[NKT];
  bool any_bias = false;
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    bias_t[t] = key_bias(mb, t * 16 + fr, Lk);
    any_bias = any_bias || bias_t[t] != 0.f;
  }
  const bool wave_masked = __any(any_bias);
  const uint32_t field_shift = (uint32_t)(fr & 1) * 16u;
  f32x4 ak[NKT][NT], av[NKT][NT];
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      ak[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      av[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  __syncthreads();

  float *Xw = X + wave * 16 * LDX, *Dw = Del + wave * 16;
  for (int qt = 0; qt < q_tiles_per_wg; ++qt) {
    const int q0 = (wg.t * q_tiles_per_wg + qt) * 64 + wave * 16;
    if (q0 >= Lq) continue;            // (wave-uniform; nothing below synchronises across waves)
#ifdef ATTN_SK_ABL
    if ((ATTN_SK_ABL & 4) && q0 >= 0) continue;
#endif
    const int qi = q0 + fr;
    float qf[NS], gf[NS];
    load_row_frag<NS>(qf, qb, E, qi, Lq, fg, D);
    load_row_frag<NS>(gf, gb, E, qi, Lq, fg, D);
    {
      float of[NS];
      load_row_frag<NS>(of, ob, E, qi, Lq, fg, D);
      float part = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s) part += gf[s] * of[s];
      const float my_delta = quad_sum(part);
      if (fg == 0) {
        Dw[fr] = my_delta;
        if (qi < Lq) delta[((long)b * H + h) * Lq + qi] = my_delta;
      }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) gf[s] *= inv_keep;    // dP is only ever used as keep * dP / (1 - p)
    // operands of the transposed products: element n of queries 4g .. 4g+3 (this wave's 16 queries)
    float qT[NT][4], gT[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = (THIN && nt == 2) ? 32 + (fr & 3) : nt * 16 + fr;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int qq = q0 + fg * 4 + s;
        const bool ok = n < D && qq < Lq;
        qT[nt][s] = ok ? qb[(long)qq * E + n] : 0.f;
        gT[nt][s] = ok ? gb[(long)qq * E + n] : 0.f;
      }
    }
    float lq[4], dl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int qq = q0 + fg * 4 + i;
      lq[i] = qq < Lq ? lse[((long)b * H + h) * Lq + qq] * kLog2e : INFINITY;   // +inf -> probability 0
    }
    __builtin_amdgcn_wave_barrier();
    {
      const float4 d4 = *reinterpret_cast<const float4 *>(Dw + fg * 4);
      dl[0] = d4.x; dl[1] = d4.y; dl[2] = d4.z; dl[3] = d4.w;
    }
    const uint32_t pair_q = (uint32_t)(((long)b * H + h) * Lq + q0 + fg * 4) * LkP + (uint32_t)(fr >> 1);
    f32x4 dqa[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) dqa[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      if (t * 16 < Lk) {    // (uniform)
        float kf[NS], vf[NS];
        I::frag(kf, Kimg, t * 16 + fr, fg);
        I::frag(vf, Vimg, t * 16 + fr, fg);
        f32x4 st = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          st = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[s], kf[s], st, 0, 0, 0);   // S[q][key]
          dp = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[s], vf[s], dp, 0, 0, 0);   // dP[q][key] / (1 - p)
        }
        f32x4 ka[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          ka[nt] = *reinterpret_cast<const f32x4 *>(Kt + ((THIN && nt == 2) ? 32 + (fr & 3) : nt * 16 + fr) * LDT + t * 16 + fg * 4);
        if (wave_masked) {
          st[0] += bias_t[t]; st[1] += bias_t[t]; st[2] += bias_t[t]; st[3] += bias_t[t];
        }
        f32x4 pd, ds;
#pragma unroll
        for (int i = 0; i < 4; ++i) pd[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[i], kLog2e, -lq[i]));
        if (drop) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const uint32_t hh = pair_hash(hkey, pair_q + (uint32_t)i * LkP + (uint32_t)(t * 8));
            const bool keep = ((hh >> field_shift) & 0xffffu) >= thr;
            ds[i] = pd[i] * ((keep ? dp[i] : 0.f) - dl[i]);
            pd[i] = keep ? pd[i] : 0.f;                    // the 1/(1-p) of dV is applied once, at the end
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) ds[i] = pd[i] * (dp[i] - dl[i]);
        }
        // dS -> dS^T through the wave's LDS patch: element (q = 4g + i, key = fr) written, (q = fr, keys 4g ..) read
#pragma unroll
        for (int i = 0; i < 4; ++i) Xw[(fg * 4 + i) * LDX + fr] = ds[i];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if (THIN && nt == 2) {
              av[t][nt] = __builtin_amdgcn_mfma_f32_4x4x1f32(gT[nt][s], pd[s], av[t][nt], 0, 0, 0);
              ak[t][nt] = __builtin_amdgcn_mfma_f32_4x4x1f32(qT[nt][s], ds[s], ak[t][nt], 0, 0, 0);
            } else {
              av[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gT[nt][s], pd[s], av[t][nt], 0, 0, 0);
              ak[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qT[nt][s], ds[s], ak[t][nt], 0, 0, 0);
            }
          }
        __builtin_amdgcn_wave_barrier();
        const f32x4 dst = *reinterpret_cast<const f32x4 *>(Xw + fr * LDX + fg * 4);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if (THIN && nt == 2) dqa[nt] = __builtin_amdgcn_mfma_f32_4x4x1f32(ka[nt][s], dst[s], dqa[nt], 0, 0, 0);
            else dqa[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[nt][s], dst[s], dqa[nt], 0, 0, 0);   // dQ^T[d][q]
          }
      }
    }
    if constexpr (THIN) {
#pragma unroll
      for (int i = 0; i < 4; ++i) dqa[2][i] = quad_sum(dqa[2][i]);
    }
    if (qi < Lq) {
      float *op = dq + ((long)b * Lq + qi) * ld_dq + h * D;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int d = nt * 16 + fg * 4 + i;
          if (d < D) op[d] = dqa[nt][i] * dq_scale;
        }
    }
  }

#ifdef ATTN_SK_ABL
  if ((ATTN_SK_ABL & 1) && ak[0][0][0] != 12345.678f) return;
#endif
  if constexpr (THIN) {
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ak[t][2][i] = quad_sum(ak[t][2][i]);
        av[t][2][i] = quad_sum(av[t][2][i]);
      }
  }
  // ---- dK, dV: sum of the four waves' shares through LDS (the key images are dead), one atomic per element
  __syncthreads();
  float *Rd = smem;     // [4 waves][8 * NT values][64 lanes]
  constexpr int NV = 8 * NT;
  static_assert(4 * NV * 64 <= 2 * KR * I::LD, "reduction area must fit the key images");
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    if (t * 16 < Lk) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          Rd[(wave * NV + nt * 4 + i) * 64 + lane] = ak[t][nt][i];
          Rd[(wave * NV + 4 * NT + nt * 4 + i) * 64 + lane] = av[t][nt][i] * inv_keep;
        }
      __syncthreads();
      const int key = t * 16 + fr;
      for (int e = wave; e < NV; e += 4) {
        const float sum = (Rd[(0 * NV + e) * 64 + lane] + Rd[(1 * NV + e) * 64 + lane]) +
                          (Rd[(2 * NV + e) * 64 + lane] + Rd[(3 * NV + e) * 64 + lane]);
        const bool is_v = e >= 4 * NT;
        const int ee = is_v ? e - 4 * NT : e;
        const int d = (ee >> 2) * 16 + fg * 4 + (ee & 3);
#ifdef ATTN_SK_ABL
        if ((ATTN_SK_ABL & 2) && sum != 12345.678f) continue;
#endif
        if (d < D && key < Lk) atomicAdd((is_v ? dv : dk) + ((long)b * Lk + key) * ld_dkv + h * D + d, sum);
      }
      __syncthreads();
    }
  }
}

template <int NS, int NT, int NKT>
int launch_smallk(int B, int H, int Lq, int Lk, int D, const float *q, const float *k, const float *v,
                         const uint8_t *mask, const float *out, const float *dout, const float *lse, float *delta,
                         float *dq, float *dk, float *dv, long ld_dq, long ld_dkv, float dq_scale, float p,
                         uint32_t site, const uint64_t *rng_counter, hipStream_t s) {
  using I = Img<NS>;
  constexpr int KR = NKT * 16;
  const size_t bytes = sizeof(float) * ((size_t)2 * KR * I::LD + (size_t)NT * 16 * (KR + 4) + 4 * 16 * 20 + 64);
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_bwd_smallk_kernel<NS, NT, NKT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  const int tiles = (Lq + 63) / 64;
  // query tiles per workgroup: about one workgroup per CU (each stages the keys once and keeps the dK / dV
  // accumulators of ALL keys in registers across its tiles)
  int per = (int)(((long)tiles * H * B) / 256);
  per = per < 1 ? 1 : (per > tiles ? tiles : per);
  const dim3 grid((tiles + per - 1) / per, H, B);
  hipLaunchKernelGGL((attn_bwd_smallk_kernel<NS, NT, NKT>), grid, dim3(kAttnThreads), bytes, s, H, Lq, Lk, D, per, q, k,
                     v, mask, out, dout, lse, delta, dq, dk, dv, ld_dq, ld_dkv, dq_scale, p, site, rng_counter);
  return (int)hipGetLastError();
}


// ---- single-pass backward for LONG KEY SETS (round 5) ---------------------------------------------------------------
// The two-kernel backward computes S and dP once per kernel (72 matrix instructions per 16 x 16 score tile: 30 in the dQ
// walk, 42 in the dK / dV walk) and runs the exponentials and the dropout hashes twice; matrix and vector instructions
// do not overlap on this part (profiles/r05_mfma_valu_interleave.txt), so both counts are time.  Here a workgroup of
// EIGHT waves owns a chunk of 256 keys (a wave: two 16-key sub-tiles, K / V fragments, K^T operands and the dK / dV
// accumulators in registers for the whole kernel) and walks the queries 64 at a time through LDS images of Q and dO
// exactly like attn_bwd_dkv_kernel; per score tile it also transposes dS through a wave-private 16 x 16 LDS patch
// (attn_bwd_smallk_kernel's step) and forms dQ^T += K^T dS^T over its 32 keys: 54 matrix instructions per tile, one
// softmax pass.  The eight waves' dQ shares of a query tile are summed through LDS (one 64 x D tile per wave, read
// back linearly) and stored to the chunk's slab of a workspace; a small element-wise launch adds the Lk / 256 slabs
// in chunk order and applies dq_scale: no atomics, bit-reproducible.
// delta = rowsum(dO o O) is formed while the query tile is staged (the staged dO float4s times the same float4s of O; a
// row's quartets sit in 16 consecutive lanes: four xor-shuffles), so no kernel has to run before this one.
// WAVES x SUB: waves per workgroup x 16-key sub-tiles per wave (8 x 2: the 256-key chunks above; 4 x 1: 64-key chunks for
// key sets that would leave most of the part idle otherwise).
template <int NS>
constexpr int longk_lds_floats(int D, int waves) {
  return 2 * 64 * Img<NS>::LD + 64 + 64 + waves * 16 * 20 + waves * 64 * D;
}

template <int NS, int NT, int kLongWaves, int kLongSub>
__global__ __launch_bounds__(kLongWaves * 64) void attn_bwd_longk_kernel(
    int H, int Lq, int Lk, int D, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const uint8_t *__restrict__ mask, const float *__restrict__ out,
    const float *__restrict__ dout, const float *__restrict__ lse, float *__restrict__ dq_out, long ld_dq,
    long chunk_stride, float dq_scale, float *__restrict__ dk, float *__restrict__ dv, long ldo, long kv_stride,
    int chunks, int q_tiles_per_wg, float p_drop, uint32_t site, const uint64_t *__restrict__ rng_counter) {
  BUTD_MAIN_PRIO_SET();
  using I = Img<NS>;
  // head dimension 36 = two full 16-row tiles + FOUR rows: the third tile of the dV / dK / dQ products as
  // v_mfma_f32_4x4x1 (16 independent 4 x 4 blocks; ~10 cycles where the 16 x 16 x 4 instruction takes ~34,
  // scratch/ubench/mfma_4x4.hip).  Block (g, c / 4) of lane (c, g) multiplies rows 32 + (c & 3) of the A side with the
  // lane's own B value, so the B operands -- the score-tile registers -- are used as they are, and the accumulator of a
  // lane holds rows 32..35 of its column as a partial sum over ITS contraction group g: summed over the four groups once
  // (dK, dV: at the end of the kernel; dQ: per query tile, two lane swaps per value).
  constexpr bool THIN = NT == 3 && NS == 9;
  constexpr int kLongThreads = kLongWaves * 64, kLongChunk = kLongWaves * kLongSub * 16;
  constexpr int LDX = 20;
  constexpr int kVec = 64 * 16 / kLongThreads;         // staging: 16 lanes per query row, the first D / 4 hold a quartet
  static_assert(NS <= 16, "a row's quartets fit 16 lanes");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float(*Qimg)[I::LD] = reinterpret_cast<float(*)[I::LD]>(smem);
  float(*Gimg)[I::LD] = Qimg + 64;
  float *Lse = reinterpret_cast<float *>(Gimg + 64);   // [64]  lse * log2(e)
  float *Del = Lse + 64;                               // [64]
  float *X = Del + 64;                                 // [waves][16][LDX]
  float *Red = X + kLongWaves * 16 * LDX;              // [waves][64][D]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const TileId wg = tile_id();
  const int b = wg.b, h = wg.h;
  const long E = (long)H * D;
  // grid.x = key chunks x query splits: workgroup (chunk, split) walks q_tiles_per_wg 64-query tiles; with more than one
  // split its dK / dV are a share as well and go to the split's slab (kv_stride floats apart), folded like dQ's
  const int chunk = wg.t % chunks, qsplit = wg.t / chunks;
  const int k0 = chunk * kLongChunk + wave * (kLongSub * 16);
  const bool wave_live = k0 < Lk;                       // (uniform) waves past the last key only help staging
  const int live_waves = min(kLongWaves, (Lk - chunk * kLongChunk + kLongSub * 16 - 1) / (kLongSub * 16));
  const float *qb = q + (long)b * Lq * E + h * D;
  const float *gb = dout + (long)b * Lq * E + h * D;
  const float *ob = out + (long)b * Lq * E + h * D;
  const float *kb = k + (long)b * Lk * E + h * D;
  const float *vb = v + (long)b * Lk * E + h * D;
  const float *lb = lse + ((long)b * H + h) * Lq;
  float *dqb = dq_out + (long)chunk * chunk_stride + (long)b * Lq * ld_dq + h * D;
  const bool drop = p_drop > 0.f;
  const float inv_keep = drop ? 1.f / (1.f - p_drop) : 1.f;
  const uint32_t thr = drop_threshold(p_drop);
  const uint32_t hkey = (drop && rng_counter) ? rng::site_key(*rng_counter, site) : rng::site_key(0ull, site);
  const uint32_t LkP = (uint32_t)(Lk + 1) >> 1;
  const uint32_t field_shift = (uint32_t)(fr & 1) * 16u;      // (k0 and the sub-tile offsets are even)

  // ---- this wave's keys: fragments (S, dP), transposed operands (dQ), masks; all for the whole kernel
  float kf[kLongSub][NS], vf[kLongSub][NS], my_bias[kLongSub];
  f32x4 ka[kLongSub][NT];
  uint32_t pair_col[kLongSub];
  bool any_bias = false;
#pragma unroll
  for (int j = 0; j < kLongSub; ++j) {
    const int ki = k0 + j * 16 + fr;
    load_row_frag<NS>(kf[j], kb, E, ki, Lk, fg, D);
    load_row_frag<NS>(vf[j], vb, E, ki, Lk, fg, D);
#pragma unroll
    for (int s = 0; s < NS; ++s) vf[j][s] *= inv_keep;        // dP is only ever used as keep * dP / (1 - p)
    my_bias[j] = key_bias(mask ? mask + (long)b * Lk : nullptr, ki, Lk);
    any_bias = any_bias || my_bias[j] != 0.f;
    pair_col[j] = (uint32_t)(((long)b * H + h) * Lq) * LkP + (uint32_t)(ki >> 1);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int key = k0 + j * 16 + fg * 4 + s, d = (THIN && nt == 2) ? 32 + (fr & 3) : nt * 16 + fr;
        ka[j][nt][s] = (key < Lk && d < D) ? kb[(long)key * E + d] : 0.f;
      }
  }
  const bool wave_masked = __any(any_bias);
  constexpr int kZeroSlot = 4 * I::NSP;
  int ncol[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = (THIN && nt == 2) ? 32 + (fr & 3) : nt * 16 + fr;
    ncol[nt] = n < D ? I::col(n) : kZeroSlot;
  }
  f32x4 ak[kLongSub][NT], av[kLongSub][NT];
#pragma unroll
  for (int j = 0; j < kLongSub; ++j)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      ak[j][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      av[j][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  if (tid < 64) {
    Qimg[tid][kZeroSlot] = 0.f;
    Gimg[tid][kZeroSlot] = 0.f;
  }

  // ---- staging of a 64-query tile: this thread's float4s (row, column quartet) and where they land in the images
  const int vpr = D >> 2;
  // (row f >> 4, quartet f & 15 of float4 slot f = tid + j * threads; offsets are recomputed where they are used -- a
  //  dozen registers that would otherwise live through the matrix loop)
  const bool s_ok = (tid & 15) < vpr;
  const int s_c4 = tid & 15;
  float4 qr[kVec], gr[kVec], orr[kVec];
  float sr = 0.f;
  auto fetch = [&](int qs) {
    const int nrows = Lq - qs;
#pragma unroll
    for (int j = 0; j < kVec; ++j) {
      const int r = (tid + j * kLongThreads) >> 4;
      const bool ok = s_ok && r < nrows;
      const long o = (long)qs * E + (long)r * E + s_c4 * 4;
      qr[j] = ok ? *reinterpret_cast<const float4 *>(qb + o) : make_float4(0.f, 0.f, 0.f, 0.f);
      gr[j] = ok ? *reinterpret_cast<const float4 *>(gb + o) : make_float4(0.f, 0.f, 0.f, 0.f);
      orr[j] = ok ? *reinterpret_cast<const float4 *>(ob + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < 64) sr = (qs + tid < Lq) ? lb[qs + tid] * kLog2e : INFINITY;     // +inf -> probability 0
  };
  auto commit = [&]() {
    float *qi_ = &Qimg[0][0], *gi_ = &Gimg[0][0];
#pragma unroll
    for (int j = 0; j < kVec; ++j) {
      // (lanes without a quartet fetched zeros)
      float part = gr[j].x * orr[j].x + gr[j].y * orr[j].y + gr[j].z * orr[j].z + gr[j].w * orr[j].w;
      part += __shfl_xor(part, 8, 16);
      part += __shfl_xor(part, 4, 16);
      part += __shfl_xor(part, 2, 16);
      part += __shfl_xor(part, 1, 16);
      if (s_ok) {
        const int base = ((tid + j * kLongThreads) >> 4) * I::LD;
        const int k0_ = base + I::col(s_c4 * 4), k1_ = base + I::col(s_c4 * 4 + 1), k2_ = base + I::col(s_c4 * 4 + 2),
                  k3_ = base + I::col(s_c4 * 4 + 3);
        qi_[k0_] = qr[j].x; qi_[k1_] = qr[j].y; qi_[k2_] = qr[j].z; qi_[k3_] = qr[j].w;
        gi_[k0_] = gr[j].x; gi_[k1_] = gr[j].y; gi_[k2_] = gr[j].z; gi_[k3_] = gr[j].w;
      }
      if (((tid + j * kLongThreads) & 15) == 0) Del[(tid + j * kLongThreads) >> 4] = part;
    }
    if (tid < 64) Lse[tid] = sr;
  };
  const int tile0 = qsplit * q_tiles_per_wg;
  const int iters = min(q_tiles_per_wg, (Lq + 63) / 64 - tile0);
  fetch(tile0 * 64);
  __syncthreads();                                 // the zero slots are in place
  commit();
  __syncthreads();
  float *Xw = X + wave * 16 * LDX, *Rw = Red + wave * 64 * D;
  for (int it = 0; it < iters; ++it) {
    const int qs = (tile0 + it) * 64;
    const bool more = it + 1 < iters;
    if (wave_live) {
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
      float qf[NS], gf[NS];
      I::frag(qf, Qimg, t * 16 + fr, fg);
      I::frag(gf, Gimg, t * 16 + fr, fg);
      // operands of the transposed products for these 16 queries
      float ga[4][NT], qa[4][NT];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float *grow = Gimg[t * 16 + fg * 4 + s];
        const float *qrow = Qimg[t * 16 + fg * 4 + s];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          ga[s][nt] = grow[ncol[nt]];
          qa[s][nt] = qrow[ncol[nt]];
        }
      }
      const float4 l4 = *reinterpret_cast<const float4 *>(Lse + t * 16 + fg * 4);
      const float4 d4 = *reinterpret_cast<const float4 *>(Del + t * 16 + fg * 4);
      const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
      f32x4 dqa[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) dqa[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < kLongSub; ++j) {
        f32x4 st = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          st = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[s], kf[j][s], st, 0, 0, 0);   // S[q][key]
          dp = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[s], vf[j][s], dp, 0, 0, 0);   // dP[q][key] / (1 - p)
        }
        if (wave_masked) {
          st[0] += my_bias[j]; st[1] += my_bias[j]; st[2] += my_bias[j]; st[3] += my_bias[j];
        }
        f32x4 pd, ds;
#pragma unroll
        for (int i = 0; i < 4; ++i) pd[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[i], kLog2e, -lq[i]));
        if (drop) {
          // one hash decides a key PAIR of a row: the lanes of keys 2m and 2m + 1 need the same four words.  Each hashes
          // two of the rows and takes the other two from its neighbour (quad_perm [1,0,3,2]): two 32-bit multiplies
          // (quarter rate) less per row pair than hashing all four
          const uint32_t pair_tile = pair_col[j] + (uint32_t)(qs + t * 16 + fg * 4 + (fr & 1) * 2) * LkP;
          const uint32_t ha = pair_hash(hkey, pair_tile), hb = pair_hash(hkey, pair_tile + LkP);
          const uint32_t oa = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ha, 0xB1, 0xf, 0xf, false);
          const uint32_t ob = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hb, 0xB1, 0xf, 0xf, false);
          const bool odd = (fr & 1) != 0;
          const uint32_t hh4[4] = {odd ? oa : ha, odd ? ob : hb, odd ? ha : oa, odd ? hb : ob};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bool keep = ((hh4[i] >> field_shift) & 0xffffu) >= thr;
            ds[i] = pd[i] * ((keep ? dp[i] : 0.f) - dl[i]);
            pd[i] = keep ? pd[i] : 0.f;                      // the 1/(1-p) of dV is applied once, at the end
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) ds[i] = pd[i] * (dp[i] - dl[i]);
        }
        // dS -> dS^T through the wave's LDS patch: element (q = 4g + i, key = fr) written, (q = fr, keys 4g ..) read
#pragma unroll
        for (int i = 0; i < 4; ++i) Xw[(fg * 4 + i) * LDX + fr] = ds[i];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if (THIN && nt == 2) {
              av[j][nt] = __builtin_amdgcn_mfma_f32_4x4x1f32(ga[s][nt], pd[s], av[j][nt], 0, 0, 0);
              ak[j][nt] = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[s][nt], ds[s], ak[j][nt], 0, 0, 0);
            } else {
              av[j][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s][nt], pd[s], av[j][nt], 0, 0, 0);
              ak[j][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s][nt], ds[s], ak[j][nt], 0, 0, 0);
            }
          }
        __builtin_amdgcn_wave_barrier();
        const f32x4 dst = *reinterpret_cast<const f32x4 *>(Xw + fr * LDX + fg * 4);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if (THIN && nt == 2) dqa[nt] = __builtin_amdgcn_mfma_f32_4x4x1f32(ka[j][nt][s], dst[s], dqa[nt], 0, 0, 0);
            else dqa[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[j][nt][s], dst[s], dqa[nt], 0, 0, 0);   // dQ^T[d][q]
          }
      }
      if constexpr (THIN) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dqa[2][i] = quad_sum(dqa[2][i]);
      }
      // this wave's share of dQ for the 16 queries: row = query, D floats
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        if (nt * 16 + fg * 4 < D) *reinterpret_cast<f32x4 *>(Rw + (t * 16 + fr) * D + nt * 16 + fg * 4) = dqa[nt];
    }
    }
    if (more) fetch(qs + 64);                      // in flight across the barrier and the sum below
    __syncthreads();
    // ---- dQ of the query tile: sum of the eight shares, linear in LDS
    for (int e = tid; e < 64 * vpr; e += kLongThreads) {
      f32x4 a = *reinterpret_cast<const f32x4 *>(Red + 4 * e);
#pragma unroll
      for (int w = 1; w < kLongWaves; ++w)
        if (w < live_waves) a += *reinterpret_cast<const f32x4 *>(Red + w * 64 * D + 4 * e);
      const int r = e / vpr, c4 = e - r * vpr;
      if (qs + r < Lq) *reinterpret_cast<f32x4 *>(dqb + (long)(qs + r) * ld_dq + 4 * c4) = a * dq_scale;
    }
    if (more) commit();
    __syncthreads();
  }
  if constexpr (THIN) {
#pragma unroll
    for (int j = 0; j < kLongSub; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ak[j][2][i] = quad_sum(ak[j][2][i]);
        av[j][2][i] = quad_sum(av[j][2][i]);
      }
  }
#pragma unroll
  for (int j = 0; j < kLongSub; ++j) {
    const int ki = k0 + j * 16 + fr;
    if (ki < Lk) {
      float *okp = dk + qsplit * kv_stride + ((long)b * Lk + ki) * ldo + h * D;
      float *ovp = dv + qsplit * kv_stride + ((long)b * Lk + ki) * ldo + h * D;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int n = nt * 16 + fg * 4 + i;
          if (n < D) {
            okp[n] = ak[j][nt][i];
            ovp[n] = av[j][nt][i] * inv_keep;
          }
        }
    }
  }
}

// dst = scale * (slab 0 + slab 1 + ...), in this order, for up to three tensors in one launch (dQ over the key chunks;
// dK and dV over the query splits): rows x E floats per slab, dst rows ld floats apart
struct FoldSeg {
  const float *ws;
  float *dst;
  long slab_stride, rows, ld, first;     // first: index of the segment's first float4 in the launch
  int slabs;
  float scale;
};
struct FoldArgs {
  FoldSeg seg[3];
  int nseg, E4;
  long total;
};
__global__ __launch_bounds__(256) void attn_dq_fold_kernel(FoldArgs args) {
  BUTD_MAIN_PRIO_SET();
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= args.total) return;
  int sgi = 0;
  if (args.nseg > 1 && i >= args.seg[1].first) sgi = 1;
  if (args.nseg > 2 && i >= args.seg[2].first) sgi = 2;
  const FoldSeg &sg = args.seg[sgi];
  const long j = i - sg.first;
  const long r = j / args.E4;
  const int c = (int)(j - r * args.E4);
  f32x4 a = *reinterpret_cast<const f32x4 *>(sg.ws + 4 * j);
  for (int ch = 1; ch < sg.slabs; ++ch) a += *reinterpret_cast<const f32x4 *>(sg.ws + ch * sg.slab_stride + 4 * j);
  *reinterpret_cast<f32x4 *>(sg.dst + r * sg.ld + 4 * c) = a * sg.scale;
}

// =====================================================================================================================
// bf16 operating point (BASELINE configs[3]), round 5: bf16 LDS IMAGES.
// Round 2's bf16 kernels kept the fp32 images and rounded every operand to bf16 in registers at each use: 57 + 117 vector
// instructions around the 24 matrix instructions of a forward key tile (v_cvt_pk / v_perm / v_alignbit, the head dimension
// 36 = 9 per lane group does not cut into quartets), i.e. the kernel was bound by its conversions (96 us where the fp32
// kernel takes 142 with a 16x slower matrix instruction).  The bf16 matrix instruction DOES run next to vector work
// (profiles/r05_mfma_valu_interleave.txt), so the win is in the vector count: here every staged element is rounded ONCE,
// on its way into LDS, the images are laid out in the fragments' own shape (a lane's share of a row: 8 or 16 bf16, zero
// padded: ds_read_b128; transposed images: four consecutive rows of one head-dim element: one ds_read_b64), and only
// the probabilities / dS (produced in registers) are rounded per tile (2 v_cvt_pk per 16 x 16).  All products use
// v_mfma_f32_16x16x32_bf16 (the full-rate shape of this part; round 2 used the K = 16 one, whose matrix pipe time was
// still 45 % of the kernel: gpurun_out/r05/attn_pmc_bf16.txt): the head dimension 36 is padded to 64 (two instructions
// per 16 x 16 score tile instead of three), the key / query contractions take two 16-row tiles per instruction (a
// contraction index may be permuted freely as long as both operands use the same permutation).
// Global tensors stay fp32; softmax statistics, exponentials and accumulators stay fp32.
// =====================================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NS>
struct ImgH {
  static constexpr int NSP = (NS + 7) / 8 * 8;    // slots per lane group: whole 8-element operands of the K = 32 instruction
  static constexpr int NV = NSP / 8;              // ... how many of them
  static constexpr int LD = 4 * NSP + 8;          // halfs per row (72 for head_dim 36: 144 B, rows 36 banks apart)
  static __device__ inline int col(int d) {
    const int g = d / NS;
    return g * NSP + (d - g * NS);
  }
};
template <int NT>
struct TImgH {
  static constexpr int ROWS = NT * 16;
  static constexpr int LD = 64 + 8;               // halfs per row (144 B)
};
template <int NS, int NT>
struct StageH {
  using I = ImgH<NS>;
  static constexpr int kVec = (64 * NS + kAttnThreads - 1) / kAttnThreads;
  int goff[kVec], row[kVec], koff[kVec][4], toff[kVec];
  __device__ inline void init(int tid, int D, long E) {
    const int vpr = D >> 2;
#pragma unroll
    for (int j = 0; j < kVec; ++j) {
      const int f = tid + j * kAttnThreads;
      const int r = vpr == NS ? f / NS : f / vpr, c4 = f - r * vpr;
      const bool ok = r < 64;
      row[j] = ok ? r : 64;
      goff[j] = ok ? (int)(r * E) + c4 * 4 : 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) koff[j][i] = ok ? r * I::LD + I::col(c4 * 4 + i) : 0;
      toff[j] = ok ? (c4 * 4) * TImgH<NT>::LD + r : 0;
    }
  }
  __device__ inline void fetch(float4 (&v)[kVec], const float *__restrict__ tile, int nrows) const {
    if (nrows >= 64) {
#pragma unroll
      for (int j = 0; j < kVec; ++j)
        if (row[j] < 64) v[j] = *reinterpret_cast<const float4 *>(tile + goff[j]);
    } else {
#pragma unroll
      for (int j = 0; j < kVec; ++j)
        v[j] = row[j] < nrows ? *reinterpret_cast<const float4 *>(tile + goff[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ inline void commit_frag(__bf16 *img, const float4 (&v)[kVec], float scale = 1.f) const {
#pragma unroll
    for (int j = 0; j < kVec; ++j)
      if (row[j] < 64) {
        img[koff[j][0]] = (__bf16)(v[j].x * scale); img[koff[j][1]] = (__bf16)(v[j].y * scale);
        img[koff[j][2]] = (__bf16)(v[j].z * scale); img[koff[j][3]] = (__bf16)(v[j].w * scale);
      }
  }
  __device__ inline void commit_t(__bf16 *timg, const float4 (&v)[kVec]) const {
    constexpr int LD = TImgH<NT>::LD;
#pragma unroll
    for (int j = 0; j < kVec; ++j)
      if (row[j] < 64) {
        __bf16 *p = timg + toff[j];
        p[0] = (__bf16)v[j].x; p[LD] = (__bf16)v[j].y; p[2 * LD] = (__bf16)v[j].z; p[3 * LD] = (__bf16)v[j].w;
      }
  }
  // split-bf16 ("bf16 x 3", round 6 microbenchmark): x = hi + lo with hi = bf16(x), lo = bf16(x - hi): 16 bits of
  // mantissa from two images
  __device__ inline void commit_frag_split(__bf16 *hi, __bf16 *lo, const float4 (&v)[kVec]) const {
#pragma unroll
    for (int j = 0; j < kVec; ++j)
      if (row[j] < 64) {
        const float x[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const __bf16 h = (__bf16)x[i];
          hi[koff[j][i]] = h;
          lo[koff[j][i]] = (__bf16)(x[i] - (float)h);
        }
      }
  }
  __device__ inline void commit_t_split(__bf16 *hi, __bf16 *lo, const float4 (&v)[kVec]) const {
    constexpr int LD = TImgH<NT>::LD;
#pragma unroll
    for (int j = 0; j < kVec; ++j)
      if (row[j] < 64) {
        const float x[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const __bf16 h = (__bf16)x[i];
          hi[toff[j] + i * LD] = h;
          lo[toff[j] + i * LD] = (__bf16)(x[i] - (float)h);
        }
      }
  }
};
// one lane's share of a head-dimension operand: NV x 8 bf16 (d = g * NS + s at slot s, zero beyond NS)
template <int NS>
struct HFrag {
  bf16x8 v[ImgH<NS>::NV];
};
// ... of image row `rowp` (= &img[row * LD]), lane group g: NV ds_read_b128
template <int NS>
__device__ inline HFrag<NS> frag_h(const __bf16 *rowp, int g) {
  HFrag<NS> r;
#pragma unroll
  for (int m = 0; m < ImgH<NS>::NV; ++m)
    r.v[m] = *reinterpret_cast<const bf16x8 *>(rowp + g * ImgH<NS>::NSP + 8 * m);
  return r;
}
// ... of a row held in registers (the query / key of this lane), rounded once per kernel
template <int NS>
__device__ inline HFrag<NS> make_hfrag(const float (&f)[NS]) {
  HFrag<NS> r;
#pragma unroll
  for (int m = 0; m < ImgH<NS>::NV; ++m)
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[m][i] = (8 * m + i < NS) ? (__bf16)f[8 * m + i] : (__bf16)0.f;
  return r;
}
// the low halves of the same row: x - bf16(x), rounded to bf16
template <int NS>
__device__ inline HFrag<NS> make_hfrag_lo(const float (&f)[NS]) {
  HFrag<NS> r;
#pragma unroll
  for (int m = 0; m < ImgH<NS>::NV; ++m)
#pragma unroll
    for (int i = 0; i < 8; ++i)
      r.v[m][i] = (8 * m + i < NS) ? (__bf16)(f[8 * m + i] - (float)(__bf16)f[8 * m + i]) : (__bf16)0.f;
  return r;
}
template <int NS>
__device__ inline f32x4 mma_h(const HFrag<NS> &a, const HFrag<NS> &b, f32x4 acc) {
#pragma unroll
  for (int m = 0; m < ImgH<NS>::NV; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[m], b.v[m], acc, 0, 0, 0);
  return acc;
}
// two 16-row tiles of a contraction over rows (keys / queries) as ONE K = 32 operand: this lane's rows 4g .. 4g+3 of
// tile t (elements 0..3) and of tile t + 1 (elements 4..7) -- registers (probabilities, dS) ...
__device__ inline bf16x8 pack8(const f32x4 &p, const f32x4 &q) {
  return (bf16x8){(__bf16)p[0], (__bf16)p[1], (__bf16)p[2], (__bf16)p[3], (__bf16)q[0], (__bf16)q[1], (__bf16)q[2], (__bf16)q[3]};
}
__device__ inline bf16x8 pack8_lo(const f32x4 &p, const f32x4 &q, const bf16x8 &hi) {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r[i] = (__bf16)(p[i] - (float)hi[i]);
    r[4 + i] = (__bf16)(q[i] - (float)hi[4 + i]);
  }
  return r;
}
// ... or a transposed image row (`p` -> element 4g of tile t; tile t + 1 is 16 columns on)
__device__ inline bf16x8 pair8(const __bf16 *p) {
  const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(p), hi = *reinterpret_cast<const bf16x4 *>(p + 16);
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ inline f32x4 mma32(const bf16x8 &a, const bf16x8 &b, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
template <int N>
__device__ inline void zero_halfs(__bf16 *p, int tid) {      // N halfs (a multiple of 2), by 32-bit words
  uint32_t *w = reinterpret_cast<uint32_t *>(p);
  for (int e = tid; e < N / 2; e += kAttnThreads) w[e] = 0u;
}

// SPLIT (round 6 microbenchmark, butd_attention_fwd_split_bf16): every operand as hi + lo bf16 halves, every product as
// hi.hi + lo.hi + hi.lo on the bf16 matrix cores (the lo.lo term, 2^-16 of the product, is dropped), fp32 accumulation:
// ~16 bits of mantissa per operand instead of 8 (bf16) or 24 (fp32), three matrix instructions instead of one.
template <int NS, int NT, int NG, bool SPLIT = false>
__global__ __launch_bounds__(kAttnThreads * NG, ATTN_FWD_WAVES) void attn_fwd_h_kernel(
    int H, int Lq, int Lk, int D, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const uint8_t *__restrict__ mask, float *__restrict__ out,
    float *__restrict__ lse, float p_drop, uint32_t site, const uint64_t *__restrict__ rng_counter) {
  BUTD_MAIN_PRIO_SET();
  using I = ImgH<NS>;
  using T = TImgH<NT>;
  constexpr int NL = SPLIT ? 2 : 1;              // images per operand: hi (| lo behind it)
  constexpr int kImg1 = 64 * I::LD, kTimg1 = T::ROWS * T::LD;
  constexpr int kImg = NL * kImg1, kTimg = NL * kTimg1;
  __shared__ __attribute__((aligned(16))) __bf16 KimgG[NG][2][kImg];
  __shared__ __attribute__((aligned(16))) __bf16 VtG[NG][2][kTimg];
  __shared__ __attribute__((aligned(16))) float BiasG[NG][2][64];
  const int grp = threadIdx.x / kAttnThreads;
  __bf16(*Kimg)[kImg] = KimgG[grp];
  __bf16(*Vt)[kTimg] = VtG[grp];
  float(*Bias)[64] = BiasG[grp];
  const int tid = threadIdx.x % kAttnThreads, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const TileId wg = tile_id();
  const int b = wg.b, h = wg.h;
  const long E = (long)H * D;
  const int q0 = wg.t * 64 + wave * 16;
  const bool live = q0 < Lq;
  const float *qb = q + (long)b * Lq * E + h * D;
  const float *kb = k + (long)b * Lk * E + h * D;
  const float *vb = v + (long)b * Lk * E + h * D;
  const uint8_t *mb = mask ? mask + (long)b * Lk : nullptr;
  const bool drop = p_drop > 0.f;
  const uint32_t thr = drop_threshold(p_drop);
  const uint32_t hkey = (drop && rng_counter) ? rng::site_key(*rng_counter, site) : rng::site_key(0ull, site);
  const int qi = q0 + fr;
  const uint32_t LkP = (uint32_t)(Lk + 1) >> 1;
  const uint32_t pair_row = (uint32_t)(((long)b * H + h) * Lq + qi) * LkP + (uint32_t)fg * 2u;

  float qf[NS];
  load_row_frag<NS>(qf, qb, E, qi, Lq, fg, D);
  const HFrag<NS> qF = make_hfrag<NS>(qf);
  HFrag<NS> qL;
  if constexpr (SPLIT) qL = make_hfrag_lo<NS>(qf);
  float m = kNegInf, l = 0.f;
  f32x4 o[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  StageH<NS, NT> sg;
  sg.init(tid, D, E);
  // pads of the fragment rows and rows D.. of the transposed images: zero, once (the commits never write them)
  zero_halfs<2 * kImg>(&Kimg[0][0], tid);
  zero_halfs<2 * kTimg>(&Vt[0][0], tid);
  __syncthreads();
  auto commit = [&](int buf, const float4 (&kreg)[StageH<NS, NT>::kVec], const float4 (&vreg)[StageH<NS, NT>::kVec]) {
    if constexpr (SPLIT) {
      sg.commit_frag_split(Kimg[buf], Kimg[buf] + kImg1, kreg);
      sg.commit_t_split(Vt[buf], Vt[buf] + kTimg1, vreg);
    } else {
      sg.commit_frag(Kimg[buf], kreg);
      sg.commit_t(Vt[buf], vreg);
    }
  };

  const int iters = ((Lk + 63) / 64 + NG - 1) / NG;
  float4 kr[StageH<NS, NT>::kVec], vr[StageH<NS, NT>::kVec];
  float br = 0.f;
  {
    const int key0 = grp * 64;
    sg.fetch(kr, kb + (long)key0 * E, Lk - key0);
    sg.fetch(vr, vb + (long)key0 * E, Lk - key0);
    if (tid < 64) br = key_bias(mb, key0 + tid, Lk);
    commit(0, kr, vr);
    if (tid < 64) Bias[0][tid] = br;
  }
  __syncthreads();
  int cur = 0;
  for (int it = 0; it < iters; ++it) {
    const int key0 = (it * NG + grp) * 64;
    const bool more = it + 1 < iters;
    if (more) {
      const int nk = key0 + NG * 64;
      sg.fetch(kr, kb + (long)nk * E, Lk - nk);
      sg.fetch(vr, vb + (long)nk * E, Lk - nk);
      if (tid < 64) br = key_bias(mb, nk + tid, Lk);
    }
    if (live) {
      f32x4 st[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const HFrag<NS> kh = frag_h<NS>(&Kimg[cur][(t * 16 + fr) * I::LD], fg);
        st[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (SPLIT) {      // the small terms first
          st[t] = mma_h<NS>(frag_h<NS>(&Kimg[cur][kImg1 + (t * 16 + fr) * I::LD], fg), qF, st[t]);
          st[t] = mma_h<NS>(kh, qL, st[t]);
        }
        st[t] = mma_h<NS>(kh, qF, st[t]);
      }
      // the first V operands (keys of tiles 0 and 1) travel while the softmax runs
      bf16x8 va[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) va[nt] = pair8(&Vt[cur][(nt * 16 + fr) * T::LD + fg * 4]);

      const bool masked_tile = mb != nullptr || key0 + 64 > Lk;
      float m_new, alpha, m2;
      if (masked_tile) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float4 bb = *reinterpret_cast<const float4 *>(&Bias[cur][t * 16 + fg * 4]);
          st[t][0] += bb.x; st[t][1] += bb.y; st[t][2] += bb.z; st[t][3] += bb.w;
        }
      }
      float tmax = fmaxf(fmaxf(st[0][0], st[0][1]), fmaxf(st[0][2], st[0][3]));
#pragma unroll
      for (int t = 1; t < 4; ++t) tmax = fmaxf(tmax, fmaxf(fmaxf(st[t][0], st[t][1]), fmaxf(st[t][2], st[t][3])));
      tmax = quad_max(tmax);
      m_new = fmaxf(m, tmax);
      m2 = m_new * kLog2e;
      alpha = __builtin_amdgcn_exp2f(__builtin_fmaf(m, kLog2e, -m2));
      if (masked_tile) {
        const bool dead = m_new == kNegInf;
        m2 = dead ? 0.f : m2;
        alpha = dead ? 1.f : alpha;
      }
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[t][i], kLog2e, -m2));
          psum += p;
          st[t][i] = p;
        }
      if (drop) {
        const uint32_t pair0 = pair_row + (uint32_t)(key0 >> 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const uint32_t h0 = pair_hash(hkey, pair0 + t * 8), h1 = pair_hash(hkey, pair0 + t * 8 + 1);
          st[t][0] = (h0 & 0xffffu) >= thr ? st[t][0] : 0.f;
          st[t][1] = (h0 >> 16) >= thr ? st[t][1] : 0.f;
          st[t][2] = (h1 & 0xffffu) >= thr ? st[t][2] : 0.f;
          st[t][3] = (h1 >> 16) >= thr ? st[t][3] : 0.f;
        }
      }
      l = l * alpha + psum;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) o[nt] *= alpha;
      {   // O^T[n][q] += V^T[n][key] P^T[key][q]: two instructions per head-dim tile and 64 keys
        bf16x8 vb[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) vb[nt] = pair8(&Vt[cur][(nt * 16 + fr) * T::LD + 32 + fg * 4]);
        const bf16x8 p01 = pack8(st[0], st[1]), p23 = pack8(st[2], st[3]);
        if constexpr (SPLIT) {
          const bf16x8 q01 = pack8_lo(st[0], st[1], p01), q23 = pack8_lo(st[2], st[3], p23);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            o[nt] = mma32(pair8(&Vt[cur][kTimg1 + (nt * 16 + fr) * T::LD + fg * 4]), p01, o[nt]);
            o[nt] = mma32(pair8(&Vt[cur][kTimg1 + (nt * 16 + fr) * T::LD + 32 + fg * 4]), p23, o[nt]);
            o[nt] = mma32(va[nt], q01, o[nt]);
            o[nt] = mma32(vb[nt], q23, o[nt]);
          }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) o[nt] = mma32(va[nt], p01, o[nt]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) o[nt] = mma32(vb[nt], p23, o[nt]);
      }
      m = m_new;
    }
    if (more) {
      commit(cur ^ 1, kr, vr);
      if (tid < 64) Bias[cur ^ 1][tid] = br;
    }
    __syncthreads();
    cur ^= 1;
  }
  if constexpr (NG == 2) {
    constexpr int kX = 4 * NT + 2;
    __shared__ float xch[256 * kX];
    if (grp == 1 && live) {
      float *px = xch + (wave * 64 + lane) * kX;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) px[nt * 4 + i] = o[nt][i];
      px[4 * NT] = m;
      px[4 * NT + 1] = l;
    }
    __syncthreads();
    if (grp == 1) return;
    if (live) {
      const float *px = xch + (wave * 64 + lane) * kX;
      const float m1 = px[4 * NT], l1 = px[4 * NT + 1];
      const float m_new = fmaxf(m, m1);
      const bool dead = m_new == kNegInf;
      const float a0 = dead ? 1.f : __expf(m - m_new), a1 = dead ? 1.f : __expf(m1 - m_new);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) o[nt][i] = o[nt][i] * a0 + px[nt * 4 + i] * a1;
      l = l * a0 + l1 * a1;
      m = m_new;
    }
  }
  if (live) {
    l = quad_sum(l);
    if (qi < Lq) {
      const float inv_l = (drop ? 1.f / (1.f - p_drop) : 1.f) / l;
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
}

template <int NS, int NT, int NG>
__global__ __launch_bounds__(kAttnThreads * NG) void attn_bwd_dq_h_kernel(
    int H, int Lq, int Lk, int D, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const uint8_t *__restrict__ mask, const float *__restrict__ out,
    const float *__restrict__ dout, const float *__restrict__ lse, float *__restrict__ delta,
    float *__restrict__ dq, long ldo, float dq_scale,
    float p_drop, uint32_t site, const uint64_t *__restrict__ rng_counter) {
  BUTD_MAIN_PRIO_SET();
  using I = ImgH<NS>;
  using T = TImgH<NT>;
  constexpr int kImg = 64 * I::LD, kTimg = T::ROWS * T::LD;
  __shared__ __attribute__((aligned(16))) __bf16 KimgG[NG][2][kImg];
  __shared__ __attribute__((aligned(16))) __bf16 VimgG[NG][2][kImg];
  __shared__ __attribute__((aligned(16))) __bf16 KtG[NG][2][kTimg];
  __shared__ __attribute__((aligned(16))) float BiasG[NG][2][64];
  const int grp = threadIdx.x / kAttnThreads;
  __bf16(*Kimg)[kImg] = KimgG[grp];
  __bf16(*Vimg)[kImg] = VimgG[grp];
  __bf16(*Kt)[kTimg] = KtG[grp];
  float(*Bias)[64] = BiasG[grp];
  const int tid = threadIdx.x % kAttnThreads, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const TileId wg = tile_id();
  const int b = wg.b, h = wg.h;
  const long E = (long)H * D;
  const int q0 = wg.t * 64 + wave * 16;
  const bool live = q0 < Lq;
  const float *qb = q + (long)b * Lq * E + h * D;
  const float *gb = dout + (long)b * Lq * E + h * D;
  const float *kb = k + (long)b * Lk * E + h * D;
  const float *vb = v + (long)b * Lk * E + h * D;
  const uint8_t *mb = mask ? mask + (long)b * Lk : nullptr;
  const bool drop = p_drop > 0.f;
  const float inv_keep = drop ? 1.f / (1.f - p_drop) : 1.f;
  const uint32_t thr = drop_threshold(p_drop);
  const uint32_t hkey = (drop && rng_counter) ? rng::site_key(*rng_counter, site) : rng::site_key(0ull, site);
  const int qi = q0 + fr;
  const uint32_t LkP = (uint32_t)(Lk + 1) >> 1;
  const uint32_t pair_row = (uint32_t)(((long)b * H + h) * Lq + qi) * LkP + (uint32_t)fg * 2u;

  float qf[NS], gf[NS];
  load_row_frag<NS>(qf, qb, E, qi, Lq, fg, D);
  load_row_frag<NS>(gf, gb, E, qi, Lq, fg, D);
  const float my_lse = qi < Lq ? lse[((long)b * H + h) * Lq + qi] : INFINITY;
  const float lse2 = my_lse * kLog2e;
  float my_delta;
  {
    float of[NS];
    load_row_frag<NS>(of, out + (long)b * Lq * E + h * D, E, qi, Lq, fg, D);
    float part = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) part += gf[s] * of[s];
    my_delta = quad_sum(part);
    if (grp == 0 && fg == 0 && qi < Lq) delta[((long)b * H + h) * Lq + qi] = my_delta;
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) gf[s] *= inv_keep;
  const HFrag<NS> qF = make_hfrag<NS>(qf), gF = make_hfrag<NS>(gf);
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  StageH<NS, NT> sg;
  sg.init(tid, D, E);
  zero_halfs<2 * kImg>(&Kimg[0][0], tid);
  zero_halfs<2 * kImg>(&Vimg[0][0], tid);
  zero_halfs<2 * kTimg>(&Kt[0][0], tid);
  __syncthreads();

  const int iters = ((Lk + 63) / 64 + NG - 1) / NG;
  float4 kr[StageH<NS, NT>::kVec], vr[StageH<NS, NT>::kVec];
  float br = 0.f;
  {
    const int key0 = grp * 64;
    sg.fetch(kr, kb + (long)key0 * E, Lk - key0);
    sg.fetch(vr, vb + (long)key0 * E, Lk - key0);
    if (tid < 64) br = key_bias(mb, key0 + tid, Lk);
    sg.commit_frag(Kimg[0], kr);
    sg.commit_t(Kt[0], kr);
    sg.commit_frag(Vimg[0], vr);
    if (tid < 64) Bias[0][tid] = br;
  }
  __syncthreads();
  int cur = 0;
  for (int it = 0; it < iters; ++it) {
    const int key0 = (it * NG + grp) * 64;
    const bool more = it + 1 < iters;
    if (more) {
      const int nk = key0 + NG * 64;
      sg.fetch(kr, kb + (long)nk * E, Lk - nk);
      sg.fetch(vr, vb + (long)nk * E, Lk - nk);
      if (tid < 64) br = key_bias(mb, nk + tid, Lk);
    }
    if (live) {
      const bool masked_tile = mb != nullptr || key0 + 64 > Lk;
      f32x4 ds[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 st = mma_h<NS>(frag_h<NS>(&Kimg[cur][(t * 16 + fr) * I::LD], fg), qF, zero);   // S^T
        f32x4 dp = mma_h<NS>(frag_h<NS>(&Vimg[cur][(t * 16 + fr) * I::LD], fg), gF, zero);   // dP^T
        if (masked_tile) {
          const float4 bb = *reinterpret_cast<const float4 *>(&Bias[cur][t * 16 + fg * 4]);
          st[0] += bb.x; st[1] += bb.y; st[2] += bb.z; st[3] += bb.w;
        }
        if (drop) {
          const uint32_t pair0 = pair_row + (uint32_t)(key0 >> 1) + t * 8;
          const uint32_t h0 = pair_hash(hkey, pair0), h1 = pair_hash(hkey, pair0 + 1);
          dp[0] = (h0 & 0xffffu) >= thr ? dp[0] : 0.f;
          dp[1] = (h0 >> 16) >= thr ? dp[1] : 0.f;
          dp[2] = (h1 & 0xffffu) >= thr ? dp[2] : 0.f;
          dp[3] = (h1 >> 16) >= thr ? dp[3] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[i], kLog2e, -lse2));
          ds[t][i] = p * (dp[i] - my_delta);
        }
      }
      // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf16x8 db = pack8(ds[2 * u], ds[2 * u + 1]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[nt] = mma32(pair8(&Kt[cur][(nt * 16 + fr) * T::LD + u * 32 + fg * 4]), db, acc[nt]);
      }
    }
    if (more) {
      sg.commit_frag(Kimg[cur ^ 1], kr);
      sg.commit_t(Kt[cur ^ 1], kr);
      sg.commit_frag(Vimg[cur ^ 1], vr);
      if (tid < 64) Bias[cur ^ 1][tid] = br;
    }
    __syncthreads();
    cur ^= 1;
  }
  if constexpr (NG == 2) {
    __shared__ float xch[256 * 4 * NT];
    if (grp == 1 && live) {
      float *px = xch + (wave * 64 + lane) * (4 * NT);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) px[nt * 4 + i] = acc[nt][i];
    }
    __syncthreads();
    if (grp == 1) return;
    if (live) {
      const float *px = xch + (wave * 64 + lane) * (4 * NT);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[nt][i] += px[nt * 4 + i];
    }
  }
  if (live && qi < Lq) {
    float *ob = dq + ((long)b * Lq + qi) * ldo + h * D;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int d = nt * 16 + fg * 4 + i;
        if (d < D) ob[d] = acc[nt][i] * dq_scale;
      }
  }
}

template <int NS, int NT, int NG>
__global__ __launch_bounds__(kAttnThreads * NG) void attn_bwd_dkv_h_kernel(
    int H, int Lq, int Lk, int D, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const uint8_t *__restrict__ mask, const float *__restrict__ dout,
    const float *__restrict__ lse, const float *__restrict__ delta, float *__restrict__ dk,
    float *__restrict__ dv, long ldo, float p_drop, uint32_t site,
    const uint64_t *__restrict__ rng_counter) {
  BUTD_MAIN_PRIO_SET();
  using I = ImgH<NS>;
  using T = TImgH<NT>;
  constexpr int kImg = 64 * I::LD, kTimg = T::ROWS * T::LD;
  __shared__ __attribute__((aligned(16))) __bf16 QimgG[NG][2][kImg];
  __shared__ __attribute__((aligned(16))) __bf16 GimgG[NG][2][kImg];
  __shared__ __attribute__((aligned(16))) __bf16 QtG[NG][2][kTimg];     // Q, dO transposed: the dK^T / dV^T products' operands
  __shared__ __attribute__((aligned(16))) __bf16 GtG[NG][2][kTimg];
  __shared__ __attribute__((aligned(16))) float LseG[NG][2][64];
  __shared__ __attribute__((aligned(16))) float DelG[NG][2][64];
  const int grp = threadIdx.x / kAttnThreads;
  __bf16(*Qimg)[kImg] = QimgG[grp];
  __bf16(*Gimg)[kImg] = GimgG[grp];
  __bf16(*Qt)[kTimg] = QtG[grp];
  __bf16(*Gt)[kTimg] = GtG[grp];
  float(*Lse)[64] = LseG[grp];
  float(*Del)[64] = DelG[grp];
  const int tid = threadIdx.x % kAttnThreads, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const TileId wg = tile_id();
  const int b = wg.b, h = wg.h;
  const long E = (long)H * D;
  const int k0 = wg.t * 64 + wave * 16;
  const bool live = k0 < Lk;
  const float *qb = q + (long)b * Lq * E + h * D;
  const float *gb = dout + (long)b * Lq * E + h * D;
  const float *kb = k + (long)b * Lk * E + h * D;
  const float *vb = v + (long)b * Lk * E + h * D;
  const float *lb = lse + ((long)b * H + h) * Lq;
  const float *db = delta + ((long)b * H + h) * Lq;
  const bool drop = p_drop > 0.f;
  const float inv_keep = drop ? 1.f / (1.f - p_drop) : 1.f;
  const uint32_t thr = drop_threshold(p_drop);
  const uint32_t hkey = (drop && rng_counter) ? rng::site_key(*rng_counter, site) : rng::site_key(0ull, site);
  const uint32_t LkP = (uint32_t)(Lk + 1) >> 1;
  const int ki = k0 + fr;
  const float my_bias = key_bias(mask ? mask + (long)b * Lk : nullptr, ki, Lk);
  const bool wave_masked = __any(my_bias != 0.f);
  const uint32_t pair_col = (uint32_t)(((long)b * H + h) * Lq) * LkP + (uint32_t)(ki >> 1);
  const uint32_t field_shift = (uint32_t)(ki & 1) * 16u;

  float kf[NS], vf[NS];
  load_row_frag<NS>(kf, kb, E, ki, Lk, fg, D);
  load_row_frag<NS>(vf, vb, E, ki, Lk, fg, D);
#pragma unroll
  for (int s = 0; s < NS; ++s) vf[s] *= inv_keep;
  const HFrag<NS> kF = make_hfrag<NS>(kf), vF = make_hfrag<NS>(vf);
  f32x4 ak[NT], av[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    ak[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    av[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  StageH<NS, NT> sg;
  sg.init(tid, D, E);
  zero_halfs<2 * kImg>(&Qimg[0][0], tid);
  zero_halfs<2 * kImg>(&Gimg[0][0], tid);
  zero_halfs<2 * kTimg>(&Qt[0][0], tid);
  zero_halfs<2 * kTimg>(&Gt[0][0], tid);
  __syncthreads();
  float4 qr[StageH<NS, NT>::kVec], gr[StageH<NS, NT>::kVec];
  float sr = 0.f;
  auto fetch_stats = [&](int qs) {
    if (tid < 64) sr = (qs + tid < Lq) ? lb[qs + tid] * kLog2e : INFINITY;
    else if (tid < 128) sr = (qs + tid - 64 < Lq) ? db[qs + tid - 64] : 0.f;
  };
  auto commit_stats = [&](int buf) {
    if (tid < 64) Lse[buf][tid] = sr;
    else if (tid < 128) Del[buf][tid - 64] = sr;
  };
  const int iters = ((Lq + 63) / 64 + NG - 1) / NG;
  {
    const int qs = grp * 64;
    sg.fetch(qr, qb + (long)qs * E, Lq - qs);
    sg.fetch(gr, gb + (long)qs * E, Lq - qs);
    fetch_stats(qs);
    sg.commit_frag(Qimg[0], qr);
    sg.commit_t(Qt[0], qr);
    sg.commit_frag(Gimg[0], gr);
    sg.commit_t(Gt[0], gr);
    commit_stats(0);
  }
  __syncthreads();
  int cur = 0;
  for (int it = 0; it < iters; ++it) {
    const int qs = (it * NG + grp) * 64;
    const bool more = it + 1 < iters;
    if (more) {
      const int nq = qs + NG * 64;
      sg.fetch(qr, qb + (long)nq * E, Lq - nq);
      sg.fetch(gr, gb + (long)nq * E, Lq - nq);
      fetch_stats(nq);
    }
    if (live) {
      const uint32_t pair_tile = pair_col + (uint32_t)(qs + fg * 4) * LkP;
#pragma unroll
      for (int u = 0; u < 2; ++u) {       // two 16-query tiles per K = 32 instruction of the transposed products
        f32x4 pd2[2], ds2[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const int t = 2 * u + w;
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
          f32x4 st = mma_h<NS>(frag_h<NS>(&Qimg[cur][(t * 16 + fr) * I::LD], fg), kF, zero);   // S[q][key]
          f32x4 dp = mma_h<NS>(frag_h<NS>(&Gimg[cur][(t * 16 + fr) * I::LD], fg), vF, zero);   // dP[q][key] / (1 - p)
          const float4 l4 = *reinterpret_cast<const float4 *>(&Lse[cur][t * 16 + fg * 4]);
          const float4 d4 = *reinterpret_cast<const float4 *>(&Del[cur][t * 16 + fg * 4]);
          const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq4[4] = {d4.x, d4.y, d4.z, d4.w};
          if (wave_masked) {
            st[0] += my_bias; st[1] += my_bias; st[2] += my_bias; st[3] += my_bias;
          }
          f32x4 pd, ds;
#pragma unroll
          for (int i = 0; i < 4; ++i) pd[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[i], kLog2e, -lq[i]));
          if (drop) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const uint32_t hh = pair_hash(hkey, pair_tile + (uint32_t)(t * 16 + i) * LkP);
              const bool keep = ((hh >> field_shift) & 0xffffu) >= thr;
              ds[i] = pd[i] * ((keep ? dp[i] : 0.f) - dq4[i]);
              pd[i] = keep ? pd[i] : 0.f;
            }
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) ds[i] = pd[i] * (dp[i] - dq4[i]);
          }
          pd2[w] = pd;
          ds2[w] = ds;
        }
        const bf16x8 pb = pack8(pd2[0], pd2[1]), sb = pack8(ds2[0], ds2[1]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {   // rows n of dO^T / Q^T at the queries of tiles 2u, 2u + 1
          av[nt] = mma32(pair8(&Gt[cur][(nt * 16 + fr) * T::LD + u * 32 + fg * 4]), pb, av[nt]);
          ak[nt] = mma32(pair8(&Qt[cur][(nt * 16 + fr) * T::LD + u * 32 + fg * 4]), sb, ak[nt]);
        }
      }
    }
    if (more) {
      sg.commit_frag(Qimg[cur ^ 1], qr);
      sg.commit_t(Qt[cur ^ 1], qr);
      sg.commit_frag(Gimg[cur ^ 1], gr);
      sg.commit_t(Gt[cur ^ 1], gr);
      commit_stats(cur ^ 1);
    }
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) av[nt] *= inv_keep;
  if constexpr (NG == 2) {
    __shared__ float xch[256 * 8 * NT];
    if (grp == 1 && live) {
      float *px = xch + (wave * 64 + lane) * (8 * NT);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          px[nt * 8 + i] = ak[nt][i];
          px[nt * 8 + 4 + i] = av[nt][i];
        }
    }
    __syncthreads();
    if (grp == 1) return;
    if (live) {
      const float *px = xch + (wave * 64 + lane) * (8 * NT);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ak[nt][i] += px[nt * 8 + i];
          av[nt][i] += px[nt * 8 + 4 + i];
        }
    }
  }
  if (live && ki < Lk) {
    float *okp = dk + ((long)b * Lk + ki) * ldo + h * D;
    float *ovp = dv + ((long)b * Lk + ki) * ldo + h * D;
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

// ---- the one-pass backward on the bf16 matrix cores (bf16 images of section "bf16 operating point") -------------------
// attn_bwd_longk_kernel's walk -- 8 waves own 256 keys (32 per wave: two 16-key sub-tiles), the queries pass 64 at a time
// through LDS images, dQ shares are summed through LDS and stored to the chunk's slab -- with the products of the bf16
// kernels above: S / dP as two K = 32 instructions per 16 x 16 tile over the padded head dimension, dV^T / dK^T with TWO
// query sub-tiles per instruction (transposed images of dO and Q), and dQ^T with the wave's TWO key sub-tiles per
// instruction (K^T operands in registers, dS transposed through two wave-private LDS patches): 8.5 matrix instructions per
// score tile where the dQ walk + the dK/dV walk issue 12.5, and ONE pass of the exponentials, dropout hashes and
// roundings those kernels are bound by.  fp32: statistics, exponentials, accumulators, everything in memory.
template <int NS, int NT, int kWaves, int kSub>
__global__ __launch_bounds__(kWaves * 64) void attn_bwd_longk_h_kernel(
    int H, int Lq, int Lk, int D, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const uint8_t *__restrict__ mask, const float *__restrict__ out,
    const float *__restrict__ dout, const float *__restrict__ lse, float *__restrict__ dq_out, long ld_dq,
    long chunk_stride, float dq_scale, float *__restrict__ dk, float *__restrict__ dv, long ldo, long kv_stride,
    int chunks, int q_tiles_per_wg, float p_drop, uint32_t site, const uint64_t *__restrict__ rng_counter) {
  BUTD_MAIN_PRIO_SET();
  using I = ImgH<NS>;
  using T = TImgH<NT>;
  static_assert(kSub == 1 || kSub == 2, "one K = 32 contraction per wave (a single sub-tile leaves its upper half zero)");
  constexpr int kThreads = kWaves * 64, kChunk = kWaves * kSub * 16;
  constexpr int kImg = 64 * I::LD, kTimg = T::ROWS * T::LD, LDX = 20;
  constexpr int kVec = 64 * 16 / kThreads;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Lse = smem;                                   // [64]  lse * log2(e)
  float *Del = Lse + 64;                               // [64]
  float *X = Del + 64;                                 // [waves][sub-tiles][16][LDX]
  float *Red = X + kWaves * kSub * 16 * LDX;           // [waves][64][D]
  __bf16 *Qimg = reinterpret_cast<__bf16 *>(Red + kWaves * 64 * D);
  __bf16 *Gimg = Qimg + kImg;
  __bf16 *Qt = Gimg + kImg;
  __bf16 *Gt = Qt + kTimg;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const TileId wg = tile_id();
  const int b = wg.b, h = wg.h, chunk = wg.t % chunks, qsplit = wg.t / chunks;     // (see attn_bwd_longk_kernel)
  const long E = (long)H * D;
  const int k0 = chunk * kChunk + wave * (kSub * 16);
  const bool wave_live = k0 < Lk;
  const int live_waves = min(kWaves, (Lk - chunk * kChunk + kSub * 16 - 1) / (kSub * 16));
  const float *qb = q + (long)b * Lq * E + h * D;
  const float *gb = dout + (long)b * Lq * E + h * D;
  const float *ob = out + (long)b * Lq * E + h * D;
  const float *kb = k + (long)b * Lk * E + h * D;
  const float *vb = v + (long)b * Lk * E + h * D;
  const float *lb = lse + ((long)b * H + h) * Lq;
  float *dqb = dq_out + (long)chunk * chunk_stride + (long)b * Lq * ld_dq + h * D;
  const bool drop = p_drop > 0.f;
  const float inv_keep = drop ? 1.f / (1.f - p_drop) : 1.f;
  const uint32_t thr = drop_threshold(p_drop);
  const uint32_t hkey = (drop && rng_counter) ? rng::site_key(*rng_counter, site) : rng::site_key(0ull, site);
  const uint32_t LkP = (uint32_t)(Lk + 1) >> 1;
  const uint32_t field_shift = (uint32_t)(fr & 1) * 16u;

  HFrag<NS> kF[kSub], vF[kSub];
  float my_bias[kSub];
  uint32_t pair_col[kSub];
  bool any_bias = false;
#pragma unroll
  for (int j = 0; j < kSub; ++j) {
    const int ki = k0 + j * 16 + fr;
    float kf[NS], vf[NS];
    load_row_frag<NS>(kf, kb, E, ki, Lk, fg, D);
    load_row_frag<NS>(vf, vb, E, ki, Lk, fg, D);
#pragma unroll
    for (int s = 0; s < NS; ++s) vf[s] *= inv_keep;
    kF[j] = make_hfrag<NS>(kf);
    vF[j] = make_hfrag<NS>(vf);
    my_bias[j] = key_bias(mask ? mask + (long)b * Lk : nullptr, ki, Lk);
    any_bias = any_bias || my_bias[j] != 0.f;
    pair_col[j] = (uint32_t)(((long)b * H + h) * Lq) * LkP + (uint32_t)(ki >> 1);
  }
  const bool wave_masked = __any(any_bias);
  // dQ^T += K^T dS^T over the wave's 32 keys in ONE instruction: elements 0..3 = keys 4g .. 4g+3 of sub-tile 0, 4..7 of sub-tile 1
  bf16x8 kaH[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int key = k0 + (e >> 2) * 16 + fg * 4 + (e & 3), d = nt * 16 + fr;
      kaH[nt][e] = (__bf16)(((e >> 2) < kSub && key < Lk && d < D) ? kb[(long)key * E + d] : 0.f);
    }
  f32x4 ak[kSub][NT], av[kSub][NT];
#pragma unroll
  for (int j = 0; j < kSub; ++j)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      ak[j][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      av[j][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  {   // pads of the fragment rows and rows D.. of the transposed images: zero, once (the commits never write them)
    uint32_t *w = reinterpret_cast<uint32_t *>(Qimg);
    for (int e = tid; e < (2 * kImg + 2 * kTimg) / 2; e += kThreads) w[e] = 0u;
  }
  const int vpr = D >> 2;
  const bool s_ok = (tid & 15) < vpr;
  const int s_c4 = tid & 15;
  float4 qr[kVec], gr[kVec], orr[kVec];
  float sr = 0.f;
  auto fetch = [&](int qs) {
    const int nrows = Lq - qs;
#pragma unroll
    for (int j = 0; j < kVec; ++j) {
      const int r = (tid + j * kThreads) >> 4;
      const bool ok = s_ok && r < nrows;
      const long o = (long)qs * E + (long)r * E + s_c4 * 4;
      qr[j] = ok ? *reinterpret_cast<const float4 *>(qb + o) : make_float4(0.f, 0.f, 0.f, 0.f);
      gr[j] = ok ? *reinterpret_cast<const float4 *>(gb + o) : make_float4(0.f, 0.f, 0.f, 0.f);
      orr[j] = ok ? *reinterpret_cast<const float4 *>(ob + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < 64) sr = (qs + tid < Lq) ? lb[qs + tid] * kLog2e : INFINITY;
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < kVec; ++j) {
      float part = gr[j].x * orr[j].x + gr[j].y * orr[j].y + gr[j].z * orr[j].z + gr[j].w * orr[j].w;
      part += __shfl_xor(part, 8, 16);
      part += __shfl_xor(part, 4, 16);
      part += __shfl_xor(part, 2, 16);
      part += __shfl_xor(part, 1, 16);
      const int r = (tid + j * kThreads) >> 4;
      if (s_ok) {
        const float qe[4] = {qr[j].x, qr[j].y, qr[j].z, qr[j].w}, ge[4] = {gr[j].x, gr[j].y, gr[j].z, gr[j].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int d = s_c4 * 4 + i;
          Qimg[r * I::LD + I::col(d)] = (__bf16)qe[i];
          Gimg[r * I::LD + I::col(d)] = (__bf16)ge[i];
          Qt[d * T::LD + r] = (__bf16)qe[i];
          Gt[d * T::LD + r] = (__bf16)ge[i];
        }
      }
      if (((tid + j * kThreads) & 15) == 0) Del[r] = part;
    }
    if (tid < 64) Lse[tid] = sr;
  };
  const int tile0 = qsplit * q_tiles_per_wg;
  const int iters = min(q_tiles_per_wg, (Lq + 63) / 64 - tile0);
  fetch(tile0 * 64);
  __syncthreads();
  commit();
  __syncthreads();
  float *Xw = X + wave * kSub * 16 * LDX, *Rw = Red + wave * 64 * D;
  for (int it = 0; it < iters; ++it) {
    const int qs = (tile0 + it) * 64;
    const bool more = it + 1 < iters;
    if (wave_live) {
#pragma unroll 1
      for (int u = 0; u < 2; ++u) {       // two 16-query sub-tiles per K = 32 instruction of the dV^T / dK^T products
        f32x4 pd2[2][kSub], ds2[2][kSub];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const int t = 2 * u + w;
          const HFrag<NS> qF = frag_h<NS>(Qimg + (t * 16 + fr) * I::LD, fg), gF = frag_h<NS>(Gimg + (t * 16 + fr) * I::LD, fg);
          const float4 l4 = *reinterpret_cast<const float4 *>(Lse + t * 16 + fg * 4);
          const float4 d4 = *reinterpret_cast<const float4 *>(Del + t * 16 + fg * 4);
          const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int j = 0; j < kSub; ++j) {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            f32x4 st = mma_h<NS>(qF, kF[j], zero);   // S[q][key]
            f32x4 dp = mma_h<NS>(gF, vF[j], zero);   // dP[q][key] / (1 - p)
            if (wave_masked) {
              st[0] += my_bias[j]; st[1] += my_bias[j]; st[2] += my_bias[j]; st[3] += my_bias[j];
            }
            f32x4 pd, ds;
#pragma unroll
            for (int i = 0; i < 4; ++i) pd[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[i], kLog2e, -lq[i]));
            if (drop) {
              const uint32_t pair_tile = pair_col[j] + (uint32_t)(qs + t * 16 + fg * 4) * LkP;
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const uint32_t hh = pair_hash(hkey, pair_tile + (uint32_t)i * LkP);
                const bool keep = ((hh >> field_shift) & 0xffffu) >= thr;
                ds[i] = pd[i] * ((keep ? dp[i] : 0.f) - dl[i]);
                pd[i] = keep ? pd[i] : 0.f;
              }
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i) ds[i] = pd[i] * (dp[i] - dl[i]);
            }
            pd2[w][j] = pd;
            ds2[w][j] = ds;
#pragma unroll
            for (int i = 0; i < 4; ++i) Xw[j * 16 * LDX + (fg * 4 + i) * LDX + fr] = ds[i];
          }
          __builtin_amdgcn_wave_barrier();
          const f32x4 dst0 = *reinterpret_cast<const f32x4 *>(Xw + fr * LDX + fg * 4);
          f32x4 dst1 = {0.f, 0.f, 0.f, 0.f};
          if constexpr (kSub == 2) dst1 = *reinterpret_cast<const f32x4 *>(Xw + 16 * LDX + fr * LDX + fg * 4);
          __builtin_amdgcn_wave_barrier();
          const bf16x8 sT = pack8(dst0, dst1);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const f32x4 dqa = mma32(kaH[nt], sT, (f32x4){0.f, 0.f, 0.f, 0.f});   // dQ^T[d][q] over this wave's 32 keys
            if (nt * 16 + fg * 4 < D) *reinterpret_cast<f32x4 *>(Rw + (t * 16 + fr) * D + nt * 16 + fg * 4) = dqa;
          }
        }
#pragma unroll
        for (int j = 0; j < kSub; ++j) {
          const bf16x8 pb = pack8(pd2[0][j], pd2[1][j]), sb = pack8(ds2[0][j], ds2[1][j]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            av[j][nt] = mma32(pair8(Gt + (nt * 16 + fr) * T::LD + u * 32 + fg * 4), pb, av[j][nt]);
            ak[j][nt] = mma32(pair8(Qt + (nt * 16 + fr) * T::LD + u * 32 + fg * 4), sb, ak[j][nt]);
          }
        }
      }
    }
    if (more) fetch(qs + 64);
    __syncthreads();
    for (int e = tid; e < 64 * vpr; e += kThreads) {
      f32x4 a = *reinterpret_cast<const f32x4 *>(Red + 4 * e);
#pragma unroll
      for (int w = 1; w < kWaves; ++w)
        if (w < live_waves) a += *reinterpret_cast<const f32x4 *>(Red + w * 64 * D + 4 * e);
      const int r = e / vpr, c4 = e - r * vpr;
      if (qs + r < Lq) *reinterpret_cast<f32x4 *>(dqb + (long)(qs + r) * ld_dq + 4 * c4) = a * dq_scale;
    }
    if (more) commit();
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < kSub; ++j) {
    const int ki = k0 + j * 16 + fr;
    if (ki < Lk) {
      float *okp = dk + qsplit * kv_stride + ((long)b * Lk + ki) * ldo + h * D;
      float *ovp = dv + qsplit * kv_stride + ((long)b * Lk + ki) * ldo + h * D;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int n = nt * 16 + fg * 4 + i;
          if (n < D) {
            okp[n] = ak[j][nt][i];
            ovp[n] = av[j][nt][i] * inv_keep;
          }
        }
    }
  }
}

template <int WAVES, int SUB>
int launch_longk(int chunks, int q_splits, int q_tiles_per_wg, int B, int H, int Lq, int Lk, int D, const float *q,
                 const float *k, const float *v, const uint8_t *mask, const float *out, const float *dout,
                 const float *lse, float *dq_out, long ld_dq, long chunk_stride, float dq_scale, float *dk, float *dv,
                 long ld_dkv, long kv_stride, float p, uint32_t site, const uint64_t *rng_counter, hipStream_t s) {
  const size_t bytes = sizeof(float) * (size_t)longk_lds_floats<9>(36, WAVES);
  static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_bwd_longk_kernel<9, 3, WAVES, SUB>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (attr != hipSuccess) return (int)attr;
  hipLaunchKernelGGL((attn_bwd_longk_kernel<9, 3, WAVES, SUB>), dim3(chunks * q_splits, H, B), dim3(WAVES * 64), bytes, s, H, Lq,
                     Lk, D, q, k, v, mask, out, dout, lse, dq_out, ld_dq, chunk_stride, dq_scale, dk, dv, ld_dkv, kv_stride,
                     chunks, q_tiles_per_wg, p, site, rng_counter);
  return 0;
}

}  // namespace

extern "C" {

// kernels with the key-group parameter: NG = 2 for grids that leave the SIMDs a single wave each;
// BF: the bf16 matrix steps
#define ATTN_DISPATCH_K(KERNEL, split, grid, ...)                                                   \
  do {                                                                                              \
    if (split) {                                                                                    \
      const dim3 blk(kAttnThreads * 2);                                                             \
      if (D <= 16) hipLaunchKernelGGL((KERNEL<4, 1, 2>), grid, blk, 0, s, __VA_ARGS__);             \
      else if (D <= 32) hipLaunchKernelGGL((KERNEL<8, 2, 2>), grid, blk, 0, s, __VA_ARGS__);        \
      else if (D <= 36) hipLaunchKernelGGL((KERNEL<9, 3, 2>), grid, blk, 0, s, __VA_ARGS__);        \
      else hipLaunchKernelGGL((KERNEL<12, 3, 2>), grid, blk, 0, s, __VA_ARGS__);                    \
    } else {                                                                                        \
      const dim3 blk(kAttnThreads);                                                                 \
      if (D <= 16) hipLaunchKernelGGL((KERNEL<4, 1, 1>), grid, blk, 0, s, __VA_ARGS__);             \
      else if (D <= 32) hipLaunchKernelGGL((KERNEL<8, 2, 1>), grid, blk, 0, s, __VA_ARGS__);        \
      else if (D <= 36) hipLaunchKernelGGL((KERNEL<9, 3, 1>), grid, blk, 0, s, __VA_ARGS__);        \
      else hipLaunchKernelGGL((KERNEL<12, 3, 1>), grid, blk, 0, s, __VA_ARGS__);                    \
    }                                                                                               \
  } while (0)
// (f32: the exact fp32 kernels; bf16: the kernels with bf16 LDS images)
#define ATTN_DISPATCH_G(KERNEL, KERNEL_H, split, grid, ...)                                         \
  do {                                                                                              \
    if (bf16) ATTN_DISPATCH_K(KERNEL_H, split, grid, __VA_ARGS__);                                  \
    else ATTN_DISPATCH_K(KERNEL, split, grid, __VA_ARGS__);                                         \
  } while (0)
static bool split_keys(const dim3 &g, int Lk) {
  return (long)g.x * g.y * g.z <= 512 && Lk >= 128;     // (<= 512 workgroups: round 1's measurement, -23 % on 256 x 1024)
}

static int attention_fwd_impl(bool bf16, int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                              const float *v, const uint8_t *key_padding_mask, float *out, float *lse,
                              float dropout_p, uint32_t dropout_site, const uint64_t *rng_counter,
                              butd_stream_t stream) {
  if (B <= 0 || H <= 0 || Lq <= 0) return 0;
  if (D <= 0 || D > 48 || (D & 3) || Lk <= 0) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((Lq + 63) / 64, H, B);
  ATTN_DISPATCH_G(attn_fwd_kernel, attn_fwd_h_kernel, split_keys(grid, Lk), grid, H, Lq, Lk, D, q, k, v, key_padding_mask, out, lse, dropout_p,
                dropout_site, rng_counter);
  return (int)hipGetLastError();
}

static int attention_bwd_impl(bool bf16, int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                              const float *v, const uint8_t *key_padding_mask, const float *out,
                              const float *dout, const float *lse, float *delta, float *dq, float *dk,
                              float *dv, long ld_dq, long ld_dkv, float dq_scale, float dropout_p,
                              uint32_t dropout_site, const uint64_t *rng_counter, butd_stream_t stream) {
  if (B <= 0 || H <= 0 || Lq <= 0) return 0;
  if (D <= 0 || D > 48 || (D & 3) || Lk <= 0) return (int)hipErrorInvalidValue;
  if (ld_dq == 0) ld_dq = (long)H * D;
  if (ld_dkv == 0) ld_dkv = (long)H * D;
  if (ld_dq < (long)H * D || ld_dkv < (long)H * D) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const dim3 gq((Lq + 63) / 64, H, B), gk((Lk + 63) / 64, H, B);
  ATTN_DISPATCH_G(attn_bwd_dq_kernel, attn_bwd_dq_h_kernel, split_keys(gq, Lk), gq, H, Lq, Lk, D, q, k, v, key_padding_mask, out, dout, lse, delta, dq,
                ld_dq, dq_scale, dropout_p, dropout_site, rng_counter);
  ATTN_DISPATCH_G(attn_bwd_dkv_kernel, attn_bwd_dkv_h_kernel, split_keys(gk, Lq), gk, H, Lq, Lk, D, q, k, v, key_padding_mask, dout, lse, delta,
                dk, dv, ld_dkv, dropout_p, dropout_site, rng_counter);
  return (int)hipGetLastError();
}

int butd_attention_bwd_short_keys_max(void) { return 144; }

int butd_attention_bwd_short_keys(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                                  const float *v, const uint8_t *key_padding_mask, const float *out,
                                  const float *dout, const float *lse, float *delta, float *dq, float *dk,
                                  float *dv, long ld_dq, long ld_dkv, float dq_scale, float dropout_p,
                                  uint32_t dropout_site, const uint64_t *rng_counter, butd_stream_t stream) {
  if (B <= 0 || H <= 0 || Lq <= 0) return 0;
  if (D <= 0 || D > 48 || (D & 3) || Lk <= 0 || Lk > 144) return (int)hipErrorInvalidValue;
  if (ld_dq == 0) ld_dq = (long)H * D;
  if (ld_dkv == 0) ld_dkv = (long)H * D;
  if (ld_dq < (long)H * D || ld_dkv < (long)H * D) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
#define SMALLK(NS, NT)                                                                                              \
  do {                                                                                                              \
    if (Lk <= 80)                                                                                                   \
      return launch_smallk<NS, NT, 5>(B, H, Lq, Lk, D, q, k, v, key_padding_mask, out, dout, lse, delta, dq, dk, dv, \
                                      ld_dq, ld_dkv, dq_scale, dropout_p, dropout_site, rng_counter, s);            \
    return launch_smallk<NS, NT, 9>(B, H, Lq, Lk, D, q, k, v, key_padding_mask, out, dout, lse, delta, dq, dk, dv,   \
                                    ld_dq, ld_dkv, dq_scale, dropout_p, dropout_site, rng_counter, s);              \
  } while (0)
  if (D <= 16) SMALLK(4, 1);
  else if (D <= 32) SMALLK(8, 2);
  else if (D <= 36) SMALLK(9, 3);
  else SMALLK(12, 3);
#undef SMALLK
  return 0;
}

/* (include/butd_attention.h)  The plan of a call: keys per workgroup and query splits.  Measured at 8 x 8 heads
 * (scratch/attn_longk_bench.py, dropout 0.1, us).  Keys per workgroup, one split -- two kernels | 64 | 128 | 256:
 *   1024 x 1024: 426 | 372 | 341 | 312;  256 x 1024: 135 | 109 | 108 | 100;  80 x 1024: 102 | 68 | 68 | 63;
 *   256 x 256: 49 | 42 | 53 | 91;  256 x 132: 47 | 41 | 54 | 84;  1024 x 132: 126 | 126 | 168 | 298.
 * Query splits with 64-key chunks -- two kernels (short-key kernel) | 1 | 2 | 4 | 8 | 16 splits:
 *   1024 x 132: 118 (133) | 126 | 104 | 93 | 103 | 123;  1024 x 80: 108 (89) | 123 | 73 | 64 | 72 | 90;
 *   256 x 80: 37 (60) | 41 | 29 | 30 | 30;  256 x 132: 45 (94) | 42 | 40 | 46;  256 x 256: 49 | 45 | 46 | 55;  80 x 80: 29 | 25 | 20 | 20.
 * So: 256 keys (8 waves x 2 sub-tiles), one split, where that gives >= 192 workgroups; else 64 keys (4 waves x 1) with
 * the queries split so that about 512-640 workgroups exist (dK / dV then go through per-split slabs like dQ).
 * g_longk_force_*: tuning hook (0 = the rule). */
struct LongkPlan { int chunk, chunks, q_splits, q_tiles_per_wg; };
static int g_longk_force_chunk = 0, g_longk_force_splits = 0;
int butd_attention_bwd_long_keys_set_chunk(int keys, int q_splits) {   /* timing experiments only */
  g_longk_force_chunk = keys;
  g_longk_force_splits = q_splits;
  return 0;
}
static LongkPlan longk_plan(int B, int H, int Lq, int Lk, int D) {
  LongkPlan p = {0, 0, 1, 0};
  if (D != 36) return p;
  const long bh = (long)B * H;
  const int tiles = (Lq + 63) / 64;
  int want = 1;
  if (g_longk_force_chunk) p.chunk = g_longk_force_chunk;
  else if ((long)((Lk + 255) / 256) * bh >= 192) p.chunk = 256;
  else {
    p.chunk = 64;
    const long x = (long)((Lk + 63) / 64) * bh, target = tiles > 4 ? 640 : 512;   // (1024 x 132: 4 splits, 256 x 132: 2)
    want = (int)((target + x - 1) / x);
  }
  if (g_longk_force_splits) want = g_longk_force_splits;
  p.chunks = (Lk + p.chunk - 1) / p.chunk;
  want = want < 1 ? 1 : (want > tiles ? tiles : want);
  p.q_tiles_per_wg = (tiles + want - 1) / want;
  p.q_splits = (tiles + p.q_tiles_per_wg - 1) / p.q_tiles_per_wg;
  if (!g_longk_force_chunk && (long)p.chunks * p.q_splits * bh < 128) p.chunk = 0;     // would leave the part idle
  return p;
}

long butd_attention_bwd_long_keys_scratch(int B, int H, int Lq, int Lk, int D, long ld_dq) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return -1;
  const LongkPlan p = longk_plan(B, H, Lq, Lk, D);
  if (!p.chunk) return -1;
  if (ld_dq == 0) ld_dq = (long)H * D;
  if (ld_dq & 3) return -1;
  const long E = (long)H * D;
  return (p.chunks > 1 ? (long)p.chunks * B * Lq * E : 0) + (p.q_splits > 1 ? 2L * p.q_splits * B * Lk * E : 0);
}

int butd_attention_bwd_long_keys(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                                 const float *v, const uint8_t *key_padding_mask, const float *out,
                                 const float *dout, const float *lse, float *dq, float *dk, float *dv, long ld_dq,
                                 long ld_dkv, float dq_scale, float dropout_p, uint32_t dropout_site,
                                 const uint64_t *rng_counter, float *ws, long ws_floats, butd_stream_t stream) {
  const long need = butd_attention_bwd_long_keys_scratch(B, H, Lq, Lk, D, ld_dq);
  if (need < 0 || (need > 0 && !ws) || ws_floats < need) return (int)hipErrorInvalidValue;
  if (ld_dq == 0) ld_dq = (long)H * D;
  if (ld_dkv == 0) ld_dkv = (long)H * D;
  if (ld_dq < (long)H * D || ld_dkv < (long)H * D || (ld_dkv & 3)) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const LongkPlan p = longk_plan(B, H, Lq, Lk, D);
  const long E = (long)H * D, q_slab = (long)B * Lq * E, kv_slab = (long)B * Lk * E;
  const bool dq_slabs = p.chunks > 1, kv_slabs = p.q_splits > 1;
  float *ws_q = ws, *ws_k = ws + (dq_slabs ? p.chunks * q_slab : 0), *ws_v = ws_k + (kv_slabs ? p.q_splits * kv_slab : 0);
  int err;
#define LONGK(W, S)                                                                                                          \
  launch_longk<W, S>(p.chunks, p.q_splits, p.q_tiles_per_wg, B, H, Lq, Lk, D, q, k, v, key_padding_mask, out, dout, lse,      \
                     dq_slabs ? ws_q : dq, dq_slabs ? E : ld_dq, dq_slabs ? q_slab : 0, dq_slabs ? 1.f : dq_scale,            \
                     kv_slabs ? ws_k : dk, kv_slabs ? ws_v : dv, kv_slabs ? E : ld_dkv, kv_slabs ? kv_slab : 0, dropout_p,    \
                     dropout_site, rng_counter, s)
  if (p.chunk == 256) err = LONGK(8, 2);
  else if (p.chunk == 128) err = LONGK(8, 1);
  else if (p.chunk == 64) err = LONGK(4, 1);
  else return (int)hipErrorInvalidValue;
#undef LONGK
  if (err) return err;
  if (dq_slabs || kv_slabs) {
    FoldArgs fa;
    fa.nseg = 0;
    fa.E4 = (int)(E / 4);
    fa.total = 0;
    auto add = [&](const float *src, float *dst, int slabs, long stride, long rows, long ld, float scale) {
      FoldSeg &g = fa.seg[fa.nseg++];
      g.ws = src; g.dst = dst; g.slabs = slabs; g.slab_stride = stride; g.rows = rows; g.ld = ld; g.scale = scale;
      g.first = fa.total;
      fa.total += rows * (E / 4);
    };
    if (dq_slabs) add(ws_q, dq, p.chunks, q_slab, (long)B * Lq, ld_dq, dq_scale);
    if (kv_slabs) {
      add(ws_k, dk, p.q_splits, kv_slab, (long)B * Lk, ld_dkv, 1.f);
      add(ws_v, dv, p.q_splits, kv_slab, (long)B * Lk, ld_dkv, 1.f);
    }
    hipLaunchKernelGGL(attn_dq_fold_kernel, dim3((unsigned)((fa.total + 255) / 256)), dim3(256), 0, s, fa);
  }
  return (int)hipGetLastError();
}

/* the same walk on the bf16 matrix cores (BASELINE configs[3]): the fp32 plan's 256- and 64-key chunks (8 waves x 2 / 4 waves
 * x 1 sub-tiles: a single sub-tile fills half of the dQ product's K = 32 contraction) with the same slabs and fold */
long butd_attention_bwd_long_keys_bf16_scratch(int B, int H, int Lq, int Lk, int D, long ld_dq) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return -1;
  const LongkPlan p = longk_plan(B, H, Lq, Lk, D);
  if (p.chunk != 256 && p.chunk != 64) return -1;
  // measured (8 x 8 heads, us, two bf16 kernels | one pass): 1024 x 1024: 256 | 135; 256 x 1024: 89 | 58; 80 x 1024: 71 | 44;
  // 1024 x 80: 77 | 47; 1024 x 132: 83 | 78; 256 x 132: 37 | 34; 256 x 80: 32 | 29; 80 x 80: 27 | 20; 256 x 256: 38 | 40
  if (!g_longk_force_chunk && p.chunk == 64 && p.chunks >= 4 && (Lq + 63) / 64 <= 4) return -1;
  return butd_attention_bwd_long_keys_scratch(B, H, Lq, Lk, D, ld_dq);
}

extern "C++" {
template <int WAVES, int SUB>
static int launch_longk_h(const LongkPlan &p, int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                          const float *v, const uint8_t *mask, const float *out, const float *dout, const float *lse,
                          float *dq_out, long ld_dq, long chunk_stride, float dq_scale, float *dk, float *dv, long ld_dkv,
                          long kv_stride, float pd, uint32_t site, const uint64_t *rng_counter, hipStream_t s) {
  using I = ImgH<9>;
  using T = TImgH<3>;
  const size_t bytes = sizeof(float) * (size_t)(64 + 64 + WAVES * SUB * 16 * 20 + WAVES * 64 * 36) +
                       sizeof(__bf16) * (size_t)(2 * 64 * I::LD + 2 * T::ROWS * T::LD);
  static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_bwd_longk_h_kernel<9, 3, WAVES, SUB>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (attr != hipSuccess) return (int)attr;
  hipLaunchKernelGGL((attn_bwd_longk_h_kernel<9, 3, WAVES, SUB>), dim3(p.chunks * p.q_splits, H, B), dim3(WAVES * 64), bytes, s, H,
                     Lq, Lk, D, q, k, v, mask, out, dout, lse, dq_out, ld_dq, chunk_stride, dq_scale, dk, dv, ld_dkv, kv_stride,
                     p.chunks, p.q_tiles_per_wg, pd, site, rng_counter);
  return 0;
}
}  // extern "C++"

int butd_attention_bwd_long_keys_bf16(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                                      const float *v, const uint8_t *key_padding_mask, const float *out,
                                      const float *dout, const float *lse, float *dq, float *dk, float *dv, long ld_dq,
                                      long ld_dkv, float dq_scale, float dropout_p, uint32_t dropout_site,
                                      const uint64_t *rng_counter, float *ws, long ws_floats, butd_stream_t stream) {
  const long need = butd_attention_bwd_long_keys_bf16_scratch(B, H, Lq, Lk, D, ld_dq);
  if (need < 0 || (need > 0 && !ws) || ws_floats < need) return (int)hipErrorInvalidValue;
  if (ld_dq == 0) ld_dq = (long)H * D;
  if (ld_dkv == 0) ld_dkv = (long)H * D;
  if (ld_dq < (long)H * D || ld_dkv < (long)H * D || (ld_dkv & 3)) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const LongkPlan p = longk_plan(B, H, Lq, Lk, D);
  const long E = (long)H * D, q_slab = (long)B * Lq * E, kv_slab = (long)B * Lk * E;
  const bool dq_slabs = p.chunks > 1, kv_slabs = p.q_splits > 1;
  float *ws_q = ws, *ws_k = ws + (dq_slabs ? p.chunks * q_slab : 0), *ws_v = ws_k + (kv_slabs ? p.q_splits * kv_slab : 0);
  int err;
#define LONGKH(W, S)                                                                                                          \
  launch_longk_h<W, S>(p, B, H, Lq, Lk, D, q, k, v, key_padding_mask, out, dout, lse, dq_slabs ? ws_q : dq,                    \
                       dq_slabs ? E : ld_dq, dq_slabs ? q_slab : 0, dq_slabs ? 1.f : dq_scale, kv_slabs ? ws_k : dk,          \
                       kv_slabs ? ws_v : dv, kv_slabs ? E : ld_dkv, kv_slabs ? kv_slab : 0, dropout_p, dropout_site,         \
                       rng_counter, s)
  if (p.chunk == 256) err = LONGKH(8, 2);
  else err = LONGKH(4, 1);
#undef LONGKH
  if (err) return err;
  if (dq_slabs || kv_slabs) {
    FoldArgs fa;
    fa.nseg = 0;
    fa.E4 = (int)(E / 4);
    fa.total = 0;
    auto add = [&](const float *src, float *dst, int slabs, long stride, long rows, long ld, float scale) {
      FoldSeg &g = fa.seg[fa.nseg++];
      g.ws = src; g.dst = dst; g.slabs = slabs; g.slab_stride = stride; g.rows = rows; g.ld = ld; g.scale = scale;
      g.first = fa.total;
      fa.total += rows * (E / 4);
    };
    if (dq_slabs) add(ws_q, dq, p.chunks, q_slab, (long)B * Lq, ld_dq, dq_scale);
    if (kv_slabs) {
      add(ws_k, dk, p.q_splits, kv_slab, (long)B * Lk, ld_dkv, 1.f);
      add(ws_v, dv, p.q_splits, kv_slab, (long)B * Lk, ld_dkv, 1.f);
    }
    hipLaunchKernelGGL(attn_dq_fold_kernel, dim3((unsigned)((fa.total + 255) / 256)), dim3(256), 0, s, fa);
  }
  return (int)hipGetLastError();
}

int butd_attention_fwd(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                       const float *v, const uint8_t *key_padding_mask, float *out, float *lse,
                       float dropout_p, uint32_t dropout_site, const uint64_t *rng_counter,
                       butd_stream_t stream) {
  return attention_fwd_impl(false, B, H, Lq, Lk, D, q, k, v, key_padding_mask, out, lse, dropout_p, dropout_site,
                            rng_counter, stream);
}

int butd_attention_fwd_bf16(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                            const float *v, const uint8_t *key_padding_mask, float *out, float *lse,
                            float dropout_p, uint32_t dropout_site, const uint64_t *rng_counter,
                            butd_stream_t stream) {
  return attention_fwd_impl(true, B, H, Lq, Lk, D, q, k, v, key_padding_mask, out, lse, dropout_p, dropout_site,
                            rng_counter, stream);
}

int butd_attention_fwd_split_bf16(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                                  const float *v, const uint8_t *key_padding_mask, float *out, float *lse,
                                  float dropout_p, uint32_t dropout_site, const uint64_t *rng_counter,
                                  butd_stream_t stream) {
  if (B <= 0 || H <= 0 || Lq <= 0) return 0;
  if (D != 36 || Lk <= 0) return (int)hipErrorInvalidValue;       // (the microbenchmark's instance: head dimension 36)
  const dim3 grid((Lq + 63) / 64, H, B);
  hipLaunchKernelGGL((attn_fwd_h_kernel<9, 3, 1, true>), grid, dim3(kAttnThreads), 0, (hipStream_t)stream, H, Lq, Lk, D, q, k,
                     v, key_padding_mask, out, lse, dropout_p, dropout_site, rng_counter);
  return (int)hipGetLastError();
}

int butd_attention_bwd(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                       const float *v, const uint8_t *key_padding_mask, const float *out,
                       const float *dout, const float *lse, float *delta, float *dq, float *dk,
                       float *dv, long ld_dq, long ld_dkv, float dq_scale, float dropout_p,
                       uint32_t dropout_site, const uint64_t *rng_counter, butd_stream_t stream) {
  return attention_bwd_impl(false, B, H, Lq, Lk, D, q, k, v, key_padding_mask, out, dout, lse, delta, dq, dk, dv, ld_dq,
                            ld_dkv, dq_scale, dropout_p, dropout_site, rng_counter, stream);
}

int butd_attention_bwd_bf16(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                            const float *v, const uint8_t *key_padding_mask, const float *out,
                            const float *dout, const float *lse, float *delta, float *dq, float *dk,
                            float *dv, long ld_dq, long ld_dkv, float dq_scale, float dropout_p,
                            uint32_t dropout_site, const uint64_t *rng_counter, butd_stream_t stream) {
  return attention_bwd_impl(true, B, H, Lq, Lk, D, q, k, v, key_padding_mask, out, dout, lse, delta, dq, dk, dv, ld_dq,
                            ld_dkv, dq_scale, dropout_p, dropout_site, rng_counter, stream);
}

}  // extern "C"

#ifdef ATTN_PROF
extern "C" int butd_attention_prof_read(unsigned long long *host8) {
  return (int)hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_attn_prof), sizeof(unsigned long long) * 8);
}
#endif
