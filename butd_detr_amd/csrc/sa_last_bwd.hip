// sa_last_bwd.hip -- backward of a set-abstraction level's LAST shared-MLP layer + max-pool in training mode without
// ever forming its dense gradient (include/butd_sa.h, butd_sa_last_bwd).
//
// Reference arithmetic: pointnet2_modules.py:243-257 (SharedMLP -> F.max_pool2d over nsample), pytorch_utils.py:11-36
// (Conv2d -> BatchNorm2d(batch statistics) -> ReLU).  With H = relu(bn2(Z2)) (P x C2, recomputed from Z2), Z3 = H W3^T,
// zhat = (Z3 - mu) rstd, s = gamma rstd, and g = the pooled gradient scattered to each (group, channel)'s arg-max row
// (one non-zero per group and channel), the BatchNorm backward is
//     dZ3 = s (g - m1 - zhat m2),      m1 = sum(g) / P,  m2 = sum(g zhat) / P        (dense only through m1, m2)
// and both products that consume dZ3 are linear in its three terms:
//     dH  = dZ3 W3      =  [sparse: row r gets sum_{c: argmax(group, c) = r} g s_c W3[c,:]]  -  H A  +  d
//                          A = W3^T diag(s m2 rstd) W3  (C2 x C2),   d = W3^T (s m2 rstd mu - s m1)
//     dW3 = dZ3^T H     =  s (T - m1 S^T - m2 rstd (W3 Gram - mu S^T))
//                          T[c,:] = sum_groups g H[argmax row,:],  S = column sums of H,  Gram = H^T H  (C2 x C2)
// So the level's 10^5..10^6-row tensors are touched as: Z2 read (twice), the layer-2 gradient written and re-read once
// -- instead of Z3 read, dZ3 written and read twice, Z2 read, dH read and re-read (3.25 GB -> 1.34 GB at SA1, B = 8),
// and the matrix work halves (2 P C2^2 + 2 P C2^2 flops instead of 4 P C2 C3).  The ReLU gate of layer 2 and the sums
// its BatchNorm backward needs are taken in the same pass (they were butd_sa_mask_stats).
//
//   sa_last_coeffs_kernel   A (negated), d                                   (tiny, double accumulation)
//   sa_last_mfma_kernel     O = H (-A) + d  -> dH buffer;  Gram partial per workgroup      (fp32 MFMA 16x16x4)
//   sa_last_sparse_kernel   O += sparse rows; gate; layer-2 sums; T, S partials per workgroup; writes the gated gradient
//   sa_last_reduce_kernel   partials -> double totals (deterministic: no atomics anywhere on this path)
//   sa_last_dw_kernel       dW3 and the layer-2 sums
#include <hip/hip_runtime.h>
// (-DBUTD_MAIN_PRIO=n raises the wave priority of this unit's kernels: the captured step's prefetch branches share CUs
// with them; profiles/r06_side_branches.txt)
#ifdef BUTD_MAIN_PRIO
#define BUTD_MAIN_PRIO_SET() __builtin_amdgcn_s_setprio(BUTD_MAIN_PRIO)
#else
#define BUTD_MAIN_PRIO_SET()
#endif

#include <math.h>
#include <stdint.h>

#include "../../include/butd_sa.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kRows = 64;   // rows per block of the two streaming kernels (a multiple of every nsample: 16, 32, 64)

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// z1 = X W1^T of one element: THE order of operations of butd_sa_thin_conv and of every kernel here that recomputes it
__device__ __forceinline__ float z1_dot(const float *__restrict__ x, const float *__restrict__ w) {
  float a = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) a += x[k] * w[k];
  return a;
}

// --------------------------------------------------------------------------------- forward of the last layer + pool
// When the backward below is in use nothing reads Z3 = H2 W3^T after the pooling, so the forward need not write it:
// per 64-row block  H tile -> LDS (two buffers: one barrier per block);  Z3 tiles on the matrix cores with the ROWS in
// the accumulator registers of a lane (D[row = 4 (lane/16) + i][column = lane % 16]): a lane then owns one channel of
// each of its tiles and folds its rows in registers -- BatchNorm sums (fp32 over the block's 16 rows per lane, then
// double for the whole kernel) and, per group of ns rows, max / min with their FIRST positions (strict compares in row
// order, ties across lanes to the smaller position: butd_sa_colstats' rule).  Two xor-shuffles per group finish a
// channel.  Replaces the layer's product launch + butd_sa_colstats: reads Z2 once, writes 10 bytes per (group, channel).
template <int C2, int C3, int NW>
__global__ __launch_bounds__(NW * 64) void sa_last_fwd_kernel(
    long P, long nblk, int ns, long G, const float *__restrict__ Z2, const float *__restrict__ sc2,
    const float *__restrict__ sh2, const float *__restrict__ W3, double *__restrict__ sum,
    double *__restrict__ sumsq, float *__restrict__ zmax, float *__restrict__ zmin, uint8_t *__restrict__ amax,
    uint8_t *__restrict__ amin, unsigned int *__restrict__ sched, int chunk) {
  BUTD_MAIN_PRIO_SET();
  constexpr int NT = NW * 64;
  constexpr int ST = C2 + 36;
  constexpr int NTO = C3 / (16 * NW);    // 16-column output tiles per wave
  constexpr int KG = C2 / 16;
  constexpr int QN = C2 / 4;
  constexpr int RP = NT / QN;
  constexpr int NP = kRows / RP;
  static_assert(C3 % (16 * NW) == 0 && kRows % RP == 0, "decomposition");
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][64][ST]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lm = lane & 15, lq = lane >> 4;
  const int q = tid % QN, rsub = tid / QN;
  const int n0 = wave * (C3 / NW);
  const int rtg = ns / 16;               // 16-row tiles per group (1, 2 or 4)
  f4 wreg[NTO][KG];                      // B operand: W3[column][16 g + 4 lq + i]
#pragma unroll
  for (int t = 0; t < NTO; ++t)
#pragma unroll
    for (int g = 0; g < KG; ++g)
      wreg[t][g] = *reinterpret_cast<const f4 *>(W3 + (long)(n0 + 16 * t + lm) * C2 + 16 * g + 4 * lq);
  const f4 sc = *reinterpret_cast<const f4 *>(sc2 + 4 * q), sh = *reinterpret_cast<const f4 *>(sh2 + 4 * q);
  double dsum[NTO], dsq[NTO];
#pragma unroll
  for (int t = 0; t < NTO; ++t) dsum[t] = dsq[t] = 0.0;
  f4 zn[NP];
  auto fetch = [&](long b) {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const long p = b * kRows + rsub + ps * RP;
      zn[ps] = p < P ? *reinterpret_cast<const f4 *>(Z2 + p * C2 + 4 * q) : f4{0.f, 0.f, 0.f, 0.f};
    }
  };
  // Blocks are handed out by an atomic counter, not by blockIdx: the step runs this kernel next to a side queue that
  // holds a few CUs for milliseconds (the next batch's sampling chain); a workgroup that is placed late must find the
  // work already taken instead of owning a fixed share of it (measured: 344 us with fixed shares, 185 us alone).  The
  // counter hands out chunks of blocks (one fetch per block costs ~15 ns each on ONE address: 16 384 of them were the
  // kernel's duration at SA1); the chunk after next is fetched while the current one is processed; sched[0] = next
  // chunk, sched[1] = workgroups done (the last one zeroes both for the next launch).
  long *s_idx = reinterpret_cast<long *>(lds + 2 * kRows * ST);     // (after the two H buffers; 16-byte aligned)
  const long nchunk = (nblk + chunk - 1) / chunk;                   // the counter hands out chunks of `chunk` blocks
  long fetched = 0;
  const bool dyn = sched != nullptr;       // NULL: fixed shares (blockIdx, + gridDim, ...): grids of several workgroups per CU
  if (dyn && tid == 0) s_idx[0] = (long)atomicAdd(sched, 2u);        // the first two chunks in one fetch
  __syncthreads();
  long cur = dyn ? s_idx[0] : (long)blockIdx.x, nxt = dyn ? cur + 1 : cur + gridDim.x, after = 0;
  __syncthreads();
  int j = 0;
  long blk = cur;                                                    // chunk c = blocks c, c + nchunk, c + 2 nchunk, ...:
  if (cur < nchunk) fetch(blk);                                      // workgroups stream NEIGHBOURING blocks at any time
  if (dyn && tid == 0) fetched = (long)atomicAdd(sched, 1u);         // (issued after the loads: they return first)
  int buf = 0;
  for (; cur < nchunk; buf ^= 1) {
    float *Ht = lds + buf * (kRows * ST);
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int r = rsub + ps * RP;
      f4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = fmaxf(sc[e] * zn[ps][e] + sh[e], 0.f);
      *reinterpret_cast<f4 *>(Ht + r * ST + 4 * q) = h;
    }
    if (dyn && j == 0 && tid == 0) s_idx[buf] = fetched;
    __syncthreads();      // (the other buffer is free again: every wave finished reading it before it got here)
    if (j == 0) after = dyn ? s_idx[buf] : nxt + gridDim.x;
    int nj = j + 1;
    long ncur = cur;
    if (nj == chunk || cur + nj * nchunk >= nblk) { nj = 0; ncur = nxt; }
    const long nblkidx = ncur + nj * nchunk;
    if (ncur < nchunk) fetch(nblkidx);
    if (dyn && j == 0 && tid == 0) fetched = (long)atomicAdd(sched, 1u);
    float s32[NTO], q32[NTO], mx[NTO], mn[NTO];
    int ax[NTO], an[NTO];
#pragma unroll
    for (int t = 0; t < NTO; ++t) s32[t] = q32[t] = 0.f;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      if (rt % rtg == 0) {
#pragma unroll
        for (int t = 0; t < NTO; ++t) {
          mx[t] = -INFINITY; mn[t] = INFINITY; ax[t] = an[t] = 0;
        }
      }
      f4 oacc[NTO];
#pragma unroll
      for (int t = 0; t < NTO; ++t) oacc[t] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        const f4 hv = *reinterpret_cast<const f4 *>(Ht + (16 * rt + lm) * ST + 16 * g + 4 * lq);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < NTO; ++t) oacc[t] = mfma4(hv[i], wreg[t][g][i], oacc[t]);
      }
      // lane: rows 16 rt + 4 lq + i (i = 0..3) of column n0 + 16 t + lm
      const bool live = blk * kRows + 16 * rt < P;      // (P is a multiple of ns >= 16: a 16-row tile is all in or all out)
      const int kbase = (16 * rt) % ns + 4 * lq;        // position inside the group
#pragma unroll
      for (int t = 0; t < NTO; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = live ? oacc[t][i] : 0.f;
          s32[t] += v;
          q32[t] += v * v;
          if (v > mx[t]) { mx[t] = v; ax[t] = kbase + i; }
          if (v < mn[t]) { mn[t] = v; an[t] = kbase + i; }
        }
      if ((rt + 1) % rtg == 0) {       // the group is complete: fold the four row quarters (lanes 16 and 32 apart)
        const long g = blk * (kRows / ns) + rt / rtg;
#pragma unroll
        for (int t = 0; t < NTO; ++t) {
          float a = mx[t], b2 = mn[t];
          int ia = ax[t], ib = an[t];
#pragma unroll
          for (int d = 16; d <= 32; d <<= 1) {
            const float oa = __shfl_xor(a, d), ob = __shfl_xor(b2, d);
            const int oia = __shfl_xor(ia, d), oib = __shfl_xor(ib, d);
            if (oa > a || (oa == a && oia < ia)) { a = oa; ia = oia; }
            if (ob < b2 || (ob == b2 && oib < ib)) { b2 = ob; ib = oib; }
          }
          if (lq == 0 && g < G && live) {
            const long o = g * C3 + n0 + 16 * t + lm;
            zmax[o] = a; zmin[o] = b2; amax[o] = (uint8_t)ia; amin[o] = (uint8_t)ib;
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < NTO; ++t) {
      dsum[t] += (double)s32[t];
      dsq[t] += (double)q32[t];
    }
    if (nj == 0) { cur = nxt; nxt = after; }
    j = nj;
    blk = nblkidx;
  }
  if (dyn && tid == 0) {
    __threadfence();
    if (atomicAdd(sched + 1, 1u) == gridDim.x - 1) {    // the last workgroup out: every fetch has happened
      sched[0] = 0u;
      sched[1] = 0u;
    }
  }
#pragma unroll
  for (int t = 0; t < NTO; ++t) {
    double a = dsum[t], b2 = dsq[t];
    a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
    b2 += __shfl_xor(b2, 16); b2 += __shfl_xor(b2, 32);
    if (lq == 0) {
      atomicAdd(sum + n0 + 16 * t + lm, a);
      atomicAdd(sumsq + n0 + 16 * t + lm, b2);
    }
  }
}

// ------------------------------------------------------------------------------------------------ coefficients
// grid (C2/16, C2/16) x 256 threads: a 16 x 16 tile of An = -W3^T diag(u) W3 per block, the two 16-column slices of W3
// staged in LDS; the blocks of the first tile row also write d = W3^T v for their 16 columns.  Double accumulation.
template <int C3>
__global__ __launch_bounds__(256) void sa_last_coeffs_kernel(int C2, long P, const float *__restrict__ W3,
                                                             const float *__restrict__ scale3,
                                                             const float *__restrict__ mean3,
                                                             const float *__restrict__ rstd3,
                                                             const double *__restrict__ S1,
                                                             const double *__restrict__ S2, float *__restrict__ An,
                                                             float *__restrict__ dvec) {
  BUTD_MAIN_PRIO_SET();
  __shared__ float Wj[C3][17], Wk[C3][17];
  __shared__ double u[C3], v[C3];
  const double invP = 1.0 / (double)P;
  const int tid = threadIdx.x, j0 = blockIdx.x * 16, k0 = blockIdx.y * 16;
  for (int c = tid; c < C3; c += 256) {
    const double s = (double)scale3[c], m1 = S1[c] * invP, m2 = S2[c] * invP, rs = (double)rstd3[c];
    u[c] = s * m2 * rs;
    v[c] = u[c] * (double)mean3[c] - s * m1;
  }
  for (int e = tid; e < C3 * 16; e += 256) {
    const int c = e >> 4, t = e & 15;
    Wj[c][t] = W3[(long)c * C2 + j0 + t];
    Wk[c][t] = W3[(long)c * C2 + k0 + t];
  }
  __syncthreads();
  const int tj = tid >> 4, tk = tid & 15;
  double a = 0.0, d = 0.0;
#pragma unroll 8
  for (int c = 0; c < C3; ++c) {
    const double wk = (double)Wk[c][tk];
    a += (double)Wj[c][tj] * u[c] * wk;
    d += wk * v[c];
  }
  An[(long)(j0 + tj) * C2 + k0 + tk] = (float)(-a);
  if (blockIdx.x == 0 && tj == 0) dvec[k0 + tk] = (float)d;
}

// ------------------------------------------------------------------------------------------- O = H An + d, Gram
// Persistent workgroups of 4 waves over 64-row blocks.  LDS: the H tile [64][C2 + 36] (row stride = 36 mod 64 banks: the
// 16-row x 16-byte operand reads of the O product are conflict-free).  O^T tile (16 output columns x 16 rows) = An-rows
// (A operand, in REGISTERS for the whole kernel: the wave owns C2/4 output columns) x H^T (B operand: lane (row, kq)
// reads 4 consecutive channels = the k indices of 4 consecutive matrix instructions).  Gram rows owned per wave likewise;
// its operands are single-float reads of H[4s + kq][column].
template <int C2>
__global__ __launch_bounds__(kThreads) void sa_last_mfma_kernel(
    long P, long nblk, const float *__restrict__ Z2, const float *__restrict__ sc2, const float *__restrict__ sh2,
    const float *__restrict__ An, const float *__restrict__ dvec, float *__restrict__ O, float *__restrict__ ws) {
  BUTD_MAIN_PRIO_SET();
  constexpr int ST = C2 + 36;
  constexpr int NTO = C2 / 64;   // output-column tiles of 16 per wave
  constexpr int KG = C2 / 16;    // contraction groups of 16 channels
  constexpr int GM = C2 / 64;    // Gram row tiles per wave
  constexpr int GN = C2 / 16;    // Gram column tiles
  constexpr int QN = C2 / 4;     // float4 chunks per row
  constexpr int RP = kThreads / QN;   // rows per load pass
  constexpr int NP = kRows / RP;      // load passes per block
  __shared__ float Ht[kRows * ST];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lm = lane & 15, lq = lane >> 4;
  const int n0 = wave * (C2 / 4);
  f4 areg[NTO][KG];
#pragma unroll
  for (int t = 0; t < NTO; ++t)
#pragma unroll
    for (int g = 0; g < KG; ++g)
      areg[t][g] = *reinterpret_cast<const f4 *>(An + (long)(n0 + 16 * t + lm) * C2 + 16 * g + 4 * lq);
  f4 dinit[NTO];
#pragma unroll
  for (int t = 0; t < NTO; ++t) dinit[t] = *reinterpret_cast<const f4 *>(dvec + n0 + 16 * t + 4 * lq);
  f4 gacc[GM][GN];
#pragma unroll
  for (int m = 0; m < GM; ++m)
#pragma unroll
    for (int n = 0; n < GN; ++n) gacc[m][n] = f4{0.f, 0.f, 0.f, 0.f};
  const int q = tid % QN, rsub = tid / QN;
  const f4 sc = *reinterpret_cast<const f4 *>(sc2 + 4 * q), sh = *reinterpret_cast<const f4 *>(sh2 + 4 * q);
  f4 znext[NP];
  long blk = blockIdx.x;
  auto load_block = [&](long b) {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const long p = b * kRows + rsub + ps * RP;
      znext[ps] = p < P ? *reinterpret_cast<const f4 *>(Z2 + p * C2 + 4 * q) : f4{0.f, 0.f, 0.f, 0.f};
    }
  };
  if (blk < nblk) load_block(blk);
  for (; blk < nblk; blk += gridDim.x) {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int r = rsub + ps * RP;
      const bool in = blk * kRows + r < P;
      f4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = in ? fmaxf(sc[e] * znext[ps][e] + sh[e], 0.f) : 0.f;
      *reinterpret_cast<f4 *>(Ht + r * ST + 4 * q) = h;
    }
    __syncthreads();
    if (blk + gridDim.x < nblk) load_block(blk + gridDim.x);   // in flight under the matrix work below
    // ---- O^T tiles
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      f4 oacc[NTO];
#pragma unroll
      for (int t = 0; t < NTO; ++t) oacc[t] = dinit[t];
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        const f4 hv = *reinterpret_cast<const f4 *>(Ht + (16 * rt + lm) * ST + 16 * g + 4 * lq);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < NTO; ++t) oacc[t] = mfma4(areg[t][g][i], hv[i], oacc[t]);
      }
      const long p = blk * kRows + 16 * rt + lm;
      if (p < P) {
#pragma unroll
        for (int t = 0; t < NTO; ++t) *reinterpret_cast<f4 *>(O + p * C2 + n0 + 16 * t + 4 * lq) = oacc[t];
      }
    }
    // ---- Gram += H^T H over the block's rows
#pragma unroll 4
    for (int s = 0; s < kRows / 4; ++s) {
      const float *hrow = Ht + (4 * s + lq) * ST + lm;
      float a[GM], b[GN];
#pragma unroll
      for (int m = 0; m < GM; ++m) a[m] = hrow[n0 + 16 * m];
#pragma unroll
      for (int n = 0; n < GN; ++n) b[n] = hrow[16 * n];
#pragma unroll
      for (int m = 0; m < GM; ++m)
#pragma unroll
        for (int n = 0; n < GN; ++n) gacc[m][n] = mfma4(a[m], b[n], gacc[m][n]);
    }
    __syncthreads();
  }
  float *out = ws + (long)blockIdx.x * C2 * C2;
#pragma unroll
  for (int m = 0; m < GM; ++m)
#pragma unroll
    for (int n = 0; n < GN; ++n)
#pragma unroll
      for (int i = 0; i < 4; ++i) out[(long)(n0 + 16 * m + 4 * lq + i) * C2 + 16 * n + lm] = gacc[m][n][i];
}

// --------------------------------------------------------------------- sparse rows, gate, layer-2 sums, T and S
// Same block decomposition, NW waves per workgroup.  LDS: the O tile and the RAW Z2 tile [64][C2 + 4].  A wave owns
// C3/NW channels: its W3 rows and its T accumulators live in registers (lane = column, C2/64 columns per lane).  The
// (group, channel) records -- pooled gradient where the pooled activation is > 0, arg-max row -- are loaded one per LANE
// (lane ci = the wave's channel ci) and broadcast with v_readlane: the loop over the wave's channels is branch-free
// straight-line code, so its LDS operations (one float atomic into the arg-max row of the O tile -- other waves may
// hit the same row -- and one read of that row's activation for T) are all in flight together.  Then the element-wise
// pass: gate by H > 0, the two BatchNorm sums of layer 2 and the column sums of H per thread, the gated gradient to
// memory.  The next block's rows and records are fetched while the current block is processed.
template <int C2, int C3, int NW>
__global__ __launch_bounds__(NW * 64) void sa_last_sparse_kernel(
    long P, long nblk, int ns, long G, float *__restrict__ O, const float *__restrict__ Z2,
    const float *__restrict__ sc2, const float *__restrict__ sh2, const float *__restrict__ mean2,
    const float *__restrict__ rstd2, const float *__restrict__ W3, const float *__restrict__ d_out,
    const float *__restrict__ zsel, const uint8_t *__restrict__ asel, const float *__restrict__ sc3,
    const float *__restrict__ sh3, float *__restrict__ ws, long ws_stride, int abl) {
  BUTD_MAIN_PRIO_SET();
  constexpr int NT = NW * 64;
  constexpr int ST = C2 + 4;
  constexpr int CPW = C3 / NW;       // channels per wave (<= 64: one record per lane)
  constexpr int KL = C2 / 64;
  constexpr int QN = C2 / 4;
  constexpr int RP = NT / QN;
  constexpr int NP = kRows / RP;
  static_assert(CPW <= 64 && kRows % RP == 0, "decomposition");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *Os = lds;                   // [64][ST]
  float *Zs = Os + kRows * ST;       // [64][ST]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int q = tid % QN, rsub = tid / QN;
  const int nc = kRows / ns;         // groups per block (1, 2 or 4)
  const int myc = wave * CPW + (lane % CPW);
  float w3r[CPW][KL], tacc[CPW][KL];
#pragma unroll
  for (int ci = 0; ci < CPW; ++ci) {
    const int c = wave * CPW + ci;
    const float s3 = sc3[c];
#pragma unroll
    for (int kk = 0; kk < KL; ++kk) {
      w3r[ci][kk] = s3 * W3[(long)c * C2 + lane + 64 * kk];      // s W3[c,:]
      tacc[ci][kk] = 0.f;
    }
  }
  float sck[KL], shk[KL];
#pragma unroll
  for (int kk = 0; kk < KL; ++kk) {
    sck[kk] = sc2[lane + 64 * kk];
    shk[kk] = sh2[lane + 64 * kk];
  }
  const float my_sc3 = sc3[myc], my_sh3 = sh3[myc];
  const f4 sc = *reinterpret_cast<const f4 *>(sc2 + 4 * q), sh = *reinterpret_cast<const f4 *>(sh2 + 4 * q);
  const f4 mu = *reinterpret_cast<const f4 *>(mean2 + 4 * q), rs = *reinterpret_cast<const f4 *>(rstd2 + 4 * q);
  f4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, scol = {0.f, 0.f, 0.f, 0.f};
  f4 zn[NP], on[NP];
  float rec_g[4];
  int rec_row[4];
  auto fetch = [&](long b) {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const long p = b * kRows + rsub + ps * RP;
      const bool in = p < P;
      zn[ps] = in ? *reinterpret_cast<const f4 *>(Z2 + p * C2 + 4 * q) : f4{0.f, 0.f, 0.f, 0.f};
      on[ps] = in ? *reinterpret_cast<const f4 *>(O + p * C2 + 4 * q) : f4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      const long g = b * nc + gi;
      float gv = 0.f;
      int row = 0;
      if (gi < nc && g < G) {
        const float z = zsel[g * C3 + myc];
        const float dy = d_out[g * C3 + myc];
        row = gi * ns + (int)asel[g * C3 + myc];
        gv = my_sc3 * z + my_sh3 > 0.f ? dy : 0.f;
      }
      rec_g[gi] = gv;
      rec_row[gi] = row;
    }
  };
  long blk = blockIdx.x;
  if (blk < nblk) fetch(blk);
  for (; blk < nblk; blk += gridDim.x) {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int r = rsub + ps * RP;
      *reinterpret_cast<f4 *>(Zs + r * ST + 4 * q) = zn[ps];
      *reinterpret_cast<f4 *>(Os + r * ST + 4 * q) = on[ps];
    }
    float cg[4];
    int cr[4];
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      cg[gi] = rec_g[gi];
      cr[gi] = rec_row[gi];
    }
    __syncthreads();
    if (blk + gridDim.x < nblk) fetch(blk + gridDim.x);          // in flight under the work below
    // ---- T[c,:] += g H[arg-max row,:]   (reads only; branch-free)
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      if (gi < nc && !(abl & 2)) {
#pragma unroll
        for (int ci = 0; ci < CPW; ++ci) {
          const float gv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cg[gi]), ci));
          const int ro = __builtin_amdgcn_readlane(cr[gi], ci) * ST;
#pragma unroll
          for (int kk = 0; kk < KL; ++kk)
            tacc[ci][kk] += gv * fmaxf(sck[kk] * Zs[ro + lane + 64 * kk] + shk[kk], 0.f);
        }
      }
    }
    // ---- O[arg-max row,:] += g s W3[c,:].  Waves own channels, so two waves may target the same row; LDS float
    // atomics serialise per lane on this part (measured: 0.55 ms of the 0.94 ms this kernel took at SA1 with them).
    // Instead the rows are cut into NW slots and the update runs in NW phases: in phase p wave w applies its records
    // whose row lies in slot (w + p) mod NW -- disjoint rows per wave within a phase, plain read-add-write, and the
    // LDS queue of a wave is in order, so two of its records on one row are applied one after the other.
    for (int ph = 0; ph < NW && !(abl & 1); ++ph) {
      const int slot = (wave + ph) % NW;
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        if (gi < nc) {
          const unsigned long long m = __ballot((cr[gi] & 63) / (kRows / NW) == slot && cg[gi] != 0.f);
          if (m != 0ull) {
#pragma unroll
            for (int ci = 0; ci < CPW; ++ci) {
              if ((m >> ci) & 1ull) {
                const float gv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cg[gi]), ci));
                const int ro = __builtin_amdgcn_readlane(cr[gi], ci) * ST;
#pragma unroll
                for (int kk = 0; kk < KL; ++kk) {
                  float *a = Os + ro + lane + 64 * kk;
                  *a = *a + gv * w3r[ci][kk];
                }
              }
            }
          }
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int r = rsub + ps * RP;
      const long p = blk * kRows + r;
      if (p < P && !(abl & 8)) {
        const f4 o = *reinterpret_cast<const f4 *>(Os + r * ST + 4 * q);
        const f4 z = *reinterpret_cast<const f4 *>(Zs + r * ST + 4 * q);
        f4 g2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float h = fmaxf(sc[e] * z[e] + sh[e], 0.f);
          g2[e] = h > 0.f ? o[e] : 0.f;
          s1[e] += g2[e];
          s2[e] += g2[e] * (z[e] - mu[e]) * rs[e];
          scol[e] += h;
        }
        *reinterpret_cast<f4 *>(O + p * C2 + 4 * q) = g2;
      }
    }
    __syncthreads();
  }
  // ---- partials of this workgroup: T (C3 x C2), then s1, s2, S (C2 each)
  float *out = ws + (long)blockIdx.x * ws_stride;
#pragma unroll
  for (int ci = 0; ci < CPW; ++ci)
#pragma unroll
    for (int kk = 0; kk < KL; ++kk) out[(long)(wave * CPW + ci) * C2 + lane + 64 * kk] = tacc[ci][kk];
  float *red = Os;    // [3][RP][C2]
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[(0 * RP + rsub) * C2 + 4 * q + e] = s1[e];
    red[(1 * RP + rsub) * C2 + 4 * q + e] = s2[e];
    red[(2 * RP + rsub) * C2 + 4 * q + e] = scol[e];
  }
  __syncthreads();
  for (int e = tid; e < 3 * C2; e += NT) {
    const int which = e / C2, col = e - which * C2;
    float a = 0.f;
    for (int t = 0; t < RP; ++t) a += red[(which * RP + t) * C2 + col];
    out[(long)C3 * C2 + e] = a;
  }
}

// ------------------------------------------------------------------------------------------------ the fused pass
// Both kernels above in ONE pass over Z2 (read once, 268 MB at SA1; the gated gradient written once): per 64-row block
//   H tile -> LDS;  O = H An + d (matrix cores) -> LDS O tile;  Gram += H^T H (matrix cores);  T += g H[arg-max rows];
//   O rows += g s W3 (row-slot phases);  gate, layer-2 sums, column sums, store.
// NW waves: wave w owns 16 output columns of O, 16 rows of Gram (C2 = 16 NW) and C3 / NW channels of W3 / T.
template <int C2, int C3, int NW>
__global__ __launch_bounds__(NW * 64) void sa_last_fused_kernel(
    long P, long nblk, int ns, long G, float *__restrict__ O, const float *__restrict__ Z2,
    const float *__restrict__ sc2, const float *__restrict__ sh2, const float *__restrict__ mean2,
    const float *__restrict__ rstd2, const float *__restrict__ W3, const float *__restrict__ An,
    const float *__restrict__ dvec, const float *__restrict__ d_out, const float *__restrict__ zsel,
    const uint8_t *__restrict__ asel, const float *__restrict__ sc3, const float *__restrict__ sh3,
    float *__restrict__ ws_gram, float *__restrict__ ws, long ws_stride) {
  BUTD_MAIN_PRIO_SET();
  static_assert(C2 == 16 * NW, "a wave owns 16 columns");
  constexpr int NT = NW * 64;
  constexpr int ST = C2 + 36;        // H tile (matrix operand reads, see sa_last_mfma_kernel)
  constexpr int SO = C2 + 4;         // O tile
  constexpr int KG = C2 / 16;
  constexpr int GN = C2 / 16;
  constexpr int CPW = C3 / NW;
  constexpr int KL = C2 / 64;
  constexpr int QN = C2 / 4;
  constexpr int RP = NT / QN;
  constexpr int NP = kRows / RP;
  static_assert(CPW <= 64 && kRows % RP == 0, "decomposition");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *Ht = lds;                   // [64][ST]
  float *Os = Ht + kRows * ST;       // [64][SO]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lm = lane & 15, lq = lane >> 4;
  const int q = tid % QN, rsub = tid / QN;
  const int n0 = wave * 16;
  const int nc = kRows / ns;
  const int myc = wave * CPW + (lane % CPW);
  f4 areg[KG];
#pragma unroll
  for (int g = 0; g < KG; ++g) areg[g] = *reinterpret_cast<const f4 *>(An + (long)(n0 + lm) * C2 + 16 * g + 4 * lq);
  const f4 dinit = *reinterpret_cast<const f4 *>(dvec + n0 + 4 * lq);
  f4 gacc[GN];
#pragma unroll
  for (int n = 0; n < GN; ++n) gacc[n] = f4{0.f, 0.f, 0.f, 0.f};
  float w3r[CPW][KL], tacc[CPW][KL];
#pragma unroll
  for (int ci = 0; ci < CPW; ++ci) {
    const int c = wave * CPW + ci;
    const float s3 = sc3[c];
#pragma unroll
    for (int kk = 0; kk < KL; ++kk) {
      w3r[ci][kk] = s3 * W3[(long)c * C2 + lane + 64 * kk];
      tacc[ci][kk] = 0.f;
    }
  }
  const float my_sc3 = sc3[myc], my_sh3 = sh3[myc];
  f4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, scol = {0.f, 0.f, 0.f, 0.f};
  f4 zn[NP];
  float rec_g[4];
  int rec_row[4];
  auto fetch = [&](long b) {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const long p = b * kRows + rsub + ps * RP;
      zn[ps] = p < P ? *reinterpret_cast<const f4 *>(Z2 + p * C2 + 4 * q) : f4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      const long g = b * nc + gi;
      float gv = 0.f;
      int row = 0;
      if (gi < nc && g < G) {
        const float z = zsel[g * C3 + myc];
        const float dy = d_out[g * C3 + myc];
        row = gi * ns + (int)asel[g * C3 + myc];
        gv = my_sc3 * z + my_sh3 > 0.f ? dy : 0.f;
      }
      rec_g[gi] = gv;
      rec_row[gi] = row;
    }
  };
  long blk = blockIdx.x;
  if (blk < nblk) fetch(blk);
  for (; blk < nblk; blk += gridDim.x) {
    f4 zk[NP];
    {
      const f4 sc = *reinterpret_cast<const f4 *>(sc2 + 4 * q), sh = *reinterpret_cast<const f4 *>(sh2 + 4 * q);
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) {
        const int r = rsub + ps * RP;
        const bool in = blk * kRows + r < P;
        zk[ps] = zn[ps];
        f4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = in ? fmaxf(sc[e] * zn[ps][e] + sh[e], 0.f) : 0.f;
        *reinterpret_cast<f4 *>(Ht + r * ST + 4 * q) = h;
      }
    }
    float cg[4];
    int cr[4];
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      cg[gi] = rec_g[gi];
      cr[gi] = rec_row[gi];
    }
    __syncthreads();
    if (blk + gridDim.x < nblk) fetch(blk + gridDim.x);          // in flight under the work below
    // ---- O^T tiles (16 columns of this wave x 16 rows) -> LDS
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      f4 oacc = dinit;
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        const f4 hv = *reinterpret_cast<const f4 *>(Ht + (16 * rt + lm) * ST + 16 * g + 4 * lq);
#pragma unroll
        for (int i = 0; i < 4; ++i) oacc = mfma4(areg[g][i], hv[i], oacc);
      }
      *reinterpret_cast<f4 *>(Os + (16 * rt + lm) * SO + n0 + 4 * lq) = oacc;
    }
    // ---- Gram rows of this wave
#pragma unroll 4
    for (int s = 0; s < kRows / 4; ++s) {
      const float *hrow = Ht + (4 * s + lq) * ST + lm;
      const float a = hrow[n0];
      float b[GN];
#pragma unroll
      for (int n = 0; n < GN; ++n) b[n] = hrow[16 * n];
#pragma unroll
      for (int n = 0; n < GN; ++n) gacc[n] = mfma4(a, b[n], gacc[n]);
    }
    // ---- T[c,:] += g H[arg-max row,:]
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      if (gi < nc) {
#pragma unroll
        for (int ci = 0; ci < CPW; ++ci) {
          const float gv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cg[gi]), ci));
          const int ro = __builtin_amdgcn_readlane(cr[gi], ci) * ST;
#pragma unroll
          for (int kk = 0; kk < KL; ++kk) tacc[ci][kk] += gv * Ht[ro + lane + 64 * kk];
        }
      }
    }
    __syncthreads();                                             // the O tile is complete
    // ---- O[arg-max row,:] += g s W3[c,:] in NW row-slot phases (see sa_last_sparse_kernel)
    for (int ph = 0; ph < NW; ++ph) {
      const int slot = (wave + ph) % NW;
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        if (gi < nc) {
          const unsigned long long m = __ballot((cr[gi] & 63) / (kRows / NW) == slot && cg[gi] != 0.f);
          if (m != 0ull) {
#pragma unroll
            for (int ci = 0; ci < CPW; ++ci) {
              if ((m >> ci) & 1ull) {
                const float gv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cg[gi]), ci));
                const int ro = __builtin_amdgcn_readlane(cr[gi], ci) * SO;
#pragma unroll
                for (int kk = 0; kk < KL; ++kk) {
                  float *a = Os + ro + lane + 64 * kk;
                  *a = *a + gv * w3r[ci][kk];
                }
              }
            }
          }
        }
      }
      __syncthreads();
    }
    // ---- gate, layer-2 sums, column sums of H, store
    {
      const f4 mu = *reinterpret_cast<const f4 *>(mean2 + 4 * q), rs = *reinterpret_cast<const f4 *>(rstd2 + 4 * q);
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) {
        const int r = rsub + ps * RP;
        const long p = blk * kRows + r;
        if (p < P) {
          const f4 o = *reinterpret_cast<const f4 *>(Os + r * SO + 4 * q);
          const f4 h = *reinterpret_cast<const f4 *>(Ht + r * ST + 4 * q);
          f4 g2;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            g2[e] = h[e] > 0.f ? o[e] : 0.f;
            s1[e] += g2[e];
            s2[e] += g2[e] * (zk[ps][e] - mu[e]) * rs[e];
            scol[e] += h[e];
          }
          *reinterpret_cast<f4 *>(O + p * C2 + 4 * q) = g2;
        }
      }
    }
    __syncthreads();
  }
  // ---- partials of this workgroup
  float *og = ws_gram + (long)blockIdx.x * C2 * C2;
#pragma unroll
  for (int n = 0; n < GN; ++n)
#pragma unroll
    for (int i = 0; i < 4; ++i) og[(long)(n0 + 4 * lq + i) * C2 + 16 * n + lm] = gacc[n][i];
  float *out = ws + (long)blockIdx.x * ws_stride;
#pragma unroll
  for (int ci = 0; ci < CPW; ++ci)
#pragma unroll
    for (int kk = 0; kk < KL; ++kk) out[(long)(wave * CPW + ci) * C2 + lane + 64 * kk] = tacc[ci][kk];
  float *red = Os;    // [3][RP][C2]
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[(0 * RP + rsub) * C2 + 4 * q + e] = s1[e];
    red[(1 * RP + rsub) * C2 + 4 * q + e] = s2[e];
    red[(2 * RP + rsub) * C2 + 4 * q + e] = scol[e];
  }
  __syncthreads();
  for (int e = tid; e < 3 * C2; e += NT) {
    const int which = e / C2, col = e - which * C2;
    float a = 0.f;
    for (int t = 0; t < RP; ++t) a += red[(which * RP + t) * C2 + col];
    out[(long)C3 * C2 + e] = a;
  }
}

// (A one-pass version for the 128-wide levels -- a workgroup per 64-row block AND column half, 244 registers, one
//  workgroup per CU -- was built and measured: 561 us at SA2 against 434 us for the two kernels above; with a single
//  workgroup per CU nothing overlaps its eleven barrier-separated phases per block.  profiles/r04_sa_last_layer.txt.)

// ----------------------------------------------------------------------------------------------- partials -> totals
// tot[n] = sum over parts of part[w][n] (double), for two groups of partials laid one after the other in tot.  A block
// = 16 elements x 16 part-lanes: a thread sums every 16th partial, the 16 sums fold in LDS in a fixed order.
__global__ __launch_bounds__(256) void sa_last_reduce_kernel(const float *__restrict__ p1, long n1, int parts1,
                                                             const float *__restrict__ p2, long n2, int parts2,
                                                             double *__restrict__ tot) {
  BUTD_MAIN_PRIO_SET();
  __shared__ double red[16][17];
  const long i = (long)blockIdx.x * 16 + (threadIdx.x & 15);
  const int pl = threadIdx.x >> 4;
  const float *src = nullptr;
  long n = 0, j = 0;
  int parts = 0;
  if (i < n1) {
    src = p1; n = n1; j = i; parts = parts1;
  } else if (i < n1 + n2) {
    src = p2; n = n2; j = i - n1; parts = parts2;
  }
  double a = 0.0;
  if (src) {
    int w = pl;
    for (; w + 48 < parts; w += 64) {
      const float v0 = src[(long)w * n + j], v1 = src[(long)(w + 16) * n + j], v2 = src[(long)(w + 32) * n + j],
                  v3 = src[(long)(w + 48) * n + j];
      a += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
    }
    for (; w < parts; w += 16) a += (double)src[(long)w * n + j];
  }
  red[pl][threadIdx.x & 15] = a;
  __syncthreads();
  if (threadIdx.x < 16 && src) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
    tot[i] = t;
  }
}

// ------------------------------------------------------------------------------------------------------- dW3, sums
// grid C3 x block C2.  tot = [Gram C2*C2 | T C3*C2 | s1 C2 | s2 C2 | S C2] (double).  The centring uses mu' = W3 S / P
// (the batch mean of Z3 = H W3^T written through the same sums as Gram: the covariance W3 (Gram - S S^T / P) is then
// centred exactly).
__global__ void sa_last_dw_kernel(int C2, int C3, long P, const float *__restrict__ W3, const float *__restrict__ scale3,
                                  const float *__restrict__ rstd3, const double *__restrict__ S1_3,
                                  const double *__restrict__ S2_3, const double *__restrict__ tot,
                                  float *__restrict__ dW3, double *__restrict__ S1_2, double *__restrict__ S2_2) {
  BUTD_MAIN_PRIO_SET();
  const int c = blockIdx.x, k = threadIdx.x;
  const double *gram = tot, *T = tot + (long)C2 * C2, *s1 = T + (long)C3 * C2, *s2 = s1 + C2, *S = s2 + C2;
  const double invP = 1.0 / (double)P;
  double wg = 0.0, ws = 0.0;
#pragma unroll 16
  for (int j = 0; j < C2; ++j) {
    const double w = (double)W3[(long)c * C2 + j];
    wg += w * gram[(long)j * C2 + k];
    ws += w * S[j];
  }
  const double mu = ws * invP, m1 = S1_3[c] * invP, m2 = S2_3[c] * invP;
  const double v = T[(long)c * C2 + k] - m1 * S[k] - m2 * (double)rstd3[c] * (wg - mu * S[k]);
  dW3[(long)c * C2 + k] = (float)((double)scale3[c] * v);
  if (c == 0) {
    S1_2[k] = s1[k];
    S2_2[k] = s2[k];
  }
}

template <int C2>
constexpr size_t fused_lds() { return (size_t)(kRows * (C2 + 36) + kRows * (C2 + 4)) * sizeof(float); }

// ---------------------------------------------------------------- the FIRST layer when no input gradient is wanted
// SA1's input features carry no gradient (bdetr.py:151: xyz + colour), so layer 1's backward is its weight gradient
// alone, and that too is linear in the three terms of the BatchNorm backward:
//   dW1 = dZ1^T X = s (GX - m1 SX^T - m2 rstd (W1 XX - mu SX^T)),   GX = g^T X,  SX = column sums of X,  XX = X^T X
// with g = dH1 gated by layer 1's ReLU and X the grouped input (P x 8).  One streaming pass over (dH1, Z1, X) yields
// S1, S2 (what butd_sa_mask_stats did), GX (C1 x 8), SX, XX as per-workgroup partials; butd_sa_dz_mid's pass over the
// 10^6-row tensor and the thin weight-gradient product (64 x 8 x 10^6: 113 us) are not run at all.
constexpr int kFirstChunk = 1024;
template <int KP>
__global__ __launch_bounds__(kThreads) void sa_first_stats_kernel(
    long P, int C, const float *__restrict__ dH, const float *__restrict__ Z, const float *__restrict__ X,
    const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ mean,
    const float *__restrict__ rstd, float *__restrict__ ws, long ws_stride) {
  BUTD_MAIN_PRIO_SET();
  static_assert(KP == 8, "grouped xyz + colour, padded to 8 columns");
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [tpg][C][KP + 2] then the X sums
  const int c4n = C >> 2, tpg = kThreads / c4n;
  const int cq = threadIdx.x % c4n, sub = threadIdx.x / c4n;
  const long row0 = (long)blockIdx.x * kFirstChunk;
  const long rows = min((long)kFirstChunk, P - row0);
  const f4 sc = *reinterpret_cast<const f4 *>(scale + 4 * cq), sh = *reinterpret_cast<const f4 *>(shift + 4 * cq);
  const f4 mu = *reinterpret_cast<const f4 *>(mean + 4 * cq), rs = *reinterpret_cast<const f4 *>(rstd + 4 * cq);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  float gx[4][KP];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int k = 0; k < KP; ++k) gx[e][k] = 0.f;
  float sxr = 0.f, xxr[KP];            // threads cq < KP of every row phase: row cq of X^T X and SX[cq]
#pragma unroll
  for (int k = 0; k < KP; ++k) xxr[k] = 0.f;
  for (long r = sub; r < rows; r += tpg) {
    const long p = row0 + r;
    const f4 z = *reinterpret_cast<const f4 *>(Z + p * C + 4 * cq);
    const f4 d = *reinterpret_cast<const f4 *>(dH + p * C + 4 * cq);
    const f4 xa = *reinterpret_cast<const f4 *>(X + p * KP), xb = *reinterpret_cast<const f4 *>(X + p * KP + 4);
    const float x[KP] = {xa[0], xa[1], xa[2], xa[3], xb[0], xb[1], xb[2], xb[3]};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float g = sc[e] * z[e] + sh[e] > 0.f ? d[e] : 0.f;
      s1[e] += g;
      s2[e] += g * (z[e] - mu[e]) * rs[e];
#pragma unroll
      for (int k = 0; k < KP; ++k) gx[e][k] += g * x[k];
    }
    if (cq < KP) {
      const float xc = cq < 4 ? xa[cq & 3] : xb[cq & 3];
      sxr += xc;
#pragma unroll
      for (int k2 = 0; k2 < KP; ++k2) xxr[k2] += xc * x[k2];
    }
  }
  // fold the tpg row phases in LDS, fixed order
  constexpr int W = KP + 2;
  float *red = lds;                                // [tpg][C][W]: gx[0..KP), s1, s2
  float *redx = lds + (long)tpg * C * W;           // [tpg][KP + KP*KP]
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float *o = red + ((long)sub * C + 4 * cq + e) * W;
#pragma unroll
    for (int k = 0; k < KP; ++k) o[k] = gx[e][k];
    o[KP] = s1[e];
    o[KP + 1] = s2[e];
  }
  if (cq < KP) {
    float *o = redx + (long)sub * (KP + KP * KP);
    o[cq] = sxr;
#pragma unroll
    for (int k2 = 0; k2 < KP; ++k2) o[KP + cq * KP + k2] = xxr[k2];
  }
  __syncthreads();
  float *out = ws + (long)blockIdx.x * ws_stride;  // [C][W] then [KP + KP*KP]
  for (int i = threadIdx.x; i < C * W; i += kThreads) {
    float a = 0.f;
    for (int t = 0; t < tpg; ++t) a += red[(long)t * C * W + i];
    out[i] = a;
  }
  for (int i = threadIdx.x; i < KP + KP * KP; i += kThreads) {
    float a = 0.f;
    for (int t = 0; t < tpg; ++t) a += redx[(long)t * (KP + KP * KP) + i];
    out[C * W + i] = a;
  }
}

// grid C x block KP: dW1[c][k]; tot = [C][KP + 2] then SX [KP], XX [KP][KP] (double).  Also hands S1, S2 out.
template <int KP>
__global__ void sa_first_dw_kernel(int C, long P, int ldw, const float *__restrict__ W1, const float *__restrict__ scale,
                                   const float *__restrict__ rstd, const double *__restrict__ tot,
                                   float *__restrict__ dW1, double *__restrict__ S1, double *__restrict__ S2) {
  BUTD_MAIN_PRIO_SET();
  constexpr int W = KP + 2;
  const int c = blockIdx.x, k = threadIdx.x;
  const double *row = tot + (long)c * W, *SX = tot + (long)C * W, *XX = SX + KP;
  const double invP = 1.0 / (double)P;
  const double m1 = row[KP] * invP, m2 = row[KP + 1] * invP;
  double wx = 0.0, wsx = 0.0;
#pragma unroll
  for (int j = 0; j < KP; ++j) {
    const double w = (double)W1[(long)c * ldw + j];
    wx += w * XX[j * KP + k];
    wsx += w * SX[j];
  }
  const double v = row[k] - m1 * SX[k] - m2 * (double)rstd[c] * (wx - wsx * invP * SX[k]);
  dW1[(long)c * ldw + k] = (float)((double)scale[c] * v);
  if (k == 0) {
    S1[c] = row[KP];
    S2[c] = row[KP + 1];
  }
}

// ------------------------------------------------------------- SA1: the first two layers forward without writing Z1
// Layer 1's input is the grouped (xyz, colour) row X (8 columns).  Z1 = X W1^T is linear in it, so its BatchNorm batch
// statistics follow from the moments of X alone: sum_c = W1[c] . SX,  sumsq_c = W1[c]^T XX W1[c]  (SX = column sums, XX =
// X^T X: 8 + 64 numbers, double).  No pass over a 10^6 x 64 tensor is needed to normalise layer 1 -- and Z1 need not
// exist at all: the second layer's kernel forms z1 -> relu(bn(z1)) from the X tile on the fly (8 multiply-adds per
// element), multiplies by W2 on the matrix cores, writes Z2 and takes its column sums in the same pass; the backward
// (sa_mid_first_kernel<RECOMP>) recomputes z1 the same way.  Replaces butd_sa_thin_conv (writes Z1), the layer-2 product
// launch (reads it) and butd_sa_colstats (reads Z2 back): 1.1 GB -> 0.3 GB at SA1, B = 8.
__global__ __launch_bounds__(256) void sa_x_moments_kernel(long P, const float *__restrict__ X, double *__restrict__ mom) {
  BUTD_MAIN_PRIO_SET();
  // thread = (row lane t / 8, column k = t % 8): SX[k] and row k of XX over its rows; mom = [SX 8 | XX 64], zero on entry
  __shared__ float red[32][72];
  const int k = threadIdx.x & 7, rl = threadIdx.x >> 3;
  float sx = 0.f, xx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long r = (long)blockIdx.x * 32 + rl; r < P; r += (long)gridDim.x * 32) {
    const f4 a = *reinterpret_cast<const f4 *>(X + r * 8), b = *reinterpret_cast<const f4 *>(X + r * 8 + 4);
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    const float xk = k < 4 ? a[k & 3] : b[k & 3];
    sx += xk;
#pragma unroll
    for (int j = 0; j < 8; ++j) xx[j] += xk * x[j];
  }
  red[rl][k] = sx;
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl][8 + k * 8 + j] = xx[j];
  __syncthreads();
  if (threadIdx.x < 72) {
    double a = 0.0;
    for (int t = 0; t < 32; ++t) a += (double)red[t][threadIdx.x];
    atomicAdd(mom + threadIdx.x, a);
  }
}

__global__ void sa_l1_stats_kernel(int C, const float *__restrict__ W1, const double *__restrict__ mom,
                                   double *__restrict__ sum, double *__restrict__ sumsq) {
  BUTD_MAIN_PRIO_SET();
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int k = 0; k < 8; ++k) {
    const double wk = (double)W1[c * 8 + k];
    s += wk * mom[k];
    for (int j = 0; j < 8; ++j) q += wk * (double)W1[c * 8 + j] * mom[8 + k * 8 + j];
  }
  sum[c] = s;
  sumsq[c] = q;
}

template <int C>
__global__ __launch_bounds__(256) void sa_l12_fwd_kernel(long P, long nblk, const float *__restrict__ X,
                                                         const float *__restrict__ W1, const float *__restrict__ sc1,
                                                         const float *__restrict__ sh1, const float *__restrict__ W2,
                                                         float *__restrict__ Z2, double *__restrict__ sum,
                                                         double *__restrict__ sumsq) {
  BUTD_MAIN_PRIO_SET();
  static_assert(C == 64, "SA1");
  constexpr int KP = 8, ST = C + 36, SO = C + 4, KG = C / 16, QN = C / 4, RP = 256 / QN, NP = kRows / RP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *Ht = lds;                   // [64][ST]
  float *Zs = Ht + kRows * ST;       // [64][SO]
  float *Xs = Zs + kRows * SO;       // [64][KP]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lm = lane & 15, lq = lane >> 4;
  const int q = tid % QN, rsub = tid / QN, n0 = wave * 16;
  // z1^T tile = W1 (A operand: this wave's 16 columns, K = 8: two instructions) x X^T -- sa_mid_first_kernel<RECOMP>
  // repeats exactly these instructions
  const float w1a[2] = {W1[(n0 + lm) * KP + lq], W1[(n0 + lm) * KP + 4 + lq]};
  const f4 scm = *reinterpret_cast<const f4 *>(sc1 + n0 + 4 * lq), shm = *reinterpret_cast<const f4 *>(sh1 + n0 + 4 * lq);
  f4 areg[KG];                       // A operand of the transposed second product: W2 rows = output columns of this wave
#pragma unroll
  for (int g = 0; g < KG; ++g) areg[g] = *reinterpret_cast<const f4 *>(W2 + (long)(n0 + lm) * C + 16 * g + 4 * lq);
  f4 s = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f};
  float xn = 0.f, xn2 = 0.f;
  auto fetch = [&](long b) {
    const long e0 = b * kRows * KP + tid, e1 = e0 + 256;            // the block's 64 x 8 inputs: two per thread
    xn = e0 < P * KP ? X[e0] : 0.f;
    xn2 = e1 < P * KP ? X[e1] : 0.f;
  };
  long blk = blockIdx.x;
  if (blk < nblk) fetch(blk);
  for (; blk < nblk; blk += gridDim.x) {
    Xs[tid] = xn;
    Xs[tid + 256] = xn2;
    __syncthreads();
    if (blk + gridDim.x < nblk) fetch(blk + gridDim.x);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      f4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) z = mfma4(w1a[kk], Xs[(16 * rt + lm) * KP + 4 * kk + lq], z);
      const bool in = blk * kRows + 16 * rt + lm < P;
      f4 h;
#pragma unroll
      for (int i = 0; i < 4; ++i) h[i] = in ? fmaxf(scm[i] * z[i] + shm[i], 0.f) : 0.f;
      *reinterpret_cast<f4 *>(Ht + (16 * rt + lm) * ST + n0 + 4 * lq) = h;
    }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      f4 oacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        const f4 hv = *reinterpret_cast<const f4 *>(Ht + (16 * rt + lm) * ST + 16 * g + 4 * lq);
#pragma unroll
        for (int i = 0; i < 4; ++i) oacc = mfma4(areg[g][i], hv[i], oacc);
      }
      *reinterpret_cast<f4 *>(Zs + (16 * rt + lm) * SO + n0 + 4 * lq) = oacc;
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int r = rsub + ps * RP;
      const long p = blk * kRows + r;
      if (p < P) {
        const f4 z = *reinterpret_cast<const f4 *>(Zs + r * SO + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[e] += z[e];
          sq[e] += z[e] * z[e];
        }
        *reinterpret_cast<f4 *>(Z2 + p * C + 4 * q) = z;
      }
    }
    // (no barrier here: the next block's X tile is its own buffer, H is rewritten after the next barrier, Z after two)
  }
  __syncthreads();
  float *red = lds;                  // [2][RP][C]
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[(0 * RP + rsub) * C + 4 * q + e] = s[e];
    red[(1 * RP + rsub) * C + 4 * q + e] = sq[e];
  }
  __syncthreads();
  if (tid < 2 * C) {
    const int which = tid / C, col = tid - which * C;
    double a = 0.0;
    for (int t = 0; t < RP; ++t) a += (double)red[(which * RP + t) * C + col];
    atomicAdd((which ? sumsq : sum) + col, a);
  }
}

// ------------------------------------------- SA1: layer 2 and layer 1 backward in one pass, nothing written per row
// With no input gradient wanted (SA1) everything below layer 2's gated gradient g2 ends in reductions:
//   dZ2 = s2 (g2 - m1 - zhat2 m2)          (m1, m2: the sums butd_sa_last_bwd left behind)
//   dW2 = dZ2^T H1                          (64 x 64, matrix cores)
//   dH1 = dZ2 W2  ->  g1 = dH1 gated by layer 1's ReLU  ->  S1, S2 of layer 1, GX = g1^T X (64 x 8), SX, XX
// so one pass over (g2, Z2, Z1, X) per 64-row block -- dZ2 tile and H1 tile in LDS, both products on the matrix cores,
// the fold of g1 per thread -- replaces butd_sa_dz_mid (writes dZ2), the weight- / input-gradient product pair (reads
// it twice, writes dH1) and the sa_first_stats pass (reads dH1): 2.45 GB -> 0.84 GB at SA1, B = 8.
// RECOMP: Z1 was never written (butd_sa_first_two_fwd): it is recomputed from the X tile and W1 on the matrix cores
// (K = 8: two instructions per 16 x 16 tile), with the instruction sequence of the forward kernel: the same bits.
template <int C, bool RECOMP>
__global__ __launch_bounds__(256, 2) void sa_mid_first_kernel(
    long P, long nblk, const float *__restrict__ G2, const float *__restrict__ Z2, const float *__restrict__ Z1,
    const float *__restrict__ X, const float *__restrict__ gamma2, const float *__restrict__ sc2,
    const float *__restrict__ sh2, const float *__restrict__ mean2, const float *__restrict__ rstd2,
    const double *__restrict__ S1_2, const double *__restrict__ S2_2, const float *__restrict__ sc1,
    const float *__restrict__ sh1, const float *__restrict__ mean1, const float *__restrict__ rstd1,
    const float *__restrict__ W2, const float *__restrict__ W1, float *__restrict__ ws_w, float *__restrict__ ws,
    long ws_stride) {
  BUTD_MAIN_PRIO_SET();
  static_assert(C == 64, "SA1: 64-wide layers");
  constexpr int KP = 8;
  constexpr int ST = C + 36, SO = C + 4, KG = C / 16, GN = C / 16, QN = C / 4, RP = 256 / QN, NP = kRows / RP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *Dt = lds;                   // [64][ST]  dZ2
  float *Ht = Dt + kRows * ST;       // [64][ST]  H1
  float *Os = Ht + kRows * ST;       // [64][SO]  dH1
  float *Xs = Os + kRows * SO;       // [64][KP]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lm = lane & 15, lq = lane >> 4;
  const int q = tid % QN, rsub = tid / QN;
  const int n0 = wave * 16;
  float w1a[2] = {0.f, 0.f};         // RECOMP: A operand of z1^T = W1 X^T: W1[column n0 + lm][4 kk + lq]
  f4 scm = {0.f, 0.f, 0.f, 0.f}, shm = {0.f, 0.f, 0.f, 0.f};      // layer 1's scale / shift of columns n0 + 4 lq + i
  if (RECOMP) {
    w1a[0] = W1[(n0 + lm) * KP + lq];
    w1a[1] = W1[(n0 + lm) * KP + 4 + lq];
    scm = *reinterpret_cast<const f4 *>(sc1 + n0 + 4 * lq);
    shm = *reinterpret_cast<const f4 *>(sh1 + n0 + 4 * lq);
  }
  // dH1^T tile = W2^T (A operand: rows = columns k of W2, contraction over c) x dZ2^T
  f4 areg[KG];
#pragma unroll
  for (int g = 0; g < KG; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) areg[g][i] = W2[(long)(16 * g + 4 * lq + i) * C + n0 + lm];
  f4 wacc[GN];                       // dW2 rows n0 .. n0 + 16 (channels c of layer 2), all 64 columns
#pragma unroll
  for (int n = 0; n < GN; ++n) wacc[n] = f4{0.f, 0.f, 0.f, 0.f};
  const double invP = 1.0 / (double)P;
  f4 ga, s2c, h2c, mu2, rs2, a1, a2, s1c, h1c, mu1, rs1;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * q + e;
    ga[e] = gamma2[c]; s2c[e] = sc2[c]; h2c[e] = sh2[c]; mu2[e] = mean2[c]; rs2[e] = rstd2[c];
    a1[e] = (float)(S1_2[c] * invP); a2[e] = (float)(S2_2[c] * invP);
    s1c[e] = sc1[c]; h1c[e] = sh1[c]; mu1[e] = mean1[c]; rs1[e] = rstd1[c];
  }
  float gx[4][KP];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int k = 0; k < KP; ++k) gx[e][k] = 0.f;
  f4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  float sxr = 0.f, xxr[KP];          // threads q < KP: row q of X^T X and SX[q]
#pragma unroll
  for (int k = 0; k < KP; ++k) xxr[k] = 0.f;
  f4 gn[NP], z2n[NP], z1n[NP];
  float xn = 0.f;                    // thread t < 128: X element (row t / 8 ... two rounds)
  float xn2 = 0.f;
  auto fetch = [&](long b) {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const long p = b * kRows + rsub + ps * RP;
      const bool in = p < P;
      gn[ps] = in ? *reinterpret_cast<const f4 *>(G2 + p * C + 4 * q) : f4{0.f, 0.f, 0.f, 0.f};
      z2n[ps] = in ? *reinterpret_cast<const f4 *>(Z2 + p * C + 4 * q) : f4{0.f, 0.f, 0.f, 0.f};
      if (!RECOMP) z1n[ps] = in ? *reinterpret_cast<const f4 *>(Z1 + p * C + 4 * q) : f4{0.f, 0.f, 0.f, 0.f};
    }
    const long e0 = b * kRows * KP + tid, e1 = e0 + 256;          // the block's 64 x 8 inputs: two per thread
    xn = e0 < P * KP ? X[e0] : 0.f;
    xn2 = e1 < P * KP ? X[e1] : 0.f;
  };
  long blk = blockIdx.x;
  if (blk < nblk) fetch(blk);
  for (; blk < nblk; blk += gridDim.x) {
    f4 zk1[NP];
    Xs[tid] = xn;
    Xs[tid + 256] = xn2;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int r = rsub + ps * RP;
      const bool in = blk * kRows + r < P;
      if (!RECOMP) zk1[ps] = z1n[ps];
      f4 d, h;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gv = s2c[e] * z2n[ps][e] + h2c[e] > 0.f ? gn[ps][e] : 0.f;      // (idempotent: g2 arrives gated)
        d[e] = in ? ga[e] * rs2[e] * (gv - a1[e] - (z2n[ps][e] - mu2[e]) * rs2[e] * a2[e]) : 0.f;
        if (!RECOMP) h[e] = in ? fmaxf(s1c[e] * zk1[ps][e] + h1c[e], 0.f) : 0.f;
      }
      *reinterpret_cast<f4 *>(Dt + r * ST + 4 * q) = d;
      if (!RECOMP) *reinterpret_cast<f4 *>(Ht + r * ST + 4 * q) = h;
    }
    if (RECOMP) {
      __syncthreads();                 // the X tile
      // z1^T tiles = W1 (A operand: this wave's 16 columns) x X^T on the matrix cores (K = 8: two instructions), the
      // BatchNorm + ReLU of layer 1 in the accumulator registers, straight into the H1 tile
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        f4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) z = mfma4(w1a[kk], Xs[(16 * rt + lm) * KP + 4 * kk + lq], z);
        const bool in = blk * kRows + 16 * rt + lm < P;
        f4 h;
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = in ? fmaxf(scm[i] * z[i] + shm[i], 0.f) : 0.f;
        *reinterpret_cast<f4 *>(Ht + (16 * rt + lm) * ST + n0 + 4 * lq) = h;
      }
    }
    __syncthreads();
    if (blk + gridDim.x < nblk) fetch(blk + gridDim.x);
    // ---- dH1^T tiles (16 columns of this wave x 16 rows) -> LDS
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      f4 oacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        const f4 dv = *reinterpret_cast<const f4 *>(Dt + (16 * rt + lm) * ST + 16 * g + 4 * lq);
#pragma unroll
        for (int i = 0; i < 4; ++i) oacc = mfma4(areg[g][i], dv[i], oacc);
      }
      *reinterpret_cast<f4 *>(Os + (16 * rt + lm) * SO + n0 + 4 * lq) = oacc;
    }
    // ---- dW2[c = n0 + .., k] += sum_r dZ2[r, c] H1[r, k]
#pragma unroll 4
    for (int s = 0; s < kRows / 4; ++s) {
      const float a = Dt[(4 * s + lq) * ST + n0 + lm];
      const float *hrow = Ht + (4 * s + lq) * ST + lm;
      float b[GN];
#pragma unroll
      for (int n = 0; n < GN; ++n) b[n] = hrow[16 * n];
#pragma unroll
      for (int n = 0; n < GN; ++n) wacc[n] = mfma4(a, b[n], wacc[n]);
    }
    __syncthreads();
    if (RECOMP) {                      // the dZ2 tile is spent: raw z1 (the same instructions, the same bits) takes its place
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        f4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) z = mfma4(w1a[kk], Xs[(16 * rt + lm) * KP + 4 * kk + lq], z);
        *reinterpret_cast<f4 *>(Dt + (16 * rt + lm) * ST + n0 + 4 * lq) = z;
      }
      __syncthreads();
    }
    // ---- g1 = dH1 gated; layer-1 sums; GX += g1^T X; SX, XX
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int r = rsub + ps * RP;
      if (RECOMP) zk1[ps] = *reinterpret_cast<const f4 *>(Dt + r * ST + 4 * q);
      if (blk * kRows + r < P) {
        const f4 o = *reinterpret_cast<const f4 *>(Os + r * SO + 4 * q);
        const f4 xa = *reinterpret_cast<const f4 *>(Xs + r * KP), xb = *reinterpret_cast<const f4 *>(Xs + r * KP + 4);
        const float x[KP] = {xa[0], xa[1], xa[2], xa[3], xb[0], xb[1], xb[2], xb[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g = s1c[e] * zk1[ps][e] + h1c[e] > 0.f ? o[e] : 0.f;
          s1[e] += g;
          s2[e] += g * (zk1[ps][e] - mu1[e]) * rs1[e];
#pragma unroll
          for (int k = 0; k < KP; ++k) gx[e][k] += g * x[k];
        }
        if (q < KP) {
          const float xc = q < 4 ? xa[q & 3] : xb[q & 3];
          sxr += xc;
#pragma unroll
          for (int k2 = 0; k2 < KP; ++k2) xxr[k2] += xc * x[k2];
        }
      }
    }
    __syncthreads();
  }
  // ---- partials: dW2 (C x C) per workgroup; then the first-layer block [C][KP + 2] | SX | XX (layout of sa_first_stats)
  float *ow = ws_w + (long)blockIdx.x * C * C;
#pragma unroll
  for (int n = 0; n < GN; ++n)
#pragma unroll
    for (int i = 0; i < 4; ++i) ow[(long)(n0 + 4 * lq + i) * C + 16 * n + lm] = wacc[n][i];
  constexpr int W = KP + 2;
  float *red = lds;                                  // [RP][C][W] = 16 * 64 * 10 floats = 40 KB (Dt + Ht hold 51 KB)
  float *redx = red + (long)RP * C * W;              // [RP][KP + KP*KP]
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float *o = red + ((long)rsub * C + 4 * q + e) * W;
#pragma unroll
    for (int k = 0; k < KP; ++k) o[k] = gx[e][k];
    o[KP] = s1[e];
    o[KP + 1] = s2[e];
  }
  if (q < KP) {
    float *o = redx + (long)rsub * (KP + KP * KP);
    o[q] = sxr;
#pragma unroll
    for (int k2 = 0; k2 < KP; ++k2) o[KP + q * KP + k2] = xxr[k2];
  }
  __syncthreads();
  float *out = ws + (long)blockIdx.x * ws_stride;
  for (int i = tid; i < C * W; i += 256) {
    float a = 0.f;
    for (int t = 0; t < RP; ++t) a += red[(long)t * C * W + i];
    out[i] = a;
  }
  for (int i = tid; i < KP + KP * KP; i += 256) {
    float a = 0.f;
    for (int t = 0; t < RP; ++t) a += redx[(long)t * (KP + KP * KP) + i];
    out[C * W + i] = a;
  }
}

// ------------------------------------ SA2-4 (128-wide layers): layer 2's backward in one pass, only the gated gradient written
// Round 5.  The levels with an input gradient keep layer 1's backward as it is, but everything between layer 2's gated
// gradient g2 (what butd_sa_last_bwd wrote) and layer 1's gated gradient g1 happens per 32-row block without a dense dZ2 or
// an ungated dH1 in memory:
//   dZ2 = gamma2 rstd2 (g2 - S1_2/P - zhat2 S2_2/P)      -> LDS tile
//   H1  = relu(scale1 Z1 + shift1)                         -> LDS tile
//   dW2 += dZ2^T H1                                        (matrix cores; a wave owns 16 channels x all 128 columns)
//   dH1 = dZ2 W2 -> g1 = dH1 gated by layer 1's ReLU       (matrix cores; W2^T fragments in registers) -> WRITTEN
//   S1_1 += g1, S2_1 += g1 zhat1                           (per thread, per-workgroup partials, summed in double)
// Replaces butd_sa_dz_mid (reads g2, Z2, writes dZ2), the layer's weight- / input-gradient product pair (reads dZ2 twice
// and Z1 twice, writes dH1) and the statistics in that product's epilogue: 958 MB -> 537 MB at SA2, B = 8.
// 512 threads = 8 waves (a wave: 16 of the 128 columns / channels), 59 KB of LDS: two workgroups per CU.
constexpr int kRowsW = 32;
template <int C>
__global__ __launch_bounds__(512, 2) void sa_mid_wide_kernel(
    long P, long nblk, const float *__restrict__ G2, const float *__restrict__ Z2, const float *__restrict__ Z1,
    const float *__restrict__ gamma2, const float *__restrict__ sc2, const float *__restrict__ sh2,
    const float *__restrict__ mean2, const float *__restrict__ rstd2, const double *__restrict__ S1_2,
    const double *__restrict__ S2_2, const float *__restrict__ sc1, const float *__restrict__ sh1,
    const float *__restrict__ mean1, const float *__restrict__ rstd1, const float *__restrict__ W2,
    float *__restrict__ G1, float *__restrict__ ws_w, float *__restrict__ ws) {
  BUTD_MAIN_PRIO_SET();
  static_assert(C == 128, "SA2-4: 128-wide layers");
  constexpr int NT = 512;
  constexpr int ST = C + 36, SO = C + 4, KG = C / 16, GN = C / 16, QN = C / 4, RP = NT / QN, NP = kRowsW / RP;
  static_assert(NP * RP == kRowsW, "row passes");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *Dt = lds;                    // [32][ST]  dZ2
  float *Ht = Dt + kRowsW * ST;       // [32][ST]  H1
  float *Os = Ht + kRowsW * ST;       // [32][SO]  dH1
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lm = lane & 15, lq = lane >> 4;
  const int q = tid % QN, rsub = tid / QN;
  const int n0 = wave * 16;
  // dH1^T tile = W2^T (A operand: rows = columns k of W2, contraction over c) x dZ2^T
  f4 areg[KG];
#pragma unroll
  for (int g = 0; g < KG; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) areg[g][i] = W2[(long)(16 * g + 4 * lq + i) * C + n0 + lm];
  f4 wacc[GN];                        // dW2 rows n0 .. n0 + 16 (channels c of layer 2), all 128 columns
#pragma unroll
  for (int n = 0; n < GN; ++n) wacc[n] = f4{0.f, 0.f, 0.f, 0.f};
  const double invP = 1.0 / (double)P;
  f4 ga, s2c, h2c, mu2, rs2, a1, a2, s1c, h1c, mu1, rs1;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * q + e;
    ga[e] = gamma2[c]; s2c[e] = sc2[c]; h2c[e] = sh2[c]; mu2[e] = mean2[c]; rs2[e] = rstd2[c];
    a1[e] = (float)(S1_2[c] * invP); a2[e] = (float)(S2_2[c] * invP);
    s1c[e] = sc1[c]; h1c[e] = sh1[c]; mu1[e] = mean1[c]; rs1[e] = rstd1[c];
  }
  f4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  f4 gn[NP], z2n[NP], z1n[NP];
  auto fetch = [&](long b) {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const long p = b * kRowsW + rsub + ps * RP;
      const bool in = p < P;
      gn[ps] = in ? *reinterpret_cast<const f4 *>(G2 + p * C + 4 * q) : f4{0.f, 0.f, 0.f, 0.f};
      z2n[ps] = in ? *reinterpret_cast<const f4 *>(Z2 + p * C + 4 * q) : f4{0.f, 0.f, 0.f, 0.f};
      z1n[ps] = in ? *reinterpret_cast<const f4 *>(Z1 + p * C + 4 * q) : f4{0.f, 0.f, 0.f, 0.f};
    }
  };
  long blk = blockIdx.x;
  if (blk < nblk) fetch(blk);
  for (; blk < nblk; blk += gridDim.x) {
    f4 zk1[NP];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int r = rsub + ps * RP;
      const bool in = blk * kRowsW + r < P;
      zk1[ps] = z1n[ps];
      f4 d, h;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gv = s2c[e] * z2n[ps][e] + h2c[e] > 0.f ? gn[ps][e] : 0.f;      // (idempotent: g2 arrives gated)
        d[e] = in ? ga[e] * rs2[e] * (gv - a1[e] - (z2n[ps][e] - mu2[e]) * rs2[e] * a2[e]) : 0.f;
        h[e] = in ? fmaxf(s1c[e] * zk1[ps][e] + h1c[e], 0.f) : 0.f;
      }
      *reinterpret_cast<f4 *>(Dt + r * ST + 4 * q) = d;
      *reinterpret_cast<f4 *>(Ht + r * ST + 4 * q) = h;
    }
    __syncthreads();
    if (blk + gridDim.x < nblk) fetch(blk + gridDim.x);
    // ---- dH1^T tiles (16 columns of this wave x 16 rows) -> LDS
#pragma unroll
    for (int rt = 0; rt < kRowsW / 16; ++rt) {
      f4 oacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        const f4 dv = *reinterpret_cast<const f4 *>(Dt + (16 * rt + lm) * ST + 16 * g + 4 * lq);
#pragma unroll
        for (int i = 0; i < 4; ++i) oacc = mfma4(areg[g][i], dv[i], oacc);
      }
      *reinterpret_cast<f4 *>(Os + (16 * rt + lm) * SO + n0 + 4 * lq) = oacc;
    }
    // ---- dW2[c = n0 + .., k] += sum_r dZ2[r, c] H1[r, k]
#pragma unroll 4
    for (int s = 0; s < kRowsW / 4; ++s) {
      const float a = Dt[(4 * s + lq) * ST + n0 + lm];
      const float *hrow = Ht + (4 * s + lq) * ST + lm;
      float b[GN];
#pragma unroll
      for (int n = 0; n < GN; ++n) b[n] = hrow[16 * n];
#pragma unroll
      for (int n = 0; n < GN; ++n) wacc[n] = mfma4(a, b[n], wacc[n]);
    }
    __syncthreads();
    // ---- g1 = dH1 gated -> memory; layer-1 sums
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int r = rsub + ps * RP;
      const long p = blk * kRowsW + r;
      if (p < P) {
        const f4 o = *reinterpret_cast<const f4 *>(Os + r * SO + 4 * q);
        f4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          g[e] = s1c[e] * zk1[ps][e] + h1c[e] > 0.f ? o[e] : 0.f;
          s1[e] += g[e];
          s2[e] += g[e] * (zk1[ps][e] - mu1[e]) * rs1[e];
        }
        *reinterpret_cast<f4 *>(G1 + p * C + 4 * q) = g;
      }
    }
    // (the next block's tiles are written after this point only by threads that have left the loop body above; its
    //  matrix phase starts behind the next barrier: Os is not touched before every thread is through its gating pass)
  }
  __syncthreads();
  // ---- partials: dW2 (C x C) per workgroup, then [S1 | S2] (2 C)
  float *ow = ws_w + (long)blockIdx.x * C * C;
#pragma unroll
  for (int n = 0; n < GN; ++n)
#pragma unroll
    for (int i = 0; i < 4; ++i) ow[(long)(n0 + 4 * lq + i) * C + 16 * n + lm] = wacc[n][i];
  float *red = lds;                                  // [RP][2][C] = 16 * 256 floats
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[(rsub * 2 + 0) * C + 4 * q + e] = s1[e];
    red[(rsub * 2 + 1) * C + 4 * q + e] = s2[e];
  }
  __syncthreads();
  float *out = ws + (long)blockIdx.x * 2 * C;
  for (int i = tid; i < 2 * C; i += NT) {
    float a = 0.f;
    for (int t = 0; t < RP; ++t) a += red[(long)t * 2 * C + i];
    out[i] = a;
  }
}

// tot = [dW2 C*C | S1 C | S2 C] (double) -> dW2 (float), S1 / S2 (double: what butd_sa_dz_mid reads)
__global__ void sa_mid_wide_finish_kernel(int C, const double *__restrict__ tot, float *__restrict__ dW2,
                                          double *__restrict__ S1, double *__restrict__ S2) {
  BUTD_MAIN_PRIO_SET();
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x, nw = (long)C * C;
  if (i < nw) dW2[i] = (float)tot[i];
  else if (i < nw + C) S1[i - nw] = tot[i];
  else if (i < nw + 2 * C) S2[i - nw - C] = tot[i];
}

__global__ void sa_d2f_kernel(const double *__restrict__ src, float *__restrict__ dst, long n) {
  BUTD_MAIN_PRIO_SET();
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)src[i];
}

template <int C2, int C3>
constexpr size_t sparse_lds() { return (size_t)(2 * kRows * (C2 + 4)) * sizeof(float); }

hipError_t sparse_attr() {
  static hipError_t err = []() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_last_sparse_kernel<64, 128, 4>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_last_sparse_kernel<128, 256, 8>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_last_fwd_kernel<64, 128, 4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_last_fwd_kernel<128, 256, 8>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_last_fused_kernel<64, 128, 4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e;
  }();
  return err;
}

int g_abl = 0;
template <int C2, int C3>
constexpr size_t fwd_lds() { return (size_t)(2 * kRows * (C2 + 36)) * sizeof(float) + 2 * sizeof(long); }

int grid_mfma(int C2, long nblk) { return (int)(nblk < (C2 == 64 ? 768 : 512) ? nblk : (C2 == 64 ? 768 : 512)); }
int grid_sparse(int C2, long nblk) { return (int)(nblk < (C2 == 64 ? 512 : 256) ? nblk : (C2 == 64 ? 512 : 256)); }

}  // namespace

extern "C" {

int butd_sa_last_bwd_set_ablation(int a) { g_abl = a; return 0; }   /* tuning hook (timing experiments only) */

int butd_sa_last_bwd_supported(int ns, int C2, int C3) {
  return (ns == 16 || ns == 32 || ns == 64) && ((C2 == 64 && C3 == 128) || (C2 == 128 && C3 == 256));
}

int butd_sa_last_fwd(int B, int np, int ns, int C2, int C3, const float *Z2, const float *scale2,
                     const float *shift2, const float *W3, double *sum, double *sumsq, float *zmax, float *zmin,
                     uint8_t *amax, uint8_t *amin, unsigned int *sched, butd_stream_t stream) {
  const long G = (long)B * np, P = G * ns;
  if (P <= 0) return 0;
  if (!butd_sa_last_bwd_supported(ns, C2, C3)) return (int)hipErrorInvalidValue;
  if (hipError_t e = sparse_attr(); e != hipSuccess) return (int)e;
  hipStream_t st = (hipStream_t)stream;
  const long nblk = (P + kRows - 1) / kRows;
  const int chunk = (int)(nblk / 4096 < 1 ? 1 : (nblk / 4096 > 16 ? 16 : nblk / 4096));   // (one workgroup per CU: <= ~4096 counter fetches)
  if (C2 == 64) {
    const int grid = (int)(nblk < 768 ? nblk : 768);
    // (three workgroups per CU: a late one costs 3 %; one counter fetch per 3.6-us block would saturate the counter)
    hipLaunchKernelGGL((sa_last_fwd_kernel<64, 128, 4>), dim3(grid), dim3(256), (fwd_lds<64, 128>()), st, P, nblk, ns, G, Z2,
                       scale2, shift2, W3, sum, sumsq, zmax, zmin, amax, amin, (unsigned int *)nullptr, 1);
  } else {
    const int grid = (int)(nblk < 256 ? nblk : 256);
    hipLaunchKernelGGL((sa_last_fwd_kernel<128, 256, 8>), dim3(grid), dim3(512), (fwd_lds<128, 256>()), st, P, nblk, ns, G, Z2,
                       scale2, shift2, W3, sum, sumsq, zmax, zmin, amax, amin, sched, chunk);
  }
  return (int)hipGetLastError();
}

int butd_sa_first_bwd_scratch(long P, int C1, int Kp, long *ws_floats, long *ws_doubles) {
  if (P <= 0 || !ws_floats || !ws_doubles || Kp != 8 || (C1 & 3) || C1 > 256) return (int)hipErrorInvalidValue;
  const long per = (long)C1 * (Kp + 2) + Kp + Kp * Kp, blocks = (P + kFirstChunk - 1) / kFirstChunk;
  *ws_floats = blocks * per;
  *ws_doubles = per;
  return 0;
}

int butd_sa_first_bwd(long P, int C1, int Kp, const float *dH1, const float *Z1, const float *X, const float *scale1,
                      const float *shift1, const float *mean1, const float *rstd1, const float *W1, float *dW1,
                      double *S1, double *S2, float *ws_f, double *ws_d, butd_stream_t stream) {
  if (P <= 0) return 0;
  if (Kp != 8 || (C1 & 3) || C1 > 256 || kThreads % (C1 >> 2)) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const long per = (long)C1 * (Kp + 2) + Kp + Kp * Kp, blocks = (P + kFirstChunk - 1) / kFirstChunk;
  const int tpg = kThreads / (C1 >> 2);
  const size_t lds = ((size_t)tpg * C1 * (Kp + 2) + (size_t)tpg * (Kp + Kp * Kp)) * sizeof(float);
  if (lds > 64 * 1024) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(sa_first_stats_kernel<8>, dim3((unsigned)blocks), dim3(kThreads), lds, st, P, C1, dH1, Z1, X, scale1,
                     shift1, mean1, rstd1, ws_f, per);
  hipLaunchKernelGGL(sa_last_reduce_kernel, dim3((unsigned)((per + 15) / 16)), dim3(256), 0, st, ws_f, per, (int)blocks,
                     (const float *)nullptr, 0L, 0, ws_d);
  hipLaunchKernelGGL(sa_first_dw_kernel<8>, dim3(C1), dim3(Kp), 0, st, C1, P, Kp, W1, scale1, rstd1, ws_d, dW1, S1, S2);
  return (int)hipGetLastError();
}

int butd_sa_first_two_fwd(long P, int C, int Kp, const float *X, const float *W1, const float *W2, double *mom,
                          double *sum1, double *sumsq1, const float *scale1, const float *shift1, float *Z2, double *sum2,
                          double *sumsq2, int phase, butd_stream_t stream) {
  if (P <= 0) return 0;
  if (Kp != 8 || C != 64) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  if (phase == 0) {            // layer 1's BatchNorm sums from the moments of X (mom: 72 doubles, zero on entry)
    const long blocks = (P + 31) / 32;
    hipLaunchKernelGGL(sa_x_moments_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, st, P, X, mom);
    hipLaunchKernelGGL(sa_l1_stats_kernel, dim3(1), dim3(64), 0, st, C, W1, mom, sum1, sumsq1);
  } else {                     // layer 2: Z2 and its column sums (added: zero on entry), z1 formed on the fly
    static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_l12_fwd_kernel<64>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return (int)attr;
    const long nblk = (P + kRows - 1) / kRows;
    const int grid = (int)(nblk < 768 ? nblk : 768);
    const size_t lds = (size_t)(kRows * (C + 36) + kRows * (C + 4) + kRows * Kp) * sizeof(float);
    hipLaunchKernelGGL(sa_l12_fwd_kernel<64>, dim3(grid), dim3(256), lds, st, P, nblk, X, W1, scale1, shift1, W2, Z2, sum2,
                       sumsq2);
  }
  return (int)hipGetLastError();
}

int butd_sa_mid_first_bwd_scratch(long P, int C, int Kp, long *ws_floats, long *ws_doubles) {
  if (P <= 0 || !ws_floats || !ws_doubles || Kp != 8 || C != 64) return (int)hipErrorInvalidValue;
  const long per = (long)C * (Kp + 2) + Kp + Kp * Kp;
  *ws_floats = 512L * ((long)C * C + per);
  *ws_doubles = (long)C * C + per;
  return 0;
}

int butd_sa_mid_first_bwd(long P, int C, int Kp, const float *G2, const float *Z2, const float *Z1, const float *X,
                          const float *gamma2, const float *scale2, const float *shift2, const float *mean2,
                          const float *rstd2, const double *S1_2, const double *S2_2, const float *scale1,
                          const float *shift1, const float *mean1, const float *rstd1, const float *W2, const float *W1,
                          float *dW2, float *dW1, double *S1_1, double *S2_1, float *ws_f, double *ws_d,
                          butd_stream_t stream) {
  if (P <= 0) return 0;
  if (Kp != 8 || C != 64) return (int)hipErrorInvalidValue;
  static hipError_t attr = []() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_mid_first_kernel<64, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_mid_first_kernel<64, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }();
  if (attr != hipSuccess) return (int)attr;
  hipStream_t st = (hipStream_t)stream;
  const long nblk = (P + kRows - 1) / kRows;
  const int grid = (int)(nblk < 512 ? nblk : 512);
  const long per = (long)C * (Kp + 2) + Kp + Kp * Kp, nw = (long)C * C;
  float *ws_w = ws_f, *ws_p = ws_f + (long)grid * nw;
  const size_t lds = (size_t)(2 * kRows * (C + 36) + kRows * (C + 4) + kRows * Kp) * sizeof(float);
  if (Z1 != nullptr)
    hipLaunchKernelGGL((sa_mid_first_kernel<64, false>), dim3(grid), dim3(256), lds, st, P, nblk, G2, Z2, Z1, X, gamma2, scale2,
                       shift2, mean2, rstd2, S1_2, S2_2, scale1, shift1, mean1, rstd1, W2, W1, ws_w, ws_p, per);
  else       // Z1 was never written (butd_sa_first_two_fwd): recomputed from X and W1
    hipLaunchKernelGGL((sa_mid_first_kernel<64, true>), dim3(grid), dim3(256), lds, st, P, nblk, G2, Z2, Z1, X, gamma2, scale2,
                       shift2, mean2, rstd2, S1_2, S2_2, scale1, shift1, mean1, rstd1, W2, W1, ws_w, ws_p, per);
  hipLaunchKernelGGL(sa_last_reduce_kernel, dim3((unsigned)((nw + per + 15) / 16)), dim3(256), 0, st, ws_w, nw, grid, ws_p, per,
                     grid, ws_d);
  hipLaunchKernelGGL(sa_d2f_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, ws_d, dW2, nw);
  hipLaunchKernelGGL(sa_first_dw_kernel<8>, dim3(C), dim3(Kp), 0, st, C, P, Kp, W1, scale1, rstd1, ws_d + nw, dW1, S1_1, S2_1);
  return (int)hipGetLastError();
}

int butd_sa_mid_wide_bwd_scratch(long P, int C, long *ws_floats, long *ws_doubles) {
  if (P <= 0 || !ws_floats || !ws_doubles || C != 128) return (int)hipErrorInvalidValue;
  *ws_floats = 512L * ((long)C * C + 2 * C);
  *ws_doubles = (long)C * C + 2 * C;
  return 0;
}

int butd_sa_mid_wide_bwd(long P, int C, const float *G2, const float *Z2, const float *Z1, const float *gamma2,
                         const float *scale2, const float *shift2, const float *mean2, const float *rstd2,
                         const double *S1_2, const double *S2_2, const float *scale1, const float *shift1,
                         const float *mean1, const float *rstd1, const float *W2, float *G1, float *dW2, double *S1_1,
                         double *S2_1, float *ws_f, double *ws_d, butd_stream_t stream) {
  if (P <= 0) return 0;
  if (C != 128 || !G2 || !Z2 || !Z1 || !G1 || !dW2 || !ws_f || !ws_d) return (int)hipErrorInvalidValue;
  constexpr size_t lds = (size_t)(2 * kRowsW * (128 + 36) + kRowsW * (128 + 4)) * sizeof(float);
  static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_mid_wide_kernel<128>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (attr != hipSuccess) return (int)attr;
  hipStream_t st = (hipStream_t)stream;
  const long nblk = (P + kRowsW - 1) / kRowsW;
  const int grid = (int)(nblk < 512 ? nblk : 512);
  const long nw = (long)C * C, per = 2L * C;
  float *ws_w = ws_f, *ws_p = ws_f + (long)grid * nw;
  hipLaunchKernelGGL((sa_mid_wide_kernel<128>), dim3(grid), dim3(512), lds, st, P, nblk, G2, Z2, Z1, gamma2, scale2, shift2, mean2,
                     rstd2, S1_2, S2_2, scale1, shift1, mean1, rstd1, W2, G1, ws_w, ws_p);
  hipLaunchKernelGGL(sa_last_reduce_kernel, dim3((unsigned)((nw + per + 15) / 16)), dim3(256), 0, st, ws_w, nw, grid, ws_p, per,
                     grid, ws_d);
  hipLaunchKernelGGL(sa_mid_wide_finish_kernel, dim3((unsigned)((nw + per + 255) / 256)), dim3(256), 0, st, C, ws_d, dW2, S1_1,
                     S2_1);
  return (int)hipGetLastError();
}

int butd_sa_last_bwd_scratch(long P, int C2, int C3, long *ws_floats, long *ws_doubles) {
  if (P <= 0 || !ws_floats || !ws_doubles) return (int)hipErrorInvalidValue;
  const long nblk = (P + kRows - 1) / kRows;
  const long per_sparse = (long)C3 * C2 + 3 * C2;
  *ws_floats = (long)C2 * C2 + C2 + (long)grid_mfma(C2, nblk) * C2 * C2 + (long)grid_sparse(C2, nblk) * per_sparse;
  *ws_doubles = (long)C2 * C2 + per_sparse;
  return 0;
}

int butd_sa_last_bwd(int B, int np, int ns, int C2, int C3, const float *Z2, const float *scale2,
                     const float *shift2, const float *mean2, const float *rstd2, const float *W3,
                     const float *d_out_pm, const float *zsel, const uint8_t *asel, const float *scale3,
                     const float *shift3, const float *mean3, const float *rstd3, const double *S1_3,
                     const double *S2_3, float *dH2, float *dW3, double *S1_2, double *S2_2, float *ws_f,
                     double *ws_d, butd_stream_t stream) {
  const long G = (long)B * np, P = G * ns;
  if (P <= 0) return 0;
  if (!butd_sa_last_bwd_supported(ns, C2, C3)) return (int)hipErrorInvalidValue;
  if (hipError_t e = sparse_attr(); e != hipSuccess) return (int)e;
  hipStream_t st = (hipStream_t)stream;
  const long nblk = (P + kRows - 1) / kRows;
  const int gm = grid_mfma(C2, nblk), gs = grid_sparse(C2, nblk);
  const long per_sparse = (long)C3 * C2 + 3 * C2;
  float *An = ws_f, *dvec = An + (long)C2 * C2, *ws_gram = dvec + C2, *ws_sparse = ws_gram + (long)gm * C2 * C2;   // (gm >= gs)
  if (C3 == 128)
    hipLaunchKernelGGL(sa_last_coeffs_kernel<128>, dim3(C2 / 16, C2 / 16), dim3(256), 0, st, C2, P, W3, scale3, mean3, rstd3, S1_3, S2_3, An, dvec);
  else
    hipLaunchKernelGGL(sa_last_coeffs_kernel<256>, dim3(C2 / 16, C2 / 16), dim3(256), 0, st, C2, P, W3, scale3, mean3, rstd3, S1_3, S2_3, An, dvec);
  const long n1 = (long)C2 * C2, n2 = per_sparse;
  {
  const bool fused = !(g_abl & 16) && C2 == 64;

  if (fused) {     // one pass; both partial groups use the sparse grid
    ws_sparse = ws_gram + (long)gs * C2 * C2;
    hipLaunchKernelGGL((sa_last_fused_kernel<64, 128, 4>), dim3(gs), dim3(256), (fused_lds<64>()), st, P, nblk, ns, G, dH2,
                       Z2, scale2, shift2, mean2, rstd2, W3, An, dvec, d_out_pm, zsel, asel, scale3, shift3, ws_gram,
                       ws_sparse, per_sparse);
  } else if (C2 == 64) {
    hipLaunchKernelGGL(sa_last_mfma_kernel<64>, dim3(gm), dim3(kThreads), 0, st, P, nblk, Z2, scale2, shift2, An, dvec, dH2, ws_gram);
    hipLaunchKernelGGL((sa_last_sparse_kernel<64, 128, 4>), dim3(gs), dim3(256), (sparse_lds<64, 128>()), st, P, nblk, ns, G, dH2, Z2, scale2,
                       shift2, mean2, rstd2, W3, d_out_pm, zsel, asel, scale3, shift3, ws_sparse, per_sparse, g_abl);
  } else {
    hipLaunchKernelGGL(sa_last_mfma_kernel<128>, dim3(gm), dim3(kThreads), 0, st, P, nblk, Z2, scale2, shift2, An, dvec, dH2, ws_gram);
    hipLaunchKernelGGL((sa_last_sparse_kernel<128, 256, 8>), dim3(gs), dim3(512), (sparse_lds<128, 256>()), st, P, nblk, ns, G, dH2, Z2, scale2,
                       shift2, mean2, rstd2, W3, d_out_pm, zsel, asel, scale3, shift3, ws_sparse, per_sparse, g_abl);
  }
  hipLaunchKernelGGL(sa_last_reduce_kernel, dim3((unsigned)((n1 + n2 + 15) / 16)), dim3(256), 0, st, ws_gram, n1, fused ? gs : gm,
                     ws_sparse, n2, gs, ws_d);
  }
  hipLaunchKernelGGL(sa_last_dw_kernel, dim3(C3), dim3(C2), 0, st, C2, C3, P, W3, scale3, rstd3, S1_3, S2_3, ws_d, dW3,
                     S1_2, S2_2);
  return (int)hipGetLastError();
}

}  // extern "C"
