// gemm_ops.hip -- grouped dense products on v_mfma_f32_16x16x4_f32 (exact fp32, 157 TF peak), the
// projection / FFN / 1x1-conv products of the attention stack, the set-abstraction MLPs and every
// gradient product of them (include/butd_attention.h: butd_gemm_grouped).  gfx950 only.
//
// Shape of the kernel
//   * 256 threads = 4 waves in a 2 x 2 grid; the workgroup tile is TM x TN (multiples of 32, chosen per
//     launch from a small menu so that rectangular problems -- N = 288 = 3 x 96 -- tile without waste
//     and small ones still fill 256 CUs), a wave owns (TM/2) x (TN/2) = kMI x kNJ MFMA tiles.
//   * BK = 32 slabs, double-buffered LDS, ONE barrier per slab.  The order inside an iteration is pinned
//     with sched_barrier(0):   issue the global loads of slab i+1  |  fragments + MFMAs of slab i  |
//     wait, apply operand effects, write slab i+1 to the other LDS buffer  |  barrier.
//     (Round 1 left the order to the compiler, which sank the loads to the middle of the MFMA chain and
//     waited vmcnt(0) right behind them: the L2 round trip was exposed in every slab.)
//   * The k index of a contraction may be permuted freely as long as A and B use the same permutation;
//     MFMA step s of lane-group g = lane>>4 consumes k = 4*g + s of the current 16-wide sub-slab, which
//     makes every operand fragment 16 contiguous bytes: a contraction-contiguous operand is stored
//     [row][k] (row stride 36 floats) and a fragment is ONE ds_read_b128; a row-contiguous operand (both
//     operands of a weight-gradient product) is stored as it arrives, [k][row], float4 writes, and a
//     fragment is four ds_read_b32.
//   * One launch serves up to 8 problems (Q/K/V projections, the input- and weight-gradient products of
//     a block ...) in a 1-D grid; epilogues: bias / scale / ReLU / dropout, column sums (BatchNorm
//     statistics), accumulation into existing tensors (c_add / c2), atomics for split-K.
#include <hip/hip_runtime.h>
// (ablation hook, scratch/r6_prio.sh: -DBUTD_MAIN_PRIO=n raises the wave priority of this unit's kernels -- the captured
// step's prefetch branches share CUs with them; profiles/r06_side_branches.txt)
#ifdef BUTD_MAIN_PRIO
#define BUTD_MAIN_PRIO_SET() __builtin_amdgcn_s_setprio(BUTD_MAIN_PRIO)
#else
#define BUTD_MAIN_PRIO_SET()
#endif

#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/butd_attention.h"
#include "rng.h"

// timing ablations for scratch/ experiments (never set in the product build): bit 0 no MFMAs, 1 no global
// loads after the first slab, 2 no LDS commit after the first slab, 3 no barrier in the loop, 4 no epilogue
#ifndef GEMM_ABL
#define GEMM_ABL 0
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kThreads = 256;
constexpr int kLdH = 40;   // bf16 image: 32 k + 8 pad per row (80 B)
constexpr int kBK = 32, kLd = kBK + 4;  // slab depth; LDS row stride 36 floats = 9 x 16 B
constexpr int kMaxProblems = 32;   // (the GemmBatch is a by-value kernel argument of ~14 KB: this runtime takes 64 KB, scratch/ubench/kernarg_size.hip)
constexpr int kAffK = 320;  // contraction range whose A-operand affine is staged in LDS (fast path)

// Loads that must be emitted as global_load_*: a FLAT load also counts against lgkmcnt, so the
// s_waitcnt lgkmcnt(0) in front of the MFMAs (for the LDS fragment reads) would wait for the prefetch
// of the NEXT slab as well and serialize HBM latency with the matrix pipe.
typedef const __attribute__((address_space(1))) f32x4 *global_f4_ptr;
__device__ inline f32x4 ldg4(const float *p) {
  return *reinterpret_cast<global_f4_ptr>(reinterpret_cast<uintptr_t>(p));
}

// uniform base (scalar registers) + 32-bit BYTE offset per lane: the "saddr + voffset" form of global_load -- no 64-bit
// vector address arithmetic in front of the load
typedef const __attribute__((address_space(1))) char *global_byte_ptr;
__device__ inline f32x4 ldg4_at(const float *base, uint32_t byte_off) {
  const global_byte_ptr g = reinterpret_cast<global_byte_ptr>(reinterpret_cast<uintptr_t>(base));
  return *reinterpret_cast<global_f4_ptr>(g + byte_off);
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it waits for the
// global loads of the slabs still in flight
__device__ inline void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// C-tile store: outputs far larger than the caches (the 10^5..10^6-row set-abstraction activations) are
// written nontemporal -- their consumer reads them from HBM anyway (measured: -7 % on those launches; no
// effect on the 9 MB outputs of the attention stack, which stay plain so the next kernel finds them in L2)
__device__ inline void store_c4(float *dst, float4 v, bool streaming) {
  if (streaming) {
    const f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4 *>(dst));
  } else {
    *reinterpret_cast<float4 *>(dst) = v;
  }
}

struct GemmBatch {
  butd_gemm_problem p[kMaxProblems];
  int blk_begin[kMaxProblems + 1];  // linear workgroup range of each problem
  int tiles_n[kMaxProblems], tiles_m[kMaxProblems];
  int count;
};

// ------------------------------------------------------------------------------------------------
// generic staging (unaligned operands, ragged K, the a2 companion): element-wise predicates
// ------------------------------------------------------------------------------------------------
struct Frag4 {
  float4 a, a2;
};

// position of float4 `f` (0 .. T*8-1) of a (T rows x 32 k) slab: contraction-contiguous operands are cut
// into 8 float4 per row, row-contiguous ones into T/4 float4 per k
template <int T>
__device__ inline void slab_pos(bool kc, int f, int &row, int &k) {
  if (kc) {
    row = f >> 3;
    k = (f & 7) * 4;
  } else {
    k = f / (T / 4);
    row = (f - k * (T / 4)) * 4;
  }
}

template <bool WITH_A2>
__device__ inline Frag4 fetch_generic(const float *__restrict__ src, const float *__restrict__ src2, bool kc,
                                      long ld_row, long ld_k, int row, int nrows, int k, int kend) {
  Frag4 f;
  f.a = make_float4(0.f, 0.f, 0.f, 0.f);
  f.a2 = f.a;
  const long ld_slow = kc ? ld_row : ld_k;
  const int slow_g = kc ? row : k, fast_g = kc ? k : row;
  const int slow_lim = kc ? nrows : kend, fast_lim = kc ? kend : nrows;
  if (slow_g < slow_lim && fast_g < fast_lim) {
    const long o = (long)slow_g * ld_slow + fast_g;
    const bool vec = (fast_g + 3 < fast_lim) && ((ld_slow & 3) == 0);
    if (vec && ((((uintptr_t)src) & 15) == 0)) {
      f.a = *reinterpret_cast<const float4 *>(src + o);
    } else {
      f.a.x = src[o];
      if (fast_g + 1 < fast_lim) f.a.y = src[o + 1];
      if (fast_g + 2 < fast_lim) f.a.z = src[o + 2];
      if (fast_g + 3 < fast_lim) f.a.w = src[o + 3];
    }
    if (WITH_A2 && src2) {
      if (vec && ((((uintptr_t)src2) & 15) == 0)) {
        f.a2 = *reinterpret_cast<const float4 *>(src2 + o);
      } else {
        f.a2.x = src2[o];
        if (fast_g + 1 < fast_lim) f.a2.y = src2[o + 1];
        if (fast_g + 2 < fast_lim) f.a2.z = src2[o + 2];
        if (fast_g + 3 < fast_lim) f.a2.w = src2[o + 3];
      }
    }
  }
  return f;
}

__device__ inline float combine(float a, float a2, int mode, float gate_scale) {
  return mode == 0 ? a + a2 : a * (a2 > 0.f ? gate_scale : 0.f);
}

// Everything the generic commit applies to a staged operand besides the plain copy.
// chan_is_k: the affine's channel index is the contraction index (A operand) or the row index (B)
struct OperandFx {
  bool has2; int mode2; float scale2;          // companion operand a2
  const float *csc, *csh; bool chan_is_k;      // per-channel affine + ReLU
  float drop_p, drop_inv; uint32_t drop_key;   // dropout keyed by the element's memory offset
  long ld_row;
};

// writes the float4 at (row, k) [slab-relative] of the K-contiguous LDS image tile[row][k]
__device__ inline void commit_generic(float *tile, const Frag4 &f, const OperandFx &fx, bool kc, long ld_k,
                                      int row0, int nrows, int k0, int kend, bool ones, int row, int k) {
  float v[4] = {f.a.x, f.a.y, f.a.z, f.a.w};
  if (fx.has2) {
    v[0] = combine(v[0], f.a2.x, fx.mode2, fx.scale2); v[1] = combine(v[1], f.a2.y, fx.mode2, fx.scale2);
    v[2] = combine(v[2], f.a2.z, fx.mode2, fx.scale2); v[3] = combine(v[3], f.a2.w, fx.mode2, fx.scale2);
  }
  if (fx.csc || fx.drop_p > 0.f) {  // relu(v * scale[chan] + shift[chan]), dropout; out-of-range stays 0
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = row0 + row + (kc ? 0 : i), kk = k0 + k + (kc ? i : 0);
      if (r < nrows && kk < kend) {
        if (fx.csc) {
          const int ch = fx.chan_is_k ? kk : r;
          v[i] = fmaxf(v[i] * fx.csc[ch] + fx.csh[ch], 0.f);
        }
        if (fx.drop_p > 0.f) {
          const uint32_t off = (uint32_t)((long)r * fx.ld_row + (long)kk * ld_k);
          v[i] = rng::keep_keyed(fx.drop_key, off, fx.drop_p) ? v[i] * fx.drop_inv : 0.f;
        }
      }
    }
  }
  if (kc) {
    if (ones && row0 + row == nrows) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = (k0 + k + i < kend) ? 1.f : 0.f;
    }
    *reinterpret_cast<float4 *>(&tile[row * kLd + k]) = make_float4(v[0], v[1], v[2], v[3]);
  } else {  // transpose into the K-contiguous LDS image
    if (ones && k0 + k < kend) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (row0 + row + i == nrows) v[i] = 1.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[(row + i) * kLd + k] = v[i];
  }
}

// A FOLD problem (butd_gemm_problem.fold_src): the fixed-order sum of the partial slabs a deterministic split-K problem
// of an EARLIER launch left behind; workgroup `chunk` of the problem's range owns 2048 consecutive floats of the
// (weights | bias) vector.  Element-wise, no LDS: it rides in whatever grouped launch comes next on the stream.
constexpr int kFoldChunk = 2048;
__device__ __forceinline__ void fold_slabs(const butd_gemm_problem &P, int chunk) {
  const long len1 = P.fold_len, total = P.fold_len + P.fold_len2, stride = P.fold_stride;
  const float *const src = P.fold_src;
  const int S = P.fold_count;
  const bool add = P.c_add != 0;
  const bool vec1 = ((uintptr_t)P.c & 15) == 0, vec2 = ((uintptr_t)P.bias_grad & 15) == 0 && (len1 & 3) == 0;
#pragma unroll
  for (int u = 0; u < kFoldChunk / (4 * kThreads); ++u) {
    const long i = (long)chunk * kFoldChunk + (long)(u * kThreads + (int)threadIdx.x) * 4;
    if (i >= total) return;
    const bool first = i < len1;                 // (len1 % 4 == 0: a float4 never straddles the two destinations)
    float *const dst = first ? P.c + i : P.bias_grad + (i - len1);
    if (i + 3 < total && (first ? vec1 : vec2)) {
      f32x4 acc = *reinterpret_cast<const f32x4 *>(src + i);
      for (int sl = 1; sl < S; ++sl) acc += *reinterpret_cast<const f32x4 *>(src + (long)sl * stride + i);
      if (add) acc += *reinterpret_cast<const f32x4 *>(dst);
      *reinterpret_cast<f32x4 *>(dst) = acc;
    } else {
      for (int e = 0; e < 4 && i + e < total; ++e) {
        float a = src[i + e];
        for (int sl = 1; sl < S; ++sl) a += src[(long)sl * stride + i + e];
        dst[e] = add ? dst[e] + a : a;
      }
    }
  }
}

struct Whole { static constexpr bool ragged = false; };   // slab kinds of the fast path (see below)
struct Ragged { static constexpr bool ragged = true; };

// FAST: every problem of the launch satisfies fast_eligible() (host side): tiles stream whole float4s
// with addresses  base + slab * step  and no bounds checks; the generic instantiation handles ragged K,
// unaligned operands and the a2 companion.  Two kernels instead of one runtime branch: with both paths
// in one body the compiler merged their MFMA blocks and serialized loads behind them.
// PIPE (FAST only): 2 = slab i+2 travels global -> registers while slab i+1 is written to LDS and slab i is
// multiplied (nothing waits for a load of its own iteration; barrier without the vmcnt drain);
// 0 = one slab ahead, commit after the MFMAs (what the 10^5..10^6-slab weight-gradient launches of the
// set-abstraction backward prefer, measured: 15-20 % there)
// BF (FAST only): operands rounded to bf16 (RNE) on their way into LDS, v_mfma_f32_16x16x32_bf16 (one
// instruction per 16 x 16 tile and slab), fp32 accumulators and epilogue -- BASELINE configs[3]'s "bf16
// attention / FFN" operating point; tensors stay fp32 in HBM.  Both operand kinds use the [row][k] image (a
// row-contiguous operand is transposed by four 2-byte writes): a fragment is 8 consecutive k = one ds_read_b128.
template <int TM, int TN, bool FAST, int PIPE, bool BF = false>
__global__ __launch_bounds__(kThreads) void gemm_kernel(GemmBatch batch,
                                                        const uint64_t *__restrict__ rng_counter) {
  BUTD_MAIN_PRIO_SET();
  constexpr int kMI = TM / 32, kNJ = TN / 32;   // 16 x 16 MFMA tiles per wave: rows, columns
  constexpr int kSubA = TM / 32, kSubB = TN / 32;   // float4 per thread, operand and slab
  constexpr int kWM = TM / 2, kWN = TN / 2;     // wave tile
  static_assert(TM % 32 == 0 && TN % 32 == 0 && TM >= 32 && TN >= 32, "tile sides are multiples of 32");
  __shared__ __attribute__((aligned(16))) float lds[2 * (TM + TN) * kLd];
  __shared__ __attribute__((aligned(16))) float Asc[kAffK], Ash[kAffK];
  auto As = [&](int buf) { return lds + buf * (TM * kLd); };
  auto Bs = [&](int buf) { return lds + 2 * TM * kLd + buf * (TN * kLd); };

  // 1-D grid: every problem owns exactly tiles_n x tiles_m x split_k consecutive LOGICAL workgroups.
  // Hardware workgroup b runs on XCD b % 8 (each XCD has its own 4 MiB L2): the logical index gives every
  // XCD one contiguous range, so the column tiles that share an A row panel (consecutive logical indices)
  // hit in one L2 instead of fetching the panel into up to 8 of them (bijective for any grid size).
  const int wg = [&] {
    // ... inside windows of 64 consecutive hardware indices only: the whole chip keeps walking the grid
    // front to back (eight XCDs streaming eight far-apart regions of a 10^6-row operand collide on the same
    // HBM channels: measured 3.6x slower), while each XCD still gets runs of 8 consecutive tiles
    const int nb = (int)gridDim.x, b = (int)blockIdx.x;
    if ((b | 63) >= nb) return b;                       // ragged last window: identity
    return (b & ~63) | ((b & 7) << 3) | ((b >> 3) & 7);
  }();
  // (blk_begin is monotone and padded with the grid size: eight independent scalar loads and compares
  // instead of a chain of dependent ones in front of the first operand load)
  int pi = 0;
#pragma unroll
  for (int i = 1; i < kMaxProblems; ++i) pi += (wg >= batch.blk_begin[i]) ? 1 : 0;
  pi = min(pi, batch.count - 1);
  const butd_gemm_problem &P = batch.p[pi];
  int rel = wg - batch.blk_begin[pi];
  if (P.fold_src) {          // (uniform per workgroup) the ride-along fold of an earlier launch's split-K slabs
    fold_slabs(P, rel);
    return;
  }
  const int tn = batch.tiles_n[pi], tm = batch.tiles_m[pi];
  const int bx = rel % tn;
  rel /= tn;
  const int by = rel % tm;
  const int slice = rel / tm;
  const int m0 = by * TM, n0 = bx * TN;

  // contraction range of this split-K slice (multiples of kBK)
  const int kslab = (P.K + kBK - 1) / kBK;
  const int per = (kslab + P.split_k - 1) / P.split_k;
  const int kbeg = slice * per * kBK;
  const int kend = min(P.K, (slice + 1) * per * kBK);
  if (kbeg >= kend && slice > 0) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;

  f32x4 acc[kMI][kNJ];
#pragma unroll
  for (int i = 0; i < kMI; ++i)
#pragma unroll
    for (int j = 0; j < kNJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool ones = P.ones_col != 0;
  // operand dropout: the (step, site) halves of the hash are kernel-invariant
  const bool a_dropout = P.a_drop_p > 0.f, b_dropout = P.b_drop_p > 0.f;
  const uint64_t step_ctr = ((a_dropout || b_dropout || P.dropout_p > 0.f || P.c_bn_drop_p > 0.f) && rng_counter) ? *rng_counter : 0ull;
  const uint32_t a_key = rng::site_key(step_ctr, P.a_drop_site), b_key = rng::site_key(step_ctr, P.b_drop_site);
  const float a_inv = a_dropout ? 1.f / (1.f - P.a_drop_p) : 1.f, b_inv = b_dropout ? 1.f / (1.f - P.b_drop_p) : 1.f;

  if constexpr (FAST) {
    // The loop is specialised at compile time on the operand layouts and on "plain" vs "with effects"
    // (affine / dropout / ones-row / ragged last slab) and selected by one switch per workgroup: with
    // every mode behind run-time branches in one loop body the kernel was ~8 % slower (the path taken
    // was a few hundred instructions scattered over a 30 KB body).
    const bool rt_a_kc = P.lda_k == 1, rt_b_kc = P.ldb_k == 1;
    const bool rt_fx = P.a_chan_scale != nullptr || P.a_bn_sum != nullptr || P.b_chan_scale != nullptr || a_dropout ||
                       b_dropout || ones || ((kend - kbeg) % kBK) != 0;
    auto run_fast = [&](auto a_kc_t, auto b_kc_t, auto fx_t) {
      constexpr bool a_kc = decltype(a_kc_t)::value, b_kc = decltype(b_kc_t)::value;
      constexpr bool FX = decltype(fx_t)::value;
      constexpr int kLdTA = TM + 4, kLdTB = TN + 4;   // row stride of a [k][row] image
      static_assert(kBK * kLdTA <= TM * kLd && kBK * kLdTB <= TN * kLd, "[k][row] image must fit the [row][k] buffer");
      const bool f_ones = FX && ones, f_adrop = FX && a_dropout, f_bdrop = FX && b_dropout;
      const bool a_bn = FX && P.a_bn_sum != nullptr;        // ... computed here from the producer's BatchNorm sums
      const bool a_aff = FX && (P.a_chan_scale != nullptr || a_bn);   // channel = k (varies per slab): staged in LDS
      const bool b_aff = FX && P.b_chan_scale != nullptr;   // channel = B row: loop-invariant per thread
      const int krange = kend - kbeg;
      // per-thread staging plan: float4 u of operand X sits at (row, k) of the slab; rows outside the
      // matrix read row 0 of the operand instead (every load stays an unconditional global_load) and are
      // zeroed at commit time
      // (addresses: ONE uniform base per operand and tile -- scalar registers -- plus a 32-bit element offset per
      // float4: the loads take the "scalar base + vector offset" form and neither the plan nor the K loop does
      // 64-bit vector arithmetic; tile rows x leading dimension stays far below 2^31 elements)
      const float *const abase = a_kc ? P.a + (long)m0 * P.lda_m + kbeg : P.a + (long)kbeg * P.lda_k + m0;
      // (a column tile that holds nothing but the virtual ones-row starts AT row N: its clamped loads must still
      //  land inside the operand, so the base row is min(n0, N - 1); tiles with real rows have n0 < N)
      const int n0c = min(n0, b_kc ? P.N - 1 : P.N - 4);   // (row-contiguous operands: a float4 spans four rows)
      const float *const bbase = b_kc ? P.b + (long)n0c * P.ldb_n + kbeg : P.b + (long)kbeg * P.ldb_k + n0c;
      const bool edge = m0 + TM > P.M || n0 + TN > P.N;   // uniform: only then rows are clamped / zeroed
      uint32_t pa[kSubA], pb[kSubB];
      int a_lds[kSubA], b_lds[kSubB], a_k[kSubA], b_k[kSubB], ones_e[kSubB];
      bool a_ok[kSubA], b_ok[kSubB];
      float4 bsc[kSubB], bsh[kSubB];
      const int lda_m = (int)P.lda_m, lda_k = (int)P.lda_k, ldb_n = (int)P.ldb_n, ldb_k = (int)P.ldb_k;
#pragma unroll
      for (int u = 0; u < kSubA; ++u) {
        int row, k;
        slab_pos<TM>(a_kc, tid + u * kThreads, row, k);
        a_k[u] = k;
        a_ok[u] = !edge || m0 + row < P.M;
        const int r = a_ok[u] ? row : 0;   // rows outside the matrix read the tile's first row instead
        pa[u] = 4u * (uint32_t)(a_kc ? r * lda_m + k : k * lda_k + r);   // bytes
        a_lds[u] = BF ? row * kLdH + k : (a_kc ? row * kLd + k : k * kLdTA + row);
      }
#pragma unroll
      for (int u = 0; u < kSubB; ++u) {
        int row, k;
        slab_pos<TN>(b_kc, tid + u * kThreads, row, k);
        b_k[u] = k;
        b_ok[u] = !edge || n0 + row < P.N;
        const int r = b_ok[u] ? row : 0;
        pb[u] = 4u * (uint32_t)(b_kc ? r * ldb_n + k : k * ldb_k + r);
        b_lds[u] = BF ? row * kLdH + k : (b_kc ? row * kLd + k : k * kLdTB + row);
        // virtual ones-row of B (row index N): which of this float4's elements is it, if any
        ones_e[u] = !f_ones ? -1 : (b_kc ? (n0 + row == P.N ? 4 : -1)
                                         : ((n0 + row <= P.N && P.N < n0 + row + 4) ? P.N - (n0 + row) : -1));
        bsc[u] = make_float4(1.f, 1.f, 1.f, 1.f);
        bsh[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b_aff && b_ok[u]) {
          if (b_kc) {
            const float sc = P.b_chan_scale[n0 + row], sh = P.b_chan_shift[n0 + row];
            bsc[u] = make_float4(sc, sc, sc, sc);
            bsh[u] = make_float4(sh, sh, sh, sh);
          } else {
            bsc[u] = *reinterpret_cast<const float4 *>(P.b_chan_scale + n0 + row);
            bsh[u] = *reinterpret_cast<const float4 *>(P.b_chan_shift + n0 + row);
          }
        }
      }
      const long sa = a_kc ? kBK : (long)kBK * P.lda_k, sb = b_kc ? kBK : (long)kBK * P.ldb_k;   // per slab
      if (a_bn) {
        // BatchNorm bookkeeping of the producing layer (the arithmetic of butd_mlp_bn_finalize): every workgroup
        // derives scale / shift of its contraction range; the problem's first workgroup also leaves mean / rstd /
        // scale / shift behind for the backward pass and updates the running statistics -- exactly once
        const bool writer = wg == batch.blk_begin[pi];
        const double cnt = (double)P.a_bn_count;
        for (int k = tid; k < krange; k += kThreads) {
          const int c = kbeg + k;
          const double m = P.a_bn_sum[c] / cnt;
          double v = P.a_bn_sumsq[c] / cnt - m * m;
          if (v < 0.0) v = 0.0;
          const float mu = (float)m, var = (float)v;
          const float rs = 1.0f / sqrtf(var + P.a_bn_eps);
          const float g = P.a_bn_gamma[c];
          const float sc = g * rs, sh = P.a_bn_beta[c] - mu * g * rs;
          Asc[k] = sc;
          Ash[k] = sh;
          if (writer) {
            if (P.a_bn_out) {
              P.a_bn_out[c] = mu;
              P.a_bn_out[P.a_bn_ld + c] = rs;
              P.a_bn_out[2 * P.a_bn_ld + c] = sc;
              P.a_bn_out[3 * P.a_bn_ld + c] = sh;
            }
            if (P.a_bn_running_mean) {
              const double unbiased = P.a_bn_count > 1 ? v * cnt / (double)(P.a_bn_count - 1) : v;
              const float mom = P.a_bn_momentum;
              P.a_bn_running_mean[c] = (1.f - mom) * P.a_bn_running_mean[c] + mom * mu;
              P.a_bn_running_var[c] = (1.f - mom) * P.a_bn_running_var[c] + mom * (float)unbiased;
            }
          }
        }
        if (writer && tid == 0 && P.a_bn_nbt) *P.a_bn_nbt += 1;
        __syncthreads();
      } else if (a_aff) {
        for (int k = tid; k < krange; k += kThreads) {
          Asc[k] = P.a_chan_scale[kbeg + k];
          Ash[k] = P.a_chan_shift[kbeg + k];
        }
        __syncthreads();
      }
      const f32x4 zero_v = {0.f, 0.f, 0.f, 0.f};
      auto f4 = [](f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); };
      auto drop4 = [](float4 v, uint32_t key, uint32_t off, float p, float inv) {
        v.x = rng::keep_keyed(key, off + 0, p) ? v.x * inv : 0.f;
        v.y = rng::keep_keyed(key, off + 1, p) ? v.y * inv : 0.f;
        v.z = rng::keep_keyed(key, off + 2, p) ? v.z * inv : 0.f;
        v.w = rng::keep_keyed(key, off + 3, p) ? v.w * inv : 0.f;
        return v;
      };
      // two register sets: slab i+2 travels global -> registers while slab i+1 (fetched an iteration earlier,
      // certainly landed) is written to LDS and slab i is multiplied
      // (plain arrays selected at compile time: handed around as a struct by reference they ended up in
      // scratch memory for the one-float4-per-thread tiles)
      f32x4 ra0[kSubA], rb0[kSubB], ra1[kSubA], rb1[kSubB];   // (ext vectors: float4 structs of a 1-element array stay in scratch)
      typedef std::integral_constant<int, 0> Set0;
      typedef std::integral_constant<int, 1> Set1;
      // fetch / commit come in two flavours selected at compile time: whole slabs (the steady state: no
      // predicates at all) and the ragged last slab (K % 32 != 0, K % 4 == 0: float4s beyond the slice
      // read as zero; its predicates cost ~8 % when left in the main loop)
      auto fetch_fast = [&](auto set, int slab, auto kind) {
#pragma unroll
        for (int u = 0; u < kSubA; ++u) {
          f32x4 v;
          if constexpr (!decltype(kind)::ragged) v = ldg4_at(abase + slab * sa, pa[u]);
          else v = (slab * kBK + a_k[u] < krange) ? ldg4_at(abase + slab * sa, pa[u]) : zero_v;
          if constexpr (decltype(set)::value == 0) ra0[u] = v; else ra1[u] = v;
        }
#pragma unroll
        for (int u = 0; u < kSubB; ++u) {
          f32x4 v;
          if constexpr (!decltype(kind)::ragged) v = ldg4_at(bbase + slab * sb, pb[u]);
          else v = (slab * kBK + b_k[u] < krange) ? ldg4_at(bbase + slab * sb, pb[u]) : zero_v;
          if constexpr (decltype(set)::value == 0) rb0[u] = v; else rb1[u] = v;
        }
      };
      // one staged float4 -> LDS (fp32 image: as it is; bf16 image: rounded, transposed if row-contiguous)
      auto put = [&](float *tile, int off, float4 v, auto kc) {
        if constexpr (!BF) {
          *reinterpret_cast<float4 *>(tile + off) = v;
        } else {
          __bf16 *h = reinterpret_cast<__bf16 *>(tile) + off;
          if constexpr (decltype(kc)::value) {
            *reinterpret_cast<bf16x4 *>(h) = (bf16x4){(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
          } else {
            h[0] = (__bf16)v.x; h[kLdH] = (__bf16)v.y; h[2 * kLdH] = (__bf16)v.z; h[3 * kLdH] = (__bf16)v.w;
          }
        }
      };
      auto commit_fast = [&](auto set, int slab, int buf, auto kind) {
        const int kslab0 = slab * kBK;   // k offset (relative to kbeg) of the slab held in rg
        float *at = As(buf), *bt = Bs(buf);
#pragma unroll
        for (int u = 0; u < kSubA; ++u) {
          bool in = true;
          if constexpr (decltype(kind)::ragged) in = kslab0 + a_k[u] < krange;
          const bool live = a_ok[u] && in;
          float4 va;
          if constexpr (decltype(set)::value == 0) va = f4(ra0[u]); else va = f4(ra1[u]);
          if (decltype(kind)::ragged || edge) {   // (uniform condition; element-wise: a select between two float4
            va.x = live ? va.x : 0.f; va.y = live ? va.y : 0.f;   //  STRUCTS is a select between their addresses and
            va.z = live ? va.z : 0.f; va.w = live ? va.w : 0.f;   //  sends both to scratch memory)
          }
          if constexpr (FX) {
            if (a_aff && live) {
              float4 sc, sh;
              if (a_kc) {
                sc = *reinterpret_cast<const float4 *>(&Asc[kslab0 + a_k[u]]);
                sh = *reinterpret_cast<const float4 *>(&Ash[kslab0 + a_k[u]]);
              } else {
                const float s1 = Asc[kslab0 + a_k[u]], h1 = Ash[kslab0 + a_k[u]];
                sc = make_float4(s1, s1, s1, s1);
                sh = make_float4(h1, h1, h1, h1);
              }
              va.x = fmaxf(va.x * sc.x + sh.x, 0.f); va.y = fmaxf(va.y * sc.y + sh.y, 0.f);
              va.z = fmaxf(va.z * sc.z + sh.z, 0.f); va.w = fmaxf(va.w * sc.w + sh.w, 0.f);
            }
            if (f_adrop && live)
              va = drop4(va, a_key, (uint32_t)((abase - P.a) + slab * sa) + (pa[u] >> 2), P.a_drop_p, a_inv);
          }
          put(at, a_lds[u], va, a_kc_t);
        }
#pragma unroll
        for (int u = 0; u < kSubB; ++u) {
          bool in = true;
          if constexpr (decltype(kind)::ragged) in = kslab0 + b_k[u] < krange;
          const bool live = b_ok[u] && in;
          float4 vb;
          if constexpr (decltype(set)::value == 0) vb = f4(rb0[u]); else vb = f4(rb1[u]);
          if (decltype(kind)::ragged || edge) {
            vb.x = live ? vb.x : 0.f; vb.y = live ? vb.y : 0.f;
            vb.z = live ? vb.z : 0.f; vb.w = live ? vb.w : 0.f;
          }
          if constexpr (FX) {
            if (b_aff && live) {
              vb.x = fmaxf(vb.x * bsc[u].x + bsh[u].x, 0.f); vb.y = fmaxf(vb.y * bsc[u].y + bsh[u].y, 0.f);
              vb.z = fmaxf(vb.z * bsc[u].z + bsh[u].z, 0.f); vb.w = fmaxf(vb.w * bsc[u].w + bsh[u].w, 0.f);
            }
            if (f_bdrop && live)
              vb = drop4(vb, b_key, (uint32_t)((bbase - P.b) + slab * sb) + (pb[u] >> 2), P.b_drop_p, b_inv);
            if (in) {   // the ones-row is 1 for every k inside the slice
              if (ones_e[u] == 4) vb = make_float4(1.f, 1.f, 1.f, 1.f);
              else if (ones_e[u] == 0) vb.x = 1.f;
              else if (ones_e[u] == 1) vb.y = 1.f;
              else if (ones_e[u] == 2) vb.z = 1.f;
              else if (ones_e[u] == 3) vb.w = 1.f;
            }
          }
          put(bt, b_lds[u], vb, b_kc_t);
        }
      };
      // operand fragments of this wave: row index inside the tile, 16-wide sub-slab u16
      auto frag_a = [&](const float *t, int row, int k0) -> f32x4 {
        if constexpr (a_kc) return *reinterpret_cast<const f32x4 *>(t + row * kLd + k0);
        else return (f32x4){t[(k0 + 0) * kLdTA + row], t[(k0 + 1) * kLdTA + row], t[(k0 + 2) * kLdTA + row],
                            t[(k0 + 3) * kLdTA + row]};
      };
      auto frag_b = [&](const float *t, int row, int k0) -> f32x4 {
        if constexpr (b_kc) return *reinterpret_cast<const f32x4 *>(t + row * kLd + k0);
        else return (f32x4){t[(k0 + 0) * kLdTB + row], t[(k0 + 1) * kLdTB + row], t[(k0 + 2) * kLdTB + row],
                            t[(k0 + 3) * kLdTB + row]};
      };
      auto mfma_fast = [&](int buf) {
        if (GEMM_ABL & 1) return;
        const float *at = As(buf), *bt = Bs(buf);
        if constexpr (BF) {
          const __bf16 *ah = reinterpret_cast<const __bf16 *>(at), *bh = reinterpret_cast<const __bf16 *>(bt);
          bf16x8 af[kMI], bf[kNJ];
#pragma unroll
          for (int i = 0; i < kMI; ++i)
            af[i] = *reinterpret_cast<const bf16x8 *>(ah + (wr * kWM + i * 16 + fr) * kLdH + fg * 8);
#pragma unroll
          for (int j = 0; j < kNJ; ++j)
            bf[j] = *reinterpret_cast<const bf16x8 *>(bh + (wc * kWN + j * 16 + fr) * kLdH + fg * 8);
#pragma unroll
          for (int i = 0; i < kMI; ++i)
#pragma unroll
            for (int j = 0; j < kNJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
          return;
        }
#pragma unroll
        for (int u = 0; u < kBK / 16; ++u) {
          f32x4 af[kMI], bf[kNJ];
#pragma unroll
          for (int i = 0; i < kMI; ++i) af[i] = frag_a(at, wr * kWM + i * 16 + fr, u * 16 + fg * 4);
#pragma unroll
          for (int j = 0; j < kNJ; ++j) bf[j] = frag_b(bt, wc * kWN + j * 16 + fr, u * 16 + fg * 4);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < kMI; ++i)
#pragma unroll
              for (int j = 0; j < kNJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
      };
      const int nslab = (krange + kBK - 1) / kBK;
      const int nwhole = krange / kBK;
      auto fetch_any = [&](auto set, int slab) {
        if (!FX || slab < nwhole) fetch_fast(set, slab, Whole());
        else fetch_fast(set, slab, Ragged());
      };
      auto commit_any = [&](auto set, int slab) {
        if (!FX || slab < nwhole) commit_fast(set, slab, slab & 1, Whole());
        else commit_fast(set, slab, slab & 1, Ragged());
      };
      if constexpr (PIPE == 2) {
      // iteration i:  issue loads of slab i+2 | LDS <- slab i+1 (registers of the previous iteration) |
      // MFMAs of slab i | barrier.  Nothing ever waits for a load issued in the same iteration.
      fetch_any(Set0(), 0);
      if (nslab > 1) fetch_any(Set1(), 1);
      commit_any(Set0(), 0);
      lds_barrier();
      auto step = [&](int sl, auto r_load, auto r_commit) {
        if (sl + 2 < nslab && !(GEMM_ABL & 2)) fetch_any(r_load, sl + 2);
        if (sl + 1 < nslab && !(GEMM_ABL & 4)) commit_any(r_commit, sl + 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_fast(sl & 1);
        if (!(GEMM_ABL & 8)) lds_barrier();
      };
      for (int sl = 0; sl < nslab; sl += 2) {
        step(sl, Set0(), Set1());
        if (sl + 1 < nslab) step(sl + 1, Set1(), Set0());
      }
      } else {
      fetch_any(Set0(), 0);
      commit_any(Set0(), 0);
      __syncthreads();
      for (int sl = 0; sl < nslab; ++sl) {
        const int nx = sl + 1;
        if (nx < nslab) {
          if (!(GEMM_ABL & 2)) fetch_any(Set0(), nx);
          __builtin_amdgcn_sched_barrier(0);
          mfma_fast(sl & 1);
          __builtin_amdgcn_sched_barrier(0);
          if (!(GEMM_ABL & 4)) commit_any(Set0(), nx);
        } else {
          mfma_fast(sl & 1);
        }
        if (!(GEMM_ABL & 8)) __syncthreads();
      }
      }
    };   // run_fast
    typedef std::true_type T_;
    typedef std::false_type F_;
    switch ((rt_a_kc ? 1 : 0) | (rt_b_kc ? 2 : 0) | (rt_fx ? 4 : 0)) {
      case 0: run_fast(F_(), F_(), F_()); break;
      case 1: run_fast(T_(), F_(), F_()); break;
      case 2: run_fast(F_(), T_(), F_()); break;
      case 3: run_fast(T_(), T_(), F_()); break;
      case 4: run_fast(F_(), F_(), T_()); break;
      case 5: run_fast(T_(), F_(), T_()); break;
      case 6: run_fast(F_(), T_(), T_()); break;
      default: run_fast(T_(), T_(), T_()); break;
    }
  } else {
    // generic: double-buffered LDS, one barrier per slab: slab i+1 travels global -> registers while
    // slab i is multiplied, then lands (transposed if need be) in the other [row][k] buffer
    Frag4 fa[kSubA], fb[kSubB];
    const bool a_kc = P.lda_k == 1, b_kc = P.ldb_k == 1;
    int kfetched = kbeg;
    const OperandFx fxa = {P.a2 != nullptr, P.a2_mode, P.a2_scale, P.a_chan_scale, P.a_chan_shift, true,
                           P.a_drop_p, a_inv, a_key, P.lda_m};
    const OperandFx fxb = {false, 0, 0.f, P.b_chan_scale, P.b_chan_shift, false,
                           P.b_drop_p, b_inv, b_key, P.ldb_n};
    auto fetch = [&](int k0) {
      kfetched = k0;
#pragma unroll
      for (int u = 0; u < kSubA; ++u) {
        int row, k;
        slab_pos<TM>(a_kc, tid + u * kThreads, row, k);
        fa[u] = fetch_generic<true>(P.a, P.a2, a_kc, P.lda_m, P.lda_k, m0 + row, P.M, k0 + k, kend);
      }
#pragma unroll
      for (int u = 0; u < kSubB; ++u) {
        int row, k;
        slab_pos<TN>(b_kc, tid + u * kThreads, row, k);
        fb[u] = fetch_generic<false>(P.b, nullptr, b_kc, P.ldb_n, P.ldb_k, n0 + row, P.N, k0 + k, kend);
      }
    };
    auto commit = [&](int buf) {
#pragma unroll
      for (int u = 0; u < kSubA; ++u) {
        int row, k;
        slab_pos<TM>(a_kc, tid + u * kThreads, row, k);
        commit_generic(As(buf), fa[u], fxa, a_kc, P.lda_k, m0, P.M, kfetched, kend, false, row, k);
      }
#pragma unroll
      for (int u = 0; u < kSubB; ++u) {
        int row, k;
        slab_pos<TN>(b_kc, tid + u * kThreads, row, k);
        commit_generic(Bs(buf), fb[u], fxb, b_kc, P.ldb_k, n0, P.N, kfetched, kend, ones, row, k);
      }
    };
    auto mfma_slab = [&](int buf) {
      const float *at = As(buf), *bt = Bs(buf);
#pragma unroll
      for (int u = 0; u < kBK / 16; ++u) {
        f32x4 af[kMI], bf[kNJ];
#pragma unroll
        for (int i = 0; i < kMI; ++i)
          af[i] = *reinterpret_cast<const f32x4 *>(at + (wr * kWM + i * 16 + fr) * kLd + u * 16 + fg * 4);
#pragma unroll
        for (int j = 0; j < kNJ; ++j)
          bf[j] = *reinterpret_cast<const f32x4 *>(bt + (wc * kWN + j * 16 + fr) * kLd + u * 16 + fg * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < kMI; ++i)
#pragma unroll
            for (int j = 0; j < kNJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
      }
    };
    fetch(kbeg);
    commit(0);
    __syncthreads();
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += kBK) {
      const bool more = k0 + kBK < kend;
      if (more) fetch(k0 + kBK);
      mfma_slab(cur);
      if (more) commit(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }

  // epilogue: lane holds C[row = fg*4 + r][col = fr] of each 16x16 tile.  Everything that is LOADED
  // (bias, RNG counter) is fetched before the first store: the output may alias nothing here, but the
  // compiler cannot know, and a load issued after a store waits for it (16 serialized L2 round trips
  // made the epilogue cost more than the whole K loop).
  if ((GEMM_ABL & 16) && acc[0][0][0] != 12345.678f) return;
  const bool drop = P.dropout_p > 0.f;
  const float inv_keep = drop ? 1.f / (1.f - P.dropout_p) : 1.f;
  const uint64_t ctr = step_ctr;
  float *const cptr = P.c;
  float *const bgrad = P.bias_grad;
  const int pM = P.M, pN = P.N, relu = P.relu, accumulate = P.accumulate, ones_col = P.ones_col;
  const long ldc = P.ldc;
  const float scale = P.scale, p_drop = P.dropout_p;
  const uint32_t site = P.dropout_site;
  if (!accumulate && !ones_col) {
    // Plain stores: stage the TM x TN tile through LDS (the operand buffers are free after the last
    // barrier) so every thread writes whole float4 row segments -- the MFMA C-layout would otherwise
    // emit sixteen 4-byte stores per lane, 64 contiguous bytes per wave-instruction.
    constexpr int kLdC = TN + 4;
    constexpr int kRQ = TN / 4;                   // float4 per tile row
    constexpr int kRowPhases = kThreads / kRQ;    // rows written per pass (threads beyond kRQ * kRowPhases idle)
    static_assert(2 * (TM + TN) * kLd >= TM * kLdC && 2 * (TM + TN) * kLd >= 2 * kRowPhases * TN,
                  "C tile / statistics scratch must fit the operand buffers");
    float *Cs = lds;
#pragma unroll
    for (int i = 0; i < kMI; ++i)
#pragma unroll
      for (int j = 0; j < kNJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Cs[(wr * kWM + i * 16 + fg * 4 + r) * kLdC + wc * kWN + j * 16 + fr] = acc[i][j][r];
    __syncthreads();
    const int c4 = (tid % kRQ) * 4, rphase = tid / kRQ;
    const bool active = rphase < kRowPhases;
    const int n = n0 + c4;
    // Lean path (uniform per workgroup): a whole tile inside the matrix, 16-byte aligned rows, no output dropout, no
    // second destination.  fp32 matrix and vector instructions share their issue slots on this part
    // (scratch/ubench/mfma_valu.hip), so every instruction of this epilogue is paid in matrix time: per float4 it is
    // one ds_read_b128, 4 fma (+4 max, +8 for the column statistics, +4 for an accumulating store) and one store.
    if (m0 + TM <= pM && n0 + TN <= pN && !drop && P.c2 == nullptr && (ldc & 3) == 0 &&
        (((uintptr_t)cptr | (uintptr_t)P.bias | (uintptr_t)P.c_gate | (uintptr_t)P.c_bn_z | (uintptr_t)P.c_bn_aff) & 15) == 0 &&
        (P.c_bn_ld & 3) == 0) {
      double *const col_sum = P.col_sum, *const col_sumsq = P.col_sumsq;
      const bool c_add = P.c_add != 0;
      const bool streaming = !c_add && (long)pM * pN >= (16L << 20);   // >= 64 MB
      f32x4 bs = {0.f, 0.f, 0.f, 0.f};
      if (P.bias && slice == 0 && active) {
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(P.bias + n);   // n % 4 == 0; bias rows are 16-byte aligned
        bs = b4 * scale;
      }
      f32x4 cs = {0.f, 0.f, 0.f, 0.f}, cq = {0.f, 0.f, 0.f, 0.f};
      const float *const gate = P.c_gate;
      const float gate_scale = P.c_gate_scale;
      // c_bn: the gate is the ReLU of a BatchNorm (scale * z + shift > 0, z = the saved pre-activation in C's layout) and
      // the statistics are the two column sums its backward needs: sum g, sum g * zhat
      const float *const bnz = P.c_bn_z;
      f32x4 bn_mu = {0.f, 0.f, 0.f, 0.f}, bn_rs = bn_mu, bn_sc = bn_mu, bn_sh = bn_mu;
      if (bnz && active) {
        const float *af = P.c_bn_aff + n;
        bn_mu = *reinterpret_cast<const f32x4 *>(af);
        bn_rs = *reinterpret_cast<const f32x4 *>(af + P.c_bn_ld);
        bn_sc = *reinterpret_cast<const f32x4 *>(af + 2 * P.c_bn_ld);
        bn_sh = *reinterpret_cast<const f32x4 *>(af + 3 * P.c_bn_ld);
      }
      // ... with a Dropout behind that ReLU (models/modules.py:64-72): its mask, regenerated as butd_mlp_mask_stats did
      const float bn_p = P.c_bn_drop_p;
      const bool bn_drop = bnz && bn_p > 0.f;
      const float bn_inv = bn_drop ? 1.f / (1.f - bn_p) : 1.f;
      const uint32_t bn_key = bn_drop ? rng::site_key(ctr, P.c_bn_drop_site) : 0u;
      auto rows = [&](auto relu_t, auto stats_t, auto add_t) {
        if (!active) return;
        float *dst = cptr + (long)(m0 + rphase) * ldc + n;
        const long step = (long)kRowPhases * ldc;
#pragma unroll 4
        for (int row = rphase; row < TM; row += kRowPhases, dst += step) {
          f32x4 v = *reinterpret_cast<const f32x4 *>(&Cs[row * kLdC + c4]) * scale + bs;
          if constexpr (decltype(relu_t)::value) {
            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
          }
          if (gate) {   // (uniform) ReLU / dropout backward of the gradient this product creates
            const f32x4 gt = *reinterpret_cast<const f32x4 *>(gate + (dst - cptr));
            v[0] = gt[0] > 0.f ? v[0] * gate_scale : 0.f; v[1] = gt[1] > 0.f ? v[1] * gate_scale : 0.f;
            v[2] = gt[2] > 0.f ? v[2] * gate_scale : 0.f; v[3] = gt[3] > 0.f ? v[3] * gate_scale : 0.f;
          }
          if constexpr (decltype(stats_t)::value) {
            if (bnz) {   // (uniform)
              const f32x4 z4 = *reinterpret_cast<const f32x4 *>(bnz + (dst - cptr));
              const f32x4 pre = z4 * bn_sc + bn_sh;
              v[0] = pre[0] > 0.f ? v[0] : 0.f; v[1] = pre[1] > 0.f ? v[1] : 0.f;
              v[2] = pre[2] > 0.f ? v[2] : 0.f; v[3] = pre[3] > 0.f ? v[3] : 0.f;
              if (bn_drop) {   // (uniform)
                const uint32_t i0 = (uint32_t)(dst - cptr);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = rng::keep_keyed(bn_key, i0 + (uint32_t)e, bn_p) ? v[e] * bn_inv : 0.f;
              }
              cs += v;
              cq += v * ((z4 - bn_mu) * bn_rs);
            } else {
              cs += v;
              cq += v * v;
            }
          }
          if constexpr (decltype(add_t)::value) v += *reinterpret_cast<const f32x4 *>(dst);
          store_c4(dst, make_float4(v[0], v[1], v[2], v[3]), streaming);
        }
      };
      typedef std::true_type T_;
      typedef std::false_type F_;
      switch ((relu ? 1 : 0) | (col_sum ? 2 : 0) | (c_add ? 4 : 0)) {
        case 0: rows(F_(), F_(), F_()); break;
        case 1: rows(T_(), F_(), F_()); break;
        case 2: rows(F_(), T_(), F_()); break;
        case 3: rows(T_(), T_(), F_()); break;
        case 4: rows(F_(), F_(), T_()); break;
        case 5: rows(T_(), F_(), T_()); break;
        case 6: rows(F_(), T_(), T_()); break;
        default: rows(T_(), T_(), T_()); break;
      }
      if (col_sum) {
        float *red = lds;
        __syncthreads();
        if (active) {
          *reinterpret_cast<f32x4 *>(&red[(0 * kRowPhases + rphase) * TN + c4]) = cs;
          *reinterpret_cast<f32x4 *>(&red[(1 * kRowPhases + rphase) * TN + c4]) = cq;
        }
        __syncthreads();
        if (tid < 2 * TN) {
          const int which = tid / TN, col = tid % TN;
          double s = 0.0;
#pragma unroll
          for (int r = 0; r < kRowPhases; ++r) s += (double)red[(which * kRowPhases + r) * TN + col];
          const long slot_off = P.col_slots > 1 ? (long)(blockIdx.x & (P.col_slots - 1)) * P.col_slot_stride : 0;
          atomicAdd((which ? col_sumsq : col_sum) + slot_off + n0 + col, s);
        }
      }
      return;
    }
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (P.bias && slice == 0 && active) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < pN) bv[e] = P.bias[n + e];
    }
    const bool vec_ok = (n + 3 < pN) && ((ldc & 3) == 0) && ((((uintptr_t)cptr) & 15) == 0);
    double *const col_sum = P.col_sum, *const col_sumsq = P.col_sumsq;
    const bool c_add = P.c_add != 0;
    float *const c2ptr = P.c2;
    const bool vec2_ok = vec_ok && ((((uintptr_t)c2ptr) & 15) == 0);
    const bool streaming = !c_add && (long)pM * pN >= (16L << 20);   // >= 64 MB
    float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
#pragma unroll 4
      for (int row = rphase; row < TM; row += kRowPhases) {
        const int m = m0 + row;
        if (m >= pM || n >= pN) continue;
        const float4 cv = *reinterpret_cast<const float4 *>(&Cs[row * kLdC + c4]);
        float v[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = (v[e] + bv[e]) * scale;
          if (relu) v[e] = fmaxf(v[e], 0.f);
          if (drop)
            v[e] = rng::keep(ctr, site, (uint32_t)((long)m * pN + n + e), p_drop) ? v[e] * inv_keep : 0.f;
          if (P.c_gate && n + e < pN) v[e] = P.c_gate[(long)m * ldc + n + e] > 0.f ? v[e] * P.c_gate_scale : 0.f;
          if (n + e < pN) {
            if (P.c_bn_z) {
              const float *af = P.c_bn_aff + n + e;
              const float z = P.c_bn_z[(long)m * ldc + n + e];
              v[e] = (z * af[2 * P.c_bn_ld] + af[3 * P.c_bn_ld] > 0.f) ? v[e] : 0.f;
              if (P.c_bn_drop_p > 0.f)
                v[e] = rng::keep(ctr, P.c_bn_drop_site, (uint32_t)((long)m * ldc + n + e), P.c_bn_drop_p)
                           ? v[e] * (1.f / (1.f - P.c_bn_drop_p)) : 0.f;
              cs[e] += v[e];
              cq[e] += v[e] * ((z - af[0]) * af[P.c_bn_ld]);
            } else {
              cs[e] += v[e];
              cq[e] += v[e] * v[e];
            }
          }
        }
        float *dst = cptr + (long)m * ldc + n;
        if (c2ptr) {   // second destination accumulates the same values
          float *d2 = c2ptr + (long)m * ldc + n;
          if (vec2_ok) {
            const float4 o = *reinterpret_cast<const float4 *>(d2);
            *reinterpret_cast<float4 *>(d2) = make_float4(o.x + v[0], o.y + v[1], o.z + v[2], o.w + v[3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < pN) d2[e] += v[e];
          }
        }
        if (vec_ok) {
          if (c_add) {
            const float4 o = *reinterpret_cast<const float4 *>(dst);
            v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
          }
          store_c4(dst, make_float4(v[0], v[1], v[2], v[3]), streaming);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < pN) dst[e] = c_add ? dst[e] + v[e] : v[e];
        }
      }
    }
    if (col_sum) {
      // column sums of the tile: one partial per row phase and column through LDS (over the C tile, once
      // every thread has read its part of it), then one double atomic per column and statistic
      float *red = lds;
      __syncthreads();
      if (active) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          red[(0 * kRowPhases + rphase) * TN + c4 + e] = cs[e];
          red[(1 * kRowPhases + rphase) * TN + c4 + e] = cq[e];
        }
      }
      __syncthreads();
      if (tid < 2 * TN) {
        const int which = tid / TN, col = tid % TN;
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < kRowPhases; ++r) s += (double)red[(which * kRowPhases + r) * TN + col];
        const long slot_off = P.col_slots > 1 ? (long)(blockIdx.x & (P.col_slots - 1)) * P.col_slot_stride : 0;
        if (n0 + col < pN) atomicAdd((which ? col_sumsq : col_sum) + slot_off + n0 + col, s);
      }
    }
    return;
  }
  if (float *const part = P.c_partial) {
    // deterministic split-K: this slice's share as plain stores into its own slab (dense [M][N] + the ones-column's
    // M results behind it), staged through LDS for float4 row segments; a later launch folds the slabs in slice order
    constexpr int kLdC = TN + 4;
    constexpr int kRQ = TN / 4;
    constexpr int kRowPhases = kThreads / kRQ;
    static_assert(2 * (TM + TN) * kLd >= TM * kLdC, "C tile must fit the operand buffers");
    float *Cs = lds;
#pragma unroll
    for (int i = 0; i < kMI; ++i)
#pragma unroll
      for (int j = 0; j < kNJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Cs[(wr * kWM + i * 16 + fg * 4 + r) * kLdC + wc * kWN + j * 16 + fr] = acc[i][j][r];
    __syncthreads();
    const int c4 = (tid % kRQ) * 4, rphase = tid / kRQ;
    if (rphase >= kRowPhases) return;
    float *const base = part + (long)slice * P.c_partial_stride;
    float *const bslab = base + (long)pM * pN;
    const int n = n0 + c4;
    const bool vec = (pN & 3) == 0 && n + 3 < pN;
    if (n > pN || (n == pN && !ones_col)) return;
    for (int row = rphase; row < TM; row += kRowPhases) {
      const int m = m0 + row;
      if (m >= pM) break;
      const f32x4 v = *reinterpret_cast<const f32x4 *>(&Cs[row * kLdC + c4]) * scale;
      float *dst = base + (long)m * pN + n;
      if (vec) {
        *reinterpret_cast<f32x4 *>(dst) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e < pN) dst[e] = v[e];
          else if (ones_col && n + e == pN) bslab[m] = v[e];
        }
      }
    }
    return;
  }
  // accumulate / bias-gradient path: element-wise atomics straight from the accumulators
  float bias_v[kNJ];
#pragma unroll
  for (int j = 0; j < kNJ; ++j) {
    const int n = n0 + wc * kWN + j * 16 + fr;
    bias_v[j] = (P.bias && slice == 0 && n < pN) ? P.bias[n] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < kMI; ++i)
#pragma unroll
    for (int j = 0; j < kNJ; ++j) {
      const int n = n0 + wc * kWN + j * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wr * kWM + i * 16 + fg * 4 + r;
        if (m >= pM) continue;
        float v = acc[i][j][r];
        if (n < pN) {
          v = (v + bias_v[j]) * scale;
          if (relu) v = fmaxf(v, 0.f);
          if (drop)
            v = rng::keep(ctr, site, (uint32_t)((long)m * pN + n), p_drop) ? v * inv_keep : 0.f;
          atomicAdd(cptr + (long)m * ldc + n, v);
        } else if (ones_col && n == pN) {
          atomicAdd(bgrad + m, v * scale);
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// host side: tile menu and launch
// ------------------------------------------------------------------------------------------------
struct TileCfg {
  int tm, tn;
};
// the menu (every entry is instantiated for FAST / PIPE 2, FAST / PIPE 0 and generic)
constexpr TileCfg kMenu[] = {{32, 32}, {64, 64}, {32, 96}, {64, 96}, {96, 32}, {128, 64}, {128, 96}};
constexpr int kMenuSize = sizeof(kMenu) / sizeof(kMenu[0]);
constexpr int kCfg32x32 = 0, kCfg64x64 = 1, kCfg32x96 = 2, kCfg64x96 = 3, kCfg96x32 = 4;   // (5 = 128 x 64, 6 = 128 x 96: forced tiles only)

int g_forced_cfg = -1;   // butd_gemm_set_tile(): tuning hook

bool fast_eligible(const butd_gemm_problem &p) {
  const bool a_kc = p.lda_k == 1, b_kc = p.ldb_k == 1;
  const int kslab = (p.K + kBK - 1) / kBK, split = p.split_k < 1 ? 1 : p.split_k;
  const long per = (long)((kslab + split - 1) / split) * kBK;   // contraction range of one slice
  // (a companion operand a2 stays on the generic kernel: its extra register set cost the fast
  // instantiations a wave of occupancy, measured)
  return p.a2 == nullptr && p.K > 0 && (p.K & 3) == 0 &&   // a ragged LAST slab is predicated per float4
         (a_kc || (p.M & 3) == 0) && (b_kc || (p.N & 3) == 0) &&   // partial tiles: whole float4 in or out
         ((a_kc ? p.lda_m : p.lda_k) & 3) == 0 && ((b_kc ? p.ldb_n : p.ldb_k) & 3) == 0 &&
         ((((uintptr_t)p.a) | ((uintptr_t)p.b)) & 15) == 0 &&
         ((p.a_chan_scale == nullptr && p.a_bn_sum == nullptr) || per <= kAffK);
}

long fill_batch(GemmBatch &batch, const butd_gemm_problem *problems, const int *index, int count,
                int tile_m, int tile_n) {
  long total = 0;
  batch.count = 0;
  for (int i = 0; i < count; ++i) {
    butd_gemm_problem p = problems[index[i]];
    if (p.split_k < 1) p.split_k = 1;
    const int ncols = p.N + (p.ones_col ? 1 : 0);
    int tn = (ncols + tile_n - 1) / tile_n, tm = (p.M + tile_m - 1) / tile_m;
    if (p.fold_src) {       // element-wise: one workgroup per 2048 floats of (weights | bias)
      tn = (int)((p.fold_len + p.fold_len2 + kFoldChunk - 1) / kFoldChunk);
      tm = 1;
      p.split_k = 1;
    }
    batch.blk_begin[batch.count] = (int)total;
    batch.tiles_n[batch.count] = tn;
    batch.tiles_m[batch.count] = tm;
    batch.p[batch.count++] = p;
    total += (long)tn * tm * p.split_k;
    if (total > 0x7fffffffL) return -1;
  }
  for (int i = batch.count; i <= kMaxProblems; ++i) batch.blk_begin[i] = (int)total;
  return total;
}

// Tile and K-loop pipeline of a launch: rules read off graph-replay timings of the training step's launches
// under every (tile, pipeline) pair of the menu (scratch/gemm_cases.py -> profiles/r02_gemm_tiles.txt).
//  * launches with a split-K (weight-gradient) problem: 96 x 32, one slab ahead -- the ones-column costs a
//    narrow column tile, the [k][row] fragments (four ds_read_b32 each) favour few B fragments (a 8192-row
//    attention block's input+weight gradient pair: 43 us vs 50 us with round 1's 32 x 32); 32 x 32 when the
//    launch is a decoder-sized one (<= 2048 rows); 64 x 64 for the set-abstraction ones, which contract
//    10^5..10^6 rows per slice;
//  * tall plain-store problems (set-abstraction forward, M >= 32768): 128 x 64, one slab ahead (with the lean
//    epilogue the fixed cost per tile matters more than the number of workgroups: 294 vs 321 us on 1M x 128 x 64);
//  * projections / FFN / input gradients of the attention stack (640..8192 rows, N and K <= 864): by the
//    amount of work, counted in 32 x 32 tiles: small launches want many small workgroups (latency-bound: a
//    2048 x 288 x 288 product is 10 us on 32 x 32, 13 us on 64 x 64), large ones want the 96-wide tiles that
//    cover N = 288 without a ragged column and stage fewer bytes per flop.
void choose(const butd_gemm_problem *problems, const int *index, int count, bool fast, int &cfg, int &pipe) {
  if (!fast) {   // generic staging (a2 companion, unaligned / ragged operands): element-wise, small tiles
    cfg = kCfg32x32;
    pipe = 0;
    return;
  }
  bool any_acc = false, n96 = true;
  long max_k_acc = 0, max_m_plain = 0, tiles32 = 0;
  for (int i = 0; i < count; ++i) {
    const butd_gemm_problem &p = problems[index[i]];
    if (p.fold_src) continue;      // (element-wise riders do not vote on the tile)
    if (p.accumulate || p.split_k > 1 || p.ones_col) {
      any_acc = true;
      if (p.K > max_k_acc) max_k_acc = p.K;
    } else {
      if (p.M > max_m_plain) max_m_plain = p.M;
      if (p.N % 96) n96 = false;
      tiles32 += (long)((p.M + 31) / 32) * ((p.N + 31) / 32);
    }
  }
  if (any_acc) {
    pipe = 0;
    cfg = max_k_acc >= 32768 ? kCfg64x64 : (max_m_plain > 0 && max_m_plain <= 2048) ? kCfg32x32 : kCfg96x32;
    return;
  }
  if (max_m_plain >= 32768) {
    // round 5 (accumulators in VGPRs, profiles/r05_gemm_tiles.txt): 64 x 64 one-ahead beats 128 x 64 on every tall
    // set-abstraction forward product (1M x 128 x 64: 276 vs 308 us, 256k x 128 x 128: 114 vs 125, 64k x 256 x 128: 60 vs 65)
    pipe = 0;
    cfg = kCfg64x64;                    // (A/B in the step: profiles/r05_gemm_choose.txt)
    return;
  }
  if (tiles32 <= 1200 || (count > 1 && tiles32 <= 3000 && max_m_plain > 2048)) {
    pipe = 0;   // (a 8192-row product grouped with two 640..1056-row ones: 31 us here, 38 us on 32 x 96)
    cfg = kCfg32x32;
  } else if (tiles32 <= 2000 || !n96) {
    pipe = 2;
    cfg = kCfg64x64;
  } else {
    pipe = 2;
    cfg = tiles32 <= 3000 ? kCfg32x96 : kCfg64x96;
    if (tiles32 > 3000) pipe = 0;   // round 5: 3 x (8192 x 288 x 288): 51.4 us one-ahead, 53.6 two-ahead
  }
}

template <int TM, int TN>
void launch_cfg(const GemmBatch &batch, long total, bool fast, int pipe, bool bf16, const uint64_t *rng_counter,
                hipStream_t stream) {
  const dim3 grid((unsigned)total), blk(kThreads);
  if (fast && bf16 && pipe == 2)
    hipLaunchKernelGGL((gemm_kernel<TM, TN, true, 2, true>), grid, blk, 0, stream, batch, rng_counter);
  else if (fast && bf16)
    hipLaunchKernelGGL((gemm_kernel<TM, TN, true, 0, true>), grid, blk, 0, stream, batch, rng_counter);
  else if (fast && pipe == 2)
    hipLaunchKernelGGL((gemm_kernel<TM, TN, true, 2>), grid, blk, 0, stream, batch, rng_counter);
  else if (fast)
    hipLaunchKernelGGL((gemm_kernel<TM, TN, true, 0>), grid, blk, 0, stream, batch, rng_counter);
  else hipLaunchKernelGGL((gemm_kernel<TM, TN, false, 0>), grid, blk, 0, stream, batch, rng_counter);
}

int g_forced_pipe = -1;

int launch_group(const butd_gemm_problem *problems, const int *index, int count, bool fast,
                 const uint64_t *rng_counter, hipStream_t stream) {
  if (count == 0) return 0;
  int best = 0, pipe = 2;
  choose(problems, index, count, fast, best, pipe);
  if (g_forced_cfg >= 0) {
    best = g_forced_cfg;
    pipe = g_forced_pipe >= 0 ? g_forced_pipe : 2;
  }
  bool bf16 = fast;
  for (int i = 0; i < count; ++i)
    if (!problems[index[i]].fold_src) bf16 = bf16 && problems[index[i]].compute_bf16 != 0;
  GemmBatch batch;
  const long total = fill_batch(batch, problems, index, count, kMenu[best].tm, kMenu[best].tn);
  if (total < 0) return (int)hipErrorInvalidValue;
  if (total == 0) return 0;
  switch (best) {
    case 0: launch_cfg<32, 32>(batch, total, fast, pipe, bf16, rng_counter, stream); break;
    case 1: launch_cfg<64, 64>(batch, total, fast, pipe, bf16, rng_counter, stream); break;
    case 2: launch_cfg<32, 96>(batch, total, fast, pipe, bf16, rng_counter, stream); break;
    case 3: launch_cfg<64, 96>(batch, total, fast, pipe, bf16, rng_counter, stream); break;
    case 4: launch_cfg<96, 32>(batch, total, fast, pipe, bf16, rng_counter, stream); break;
    case 5: launch_cfg<128, 64>(batch, total, fast, pipe, bf16, rng_counter, stream); break;
    default: launch_cfg<128, 96>(batch, total, fast, pipe, bf16, rng_counter, stream); break;
  }
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int butd_gemm_set_tile(int tile_m, int tile_n) {
  if (tile_m <= 0 || tile_n == 0) {
    g_forced_cfg = g_forced_pipe = -1;
    return 0;
  }
  // a negative tile_n asks for the one-slab-ahead K loop
  const int pipe = tile_n < 0 ? 0 : 2;
  if (tile_n < 0) tile_n = -tile_n;
  for (int c = 0; c < kMenuSize; ++c)
    if (kMenu[c].tm == tile_m && kMenu[c].tn == tile_n) {
      g_forced_cfg = c;
      g_forced_pipe = pipe;
      return 0;
    }
  return (int)hipErrorInvalidValue;
}

int butd_gemm_grouped(const butd_gemm_problem *problems, int count, const uint64_t *rng_counter,
                      butd_stream_t stream) {
  if (count <= 0) return 0;
  if (count > kMaxProblems) return (int)hipErrorInvalidValue;
  // the problems of a group are independent: the fast-eligible ones and the rest run as two launches
  int fast_idx[kMaxProblems], slow_idx[kMaxProblems], fold_idx[kMaxProblems], nf = 0, ns = 0, nfold = 0;
  for (int i = 0; i < count; ++i) {
    const butd_gemm_problem &p = problems[i];
    if (p.fold_src) {
      if (!p.c || p.fold_count < 1 || p.fold_len <= 0 || (p.fold_len & 3) || p.fold_len2 < 0 || (p.fold_stride & 3) ||
          (p.fold_len2 > 0 && !p.bias_grad) || (((uintptr_t)p.fold_src) & 15) || p.c_partial)
        return (int)hipErrorInvalidValue;
      fold_idx[nfold++] = i;
      continue;
    }
    if (p.M <= 0 || p.N <= 0) continue;
    if (p.c_partial && (!p.accumulate || p.bias || p.relu || p.dropout_p > 0.f || (((uintptr_t)p.c_partial) & 15) ||
                        (p.c_partial_stride & 3) || p.c_partial_stride < (long)p.M * p.N + (p.ones_col ? p.M : 0) ||
                        p.split_k < 1))
      return (int)hipErrorInvalidValue;
    if (p.c_partial) {      // every slice must own a slab of the contraction: an empty one would leave its share unwritten
      const int kslab = (p.K + kBK - 1) / kBK, per = (kslab + p.split_k - 1) / p.split_k;
      if ((long)(p.split_k - 1) * per >= kslab) return (int)hipErrorInvalidValue;
    }
    if (p.split_k > 1 && !p.accumulate) return (int)hipErrorInvalidValue;
    if ((p.col_sum != nullptr || p.c_add || p.c2 != nullptr || p.c_gate != nullptr) && (p.accumulate || p.ones_col || p.split_k > 1))
      return (int)hipErrorInvalidValue;
    if (p.col_slots > 1 && (p.col_slots & (p.col_slots - 1))) return (int)hipErrorInvalidValue;
    if (p.c_bn_z && (!p.c_bn_aff || !p.col_sum || !p.col_sumsq || p.relu || p.dropout_p > 0.f || p.c2 != nullptr))
      return (int)hipErrorInvalidValue;
    // (the in-kernel BatchNorm bookkeeping exists on the float4 path only, on unsplit forward products)
    if (p.a_bn_sum && (!fast_eligible(p) || p.split_k > 1 || !p.a_bn_sumsq || !p.a_bn_gamma || !p.a_bn_beta ||
                       p.a_bn_count <= 0 || p.a_chan_scale))
      return (int)hipErrorInvalidValue;
    if (fast_eligible(p)) fast_idx[nf++] = i; else slow_idx[ns++] = i;
  }
  static const bool log_slow = getenv("BUTD_GEMM_LOG_SLOW") != nullptr;   // which problems miss the float4 path (stderr)
  if (log_slow)
    for (int j = 0; j < ns; ++j) {
      const butd_gemm_problem &p = problems[slow_idx[j]];
      fprintf(stderr, "butd_gemm slow: M %d N %d K %d lda (%ld,%ld) ldb (%ld,%ld) split %d a2 %d ones %d align %d of %d in group\n",
              p.M, p.N, p.K, (long)p.lda_m, (long)p.lda_k, (long)p.ldb_n, (long)p.ldb_k, p.split_k, p.a2 != nullptr,
              (int)p.ones_col, (int)((((uintptr_t)p.a) | ((uintptr_t)p.b)) & 15), count);
    }
  // the riders join the float4 launch when there is one (else the element-wise staged one, else they are the launch)
  for (int j = 0; j < nfold; ++j) {
    if (nf > 0 || ns == 0) fast_idx[nf++] = fold_idx[j];
    else slow_idx[ns++] = fold_idx[j];
  }
  int err = launch_group(problems, fast_idx, nf, true, rng_counter, (hipStream_t)stream);
  if (err) return err;
  return launch_group(problems, slow_idx, ns, false, rng_counter, (hipStream_t)stream);
}

}  // extern "C"
