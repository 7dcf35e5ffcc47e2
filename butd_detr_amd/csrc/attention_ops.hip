// attention_ops.hip -- fp32 MFMA building blocks of the fused cross-modal attention / FFN path
// (include/butd_attention.h).  gfx950 only.
//
//   gemm_kernel          grouped dense products on v_mfma_f32_16x16x4_f32 (exact fp32, 157 TF peak):
//                        64x64 output tile per 4-wave workgroup, 32x32 per wave (2x2 MFMA tiles), BK=32,
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
#include <stdlib.h>

#include <type_traits>

#include "../../include/butd_attention.h"
#include "rng.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBK = 32, kLd = kBK + 4;  // slab depth; LDS row stride 36 floats = 9 x 16 B
// Output tiles are TILE x TILE with TILE = 64 (large grids) or 32 (grids that would leave CUs idle).
// The kernel is instantiated for 256 threads (4 waves, wave tile 32x32) and 512 threads (8 waves, wave
// tile 32x16: half the MFMA chain per wave and twice the waves per SIMD for small grids).  One staging
// step moves ONE float4 per thread and operand: a (64 rows x SW k) sub-slab, SW = THREADS/16.
constexpr int kMaxProblems = 8;

// Loads that must be emitted as global_load_*: a FLAT load also counts against lgkmcnt, so the
// s_waitcnt lgkmcnt(0) in front of the MFMAs (for the LDS fragment reads) would wait for the prefetch
// of the NEXT slab as well and serialize HBM latency with the matrix pipe.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4_t *global_f4_ptr;
__device__ inline float4 ldg4(const float *p) {
  const f32x4_t v = *reinterpret_cast<global_f4_ptr>(reinterpret_cast<uintptr_t>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
constexpr int kAffK = 320;  // contraction range whose A-operand affine is staged in LDS (fast path)

struct GemmBatch {
  butd_gemm_problem p[kMaxProblems];
  int blk_begin[kMaxProblems + 1];  // linear workgroup range of each problem
  int tiles_n[kMaxProblems], tiles_m[kMaxProblems];
  int count;
};

// Staging of a (rows x 16) operand slab, split in two so the global loads of slab i+1 are in flight
// while the MFMAs of slab i run:  fetch_tile() only ISSUES loads (raw values of the operand and of its
// optional companion a2 land in registers, nothing consumes them), commit_tile() combines and writes
// the LDS image tile[row][k].   element(row, k) = src[row*ld_row + k*ld_k]; exactly one of the two
// strides is 1 and each thread moves the float4 that is contiguous in memory: 4 consecutive k of one
// row (contraction-contiguous operand) or 4 consecutive rows of one k (row-contiguous operand, which
// commit_tile transposes).  Rows >= nrows and k >= kend read as 0, except the virtual ones-row.
struct Frag4 {
  float4 a, a2;
};

struct TileIdx {
  int slow, fast;   // position along the strided / contiguous dimension inside the slab
  bool kc;          // contraction-contiguous?
};
template <int SW, int TILE>
__device__ inline TileIdx tile_idx(long ld_k, int tid) {
  TileIdx t;
  t.kc = ld_k == 1;
  t.slow = t.kc ? (tid / (SW / 4)) : (tid / (TILE / 4));
  t.fast = t.kc ? (tid % (SW / 4)) * 4 : (tid % (TILE / 4)) * 4;
  return t;
}

template <int SW, int TILE, bool WITH_A2 = true>
__device__ inline Frag4 fetch_tile(const float *__restrict__ src, const float *__restrict__ src2,
                                   long ld_row, long ld_k, int row0, int nrows, int k0, int kend,
                                   int tid) {
  const TileIdx t = tile_idx<SW, TILE>(ld_k, tid);
  const long ld_slow = t.kc ? ld_row : ld_k;
  const int slow_g = (t.kc ? row0 : k0) + t.slow, fast_g = (t.kc ? k0 : row0) + t.fast;
  const int slow_lim = t.kc ? nrows : kend, fast_lim = t.kc ? kend : nrows;
  Frag4 f;
  f.a = make_float4(0.f, 0.f, 0.f, 0.f);
  f.a2 = f.a;
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

// Everything commit_tile() applies to a staged operand besides the plain copy.
// chan_is_k: the affine's channel index is the contraction index (A operand) or the row index (B)
struct OperandFx {
  bool has2; int mode2; float scale2;          // companion operand a2
  const float *csc, *csh; bool chan_is_k;      // per-channel affine + ReLU
  float drop_p, drop_inv; uint32_t drop_key;   // dropout keyed by the element's memory offset
  long ld_row;
};

template <int SW, int TILE>
__device__ inline void commit_tile(float (*tile)[kLd], const Frag4 &f, const OperandFx &fx, long ld_k,
                                   int row0, int nrows, int k0, int kend, bool ones, int koff, int tid) {
  const TileIdx t = tile_idx<SW, TILE>(ld_k, tid);
  float v[4] = {f.a.x, f.a.y, f.a.z, f.a.w};
  if (fx.has2) {
    v[0] = combine(v[0], f.a2.x, fx.mode2, fx.scale2); v[1] = combine(v[1], f.a2.y, fx.mode2, fx.scale2);
    v[2] = combine(v[2], f.a2.z, fx.mode2, fx.scale2); v[3] = combine(v[3], f.a2.w, fx.mode2, fx.scale2);
  }
  if (fx.csc || fx.drop_p > 0.f) {  // relu(v * scale[chan] + shift[chan]), dropout; out-of-range stays 0
    const int rbase = row0 + (t.kc ? t.slow : t.fast), kbase = k0 + (t.kc ? t.fast : t.slow);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = rbase + (t.kc ? 0 : i), k = kbase + (t.kc ? i : 0);
      if (r < nrows && k < kend) {
        if (fx.csc) {
          const int ch = fx.chan_is_k ? k : r;
          v[i] = fmaxf(v[i] * fx.csc[ch] + fx.csh[ch], 0.f);
        }
        if (fx.drop_p > 0.f) {
          const uint32_t off = (uint32_t)((long)r * fx.ld_row + (long)k * ld_k);
          v[i] = rng::keep_keyed(fx.drop_key, off, fx.drop_p) ? v[i] * fx.drop_inv : 0.f;
        }
      }
    }
  }
  if (t.kc) {
    if (ones && row0 + t.slow == nrows) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = (k0 + t.fast + i < kend) ? 1.f : 0.f;
    }
    *reinterpret_cast<float4 *>(&tile[t.slow][koff + t.fast]) = make_float4(v[0], v[1], v[2], v[3]);
  } else {  // transpose into the K-contiguous LDS image
    if (ones && k0 + t.slow < kend) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (row0 + t.fast + i == nrows) v[i] = 1.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[t.fast + i][koff + t.slow] = v[i];
  }
}

struct Whole { static constexpr bool ragged = false; };   // slab kinds of the fast path (see below)
struct Ragged { static constexpr bool ragged = true; };

// FAST: every problem of the launch satisfies fast_eligible() (host side): interior tiles stream whole
// float4s with addresses  base + slab * step  and no bounds checks; the generic instantiation handles
// ragged K, unaligned operands and the a2 companion.  Two kernels instead of one runtime branch: with
// both paths in one body the compiler merged their MFMA blocks and serialized loads behind them.
template <int THREADS, int TILE, bool FAST>
__global__ __launch_bounds__(THREADS) void gemm_kernel(GemmBatch batch,
                                                       const uint64_t *__restrict__ rng_counter) {
  constexpr int kBM = TILE, kBN = TILE;
  constexpr int kSW = 4 * THREADS / TILE;  // k-width of one staging step (one float4 per thread)
  constexpr int kSub = kBK / kSW;          // staging steps per slab
  constexpr int kWavesN = THREADS / 128;   // wave grid 2 x kWavesN
  constexpr int kMI = TILE / 32;           // 16-row fragments per wave
  constexpr int kNJ = TILE / (16 * kWavesN);   // 16-column fragments per wave
  constexpr int kRQ = TILE / 4;            // float4 per tile row / per k-row of a row-contiguous operand
  constexpr int kRowPhases = THREADS / kRQ;    // rows written per epilogue pass
  static_assert(kSub >= 1 && kNJ >= 1 && kMI >= 1 && kRowPhases <= TILE, "unsupported THREADS x TILE");
  __shared__ __attribute__((aligned(16))) float As[2][kBM][kLd];
  __shared__ __attribute__((aligned(16))) float Bs[2][kBN][kLd];
  __shared__ __attribute__((aligned(16))) float Asc[kAffK], Ash[kAffK];

  // 1-D grid: every problem owns exactly tiles_n x tiles_m x split_k consecutive workgroups
  int pi = 0;
  while (pi + 1 < batch.count && (int)blockIdx.x >= batch.blk_begin[pi + 1]) ++pi;
  const butd_gemm_problem &P = batch.p[pi];
  int rel = blockIdx.x - batch.blk_begin[pi];
  const int tn = batch.tiles_n[pi], tm = batch.tiles_m[pi];
  const int bx = rel % tn;
  rel /= tn;
  const int by = rel % tm;
  const int slice = rel / tm;
  const int m0 = by * kBM, n0 = bx * kBN;

  // contraction range of this split-K slice (multiples of kBK)
  const int kslab = (P.K + kBK - 1) / kBK;
  const int per = (kslab + P.split_k - 1) / P.split_k;
  const int kbeg = slice * per * kBK;
  const int kend = min(P.K, (slice + 1) * per * kBK);
  if (kbeg >= kend && slice > 0) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / kWavesN, wc = wave % kWavesN;
  const int fr = lane & 15, fg = lane >> 4;

  f32x4 acc[kMI][kNJ];
#pragma unroll
  for (int i = 0; i < kMI; ++i)
#pragma unroll
    for (int j = 0; j < kNJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool ones = P.ones_col != 0;
  // operand dropout: the (step, site) halves of the hash are kernel-invariant
  const bool a_dropout = P.a_drop_p > 0.f, b_dropout = P.b_drop_p > 0.f;
  const uint64_t step_ctr = ((a_dropout || b_dropout || P.dropout_p > 0.f) && rng_counter) ? *rng_counter : 0ull;
  const uint32_t a_key = rng::site_key(step_ctr, P.a_drop_site), b_key = rng::site_key(step_ctr, P.b_drop_site);
  const float a_inv = a_dropout ? 1.f / (1.f - P.a_drop_p) : 1.f, b_inv = b_dropout ? 1.f / (1.f - P.b_drop_p) : 1.f;
  auto mfma_slab = [&](int buf) {
#pragma unroll
    for (int u = 0; u < kBK / 16; ++u) {
      f32x4 af[kMI], bf[kNJ];
#pragma unroll
      for (int i = 0; i < kMI; ++i)
        af[i] = *reinterpret_cast<const f32x4 *>(&As[buf][wr * (16 * kMI) + i * 16 + fr][u * 16 + fg * 4]);
#pragma unroll
      for (int j = 0; j < kNJ; ++j)
        bf[j] = *reinterpret_cast<const f32x4 *>(&Bs[buf][wc * (16 * kNJ) + j * 16 + fr][u * 16 + fg * 4]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < kMI; ++i)
#pragma unroll
          for (int j = 0; j < kNJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
  };
  // Fast path (interior tiles, the common case): every address is  base + slab * step  with the
  // per-thread bases computed once; a slab costs each thread 2*kSub float4 loads, 2*kSub LDS writes and
  // the MFMAs -- no bounds checks, no index arithmetic.  Edge tiles take the generic path below.
  if constexpr (FAST) {
    // The loop is specialised at compile time on the operand layouts and on "plain" vs "with effects"
    // (affine / dropout / ones-row / ragged last slab) and selected by one switch per workgroup: with
    // every mode behind run-time branches in one loop body the kernel was ~8 % slower (the path taken
    // was a few hundred instructions scattered over a 30 KB body).
    const bool rt_a_kc = P.lda_k == 1, rt_b_kc = P.ldb_k == 1;
    const bool rt_fx = P.a_chan_scale != nullptr || P.b_chan_scale != nullptr || a_dropout || b_dropout ||
                       ones || ((kend - kbeg) % kBK) != 0;
    auto run_fast = [&](auto a_kc_t, auto b_kc_t, auto fx_t) {
    constexpr bool a_kc = decltype(a_kc_t)::value, b_kc = decltype(b_kc_t)::value;
    constexpr bool FX = decltype(fx_t)::value;
    const bool f_ones = FX && ones, f_adrop = FX && a_dropout, f_bdrop = FX && b_dropout;
    const int a_slow = a_kc ? (tid / (kSW / 4)) : (tid / kRQ);
    const int a_fast = a_kc ? (tid % (kSW / 4)) * 4 : (tid % kRQ) * 4;
    const int b_slow = b_kc ? (tid / (kSW / 4)) : (tid / kRQ);
    const int b_fast = b_kc ? (tid % (kSW / 4)) * 4 : (tid % kRQ) * 4;
    const float *pa = a_kc ? P.a + (long)(m0 + a_slow) * P.lda_m + kbeg + a_fast
                           : P.a + (long)(kbeg + a_slow) * P.lda_k + m0 + a_fast;
    const float *pb = b_kc ? P.b + (long)(n0 + b_slow) * P.ldb_n + kbeg + b_fast
                           : P.b + (long)(kbeg + b_slow) * P.ldb_k + n0 + b_fast;
    // rows of this thread inside the matrix?  (loop-invariant; rows outside a partial tile read 0:
    // their loads are redirected to the operand base with stride 0 and discarded at commit time, so
    // every load stays an unconditional global_load)
    const bool a_ok = m0 + (a_kc ? a_slow : a_fast) < P.M;
    const bool b_ok = n0 + (b_kc ? b_slow : b_fast) < P.N;
    const long sa16 = !a_ok ? 0 : (a_kc ? kSW : kSW * P.lda_k);   // per staging step (kSW k)
    const long sb16 = !b_ok ? 0 : (b_kc ? kSW : kSW * P.ldb_k);
    if (!a_ok) pa = P.a;
    if (!b_ok) pb = P.b;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool a_aff = FX && P.a_chan_scale != nullptr;   // channel = k (varies per slab): staged in LDS
    if (a_aff) {
      for (int k = tid; k < kend - kbeg; k += THREADS) {
        Asc[k] = P.a_chan_scale[kbeg + k];
        Ash[k] = P.a_chan_shift[kbeg + k];
      }
    }
    float4 bsc4 = make_float4(1.f, 1.f, 1.f, 1.f), bsh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool b_aff = FX && P.b_chan_scale != nullptr;
    if (b_aff && !b_kc) {  // channel = B row = 4 consecutive rows of this thread: loop-invariant
      if (b_ok) {
        bsc4 = *reinterpret_cast<const float4 *>(P.b_chan_scale + n0 + b_fast);
        bsh4 = *reinterpret_cast<const float4 *>(P.b_chan_shift + n0 + b_fast);
      }
    } else if (b_aff && b_ok) {
      const float sc = P.b_chan_scale[n0 + b_slow], sh = P.b_chan_shift[n0 + b_slow];
      bsc4 = make_float4(sc, sc, sc, sc);
      bsh4 = make_float4(sh, sh, sh, sh);
    }
    if (a_aff) __syncthreads();
    // virtual ones-row of B (row index N): which of this thread's elements is it, if any
    const int ones_e = !f_ones ? -1 : (b_kc ? (n0 + b_slow == P.N ? 4 : -1)
                                          : ((n0 + b_fast <= P.N && P.N < n0 + b_fast + 4) ? P.N - (n0 + b_fast) : -1));
    float4 ra[kSub], rb[kSub];
    const long offa0 = pa - P.a, offb0 = pb - P.b;   // element offsets of this thread's first float4
    auto drop4 = [](float4 v, uint32_t key, uint32_t off, float p, float inv) {
      v.x = rng::keep_keyed(key, off + 0, p) ? v.x * inv : 0.f;
      v.y = rng::keep_keyed(key, off + 1, p) ? v.y * inv : 0.f;
      v.z = rng::keep_keyed(key, off + 2, p) ? v.z * inv : 0.f;
      v.w = rng::keep_keyed(key, off + 3, p) ? v.w * inv : 0.f;
      return v;
    };
    // k position (inside a staging step) of this thread's float4, per operand; a float4 whose k is
    // beyond the slice (ragged last slab, K % 4 == 0) is zero
    const int krange = kend - kbeg;
    const int a_k = a_kc ? a_fast : a_slow, b_k = b_kc ? b_fast : b_slow;
    // fetch / commit come in two flavours selected at compile time: whole slabs (the steady state: no
    // predicates at all) and the ragged last slab (its predicates cost ~8 % when left in the main loop)
    auto fetch_fast = [&](int slab, auto kind) {
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        if constexpr (!decltype(kind)::ragged) {
          ra[u] = ldg4(pa + (long)(slab * kSub + u) * sa16);
          rb[u] = ldg4(pb + (long)(slab * kSub + u) * sb16);
        } else {
          const int k0 = slab * kBK + u * kSW;
          ra[u] = (k0 + a_k < krange) ? ldg4(pa + (long)(slab * kSub + u) * sa16) : zero4;
          rb[u] = (k0 + b_k < krange) ? ldg4(pb + (long)(slab * kSub + u) * sb16) : zero4;
        }
      }
    };
    // LDS images.  A contraction-contiguous operand is stored [row][k] (row stride kLd) and a fragment
    // (4 consecutive k of one row) is one ds_read_b128.  A row-contiguous operand (both operands of a
    // weight-gradient product) is stored AS IT ARRIVES, [k][row] with row stride kLdT: the float4 write is
    // conflict-free and the fragment becomes four conflict-free ds_read_b32 -- transposing on the way
    // in cost sixteen 4-way-conflicting scalar writes per thread and slab and made these products run
    // at half the per-slab rate of the forward ones.
    constexpr int kLdT = kBM + 4;
    static_assert(kBK * kLdT <= kBM * kLd && kBM == kBN, "[k][row] image must fit the [row][k] buffer");
    auto put = [&](float (*tile)[kLd], bool kc, int slow, int fst, int koff, float4 v) {
      if (kc) {
        *reinterpret_cast<float4 *>(&tile[slow][koff + fst]) = v;
      } else {
        float(*t)[kLdT] = reinterpret_cast<float(*)[kLdT]>(&tile[0][0]);
        *reinterpret_cast<float4 *>(&t[koff + slow][fst]) = v;
      }
    };
    auto frag = [&](float (*tile)[kLd], bool kc, int row, int k0) -> f32x4 {
      if (kc) return *reinterpret_cast<const f32x4 *>(&tile[row][k0]);
      const float(*t)[kLdT] = reinterpret_cast<const float(*)[kLdT]>(&tile[0][0]);
      return (f32x4){t[k0 + 0][row], t[k0 + 1][row], t[k0 + 2][row], t[k0 + 3][row]};
    };
    auto mfma_fast = [&](int buf) {
#pragma unroll
      for (int u = 0; u < kBK / 16; ++u) {
        f32x4 af[kMI], bf[kNJ];
#pragma unroll
        for (int i = 0; i < kMI; ++i)
          af[i] = frag(As[buf], a_kc, wr * (16 * kMI) + i * 16 + fr, u * 16 + fg * 4);
#pragma unroll
        for (int j = 0; j < kNJ; ++j)
          bf[j] = frag(Bs[buf], b_kc, wc * (16 * kNJ) + j * 16 + fr, u * 16 + fg * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < kMI; ++i)
#pragma unroll
            for (int j = 0; j < kNJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
      }
    };
    auto commit_fast = [&](int slab, int buf, auto kind) {
      const int kslab0 = slab * kBK;   // k offset (relative to kbeg) of the slab held in ra/rb
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        bool a_in = true, b_in = true;
        if constexpr (decltype(kind)::ragged) {
          a_in = kslab0 + u * kSW + a_k < krange;
          b_in = kslab0 + u * kSW + b_k < krange;
        }
        const bool a_live = a_ok && a_in, b_live = b_ok && b_in;
        float4 va = a_live ? ra[u] : zero4, vb = b_live ? rb[u] : zero4;
        if (a_aff && a_live) {
          float4 sc, sh;
          if (a_kc) {
            sc = *reinterpret_cast<const float4 *>(&Asc[kslab0 + u * kSW + a_fast]);
            sh = *reinterpret_cast<const float4 *>(&Ash[kslab0 + u * kSW + a_fast]);
          } else {
            const float s1 = Asc[kslab0 + u * kSW + a_slow], h1 = Ash[kslab0 + u * kSW + a_slow];
            sc = make_float4(s1, s1, s1, s1);
            sh = make_float4(h1, h1, h1, h1);
          }
          va.x = fmaxf(va.x * sc.x + sh.x, 0.f); va.y = fmaxf(va.y * sc.y + sh.y, 0.f);
          va.z = fmaxf(va.z * sc.z + sh.z, 0.f); va.w = fmaxf(va.w * sc.w + sh.w, 0.f);
        }
        if (b_aff && b_live) {
          vb.x = fmaxf(vb.x * bsc4.x + bsh4.x, 0.f); vb.y = fmaxf(vb.y * bsc4.y + bsh4.y, 0.f);
          vb.z = fmaxf(vb.z * bsc4.z + bsh4.z, 0.f); vb.w = fmaxf(vb.w * bsc4.w + bsh4.w, 0.f);
        }
        if (f_adrop && a_live)
          va = drop4(va, a_key, (uint32_t)(offa0 + (long)(slab * kSub + u) * sa16), P.a_drop_p, a_inv);
        if (f_bdrop && b_live)
          vb = drop4(vb, b_key, (uint32_t)(offb0 + (long)(slab * kSub + u) * sb16), P.b_drop_p, b_inv);
        if (b_in) {   // the ones-row is 1 for every k inside the slice
          if (ones_e == 4) vb = make_float4(1.f, 1.f, 1.f, 1.f);
          else if (ones_e == 0) vb.x = 1.f;
          else if (ones_e == 1) vb.y = 1.f;
          else if (ones_e == 2) vb.z = 1.f;
          else if (ones_e == 3) vb.w = 1.f;
        }
        put(As[buf], a_kc, a_slow, a_fast, u * kSW, va);
        put(Bs[buf], b_kc, b_slow, b_fast, u * kSW, vb);
      }
    };
    const int nslab = (krange + kBK - 1) / kBK;
    // (a two-slab-deep register prefetch was measured: no gain -- the loop is not bound by the L2 round
    // trip -- so one register set it is)
    const int nwhole = krange / kBK;
    if (!FX || nwhole > 0) {
      fetch_fast(0, Whole());
      commit_fast(0, 0, Whole());
    } else {
      fetch_fast(0, Ragged());
      commit_fast(0, 0, Ragged());
    }
    __syncthreads();
    for (int sl = 0; sl < nslab; ++sl) {
      const int nx = sl + 1;
      if (nx < nwhole) {
        fetch_fast(nx, Whole());
        mfma_fast(sl & 1);
        commit_fast(nx, nx & 1, Whole());
      } else if (FX && nx < nslab) {
        fetch_fast(nx, Ragged());
        mfma_fast(sl & 1);
        commit_fast(nx, nx & 1, Ragged());
      } else {
        mfma_fast(sl & 1);
      }
      __syncthreads();
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
    // streaming: double-buffered LDS, one barrier per slab: slab i+1 travels global -> registers while
    // slab i is multiplied, then lands in the other buffer
    Frag4 fa[kSub], fb[kSub];
    int kfetched = kbeg;
    const OperandFx fxa = {P.a2 != nullptr, P.a2_mode, P.a2_scale, P.a_chan_scale, P.a_chan_shift, true,
                           P.a_drop_p, a_inv, a_key, P.lda_m};
    const OperandFx fxb = {false, 0, 0.f, P.b_chan_scale, P.b_chan_shift, false,
                           P.b_drop_p, b_inv, b_key, P.ldb_n};
    auto fetch = [&](int k0) {
      kfetched = k0;
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        fa[u] = fetch_tile<kSW, TILE>(P.a, P.a2, P.lda_m, P.lda_k, m0, P.M, k0 + u * kSW, kend, tid);
        fb[u] = fetch_tile<kSW, TILE, false>(P.b, nullptr, P.ldb_n, P.ldb_k, n0, P.N, k0 + u * kSW, kend, tid);
      }
    };
    auto commit = [&](int buf) {
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        commit_tile<kSW, TILE>(As[buf], fa[u], fxa, P.lda_k, m0, P.M, kfetched + u * kSW, kend, false, u * kSW, tid);
        commit_tile<kSW, TILE>(Bs[buf], fb[u], fxb, P.ldb_k, n0, P.N, kfetched + u * kSW, kend, ones, u * kSW, tid);
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
    // Plain stores: stage the 64x64 tile through LDS (the operand buffers are free after the last
    // barrier) so every thread writes whole float4 row segments -- the MFMA C-layout would otherwise
    // emit sixteen 4-byte stores per lane, 64 contiguous bytes per wave-instruction.
    float(*Cs)[kBN + 4] = reinterpret_cast<float(*)[kBN + 4]>(&As[0][0][0]);
    static_assert(sizeof(As) >= sizeof(float) * kBM * (kBN + 4), "C tile must fit the A buffers");
#pragma unroll
    for (int i = 0; i < kMI; ++i)
#pragma unroll
      for (int j = 0; j < kNJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Cs[wr * (16 * kMI) + i * 16 + fg * 4 + r][wc * (16 * kNJ) + j * 16 + fr] = acc[i][j][r];
    __syncthreads();
    const int c4 = (tid % kRQ) * 4, rphase = tid / kRQ;
    const int n = n0 + c4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (P.bias && slice == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < pN) bv[e] = P.bias[n + e];
    }
    const bool vec_ok = (n + 3 < pN) && ((ldc & 3) == 0) && ((((uintptr_t)cptr) & 15) == 0);
    double *const col_sum = P.col_sum, *const col_sumsq = P.col_sumsq;
    const bool c_add = P.c_add != 0;
    float *const c2ptr = P.c2;
    const bool vec2_ok = vec_ok && ((((uintptr_t)c2ptr) & 15) == 0);
    float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qq = 0; qq < kBM / kRowPhases; ++qq) {
      const int row = rphase + qq * kRowPhases;
      const int m = m0 + row;
      if (m >= pM || n >= pN) continue;
      const float4 cv = *reinterpret_cast<const float4 *>(&Cs[row][c4]);
      float v[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = (v[e] + bv[e]) * scale;
        if (relu) v[e] = fmaxf(v[e], 0.f);
        if (drop)
          v[e] = rng::keep(ctr, site, (uint32_t)((long)m * pN + n + e), p_drop) ? v[e] * inv_keep : 0.f;
        if (n + e < pN) {
          cs[e] += v[e];
          cq[e] += v[e] * v[e];
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
        *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < pN) dst[e] = c_add ? dst[e] + v[e] : v[e];
      }
    }
    if (col_sum) {
      // column sums of the tile: 16 row-phase partials per column through LDS (the B buffers are free
      // after the last barrier of the K loop), then one double atomic per column and statistic
      float *red = &Bs[0][0][0];
      static_assert(sizeof(Bs) >= sizeof(float) * 2 * kRowPhases * kBN, "statistics scratch must fit the B buffers");
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[(0 * kRowPhases + rphase) * kBN + c4 + e] = cs[e];
        red[(1 * kRowPhases + rphase) * kBN + c4 + e] = cq[e];
      }
      __syncthreads();
      if (tid < 2 * kBN) {
        const int which = tid / kBN, col = tid % kBN;
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < kRowPhases; ++r) acc += (double)red[(which * kRowPhases + r) * kBN + col];
        const long slot_off = P.col_slots > 1 ? (long)(blockIdx.x & (P.col_slots - 1)) * P.col_slot_stride : 0;
        if (n0 + col < pN) atomicAdd((which ? col_sumsq : col_sum) + slot_off + n0 + col, acc);
      }
    }
    return;
  }
  // accumulate / bias-gradient path: element-wise atomics straight from the accumulators
  float bias_v[kNJ];
#pragma unroll
  for (int j = 0; j < kNJ; ++j) {
    const int n = n0 + wc * (16 * kNJ) + j * 16 + fr;
    bias_v[j] = (P.bias && slice == 0 && n < pN) ? P.bias[n] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < kMI; ++i)
#pragma unroll
    for (int j = 0; j < kNJ; ++j) {
      const int n = n0 + wc * (16 * kNJ) + j * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wr * (16 * kMI) + i * 16 + fg * 4 + r;
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
    float dropout_p, uint32_t site, const uint64_t *__restrict__ rng_counter, int rows_per_wave) {
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
    atomicAdd(dgamma + c, a);
    atomicAdd(dbeta + c, b);
  }
}

}  // namespace

extern "C" {

static bool fast_eligible(const butd_gemm_problem &p) {
  const bool a_kc = p.lda_k == 1, b_kc = p.ldb_k == 1;
  const int kslab = (p.K + kBK - 1) / kBK, split = p.split_k < 1 ? 1 : p.split_k;
  const long per = (long)((kslab + split - 1) / split) * kBK;   // contraction range of one slice
  // (a companion operand a2 stays on the generic kernel: its extra register set cost the fast
  // instantiations a wave of occupancy, measured)
  return p.a2 == nullptr && p.K > 0 && (p.K & 3) == 0 &&   // a ragged LAST slab is predicated per float4
         (a_kc || (p.M & 3) == 0) && (b_kc || (p.N & 3) == 0) &&   // partial tiles: whole float4 in or out
         ((a_kc ? p.lda_m : p.lda_k) & 3) == 0 && ((b_kc ? p.ldb_n : p.ldb_k) & 3) == 0 &&
         ((((uintptr_t)p.a) | ((uintptr_t)p.b)) & 15) == 0 &&
         (p.a_chan_scale == nullptr || per <= kAffK);
}

static long fill_batch(GemmBatch &batch, const butd_gemm_problem *problems, const int *index, int count,
                       int tile) {
  long total = 0;
  batch.count = 0;
  for (int i = 0; i < count; ++i) {
    butd_gemm_problem p = problems[index[i]];
    if (p.split_k < 1) p.split_k = 1;
    const int ncols = p.N + (p.ones_col ? 1 : 0);
    const int tn = (ncols + tile - 1) / tile, tm = (p.M + tile - 1) / tile;
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

static int launch_group(const butd_gemm_problem *problems, const int *index, int count, bool fast,
                        const uint64_t *rng_counter, hipStream_t stream) {
  if (count == 0) return 0;
  GemmBatch batch;
  long total = fill_batch(batch, problems, index, count, 64);
  if (total < 0) return (int)hipErrorInvalidValue;
  if (total == 0) return 0;
  // Configuration by grid size (measured with the whole training step, graph replay): 32x32 tiles up to
  // 10 000 64x64-tiles' worth of work -- four times the workgroups, a quarter of the matrix phase each,
  // and the phases of co-resident workgroups overlap (a 2048x288x288 product: 13.9 -> 10.1 us, an
  // 8192-row one 28.7 -> 25.6 us, step 35.7 -> 33.0 ms); 64x64 tiles / 4 waves for the 10^5..10^6-row
  // set-abstraction products, where the larger tile's operand reuse wins.
  static const int forced = getenv("BUTD_GEMM_CFG") ? atoi(getenv("BUTD_GEMM_CFG")) : 0;
  static const long t32 = getenv("BUTD_GEMM_T32") ? atol(getenv("BUTD_GEMM_T32")) : 10000;
  const int cfg = forced ? forced : (total <= t32 ? 32 : 64);
  if (cfg == 32) {
    total = fill_batch(batch, problems, index, count, 32);
    if (total < 0) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)total);
    if (fast) hipLaunchKernelGGL((gemm_kernel<256, 32, true>), grid, dim3(256), 0, stream, batch, rng_counter);
    else hipLaunchKernelGGL((gemm_kernel<256, 32, false>), grid, dim3(256), 0, stream, batch, rng_counter);
  } else if (cfg == 512) {
    const dim3 grid((unsigned)total);
    if (fast) hipLaunchKernelGGL((gemm_kernel<512, 64, true>), grid, dim3(512), 0, stream, batch, rng_counter);
    else hipLaunchKernelGGL((gemm_kernel<512, 64, false>), grid, dim3(512), 0, stream, batch, rng_counter);
  } else {
    const dim3 grid((unsigned)total);
    if (fast) hipLaunchKernelGGL((gemm_kernel<256, 64, true>), grid, dim3(256), 0, stream, batch, rng_counter);
    else hipLaunchKernelGGL((gemm_kernel<256, 64, false>), grid, dim3(256), 0, stream, batch, rng_counter);
  }
  return (int)hipGetLastError();
}

int butd_gemm_grouped(const butd_gemm_problem *problems, int count, const uint64_t *rng_counter,
                      butd_stream_t stream) {
  if (count <= 0) return 0;
  if (count > kMaxProblems) return (int)hipErrorInvalidValue;
  // the problems of a group are independent: the fast-eligible ones and the rest run as two launches
  int fast_idx[kMaxProblems], slow_idx[kMaxProblems], nf = 0, ns = 0;
  for (int i = 0; i < count; ++i) {
    const butd_gemm_problem &p = problems[i];
    if (p.M <= 0 || p.N <= 0) continue;
    if (p.split_k > 1 && !p.accumulate) return (int)hipErrorInvalidValue;
    if ((p.col_sum != nullptr || p.c_add || p.c2 != nullptr) && (p.accumulate || p.ones_col || p.split_k > 1))
      return (int)hipErrorInvalidValue;
    if (p.col_slots > 1 && (p.col_slots & (p.col_slots - 1))) return (int)hipErrorInvalidValue;
    if (fast_eligible(p)) fast_idx[nf++] = i; else slow_idx[ns++] = i;
  }
  int err = launch_group(problems, fast_idx, nf, true, rng_counter, (hipStream_t)stream);
  if (err) return err;
  return launch_group(problems, slow_idx, ns, false, rng_counter, (hipStream_t)stream);
}

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
  // one row per wave while that still leaves workgroups for every CU to spare (a 2048-row call is 128
  // workgroups); several rows per wave only for very tall inputs, where it trims the dgamma/dbeta atomics
  static const int forced = getenv("BUTD_LN_RPW") ? atoi(getenv("BUTD_LN_RPW")) : 0;
  // (measured: the column-sum atomics are half of the kernel's time at 8192 rows -- 512 workgroups on the same
  // 576 addresses; two rows per wave halve them there: 21.2 -> 18.4 us.  Folding private copies with a
  // last-workgroup ticket needs an agent-scope release per workgroup = an L2 write-back each: 159 us.)
  const int rpw = forced ? forced : (rows >= 65536 ? kLnRowsPerWave : rows >= 8192 ? 2 : 1);
  const int rows_per_block = (kLnBwdThreads / 64) * rpw;
  const dim3 grid((rows + rows_per_block - 1) / rows_per_block);
  LN_DISPATCH_T(kLnBwdThreads, ln_bwd_kernel, rows, cols, dy, x, residual, gamma, mean, rstd, dx, d_residual, dgamma,
              dbeta, dropout_p, dropout_site, rng_counter, rpw);
  return (int)hipGetLastError();
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

// NG = 2 (small grids, e.g. the decoder's 256 queries: 256 workgroups = ONE wave per SIMD, nothing to
// hide a stall behind): two wave groups of a 512-thread workgroup walk the even / odd key tiles of the
// same 64 queries with their own LDS images and merge their (o, m, l) states through LDS at the end.
template <int NS, int NT, int NG>
__global__ __launch_bounds__(kAttnThreads * NG) void attn_fwd_kernel(
    int H, int Lq, int Lk, int D, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const uint8_t *__restrict__ mask, float *__restrict__ out,
    float *__restrict__ lse, float p_drop, uint32_t site, const uint64_t *__restrict__ rng_counter) {
  using I = Img<NS>;
  __shared__ __attribute__((aligned(16))) float KimgG[NG][2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float VimgG[NG][2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float BiasG[NG][2][64];
  const int grp = threadIdx.x / kAttnThreads;
  float(*Kimg)[64][I::LD] = KimgG[grp];
  float(*Vimg)[64][I::LD] = VimgG[grp];
  float(*Bias)[64] = BiasG[grp];
  const int tid = threadIdx.x % kAttnThreads, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const long E = (long)H * D;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const bool live = q0 < Lq;  // whole wave beyond Lq: still stages tiles and hits the barriers
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
  int vcol[NT];
  bool vok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = nt * 16 + fr;
    vok[nt] = n < D;
    vcol[nt] = I::col(vok[nt] ? n : 0);
  }
  float m = kNegInf, l = 0.f;
  f32x4 o[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // key tiles of this wave group: grp, grp + NG, ... (a tile past Lk stages zeros with -inf bias and
  // contributes nothing, so both groups run the same number of iterations and barriers)
  const int iters = ((Lk + 63) / 64 + NG - 1) / NG;
  typename I::Regs kr, vr;
  float br = 0.f;
  I::fetch(kr, kb, E, D, grp * 64, Lk, tid);
  I::fetch(vr, vb, E, D, grp * 64, Lk, tid);
  if (tid < 64) br = key_bias(mb, grp * 64 + tid, Lk);
  I::commit(Kimg[0], kr, D, tid);
  I::commit(Vimg[0], vr, D, tid);
  if (tid < 64) Bias[0][tid] = br;
  __syncthreads();
  int cur = 0;
  for (int it = 0; it < iters; ++it) {
    const int key0 = (it * NG + grp) * 64;
    const bool more = it + 1 < iters;
    if (more) {
      I::fetch(kr, kb, E, D, key0 + NG * 64, Lk, tid);
      I::fetch(vr, vb, E, D, key0 + NG * 64, Lk, tid);
      if (tid < 64) br = key_bias(mb, key0 + NG * 64 + tid, Lk);
    }
    if (live) {
      f32x4 st[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float kf[NS];
        I::frag(kf, Kimg[cur], t * 16 + fr, fg);
        st[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s)
          st[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s], qf[s], st[t], 0, 0, 0);
      }
      float tmax = kNegInf;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 bb = *reinterpret_cast<const float4 *>(&Bias[cur][t * 16 + fg * 4]);
        st[t][0] += bb.x; st[t][1] += bb.y; st[t][2] += bb.z; st[t][3] += bb.w;
        tmax = fmaxf(fmaxf(fmaxf(tmax, st[t][0]), fmaxf(st[t][1], st[t][2])), st[t][3]);
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
          const float *vrow = Vimg[cur][t * 16 + fg * 4 + s];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float a = vok[nt] ? vrow[vcol[nt]] : 0.f;
            o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, st[t][s], o[nt], 0, 0, 0);
          }
        }
      m = m_new;
    }
    if (more) {
      I::commit(Kimg[cur ^ 1], kr, D, tid);
      I::commit(Vimg[cur ^ 1], vr, D, tid);
      if (tid < 64) Bias[cur ^ 1][tid] = br;
    }
    __syncthreads();
    cur ^= 1;
  }
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
}

// dQ: same walk as the forward with K and V swapping roles.  Also produces
// delta[b,h,q] = sum_n dO[q][n] * O[q][n]  (each query row belongs to exactly one wave here) for the
// dK/dV kernel that is launched next -- it used to be a launch of its own.
template <int NS, int NT, int NG>
__global__ __launch_bounds__(kAttnThreads * NG) void attn_bwd_dq_kernel(
    int H, int Lq, int Lk, int D, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const uint8_t *__restrict__ mask, const float *__restrict__ out,
    const float *__restrict__ dout, const float *__restrict__ lse, float *__restrict__ delta,
    float *__restrict__ dq, long ldo, float dq_scale,
    float p_drop, uint32_t site, const uint64_t *__restrict__ rng_counter) {
  using I = Img<NS>;
  __shared__ __attribute__((aligned(16))) float KimgG[NG][2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float VimgG[NG][2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float BiasG[NG][2][64];
  const int grp = threadIdx.x / kAttnThreads;   // key-tile group, see attn_fwd_kernel
  float(*Kimg)[64][I::LD] = KimgG[grp];
  float(*Vimg)[64][I::LD] = VimgG[grp];
  float(*Bias)[64] = BiasG[grp];
  const int tid = threadIdx.x % kAttnThreads, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const long E = (long)H * D;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const bool live = q0 < Lq;
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
  // rows beyond Lq: lse = +inf makes every probability exp(s - inf) = 0
  const float my_lse = qi < Lq ? lse[((long)b * H + h) * Lq + qi] : INFINITY;
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
  int kcol[NT];
  bool kok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int d = nt * 16 + fr;
    kok[nt] = d < D;
    kcol[nt] = I::col(kok[nt] ? d : 0);
  }
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // key tiles of this wave group: grp, grp + NG, ... (a tile past Lk stages zeros with -inf bias and
  // contributes nothing, so both groups run the same number of iterations and barriers)
  const int iters = ((Lk + 63) / 64 + NG - 1) / NG;
  typename I::Regs kr, vr;
  float br = 0.f;
  I::fetch(kr, kb, E, D, grp * 64, Lk, tid);
  I::fetch(vr, vb, E, D, grp * 64, Lk, tid);
  if (tid < 64) br = key_bias(mb, grp * 64 + tid, Lk);
  I::commit(Kimg[0], kr, D, tid);
  I::commit(Vimg[0], vr, D, tid);
  if (tid < 64) Bias[0][tid] = br;
  __syncthreads();
  int cur = 0;
  for (int it = 0; it < iters; ++it) {
    const int key0 = (it * NG + grp) * 64;
    const bool more = it + 1 < iters;
    if (more) {
      I::fetch(kr, kb, E, D, key0 + NG * 64, Lk, tid);
      I::fetch(vr, vb, E, D, key0 + NG * 64, Lk, tid);
      if (tid < 64) br = key_bias(mb, key0 + NG * 64 + tid, Lk);
    }
    if (live) {
      f32x4 ds[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float kf[NS], vf[NS];
        I::frag(kf, Kimg[cur], t * 16 + fr, fg);
        I::frag(vf, Vimg[cur], t * 16 + fr, fg);
        f32x4 st = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s], qf[s], st, 0, 0, 0);  // S^T
          dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[s], gf[s], dp, 0, 0, 0);  // dP^T
        }
        const float4 bb = *reinterpret_cast<const float4 *>(&Bias[cur][t * 16 + fg * 4]);
        const float bias4[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p = __expf(st[i] + bias4[i] - my_lse);
          float dpe = dp[i];
          if (drop) {
            const int kk = key0 + t * 16 + fg * 4 + i;
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
          const float *krow = Kimg[cur][t * 16 + fg * 4 + s];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float a = kok[nt] ? krow[kcol[nt]] : 0.f;
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ds[t][s], acc[nt], 0, 0, 0);
          }
        }
    }
    if (more) {
      I::commit(Kimg[cur ^ 1], kr, D, tid);
      I::commit(Vimg[cur ^ 1], vr, D, tid);
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
template <int NS, int NT, int NG>
__global__ __launch_bounds__(kAttnThreads * NG) void attn_bwd_dkv_kernel(
    int H, int Lq, int Lk, int D, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const uint8_t *__restrict__ mask, const float *__restrict__ dout,
    const float *__restrict__ lse, const float *__restrict__ delta, float *__restrict__ dk,
    float *__restrict__ dv, long ldo, float p_drop, uint32_t site,
    const uint64_t *__restrict__ rng_counter) {
  using I = Img<NS>;
  __shared__ __attribute__((aligned(16))) float QimgG[NG][2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float GimgG[NG][2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float LseG[NG][2][64];
  __shared__ __attribute__((aligned(16))) float DelG[NG][2][64];
  const int grp = threadIdx.x / kAttnThreads;
  float(*Qimg)[64][I::LD] = QimgG[grp];
  float(*Gimg)[64][I::LD] = GimgG[grp];
  float(*Lse)[64] = LseG[grp];
  float(*Del)[64] = DelG[grp];
  const int tid = threadIdx.x % kAttnThreads, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const long E = (long)H * D;
  const int k0 = blockIdx.x * 64 + wave * 16;
  const bool live = k0 < Lk;
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

  float kf[NS], vf[NS];
  load_row_frag<NS>(kf, kb, E, ki, Lk, fg, D);
  load_row_frag<NS>(vf, vb, E, ki, Lk, fg, D);
  int ncol[NT];
  bool nok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = nt * 16 + fr;
    nok[nt] = n < D;
    ncol[nt] = I::col(nok[nt] ? n : 0);
  }
  f32x4 ak[NT], av[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    ak[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    av[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  typename I::Regs qr, gr;
  float sr = 0.f;  // threads 0..63: lse of row tid ; threads 64..127: delta of row tid-64
  auto fetch_stats = [&](int qs) {
    if (tid < 64) sr = (qs + tid < Lq) ? lb[qs + tid] : INFINITY;       // +inf -> probability 0
    else if (tid < 128) sr = (qs + tid - 64 < Lq) ? db[qs + tid - 64] : 0.f;
  };
  auto commit_stats = [&](int buf) {
    if (tid < 64) Lse[buf][tid] = sr;
    else if (tid < 128) Del[buf][tid - 64] = sr;
  };
  // query tiles of this wave group: grp, grp + NG, ... (a tile past Lq stages zeros with lse = +inf:
  // every probability is 0, so both groups run the same number of iterations and barriers)
  const int iters = ((Lq + 63) / 64 + NG - 1) / NG;
  I::fetch(qr, qb, E, D, grp * 64, Lq, tid);
  I::fetch(gr, gb, E, D, grp * 64, Lq, tid);
  fetch_stats(grp * 64);
  I::commit(Qimg[0], qr, D, tid);
  I::commit(Gimg[0], gr, D, tid);
  commit_stats(0);
  __syncthreads();
  int cur = 0;
  for (int it = 0; it < iters; ++it) {
    const int qs = (it * NG + grp) * 64;
    const bool more = it + 1 < iters;
    if (more) {
      I::fetch(qr, qb, E, D, qs + NG * 64, Lq, tid);
      I::fetch(gr, gb, E, D, qs + NG * 64, Lq, tid);
      fetch_stats(qs + NG * 64);
    }
    if (live) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float qf[NS], gf[NS];
        I::frag(qf, Qimg[cur], t * 16 + fr, fg);
        I::frag(gf, Gimg[cur], t * 16 + fr, fg);
        f32x4 st = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          st = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[s], kf[s], st, 0, 0, 0);  // S[q][key]
          dp = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[s], vf[s], dp, 0, 0, 0);  // dP[q][key]
        }
        const float4 l4 = *reinterpret_cast<const float4 *>(&Lse[cur][t * 16 + fg * 4]);
        const float4 d4 = *reinterpret_cast<const float4 *>(&Del[cur][t * 16 + fg * 4]);
        const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq4[4] = {d4.x, d4.y, d4.z, d4.w};
        f32x4 pd, ds;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p = __expf(st[i] + my_bias - lq[i]);
          float keepf = 1.f;
          if (drop) {
            const int qq = qs + t * 16 + fg * 4 + i;
            const uint32_t idx = (uint32_t)((((long)b * H + h) * Lq + qq) * Lk + ki);
            keepf = rng::keep(ctr, site, idx, p_drop) ? inv_keep : 0.f;
          }
          pd[i] = p * keepf;
          ds[i] = p * (dp[i] * keepf - dq4[i]);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float *grow = Gimg[cur][t * 16 + fg * 4 + s];
          const float *qrow = Qimg[cur][t * 16 + fg * 4 + s];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float ag = nok[nt] ? grow[ncol[nt]] : 0.f;
            const float aq = nok[nt] ? qrow[ncol[nt]] : 0.f;
            av[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag, pd[s], av[nt], 0, 0, 0);
            ak[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, ds[s], ak[nt], 0, 0, 0);
          }
        }
      }
    }
    if (more) {
      I::commit(Qimg[cur ^ 1], qr, D, tid);
      I::commit(Gimg[cur ^ 1], gr, D, tid);
      commit_stats(cur ^ 1);
    }
    __syncthreads();
    cur ^= 1;
  }
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

}  // namespace

extern "C" {

// kernels with the key-group parameter: NG = 2 for grids that leave the SIMDs a single wave each
#define ATTN_DISPATCH_G(KERNEL, split, grid, ...)                                                   \
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
static bool split_keys(const dim3 &g, int Lk) {
  static const int forced = getenv("BUTD_ATTN_SPLIT") ? atoi(getenv("BUTD_ATTN_SPLIT")) : -1;
  if (forced >= 0) return forced != 0 && Lk > 64;
  static const long limit = getenv("BUTD_ATTN_SPLIT_MAX") ? atol(getenv("BUTD_ATTN_SPLIT_MAX")) : 512;
  return (long)g.x * g.y * g.z <= limit && Lk >= 128;
}

int butd_attention_fwd(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                       const float *v, const uint8_t *key_padding_mask, float *out, float *lse,
                       float dropout_p, uint32_t dropout_site, const uint64_t *rng_counter,
                       butd_stream_t stream) {
  if (B <= 0 || H <= 0 || Lq <= 0) return 0;
  if (D <= 0 || D > 48 || (D & 3) || Lk <= 0) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((Lq + 63) / 64, H, B);
  ATTN_DISPATCH_G(attn_fwd_kernel, split_keys(grid, Lk), grid, H, Lq, Lk, D, q, k, v, key_padding_mask, out, lse, dropout_p,
                dropout_site, rng_counter);
  return (int)hipGetLastError();
}

int butd_attention_bwd(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
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
  ATTN_DISPATCH_G(attn_bwd_dq_kernel, split_keys(gq, Lk), gq, H, Lq, Lk, D, q, k, v, key_padding_mask, out, dout, lse, delta, dq,
                ld_dq, dq_scale, dropout_p, dropout_site, rng_counter);
  ATTN_DISPATCH_G(attn_bwd_dkv_kernel, split_keys(gk, Lq), gk, H, Lq, Lk, D, q, k, v, key_padding_mask, dout, lse, delta,
                dk, dv, ld_dkv, dropout_p, dropout_site, rng_counter);
  return (int)hipGetLastError();
}

}  // extern "C"
