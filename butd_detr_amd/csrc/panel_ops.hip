// panel_ops.hip -- row-panel chain kernels (include/butd_panel.h): dependent row-wise operators of the attention /
// FFN stack (out-projection + residual + dropout + LayerNorm + next query projection; Linear-ReLU-Linear + LayerNorm)
// as ONE launch.  gfx950 only.
//
// Shape of the kernel
//   * a workgroup (256 threads = 4 waves) owns R = 16 or 32 consecutive rows; every intermediate of the chain lives
//     in LDS "panels" of R x 292 floats (row stride 292 = 36 mod 64 banks: the 16 rows x 4 lane groups of an MFMA
//     A-operand read -- one ds_read_b128 per lane -- and the C-layout writes of an epilogue are conflict-free);
//   * one stage = one (R x K) . (K x N) product on v_mfma_f32_16x16x4_f32: the A operand comes from a panel, the B
//     operand (the weights, N x K row-major = contraction-contiguous) goes from L2 STRAIGHT into operand registers
//     -- every wave owns its own 16-column tiles (wave, wave + 4, ...), so nothing of B is shared inside the
//     workgroup and staging it through LDS would only add barriers; a ring of kDepth 16-deep k-slabs per wave
//     keeps ~1.3 KB per wave in flight, the K loop has no barrier at all;
//     the k index is permuted as in gemm_ops.hip (MFMA step s of lane group g takes k = 16 * slab + 4 g + s), so
//     both operand fragments are 16 contiguous bytes;
//   * epilogue: (acc + bias) * scale, ReLU -> the stage's output panel (C layout), barrier, then a ROW PASS (one
//     wave per row, float4 per lane, coalesced global traffic): saved pre-dropout copy, dropout (the counter hash of
//     csrc/rng.h on element row * N + col -- the masks butd_add_dropout_layernorm_bwd and the GEMM epilogue
//     regenerate), residual + LayerNorm (two-pass statistics, DPP wave sums), + pos, global stores, results back
//     into panels for the next stage; barrier.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/butd_panel.h"
#include "rng.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// timing ablations for scratch/ experiments (never set in the product build): bit 0 no MFMAs, 1 no B loads after the
// ring prefetch, 2 no epilogue stores
#ifndef PANEL_ABL
#define PANEL_ABL 0
#endif

constexpr int kThreads = 256;
constexpr int kLdP = BUTD_PANEL_MAX_COLS + 4;   // panel row stride in floats
#ifndef PANEL_DEPTH
#define PANEL_DEPTH 4
#endif
constexpr int kDepth = PANEL_DEPTH;             // k-slabs (16 deep) of B in flight per wave

typedef const __attribute__((address_space(1))) f32x4 *global_f4_ptr;
typedef const __attribute__((address_space(1))) char *global_byte_ptr;
// uniform base (scalar registers) + 32-bit byte offset per lane: global_load_dwordx4 v, v_off, s[base]
__device__ inline f32x4 ldg4_at(const float *base, uint32_t byte_off) {
  const global_byte_ptr g = reinterpret_cast<global_byte_ptr>(reinterpret_cast<uintptr_t>(base));
  return *reinterpret_cast<global_f4_ptr>(g + byte_off);
}

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

struct PanelArgs {
  butd_panel_stage st[BUTD_PANEL_MAX_STAGES];
  int nstages, rows;
  const float *in;
  int in_cols, in_buf;
  const float *in_pos;
  float *in_sum;
  int in_sum_buf;
};

// One wave's B-operand stream of a stage: byte offsets of its (up to 5) column tiles, tile ct = wave + 4 j.
// (a wave without a j-th tile re-reads the last one: the loops stay branch-free, the epilogue drops the result)
__device__ __forceinline__ void tile_offsets(uint32_t (&boff)[5], int K, int NT, int wave, int fr, int fg) {
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int ct = min(wave + 4 * j, NT - 1);
    boff[j] = 4u * (uint32_t)((ct * 16 + fr) * K + fg * 4);
  }
}

// the first kDepth k-slabs of a stage's weights -> the ring.  Issued BEFORE the previous stage's epilogue (and before
// the input panel is loaded), so the L2 round trip of a stage's first operands hides behind that work and the loads
// are older than the epilogue's stores (vmcnt retires in order: a load behind a store waits for the store as well).
template <int TPW>
__device__ __forceinline__ void ring_prefetch(f32x4 (&ring)[kDepth][5], const float *__restrict__ w, const uint32_t (&boff)[5],
                                     int nslab) {
  (void)nslab;   // (the contraction is at least kDepth slabs deep: checked on the host -- unconditional loads keep the
                 //  ring in registers; with `if (d < nslab)` around them the compiler parked one ring row in scratch)
#pragma unroll
  for (int d = 0; d < kDepth; ++d) {
#pragma unroll
    for (int j = 0; j < TPW; ++j) ring[d][j] = ldg4_at(w, boff[j] + (uint32_t)d * 64u);
#pragma unroll
    for (int j = TPW; j < 5; ++j) ring[d][j] = (f32x4){0.f, 0.f, 0.f, 0.f};   // (defined on every path: else -> scratch)
  }
}

// f(slab, ring slot) for slab = S .. N-1 as straight-line code (the ring slot is a compile-time index)
template <int S, int N, typename F>
__device__ __forceinline__ void slab_sequence(F &f) {
  if constexpr (S < N) {
    f(S, std::integral_constant<int, S % kDepth>());
    slab_sequence<S + 1, N>(f);
  }
}

// acc[j][rt] += panel rows (rt * 16 ..) x columns of tile ct = wave + 4 j, over the whole contraction; the ring
// holds slabs 0 .. kDepth-1 on entry.
// The contraction depth (NSLAB 16-deep slabs) is a compile-time constant and the loop is straight-line code: with a
// run-time loop the compiler's wait-count insertion cannot see across the back edge and drains EVERY load in flight
// (s_waitcnt vmcnt(0)) at the top of each iteration, i.e. the L2 round trip is exposed once per kDepth slabs.
template <int RT, int TPW, int NSLAB>
__device__ __forceinline__ void gemm_stage(f32x4 (&acc)[5][RT], f32x4 (&ring)[kDepth][5], const float *__restrict__ A,
                                  const float *__restrict__ w, const uint32_t (&boff)[5], int K, int fr, int fg) {
  constexpr int nslab = NSLAB;
  const float *arow = A + fr * kLdP + fg * 4;
  auto slab_step = [&](int slab, auto dsel) __attribute__((always_inline)) {
    constexpr int d = decltype(dsel)::value;
    f32x4 a[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) a[rt] = *reinterpret_cast<const f32x4 *>(arow + rt * 16 * kLdP + slab * 16);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          if (!(PANEL_ABL & 1)) acc[j][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][s], ring[d][j][s], acc[j][rt], 0, 0, 0);
          else acc[j][rt][s] += a[rt][s] * ring[d][j][s];
    if (slab + kDepth < nslab && !(PANEL_ABL & 2)) {
#pragma unroll
      for (int j = 0; j < TPW; ++j) ring[d][j] = ldg4_at(w, boff[j] + (uint32_t)(slab + kDepth) * 64u);
    }
  };
  slab_sequence<0, NSLAB>(slab_step);
}

template <int RT, int TPW>
__device__ __forceinline__ void gemm_stage_k(f32x4 (&acc)[5][RT], f32x4 (&ring)[kDepth][5], const float *__restrict__ A,
                                    const float *__restrict__ w, const uint32_t (&boff)[5], int K, int fr, int fg) {
  if (K == 288) gemm_stage<RT, TPW, 18>(acc, ring, A, w, boff, K, fr, fg);
  else if (K == 256) gemm_stage<RT, TPW, 16>(acc, ring, A, w, boff, K, fr, fg);
  else gemm_stage<RT, TPW, 8>(acc, ring, A, w, boff, K, fr, fg);   // K == 128
}

template <int R>
__global__ __launch_bounds__(kThreads) void panel_chain_kernel(PanelArgs args, const uint64_t *__restrict__ rng_counter) {
  constexpr int RT = R / 16;
  constexpr int RPW = R / 4;   // rows per wave in a row pass
  extern __shared__ __attribute__((aligned(16))) float lds[];
  auto panel = [&](int i) { return lds + i * (R * kLdP); };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int rows = args.rows;
  const int row0 = (int)blockIdx.x * R;

  uint32_t boff[5];
  f32x4 ring[kDepth][5];
  auto prefetch_stage = [&](const butd_panel_stage &S) __attribute__((always_inline)) {
    const int NT = S.N >> 4;
    tile_offsets(boff, S.K, NT, wave, fr, fg);
    if (NT <= 16) ring_prefetch<4>(ring, S.w, boff, S.K >> 4);
    else ring_prefetch<5>(ring, S.w, boff, S.K >> 4);
  };
  prefetch_stage(args.st[0]);

  // ---- input panel (+ in_pos): one wave per row, float4 per lane; all loads first, then the LDS writes
  {
    const int n4 = args.in_cols >> 2;
    float *P0 = panel(args.in_buf);
    float *P1 = args.in_pos ? panel(args.in_sum_buf) : nullptr;
    f32x4 v[RPW][2], pv[RPW][2];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int grow = min(row0 + wave + 4 * rr, rows - 1);
      const long gb = (long)grow * args.in_cols;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < n4) {
          v[rr][i] = *reinterpret_cast<const f32x4 *>(args.in + gb + c4 * 4);
          if (P1) pv[rr][i] = *reinterpret_cast<const f32x4 *>(args.in_pos + gb + c4 * 4);
        }
      }
    }
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int r = wave + 4 * rr;
      const bool live = row0 + r < rows;
      const long gb = (long)(live ? row0 + r : rows - 1) * args.in_cols;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < n4) {
          *reinterpret_cast<f32x4 *>(P0 + r * kLdP + c4 * 4) = v[rr][i];
          if (P1) {
            const f32x4 s = v[rr][i] + pv[rr][i];
            *reinterpret_cast<f32x4 *>(P1 + r * kLdP + c4 * 4) = s;
            if (args.in_sum && live) *reinterpret_cast<f32x4 *>(args.in_sum + gb + c4 * 4) = s;
          }
        }
      }
    }
  }
  __syncthreads();

  uint64_t step_ctr = 0ull;
  bool have_ctr = false;

  for (int si = 0; si < args.nstages; ++si) {
    const butd_panel_stage &S = args.st[si];
    const int N = S.N, NT = N >> 4;
    const bool ln = S.ln != 0;
    const int n4 = N >> 2;
    const bool drop = S.drop_p > 0.f;
    if (drop && !have_ctr) {
      step_ctr = rng_counter ? *rng_counter : 0ull;
      have_ctr = true;
    }
    const uint32_t dkey = rng::site_key(step_ctr, S.drop_site);
    const float inv_keep = drop ? 1.f / (1.f - S.drop_p) : 1.f;

    // what the row pass of a LayerNorm stage reads from global memory travels while the stage multiplies
    f32x4 rres[RPW][2], rpos[RPW][2];
    const bool res_g = ln && S.res_buf < 0;
    if (ln) {
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const long gb = (long)min(row0 + wave + 4 * rr, rows - 1) * N;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int c4 = lane + 64 * i;
          if (c4 < n4) {
            if (res_g) rres[rr][i] = *reinterpret_cast<const f32x4 *>(S.res + gb + c4 * 4);
            if (S.pos) rpos[rr][i] = *reinterpret_cast<const f32x4 *>(S.pos + gb + c4 * 4);
          }
        }
      }
    }

    f32x4 acc[5][RT];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[j][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (NT <= 16) gemm_stage_k<RT, 4>(acc, ring, panel(S.in_buf), S.w, boff, S.K, fr, fg);
    else gemm_stage_k<RT, 5>(acc, ring, panel(S.in_buf), S.w, boff, S.K, fr, fg);
    // (this stage's ring is drained: the next stage's first slabs start their trip now)
    float bias_v[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int ct = wave + 4 * j;
      bias_v[j] = (S.bias && ct < NT) ? S.bias[ct * 16 + fr] : 0.f;
    }
    if (si + 1 < args.nstages) prefetch_stage(args.st[si + 1]);

    const float scale = S.scale;
    const bool relu = S.relu != 0;
    float *O = S.out_buf >= 0 ? panel(S.out_buf) : nullptr;
    if (!ln) {
      // ---- plain stage: everything in the accumulators' C layout (lane: rows 4 fg + r, column fr of each tile);
      // no row pass, and no barrier unless a later stage reads the result from LDS
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int ct = wave + 4 * j;
        if (ct < NT) {
          const int n = ct * 16 + fr;
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int prow = rt * 16 + fg * 4 + r, grow = row0 + prow;
              float v = (acc[j][rt][r] + bias_v[j]) * scale;
              if (relu) v = fmaxf(v, 0.f);
              if (grow < rows) {
                const long e = (long)grow * N + n;
                if (S.pre) S.pre[e] = v;
                if (drop) v = rng::keep_keyed(dkey, (uint32_t)e, S.drop_p) ? v * inv_keep : 0.f;
                if (S.out) S.out[e] = v;
              }
              if (O) O[prow * kLdP + n] = v;
            }
        }
      }
      if (O) __syncthreads();
      continue;
    }

    // ---- LayerNorm stage: accumulators -> output panel, barrier, row pass (one wave per row), barrier
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int ct = wave + 4 * j;
      if (ct < NT) {
        const int n = ct * 16 + fr;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = (acc[j][rt][r] + bias_v[j]) * scale;
            if (relu) v = fmaxf(v, 0.f);
            O[(rt * 16 + fg * 4 + r) * kLdP + n] = v;
          }
      }
    }
    __syncthreads();

    f32x4 gm[2], bt[2];
    bool ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c4 = lane + 64 * i;
      ok[i] = c4 < n4;
      gm[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      bt[i] = gm[i];
      if (ok[i]) {
        gm[i] = *reinterpret_cast<const f32x4 *>(S.gamma + c4 * 4);
        bt[i] = *reinterpret_cast<const f32x4 *>(S.beta + c4 * 4);
      }
    }
    const float *RP = S.res_buf >= 0 ? panel(S.res_buf) : nullptr;
    float *PP = (S.pos && S.pos_buf >= 0) ? panel(S.pos_buf) : nullptr;
    const float inv_n = 1.f / (float)N;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int r = wave + 4 * rr;
      const int grow = row0 + r;
      const bool live = grow < rows;
      const long gb = (long)(live ? grow : rows - 1) * N;
      float *orow = O + r * kLdP;
      f32x4 z[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c4 = lane + 64 * i;
        z[i] = ok[i] ? *reinterpret_cast<const f32x4 *>(orow + c4 * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      if (S.pre && live) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          if (ok[i]) *reinterpret_cast<f32x4 *>(S.pre + gb + (lane + 64 * i) * 4) = z[i];
      }
      if (drop) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint32_t e0 = (uint32_t)(gb + (lane + 64 * i) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) z[i][e] = rng::keep_keyed(dkey, e0 + e, S.drop_p) ? z[i][e] * inv_keep : 0.f;
        }
      }
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c4 = lane + 64 * i;
        if (ok[i]) {
          const f32x4 rv = RP ? *reinterpret_cast<const f32x4 *>(RP + r * kLdP + c4 * 4) : rres[rr][i];
          z[i] += rv;
          sum += (z[i][0] + z[i][1]) + (z[i][2] + z[i][3]);
        }
      }
      const float mu = wave_sum(sum) * inv_n;
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if (ok[i]) {
          const f32x4 d = z[i] - mu;
          sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
      const float rs = rsqrtf(wave_sum(sq) * inv_n + S.eps);
#pragma unroll
      for (int i = 0; i < 2; ++i) z[i] = (z[i] - mu) * rs * gm[i] + bt[i];
      if (lane == 0 && live) {
        S.mean[grow] = mu;
        S.rstd[grow] = rs;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c4 = lane + 64 * i;
        if (ok[i]) {
          if (S.out && live) *reinterpret_cast<f32x4 *>(S.out + gb + c4 * 4) = z[i];
          *reinterpret_cast<f32x4 *>(orow + c4 * 4) = z[i];
          if (S.pos) {
            const f32x4 yp = z[i] + rpos[rr][i];
            if (S.out_pos && live) *reinterpret_cast<f32x4 *>(S.out_pos + gb + c4 * 4) = yp;
            if (PP) *reinterpret_cast<f32x4 *>(PP + r * kLdP + c4 * 4) = yp;
          }
        }
      }
    }
    __syncthreads();
  }
}

int g_forced_rows = 0;

template <int R>
int launch(const PanelArgs &args, int nbuf, const uint64_t *rng_counter, hipStream_t stream) {
  static bool configured = false;
  const size_t bytes = (size_t)nbuf * R * kLdP * sizeof(float);
  if (!configured) {   // dynamic LDS beyond the default 64 KB limit
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&panel_chain_kernel<R>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  const dim3 grid((unsigned)((args.rows + R - 1) / R));
  hipLaunchKernelGGL((panel_chain_kernel<R>), grid, dim3(kThreads), bytes, stream, args, rng_counter);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int butd_panel_set_rows(int r) {
  if (r != 0 && r != 16 && r != 32) return (int)hipErrorInvalidValue;
  g_forced_rows = r;
  return 0;
}

int butd_panel_chain(int rows, const float *in, int in_cols, int in_buf, const float *in_pos, float *in_sum,
                     int in_sum_buf, const butd_panel_stage *stages, int nstages, int nbuf,
                     const uint64_t *rng_counter, butd_stream_t stream) {
  if (rows <= 0 || nstages <= 0) return 0;
  auto buf_ok = [&](int b) { return b >= 0 && b < nbuf; };
  if (nstages > BUTD_PANEL_MAX_STAGES || nbuf < 2 || nbuf > BUTD_PANEL_MAX_BUFFERS || !in || in_cols <= 0 ||
      (in_cols & 3) || in_cols > BUTD_PANEL_MAX_COLS || !buf_ok(in_buf) ||
      (in_pos && (!buf_ok(in_sum_buf) || in_sum_buf == in_buf)) || (!in_pos && in_sum) ||
      ((((uintptr_t)in) | ((uintptr_t)in_pos) | ((uintptr_t)in_sum)) & 15))
    return (int)hipErrorInvalidValue;
  PanelArgs args;
  for (int i = 0; i < nstages; ++i) {
    const butd_panel_stage &s = stages[i];
    const uintptr_t ptrs = (uintptr_t)s.w | (uintptr_t)s.pre | (uintptr_t)s.res | (uintptr_t)s.gamma |
                           (uintptr_t)s.beta | (uintptr_t)s.out | (uintptr_t)s.pos | (uintptr_t)s.out_pos;
    if (!s.w || s.N <= 0 || (s.K != 288 && s.K != 256 && s.K != 128) || (s.N & 15) || s.N > BUTD_PANEL_MAX_COLS ||
        s.K > BUTD_PANEL_MAX_COLS || !buf_ok(s.in_buf) || (s.out_buf >= 0 && !buf_ok(s.out_buf)) || (s.ln && s.out_buf < 0) ||
        s.out_buf == s.in_buf || (!s.ln && (s.pos || s.mean || s.rstd)) ||
        (ptrs & 15) || s.drop_p < 0.f || s.drop_p >= 1.f ||
        (s.ln && (!s.gamma || !s.beta || !s.mean || !s.rstd || (s.res_buf < 0 && !s.res) ||
                  (s.res_buf >= 0 && (!buf_ok(s.res_buf) || s.res_buf == s.out_buf)))) ||
        (!s.pos && (s.out_pos || s.pos_buf >= 0)) ||
        (s.pos && s.pos_buf >= 0 && (!buf_ok(s.pos_buf) || s.pos_buf == s.out_buf)))
      return (int)hipErrorInvalidValue;
    args.st[i] = s;
  }
  args.nstages = nstages;
  args.rows = rows;
  args.in = in;
  args.in_cols = in_cols;
  args.in_buf = in_buf;
  args.in_pos = in_pos;
  args.in_sum = in_sum;
  args.in_sum_buf = in_sum_buf;
  const int R = g_forced_rows ? g_forced_rows : (rows >= 4096 ? 32 : 16);
  if (R == 32) return launch<32>(args, nbuf, rng_counter, (hipStream_t)stream);
  return launch<16>(args, nbuf, rng_counter, (hipStream_t)stream);
}

}  // extern "C"
