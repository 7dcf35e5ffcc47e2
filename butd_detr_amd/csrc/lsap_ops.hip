// lsap_ops.hip -- batched rectangular linear sum assignment, one wavefront per problem (gfx950).
//
// See include/butd_lsap.h.  The algorithm is the one scipy.optimize.linear_sum_assignment runs
// (rectangular_lsap.cpp, after its internal transpose to "fewer rows than columns"): for every row
// (target) in ascending order find the shortest augmenting path over the columns (queries) with the
// reduced costs  minVal + C[i][j] - u[i] - v[j],  update the duals, flip the path.  The inner scan over
// the remaining columns is the parallel part: positions of the `remaining` list are dealt to the 64
// lanes; the sequential tie rule of the scan ("a strictly smaller value wins; an equal value wins if its
// column is still unassigned") selects, among the minimum-valued positions, the LAST unassigned one if
// there is one and the FIRST otherwise -- an order-free rule, evaluated with three DPP reductions.
// All per-problem state lives in LDS; one wave = no barriers.
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

#include "../../include/butd_lsap.h"
#include "wave_ops.h"

namespace {

constexpr int kWave = 64;

__device__ inline void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

// monotone map double -> uint64 (total order of the non-NaN values)
__device__ inline unsigned long long ordered_key64(double d) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(d);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

__global__ __launch_bounds__(kWave) void lsap_kernel(int nq, int ng, const float *__restrict__ cost,
                                                     const unsigned char *__restrict__ valid,
                                                     int *__restrict__ match, int *__restrict__ status) {
  BUTD_MAIN_PRIO_SET();
  extern __shared__ unsigned char smem_raw[];
  const int lane = threadIdx.x;
  const int p = blockIdx.x;
  const float *C = cost + (size_t)p * ng * nq;
  const unsigned char *ok = valid + (size_t)p * ng;
  int *out = match + (size_t)p * ng;

  // LDS carve-up (doubles first)
  double *v = reinterpret_cast<double *>(smem_raw);          // [nq] column duals
  double *shortest = v + nq;                                 // [nq]
  double *u = shortest + nq;                                 // [ng] row duals
  int *path = reinterpret_cast<int *>(u + ng);               // [nq]
  int *row4col = path + nq;                                  // [nq]
  int *remaining = row4col + nq;                             // [nq]
  int *col4row = remaining + nq;                             // [ng]
  unsigned char *SC = reinterpret_cast<unsigned char *>(col4row + ng);  // [nq]
  unsigned char *SR = SC + nq;                                         // [ng]

  for (int j = lane; j < nq; j += kWave) {
    v[j] = 0.0;
    row4col[j] = -1;
  }
  int nvalid = 0, bad = 0;
  for (int g0 = 0; g0 < ng; g0 += kWave) {
    const int g = g0 + lane;
    const bool is_valid = g < ng && ok[g] != 0;
    if (g < ng) {
      u[g] = 0.0;
      col4row[g] = -1;
    }
    nvalid += __popcll(__ballot(is_valid));
  }
  if (nvalid > nq) bad = 1;
  wave_fence();

  for (int cur = 0; cur < ng && !bad; ++cur) {
    if (ok[cur] == 0) continue;  // wave-uniform
    const float *row_cur = C + (size_t)cur * nq;
    // scipy rejects matrices with NaN / -inf entries before solving
    int invalid = 0;
    for (int j = lane; j < nq; j += kWave) {
      const float c = row_cur[j];
      invalid |= (c != c) || (c == -INFINITY);
      remaining[j] = nq - 1 - j;  // reverse fill (rectangular_lsap.cpp: constant matrices give the identity)
      shortest[j] = INFINITY;
      SC[j] = 0;
    }
    for (int g = lane; g < ng; g += kWave) SR[g] = 0;
    if (__ballot(invalid != 0) != 0ull) {
      bad = 1;
      break;
    }
    wave_fence();

    int num_remaining = nq, i = cur, sink = -1;
    double min_val = 0.0;
    while (sink < 0) {
      if (lane == 0) SR[i] = 1;
      const double ui = u[i];
      const float *row = C + (size_t)i * nq;
      // lane-local best under the order-free rule: smallest value; then unassigned beats assigned;
      // then the larger position among unassigned / the smaller among assigned  (tie key below)
      double best = INFINITY;
      int best_key = -0x3FFFFFFF;
      for (int it = lane; it < num_remaining; it += kWave) {
        const int j = remaining[it];
        const double r = min_val + (double)row[j] - ui - v[j];
        double sp = shortest[j];
        if (r < sp) {
          path[j] = i;
          shortest[j] = r;
          sp = r;
        }
        const int key = row4col[j] < 0 ? nq + it : -it;
        if (sp < best || (sp == best && key > best_key)) {
          best = sp;
          best_key = key;
        }
      }
      // wave arg-min: 64-bit ordered key as (hi, lo) 32-bit DPP minima, then the tie key maximum
      const unsigned long long k64 = ordered_key64(best + 0.0);  // -0.0 -> +0.0: C++ compares them equal
      const unsigned hi = (unsigned)(k64 >> 32), lo = (unsigned)k64;
      const unsigned m_hi = waveops::wave_min_u32(hi);
      const unsigned m_lo = waveops::wave_min_u32(hi == m_hi ? lo : 0xFFFFFFFFu);
      const bool is_min = hi == m_hi && lo == m_lo;
      const unsigned tie = is_min ? (unsigned)(best_key + 0x40000000) : 0u;  // keys > -2^30
      const int win_key = (int)waveops::wave_max_u32(tie) - 0x40000000;
      const unsigned long long kmin = ((unsigned long long)m_hi << 32) | m_lo;  // decode the winning value
      const unsigned long long bits = (kmin >> 63) ? (kmin & 0x7FFFFFFFFFFFFFFFull) : ~kmin;
      min_val = __longlong_as_double((long long)bits);
      if (!(min_val < INFINITY)) {  // infeasible (rectangular_lsap.cpp returns -1)
        bad = 1;
        break;
      }
      const int index = win_key >= nq ? win_key - nq : -win_key;
      const int j = remaining[index];
      const int owner = row4col[j];
      if (owner < 0) sink = j; else i = owner;
      wave_fence();
      if (lane == 0) {
        SC[j] = 1;
        remaining[index] = remaining[num_remaining - 1];
      }
      --num_remaining;
      wave_fence();
    }
    if (bad) break;

    // dual update (uses the pre-augmentation col4row)
    for (int g = lane; g < ng; g += kWave) {
      if (g == cur) u[g] += min_val;
      else if (SR[g]) u[g] += min_val - shortest[col4row[g]];
    }
    for (int j = lane; j < nq; j += kWave)
      if (SC[j]) v[j] -= min_val - shortest[j];
    wave_fence();
    // augment along the path (short and sequential)
    if (lane == 0) {
      int j = sink;
      while (true) {
        const int r = path[j];
        row4col[j] = r;
        const int prev = col4row[r];
        col4row[r] = j;
        j = prev;
        if (r == cur) break;
      }
    }
    wave_fence();
  }

  for (int g = lane; g < ng; g += kWave) out[g] = (!bad && ok[g] != 0) ? col4row[g] : -1;
  if (status != nullptr && lane == 0) status[p] = bad;
}

inline size_t lsap_lds_bytes(int nq, int ng) {
  return sizeof(double) * (2 * (size_t)nq + ng) + sizeof(int) * (3 * (size_t)nq + ng) + (size_t)nq + ng + 16;
}

}  // namespace

extern "C" int butd_hungarian_match(int count, int nq, int ng, const float *cost,
                                    const unsigned char *valid, int *match, int *status,
                                    butd_stream_t stream) {
  if (count <= 0 || ng <= 0) return 0;
  if (nq <= 0 || nq > BUTD_LSAP_MAX_QUERIES || ng > BUTD_LSAP_MAX_TARGETS) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(lsap_kernel, dim3(count), dim3(kWave), lsap_lds_bytes(nq, ng), (hipStream_t)stream, nq,
                     ng, cost, valid, match, status);
  return (int)hipGetLastError();
}
