// criterion_ops.hip -- fused terms of the training criterion (include/butd_criterion.h), gfx950.
//
// Every kernel here replaces a few dozen to a few hundred tiny elementwise launches of the reference's
// models/losses.py on (prefix, scene, slot / query) tensors; all are latency / launch-bound, none touches
// more than a few MB.  Forward value and gradient are produced together: each term enters the total loss
// linearly, so its backward pass is a scaling by the incoming scalar.
// Compiled with -ffp-contract=off: the cost tensor feeds an arg-min (the assignment), so its arithmetic
// follows the reference's separate multiply / add launches.
#include <hip/hip_runtime.h>
// (-DBUTD_MAIN_PRIO=n raises the wave priority of this unit's kernels: the captured step's prefetch branches share CUs
// with them; profiles/r06_side_branches.txt)
#ifdef BUTD_MAIN_PRIO
#define BUTD_MAIN_PRIO_SET() __builtin_amdgcn_s_setprio(BUTD_MAIN_PRIO)
#else
#define BUTD_MAIN_PRIO_SET()
#endif


#include "zero_fill.h"
#include <math.h>
#include <stdint.h>

#include "../../include/butd_criterion.h"
#include "wave_ops.h"

namespace {

constexpr int kWave = 64;

__device__ inline float wave_sum_f32(float v) {
#define BUTD_ADD_STEP(CTRL, RMASK)                                                                         \
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, RMASK, 0xf, false))
  BUTD_ADD_STEP(0x111, 0xf);  // row_shr:1
  BUTD_ADD_STEP(0x112, 0xf);  // row_shr:2
  BUTD_ADD_STEP(0x114, 0xf);  // row_shr:4
  BUTD_ADD_STEP(0x118, 0xf);  // row_shr:8
  BUTD_ADD_STEP(0x142, 0xa);  // row_bcast:15
  BUTD_ADD_STEP(0x143, 0xc);  // row_bcast:31
#undef BUTD_ADD_STEP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ---------------------------------------------------------------------------------------------- boxes
struct Corners {
  float lo[3], hi[3];
};

// box_cxcyczwhd_to_xyzxyz (losses.py:27-37)
__device__ inline Corners corners_of(const float *b) {
  Corners c;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float s = fmaxf(b[3 + k], 1e-6f);
    c.lo[k] = b[k] - 0.5f * s;
    c.hi[k] = b[k] + 0.5f * s;
  }
  return c;
}

// generalized_box_iou3d of one pair (losses.py:40-91)
__device__ inline float giou_of(const Corners &a, const Corners &b) {
  float e[3], h[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    e[k] = fmaxf(fminf(a.hi[k], b.hi[k]) - fmaxf(a.lo[k], b.lo[k]), 0.0f);
    h[k] = fmaxf(fmaxf(a.hi[k], b.hi[k]) - fminf(a.lo[k], b.lo[k]), 0.0f);
  }
  const float inter = e[0] * e[1] * e[2];
  const float va = (a.hi[0] - a.lo[0]) * (a.hi[1] - a.lo[1]) * (a.hi[2] - a.lo[2]);
  const float vb = (b.hi[0] - b.lo[0]) * (b.hi[1] - b.lo[1]) * (b.hi[2] - b.lo[2]);
  const float uni = va + vb - inter;
  const float vol = h[0] * h[1] * h[2];
  return inter / uni - (vol - uni) / vol;
}

__global__ __launch_bounds__(256) void match_cost_kernel(int Q, int G, int B,
                                                         const float *__restrict__ pred_boxes,
                                                         const float *__restrict__ tgt_boxes,
                                                         const unsigned char *__restrict__ valid,
                                                         const float *__restrict__ class_cost, float w_bbox,
                                                         float w_class, float w_giou,
                                                         float *__restrict__ cost) {
  BUTD_MAIN_PRIO_SET();
  const int row = blockIdx.x;  // (p*B + b)*G + g
  const int g = row % G;
  const int pb = row / G;
  const int b = pb % B;
  float *out = cost + (size_t)row * Q;
  if (valid[(size_t)b * G + g] == 0) {
    for (int q = threadIdx.x; q < Q; q += 256) out[q] = 0.0f;
    return;
  }
  float t[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) t[k] = tgt_boxes[((size_t)b * G + g) * 6 + k];
  const Corners tc = corners_of(t);
  for (int q = threadIdx.x; q < Q; q += 256) {
    float p[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) p[k] = pred_boxes[((size_t)pb * Q + q) * 6 + k];
    float l1 = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) l1 += fabsf(t[k] - p[k]);
    const float cg = -giou_of(tc, corners_of(p));
    const float cc = class_cost != nullptr ? class_cost[(size_t)row * Q + q] : 0.0f;
    out[q] = w_bbox * l1 + w_class * cc + w_giou * cg;
  }
}

// 1 - GIoU of (src, tgt) and its gradient w.r.t. the src centre+size box (clamp / max / min pass their
// gradient like torch: to the larger / smaller argument, clamp(min) where x >= min)
__device__ inline float giou_loss_grad(const float *src, const float *tgt, float *d) {
  float s[3], ds[3], alo[3], ahi[3], tlo[3], thi[3], e[3], h[3], de_lo[3], de_hi[3], dh_lo[3], dh_hi[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    s[k] = fmaxf(src[3 + k], 1e-6f);
    ds[k] = src[3 + k] >= 1e-6f ? 1.0f : 0.0f;
    alo[k] = src[k] - 0.5f * s[k];
    ahi[k] = src[k] + 0.5f * s[k];
    const float st = fmaxf(tgt[3 + k], 1e-6f);
    tlo[k] = tgt[k] - 0.5f * st;
    thi[k] = tgt[k] + 0.5f * st;
    const float ilo = fmaxf(alo[k], tlo[k]), ihi = fminf(ahi[k], thi[k]);
    const float raw = ihi - ilo;
    e[k] = fmaxf(raw, 0.0f);
    const float pass = raw >= 0.0f ? 1.0f : 0.0f;
    de_lo[k] = alo[k] > tlo[k] ? -pass : (alo[k] == tlo[k] ? -0.5f * pass : 0.0f);
    de_hi[k] = ahi[k] < thi[k] ? pass : (ahi[k] == thi[k] ? 0.5f * pass : 0.0f);
    const float hlo = fminf(alo[k], tlo[k]), hhi = fmaxf(ahi[k], thi[k]);
    h[k] = fmaxf(hhi - hlo, 0.0f);
    dh_lo[k] = alo[k] < tlo[k] ? -1.0f : (alo[k] == tlo[k] ? -0.5f : 0.0f);
    dh_hi[k] = ahi[k] > thi[k] ? 1.0f : (ahi[k] == thi[k] ? 0.5f : 0.0f);
  }
  const float inter = e[0] * e[1] * e[2];
  const float ea[3] = {ahi[0] - alo[0], ahi[1] - alo[1], ahi[2] - alo[2]};
  const float va = ea[0] * ea[1] * ea[2];
  const float vb = (thi[0] - tlo[0]) * (thi[1] - tlo[1]) * (thi[2] - tlo[2]);
  const float uni = va + vb - inter;
  const float vol = h[0] * h[1] * h[2];
  const float loss = 1.0f - (inter / uni - (vol - uni) / vol);
  // loss = 2 - inter/uni - uni/vol
  const float dl_dinter = -1.0f / uni;
  const float dl_duni = inter / (uni * uni) - 1.0f / vol;
  const float dl_dvol = uni / (vol * vol);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
    const float oe = e[k1] * e[k2], oa = ea[k1] * ea[k2], oh = h[k1] * h[k2];
    // derivatives w.r.t. alo[k], ahi[k]
    const float dinter_lo = de_lo[k] * oe, dinter_hi = de_hi[k] * oe;
    const float dva_lo = -oa, dva_hi = oa;
    const float dvol_lo = dh_lo[k] * oh, dvol_hi = dh_hi[k] * oh;
    const float g_lo = dl_dinter * dinter_lo + dl_duni * (dva_lo - dinter_lo) + dl_dvol * dvol_lo;
    const float g_hi = dl_dinter * dinter_hi + dl_duni * (dva_hi - dinter_hi) + dl_dvol * dvol_hi;
    d[k] = g_lo + g_hi;
    d[3 + k] = 0.5f * (g_hi - g_lo) * ds[k];
  }
  return loss;
}

__global__ __launch_bounds__(256) void box_loss_kernel(int B, int Q, int G,
                                                       const float *__restrict__ pred_boxes,
                                                       const float *__restrict__ tgt_boxes,
                                                       const int *__restrict__ match,
                                                       float *__restrict__ sums, float *__restrict__ grad) {
  BUTD_MAIN_PRIO_SET();
  __shared__ float red[2][256 / kWave];
  const int p = blockIdx.x;
  float acc_l1 = 0.0f, acc_g = 0.0f;
  for (int i = threadIdx.x; i < B * G; i += 256) {
    const int b = i / G;
    const size_t slot = (size_t)p * B * G + i;
    const int q = match[slot];
    float *gr = grad + slot * 12;
    if (q < 0) {
#pragma unroll
      for (int k = 0; k < 12; ++k) gr[k] = 0.0f;
      continue;
    }
    float src[6], tgt[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      src[k] = pred_boxes[(((size_t)p * B + b) * Q + q) * 6 + k];
      tgt[k] = tgt_boxes[(size_t)i * 6 + k];
    }
    float l1 = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const float diff = src[k] - tgt[k];
      const float w = k < 3 ? 1.0f : 0.2f;
      l1 += w * fabsf(diff);
      gr[k] = diff > 0.0f ? w : (diff < 0.0f ? -w : 0.0f);
    }
    acc_l1 += l1;
    acc_g += giou_loss_grad(src, tgt, gr + 6);
  }
  const float s0 = wave_sum_f32(acc_l1), s1 = wave_sum_f32(acc_g);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s0;
    red[1][threadIdx.x >> 6] = s1;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const float *r = red[threadIdx.x];
    sums[p * 2 + threadIdx.x] = (r[0] + r[1]) + (r[2] + r[3]);
  }
}

__global__ __launch_bounds__(256) void box_loss_bwd_kernel(int total, int B, int Q, int G,
                                                           const int *__restrict__ match,
                                                           const float *__restrict__ grad,
                                                           const float *__restrict__ w,
                                                           float *__restrict__ grad_pred) {
  BUTD_MAIN_PRIO_SET();
  const int slot = blockIdx.x * 256 + threadIdx.x;  // (p*B + b)*G + g
  if (slot >= total) return;
  const int q = match[slot];
  if (q < 0) return;
  const int pb = slot / G;
  const int p = pb / B;
  const float w0 = w[p * 2], w1 = w[p * 2 + 1];
  const float *gr = grad + (size_t)slot * 12;
  float *out = grad_pred + ((size_t)pb * Q + q) * 6;
#pragma unroll
  for (int k = 0; k < 6; ++k) out[k] = w0 * gr[k] + w1 * gr[6 + k];
}

// ---------------------------------------------------------------------------------------------- rows
// slot g with match[g] == q among G slots (unique), or -1: lanes scan the slots
__device__ inline int owner_of(const int *__restrict__ match_pb, int G, int q, int lane) {
  int found = -1;
  for (int g0 = 0; g0 < G; g0 += kWave) {
    const int g = g0 + lane;
    const bool hit = g < G && match_pb[g] == q;
    const unsigned long long m = __ballot(hit);
    if (m != 0ull) found = g0 + __ffsll((long long)m) - 1;
  }
  return found;
}

constexpr int kRowsPerBlock = 4;

// losses.py:355-390; one wave per (p,b,q) row
__global__ __launch_bounds__(kWave * kRowsPerBlock) void soft_token_ce_kernel(
    int rows, int Q, int G, int C, const float *__restrict__ logits, const int *__restrict__ match,
    const float *__restrict__ positive_map, int ldpm, int B, float eos_coef, float *__restrict__ row_loss,
    float *__restrict__ dlogits) {
  BUTD_MAIN_PRIO_SET();
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int q = row % Q, pb = row / Q, b = pb % B;
  const int g = owner_of(match + (size_t)pb * G, G, q, lane);
  const float *x = logits + (size_t)row * C;
  const float *t_row = g >= 0 ? positive_map + ((size_t)b * G + g) * ldpm : nullptr;
  float mx = -INFINITY;
  for (int c = lane; c < C; c += kWave) mx = fmaxf(mx, x[c]);
  mx = waveops::wave_max_f32(mx);
  float se = 0.0f, st = 0.0f, ent = 0.0f, tx = 0.0f;
  for (int c = lane; c < C; c += kWave) {
    const float t = t_row != nullptr ? t_row[c] : (c == C - 1 ? 1.0f : 0.0f);
    se += expf(x[c] - mx);
    st += t;
    ent += logf(t + 1e-6f) * t;
    tx += t * (x[c] - mx);
  }
  se = wave_sum_f32(se);
  st = wave_sum_f32(st);
  ent = wave_sum_f32(ent);
  tx = wave_sum_f32(tx);
  const float lse = logf(se);  // relative to mx
  const float w = g >= 0 ? 1.0f : eos_coef;
  if (lane == 0) row_loss[row] = w * (ent - (tx - st * lse));
  float *dx = dlogits + (size_t)row * C;
  const float inv = 1.0f / se;
  for (int c = lane; c < C; c += kWave) {
    const float t = t_row != nullptr ? t_row[c] : (c == C - 1 ? 1.0f : 0.0f);
    dx[c] = w * (expf(x[c] - mx) * inv * st - t);
  }
}

// losses.py:420-474 ("Loss 1"); one wave per (p,b,q) row of L tokens
__global__ __launch_bounds__(kWave * kRowsPerBlock) void contrastive_rows_kernel(
    int rows, int Q, int G, int L, int B, const float *__restrict__ logits, const int *__restrict__ match,
    const float *__restrict__ positive_map, int ldpm, const int *__restrict__ last, float eos_coef,
    float *__restrict__ row_loss, float *__restrict__ dlogits, int *__restrict__ owner) {
  BUTD_MAIN_PRIO_SET();
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int q = row % Q, pb = row / Q, b = pb % B;
  const int g = owner_of(match + (size_t)pb * G, G, q, lane);
  if (lane == 0) owner[row] = g;
  const float *x = logits + (size_t)row * L;
  const float *pm = g >= 0 ? positive_map + ((size_t)b * G + g) * ldpm : nullptr;
  const int l_last = last[b];
  const int l_prev = ((l_last - 1) % L + L) % L;  // python indexing: -1 is the last column
  float mx = -INFINITY;
  for (int l = lane; l < L; l += kWave) mx = fmaxf(mx, x[l]);
  mx = waveops::wave_max_f32(mx);
  float se = 0.0f, cnt = 0.0f, ps = 0.0f;
  for (int l = lane; l < L; l += kWave) {
    const bool pos = pm != nullptr ? pm[l] > 0.0f : (l == l_last || l == l_prev);
    se += expf(x[l] - mx);
    cnt += pos ? 1.0f : 0.0f;
    ps += pos ? x[l] : 0.0f;
  }
  se = wave_sum_f32(se);
  cnt = wave_sum_f32(cnt);
  ps = wave_sum_f32(ps);
  const bool has_pos = cnt > 0.0f;
  const float nb = cnt + 1e-6f;
  const float w = 0.5f * (g >= 0 ? 1.0f : eos_coef);
  const float lse = logf(se) + mx;
  if (lane == 0) row_loss[row] = has_pos ? w * (-logf(nb + 1e-6f) / nb + (-ps) / nb + lse) : 0.0f;
  float *dx = dlogits + (size_t)row * L;
  const float inv = 1.0f / se, scale = has_pos ? w : 0.0f;
  for (int l = lane; l < L; l += kWave) {
    const bool pos = pm != nullptr ? pm[l] > 0.0f : (l == l_last || l == l_prev);
    dx[l] = scale * (expf(x[l] - mx) * inv - (pos ? 1.0f / nb : 0.0f));
  }
}

// losses.py:476-487 ("Loss 2"); one workgroup per (p,b) and 64-column chunk; thread = (column l, one of 16
// query strides), partial (max, sum-exp, count, sum) merged through LDS
constexpr int kColThreads = 1024;

__global__ __launch_bounds__(kColThreads) void contrastive_cols_kernel(
    int Q, int G, int L, int B, const float *__restrict__ logits, const int *__restrict__ owner,
    const float *__restrict__ positive_map, int ldpm, const int *__restrict__ last, float eos_coef,
    float *__restrict__ col_loss, float *__restrict__ dlogits) {
  BUTD_MAIN_PRIO_SET();
  extern __shared__ float smem[];
  const int pb = blockIdx.x, b = pb % B;
  const int l_last = last[b];
  const int l_prev = ((l_last - 1) % L + L) % L;
  const float *x = logits + (size_t)pb * Q * L;
  const int *own = owner + (size_t)pb * Q;
  const float *pm_b = positive_map + (size_t)b * G * ldpm;
  float *dx = dlogits + (size_t)pb * Q * L;
  {
    const int l0 = blockIdx.y * 64;               // one 64-column chunk per workgroup
    const int cols = (L - l0) < 64 ? (L - l0) : 64;
    const int parts = kColThreads / 64;           // 16 query strides
    const int lc = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int l = l0 + lc;
    const bool active = lc < cols;
    // pass 1: online max / sum-exp, positive count and sum over this thread's queries
    float mx = -INFINITY, se = 0.0f, cnt = 0.0f, ps = 0.0f;
    if (active) {
      for (int q = part; q < Q; q += parts) {
        const float v = x[(size_t)q * L + l];
        const int g = own[q];
        const bool pos = g >= 0 ? pm_b[(size_t)g * ldpm + l] > 0.0f : (l == l_last || l == l_prev);
        if (v > mx) {
          se = se * expf(mx - v) + 1.0f;
          mx = v;
        } else {
          se += expf(v - mx);
        }
        cnt += pos ? 1.0f : 0.0f;
        ps += pos ? v : 0.0f;
      }
    }
    float *s_mx = smem, *s_se = smem + kColThreads, *s_cnt = smem + 2 * kColThreads, *s_ps = smem + 3 * kColThreads;
    s_mx[threadIdx.x] = mx;
    s_se[threadIdx.x] = se;
    s_cnt[threadIdx.x] = cnt;
    s_ps[threadIdx.x] = ps;
    __syncthreads();
    float M = -INFINITY, S = 0.0f, N = 0.0f, PS = 0.0f;
    for (int t = 0; t < parts; ++t) M = fmaxf(M, s_mx[t * 64 + lc]);
    for (int t = 0; t < parts; ++t) {
      const float m_t = s_mx[t * 64 + lc];
      S += m_t > -INFINITY ? s_se[t * 64 + lc] * expf(m_t - M) : 0.0f;
      N += s_cnt[t * 64 + lc];
      PS += s_ps[t * 64 + lc];
    }
    __syncthreads();
    const bool has_pos = N > 0.0f;
    const float nb = N + 1e-6f;
    const float w = 0.5f * (l == l_last ? 1.0f : eos_coef);
    if (active && part == 0)
      col_loss[(size_t)pb * L + l] = has_pos ? w * (-logf(nb + 1e-6f) / nb + (-PS) / nb + (logf(S) + M)) : 0.0f;
    if (active && has_pos) {
      const float inv = 1.0f / S;
      for (int q = part; q < Q; q += parts) {
        const float v = x[(size_t)q * L + l];
        const int g = own[q];
        const bool pos = g >= 0 ? pm_b[(size_t)g * ldpm + l] > 0.0f : (l == l_last || l == l_prev);
        dx[(size_t)q * L + l] += w * (expf(v - M) * inv - (pos ? 1.0f / nb : 0.0f));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- seeds
// losses.py:171-207: one wave per (scene, box slot): distances of the K seeds in LDS, `topk` rounds of
// wave arg-min over (distance bits, seed index)
__global__ __launch_bounds__(kWave) void objectness_label_kernel(
    int K, int G, int N, int topk, const float *__restrict__ seed_xyz, const int *__restrict__ seed_inds,
    const int64_t *__restrict__ pil, const float *__restrict__ gt_center, const float *__restrict__ gt_size,
    const float *__restrict__ box_mask, unsigned char *__restrict__ label) {
  BUTD_MAIN_PRIO_SET();
  extern __shared__ float dist[];
  const int lane = threadIdx.x;
  const int bg = blockIdx.x, b = bg / G, g = bg % G;
  if (!(box_mask[bg] > 0.0f)) return;
  float c[3], inv[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    c[a] = gt_center[(size_t)bg * 3 + a];
    inv[a] = gt_size[(size_t)bg * 3 + a] + 1e-6f;
  }
  for (int k = lane; k < K; k += kWave) {
    const long long obj = pil[(size_t)b * N + seed_inds[(size_t)b * K + k]];
    const int owner = obj < 0 ? G - 1 : (int)obj;
    float d = 100.0f;
    if (owner == g) {
      float acc = 0.0f;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float t = (seed_xyz[((size_t)b * K + k) * 3 + a] - c[a]) / inv[a];
        acc += t * t;
      }
      d = sqrtf(acc + 1e-6f);
    }
    dist[k] = d;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  for (int r = 0; r < topk && r < K; ++r) {
    unsigned best_d = 0xFFFFFFFFu, best_k = 0xFFFFFFFFu;   // distances are >= 0: float bits order like uints
    for (int k = lane; k < K; k += kWave) {
      const unsigned bits = __float_as_uint(dist[k]);
      if (bits < best_d) {
        best_d = bits;
        best_k = (unsigned)k;
      }
    }
    const unsigned m_d = waveops::wave_min_u32(best_d);
    const unsigned m_k = waveops::wave_min_u32(best_d == m_d ? best_k : 0xFFFFFFFFu);
    if (m_k == 0xFFFFFFFFu) break;  // everything taken
    if (lane == 0) {
      label[(size_t)b * K + m_k] = 1;
      dist[m_k] = INFINITY;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
}

// losses.py:204-221 + SigmoidFocalClassificationLoss (:94-158), value and derivative
__global__ __launch_bounds__(256) void objectness_focal_kernel(int total, int K, int N,
                                                               const int *__restrict__ seed_inds,
                                                               const int64_t *__restrict__ pil,
                                                               const float *__restrict__ logits,
                                                               unsigned char *__restrict__ label,
                                                               float *__restrict__ elem_loss,
                                                               float *__restrict__ dlogits) {
  BUTD_MAIN_PRIO_SET();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int b = i / K;
  const bool fg = pil[(size_t)b * N + seed_inds[i]] >= 0;
  const float t = (label[i] != 0 && fg) ? 1.0f : 0.0f;
  label[i] = (unsigned char)t;
  const float x = logits[i];
  const float p = 1.0f / (1.0f + expf(-x));
  const float alpha_w = t * 0.25f + (1.0f - t) * 0.75f;
  const float pt = t * (1.0f - p) + (1.0f - t) * p;
  const float bce = fmaxf(x, 0.0f) - x * t + log1pf(expf(-fabsf(x)));
  const float w = 1.0f / (float)(K > 1 ? K : 1);
  elem_loss[i] = alpha_w * pt * pt * bce * w;
  const float dpt = (1.0f - 2.0f * t) * p * (1.0f - p);
  dlogits[i] = w * alpha_w * (2.0f * pt * dpt * bce + pt * pt * (p - t));
}

// loss = w_gen * generation + w_sum * (sum ce + w_bbox * sum bbox + sum giou + sum align), NaN if any status word != 0;
// out = [loss, sum ce, sum bbox, sum giou, sum align]  (losses.py:592-617: the sums over the prefixes + the weighting)
__global__ void loss_combine_kernel(int P, const float *ce, const float *bbox, const float *giou, const float *align,
                                    const float *generation, const int *status_words, int nstatus, float w_gen,
                                    float w_sum, float w_bbox, float *out) {
  BUTD_MAIN_PRIO_SET();
  if (threadIdx.x != 0) return;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  const float *src[4] = {ce, bbox, giou, align};
  for (int t = 0; t < 4; ++t)
    if (src[t])
      for (int i = 0; i < P; ++i) s[t] += src[t][i];
  const float gen = generation ? generation[0] : 0.f;
  float loss = w_gen * gen + w_sum * (((s[0] + w_bbox * s[1]) + s[2]) + s[3]);
  bool bad = false;
  for (int i = 0; i < nstatus; ++i) bad = bad || status_words[i] != 0;
  out[0] = bad ? nanf("") : loss;
  out[1] = s[0]; out[2] = s[1]; out[3] = s[2]; out[4] = s[3];
}

// gradients of the above for an upstream scalar g: every element of a term gets g * its weight -- and 0 when an assignment
// failed (the stock expression is torch.where(bad, nan, loss): no gradient reaches the terms of an invalid match)
__global__ void loss_combine_bwd_kernel(int P, const float *g, const int *status_words, int nstatus, float w_gen,
                                        float w_sum, float w_bbox, float *d_ce, float *d_bbox, float *d_giou,
                                        float *d_align, float *d_generation) {
  BUTD_MAIN_PRIO_SET();
  const int i = threadIdx.x;
  bool bad = false;
  for (int j = 0; j < nstatus; ++j) bad = bad || status_words[j] != 0;
  const float gv = bad ? 0.f : g[0];
  if (i < P) {
    if (d_ce) d_ce[i] = gv * w_sum;
    if (d_bbox) d_bbox[i] = gv * w_sum * w_bbox;
    if (d_giou) d_giou[i] = gv * w_sum;
    if (d_align) d_align[i] = gv * w_sum;
  }
  if (i == 0 && d_generation) d_generation[0] = gv * w_gen;
}

// ---- the whole tail of compute_hungarian_loss in one workgroup (round 5) -------------------------------------------------
// per prefix p: ce = sum ce_rows[p] / nb, bbox = box_sums[p][0] / nb, giou = box_sums[p][1] / nb,
//               align = (sum align_rows[p] + sum align_cols[p]) / nb        (nb = the device scalar num_boxes)
// generation = sum gen_elem / gen_div;  then loss_combine_kernel's expression.  One workgroup of 1024 threads walks the
// prefixes (41 K floats at the bench shape): fixed order, no atomics.  The stock graph was 3 row reductions, 4 divisions,
// the objectness reduction + division and the combine launch, each 4.6 us as a graph node.
__device__ float block_sum_1024(float v, float *red) {
  v = wave_sum_f32(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) t += red[w];
  return t;
}
__global__ __launch_bounds__(1024) void criterion_reduce_kernel(
    int P, const float *__restrict__ ce_rows, long n_ce, const float *__restrict__ box_sums,
    const float *__restrict__ align_rows, long n_ar, const float *__restrict__ align_cols, long n_ac,
    const float *__restrict__ gen_elem, long n_gen, float gen_div, const float *__restrict__ num_boxes,
    const int *__restrict__ status_words, int nstatus, float w_gen, float w_sum, float w_bbox,
    float *__restrict__ per_prefix, float *__restrict__ out6) {
  BUTD_MAIN_PRIO_SET();
  __shared__ float red[16];
  const float inv_nb = 1.f / num_boxes[0];
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int p = 0; p < P; ++p) {
    float ce = 0.f, al = 0.f;
    if (ce_rows) {
      float a = 0.f;
      for (long i = threadIdx.x; i < n_ce; i += 1024) a += ce_rows[p * n_ce + i];
      ce = block_sum_1024(a, red) * inv_nb;
    }
    if (align_rows) {
      float a = 0.f;
      for (long i = threadIdx.x; i < n_ar; i += 1024) a += align_rows[p * n_ar + i];
      float b = 0.f;
      for (long i = threadIdx.x; i < n_ac; i += 1024) b += align_cols[p * n_ac + i];
      al = (block_sum_1024(a, red) + block_sum_1024(b, red)) * inv_nb;
    }
    const float bb = box_sums[2 * p] * inv_nb, gi = box_sums[2 * p + 1] * inv_nb;
    if (threadIdx.x == 0) {
      per_prefix[4 * p] = ce; per_prefix[4 * p + 1] = bb; per_prefix[4 * p + 2] = gi; per_prefix[4 * p + 3] = al;
    }
    s[0] += ce; s[1] += bb; s[2] += gi; s[3] += al;
  }
  float gen = 0.f;
  if (gen_elem) {
    float a = 0.f;
    for (long i = threadIdx.x; i < n_gen; i += 1024) a += gen_elem[i];
    gen = block_sum_1024(a, red) / gen_div;
  }
  if (threadIdx.x == 0) {
    const float loss = w_gen * gen + w_sum * (((s[0] + w_bbox * s[1]) + s[2]) + s[3]);
    bool bad = false;
    for (int i = 0; i < nstatus; ++i) bad = bad || status_words[i] != 0;
    out6[0] = bad ? nanf("") : loss;
    out6[1] = s[0]; out6[2] = s[1]; out6[3] = s[2]; out6[4] = s[3]; out6[5] = gen;
  }
}

// its gradient for an upstream device scalar g: the saved derivatives of the row sums times c = g w_sum / nb (0 when an
// assignment failed), the objectness derivative times g w_gen / gen_div, and the (P, 2) weights of butd_box_loss_bwd --
// three element-wise tensors in one launch (float4 where the tensor allows)
__global__ __launch_bounds__(256) void criterion_scale_kernel(
    int P, const float *__restrict__ g, const float *__restrict__ num_boxes, const int *__restrict__ status_words,
    int nstatus, float w_gen, float w_sum, float w_bbox, float gen_div, const float *__restrict__ dx_ce,
    float *__restrict__ d_logits, long n_ce, const float *__restrict__ dx_al, float *__restrict__ d_align, long n_al,
    const float *__restrict__ dx_gen, float *__restrict__ d_seed, long n_gen, float *__restrict__ box_w) {
  BUTD_MAIN_PRIO_SET();
  bool bad = false;
  for (int j = 0; j < nstatus; ++j) bad = bad || status_words[j] != 0;
  const float gv = bad ? 0.f : g[0];
  const float c = gv * w_sum / num_boxes[0], cg = gv * w_gen / gen_div;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < 2 * P && box_w) box_w[i] = (i & 1) ? c : c * w_bbox;
  const long q_ce = (n_ce + 3) >> 2, q_al = (n_al + 3) >> 2, q_gen = (n_gen + 3) >> 2;
  auto scale4 = [&](const float *src, float *dst, long n, long q, float k) {
    if (4 * q + 4 <= n && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
      const float4 v = *reinterpret_cast<const float4 *>(src + 4 * q);
      *reinterpret_cast<float4 *>(dst + 4 * q) = make_float4(v.x * k, v.y * k, v.z * k, v.w * k);
    } else {
      for (long e = 4 * q; e < n && e < 4 * q + 4; ++e) dst[e] = src[e] * k;
    }
  };
  if (i < q_ce) scale4(dx_ce, d_logits, n_ce, i, c);
  else if (i < q_ce + q_al) scale4(dx_al, d_align, n_al, i - q_ce, c);
  else if (i < q_ce + q_al + q_gen) scale4(dx_gen, d_seed, n_gen, i - q_ce - q_al, cg);
}

inline int status() { return (int)hipGetLastError(); }

}  // namespace

extern "C" {

int butd_match_cost(int P, int B, int Q, int G, const float *pred_boxes, const float *tgt_boxes,
                    const unsigned char *valid, const float *class_cost, float w_bbox, float w_class,
                    float w_giou, float *cost, butd_stream_t stream) {
  if (P <= 0 || B <= 0 || Q <= 0 || G <= 0) return 0;
  if (class_cost == nullptr && w_class != 0.0f) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(match_cost_kernel, dim3((unsigned)(P * B * G)), dim3(256), 0, (hipStream_t)stream, Q, G, B,
                     pred_boxes, tgt_boxes, valid, class_cost, w_bbox, w_class, w_giou, cost);
  return status();
}

int butd_box_loss(int P, int B, int Q, int G, const float *pred_boxes, const float *tgt_boxes,
                  const int *match, float *sums, float *grad, butd_stream_t stream) {
  if (P <= 0) return 0;
  hipLaunchKernelGGL(box_loss_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, B, Q, G, pred_boxes, tgt_boxes,
                     match, sums, grad);
  return status();
}

int butd_box_loss_bwd(int P, int B, int Q, int G, const int *match, const float *grad, const float *w,
                      float *grad_pred, butd_stream_t stream) {
  if (P <= 0 || B <= 0 || Q <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = butd_zero_async(grad_pred, sizeof(float) * (size_t)P * B * Q * 6, s);
  if (e != hipSuccess) return (int)e;
  const int total = P * B * G;
  if (total > 0)
    hipLaunchKernelGGL(box_loss_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, s, total, B, Q, G, match,
                       grad, w, grad_pred);
  return status();
}

int butd_soft_token_ce(int P, int B, int Q, int G, int C, const float *logits, const int *match,
                       const float *positive_map, int ldpm, float eos_coef, float *row_loss,
                       float *dlogits, butd_stream_t stream) {
  const int rows = P * B * Q;
  if (rows <= 0 || C <= 0) return 0;
  if (ldpm < C) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(soft_token_ce_kernel, dim3((rows + kRowsPerBlock - 1) / kRowsPerBlock),
                     dim3(kWave * kRowsPerBlock), 0, (hipStream_t)stream, rows, Q, G, C, logits, match,
                     positive_map, ldpm, B, eos_coef, row_loss, dlogits);
  return status();
}

int butd_contrastive_rows(int P, int B, int Q, int G, int L, const float *logits, const int *match,
                          const float *positive_map, int ldpm, const int *last, float eos_coef,
                          float *row_loss, float *dlogits, int *owner, butd_stream_t stream) {
  const int rows = P * B * Q;
  if (rows <= 0 || L <= 0) return 0;
  if (ldpm < L) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(contrastive_rows_kernel, dim3((rows + kRowsPerBlock - 1) / kRowsPerBlock),
                     dim3(kWave * kRowsPerBlock), 0, (hipStream_t)stream, rows, Q, G, L, B, logits, match,
                     positive_map, ldpm, last, eos_coef, row_loss, dlogits, owner);
  return status();
}

int butd_contrastive_cols(int P, int B, int Q, int G, int L, const float *logits, const int *owner,
                          const float *positive_map, int ldpm, const int *last, float eos_coef,
                          float *col_loss, float *dlogits, butd_stream_t stream) {
  if (P * B <= 0 || L <= 0 || Q <= 0) return 0;
  if (ldpm < L) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(contrastive_cols_kernel, dim3((unsigned)(P * B), (unsigned)((L + 63) / 64)), dim3(kColThreads),
                     sizeof(float) * 4 * kColThreads, (hipStream_t)stream, Q, G, L, B, logits, owner,
                     positive_map, ldpm, last, eos_coef, col_loss, dlogits);
  return status();
}

int butd_seed_objectness(int B, int K, int G, int N, int topk, const float *seed_xyz, const int *seed_inds,
                         const int64_t *point_instance_label, const float *gt_center, const float *gt_size,
                         const float *box_mask, const float *logits, unsigned char *label, float *elem_loss,
                         float *dlogits, butd_stream_t stream) {
  if (B <= 0 || K <= 0) return 0;
  if (topk < 0 || topk > 32 || G <= 0) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = butd_zero_async(label, (size_t)B * K, s);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(objectness_label_kernel, dim3((unsigned)(B * G)), dim3(kWave), sizeof(float) * K, s, K, G, N,
                     topk, seed_xyz, seed_inds, point_instance_label, gt_center, gt_size, box_mask, label);
  const int total = B * K;
  hipLaunchKernelGGL(objectness_focal_kernel, dim3((total + 255) / 256), dim3(256), 0, s, total, K, N, seed_inds,
                     point_instance_label, logits, label, elem_loss, dlogits);
  return status();
}

int butd_loss_combine(int P, const float *loss_ce, const float *loss_bbox, const float *loss_giou,
                      const float *loss_align, const float *generation, const int *status_words, int nstatus,
                      float w_gen, float w_sum, float w_bbox, float *out5, butd_stream_t stream) {
  if (P <= 0 || P > 64 || !loss_bbox || !out5) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(kWave), 0, (hipStream_t)stream, P, loss_ce, loss_bbox, loss_giou,
                     loss_align, generation, status_words, status_words ? nstatus : 0, w_gen, w_sum, w_bbox, out5);
  return status();
}

int butd_loss_combine_bwd(int P, const float *g, const int *status_words, int nstatus, float w_gen, float w_sum,
                          float w_bbox, float *d_ce, float *d_bbox, float *d_giou, float *d_align, float *d_generation,
                          butd_stream_t stream) {
  if (P <= 0 || P > 64 || !g) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(loss_combine_bwd_kernel, dim3(1), dim3(kWave), 0, (hipStream_t)stream, P, g, status_words,
                     status_words ? nstatus : 0, w_gen, w_sum, w_bbox, d_ce, d_bbox, d_giou, d_align, d_generation);
  return status();
}

int butd_criterion_reduce(int P, const float *ce_rows, long n_ce, const float *box_sums, const float *align_rows,
                          long n_align_rows, const float *align_cols, long n_align_cols, const float *gen_elem,
                          long n_gen, float gen_div, const float *num_boxes, const int *status_words, int nstatus,
                          float w_gen, float w_sum, float w_bbox, float *per_prefix, float *out6,
                          butd_stream_t stream) {
  if (P <= 0 || P > 64 || !box_sums || !num_boxes || !per_prefix || !out6) return (int)hipErrorInvalidValue;
  if ((align_rows == nullptr) != (align_cols == nullptr) || (gen_elem && gen_div == 0.f)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(criterion_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, P, ce_rows, n_ce, box_sums,
                     align_rows, n_align_rows, align_cols, n_align_cols, gen_elem, n_gen, gen_div, num_boxes, status_words,
                     status_words ? nstatus : 0, w_gen, w_sum, w_bbox, per_prefix, out6);
  return status();
}

int butd_criterion_scale(int P, const float *g, const float *num_boxes, const int *status_words, int nstatus,
                         float w_gen, float w_sum, float w_bbox, float gen_div, const float *dx_ce, float *d_logits,
                         long n_ce, const float *dx_align, float *d_align, long n_align, const float *dx_gen,
                         float *d_seed, long n_gen, float *box_w, butd_stream_t stream) {
  if (P <= 0 || P > 64 || !g || !num_boxes) return (int)hipErrorInvalidValue;
  if (!dx_ce) n_ce = 0;
  if (!dx_align) n_align = 0;
  if (!dx_gen) n_gen = 0;
  if ((n_ce && !d_logits) || (n_align && !d_align) || (n_gen && (!d_seed || gen_div == 0.f))) return (int)hipErrorInvalidValue;
  long q = ((n_ce + 3) >> 2) + ((n_align + 3) >> 2) + ((n_gen + 3) >> 2);
  if (q < 2 * P) q = 2 * P;
  hipLaunchKernelGGL(criterion_scale_kernel, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P, g,
                     num_boxes, status_words, status_words ? nstatus : 0, w_gen, w_sum, w_bbox, gen_div ? gen_div : 1.f,
                     dx_ce, d_logits, n_ce, dx_align, d_align, n_align, dx_gen, d_seed, n_gen, box_w);
  return status();
}

}  // extern "C"
