// pointnet2_ops.hip -- gfx950 (MI355X, CDNA4) kernels behind include/butd_pointnet2.h.
//
// Written for 64-lane wavefronts / 256 CUs; NOT a translation of the reference's
// one-block-per-scene CUDA kernels (pointnet2/_ext_src/src/*.cu).  What is kept from the
// reference is the arithmetic contract only: fp32, the expression order of the .cu sources,
// one rounding per operation -- this TU is compiled with -ffp-contract=off so hipcc does not
// fuse a*a+b*b into v_fma_f32 (which would change low bits and hence indices).
//
// Kernels
//   fps_kernel<PPT, XYZ_IN_REGS>  furthest point sampling, one 1024-thread workgroup per scene,
//                                 running min-distance kept in VGPRs, one barrier per iteration.
//   ball_query_kernel<C>          C centres per wave (state in SGPRs), 64 points per step (one per
//                                 lane, coalesced), v_cmp mask + popcount = ordered compaction.
//   gather/group/interpolate      bandwidth kernels, coalesced on the index/output side.
#include <hip/hip_runtime.h>

#include "zero_fill.h"
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/butd_pointnet2.h"
#include "fps_common.h"

namespace {

constexpr int kWave = 64;

__host__ __device__ inline int ilog2_floor(unsigned v) {
  int r = 0;
  while (v >>= 1) ++r;
  return r;
}

// ----------------------------------------------------------------------------------------------
// FPS (small / medium clouds; large ones take the pruned path in fps_pruned.hip)
// ----------------------------------------------------------------------------------------------
// One workgroup per scene, coordinates + running min distance + tie-break key of every owned point
// in VGPRs.  The iteration is a pure latency chain (update -> in-wave arg-max -> cross-wave arg-max ->
// next sample), and every resident wave replays the reduction/selection instructions on its SIMD, so
// the kernel runs ONE wave per SIMD (256 threads) and gives each lane up to 16 points instead of
// spreading 1024 threads over the points: 2-3x shorter iterations at n <= 4096.
template <int THREADS, int PPT>
__global__ __launch_bounds__(THREADS) void fps_kernel(int n, int m, int log2bs,
                                                      const float *__restrict__ dataset,
                                                      int *__restrict__ idxs,
                                                      const int *__restrict__ verdict, int verdict_stride) {
  constexpr int kNumWaves = THREADS / kWave;
  __shared__ fps::Slot slots[2][fps::kMaxWaves];
  const int tid = threadIdx.x;
  // the prefix check (below) proved that this scene's samples are 0, 1, ..., m-1: write them, skip the serial chain
  if (verdict != nullptr && verdict[(size_t)blockIdx.x * verdict_stride] == 0) {
    for (int j = tid; j < m; j += THREADS) idxs[(size_t)blockIdx.x * m + j] = j;
    return;
  }
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float *pts = dataset + (size_t)blockIdx.x * n * 3;
  int *out = idxs + (size_t)blockIdx.x * m;

  // running min distance per owned point; -1 marks "never competes" (skipped or out of range):
  // min(d, -1) = -1 never beats a candidate, exactly like `continue` in the reference.
  float t[PPT], px[PPT], py[PPT], pz[PPT];
  unsigned key[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = tid + i * THREADS;
    float x = 0.f, y = 0.f, z = 0.f;
    if (k < n) {
      x = pts[k * 3 + 0];
      y = pts[k * 3 + 1];
      z = pts[k * 3 + 2];
    }
    t[i] = (k < n && !fps::skipped(x, y, z)) ? 1e10f : -1.0f;
    px[i] = x;
    py[i] = y;
    pz[i] = z;
    key[i] = fps::key_of((unsigned)k, log2bs);
  }
  const float p0x = pts[0], p0y = pts[1], p0z = pts[2];
  float x1 = p0x, y1 = p0y, z1 = p0z;
  if (tid == 0) out[0] = 0;

  for (int j = 1; j < m; ++j) {
    float best = -1.0f;
    unsigned bkey = 0xFFFFFFFFu;
    float bx = 0.f, by = 0.f, bz = 0.f;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float d = (px[i] - x1) * (px[i] - x1) + (py[i] - y1) * (py[i] - y1) +
                      (pz[i] - z1) * (pz[i] - z1);
      const float d2 = fminf(d, t[i]);
      t[i] = d2;
      // value descending, then key ascending (the reference's tie order, fps_common.h)
      const bool better = d2 > best || (d2 == best && key[i] < bkey);
      bkey = better ? key[i] : bkey;
      bx = better ? px[i] : bx;
      by = better ? py[i] : by;
      bz = better ? pz[i] : bz;
      best = better ? d2 : best;
    }
    fps::Slot *buf = slots[j & 1];
    fps::publish_wave_best(buf, wave, lane, best >= 0.0f, __float_as_uint(best), bkey, bx, by, bz);
    __syncthreads();
    const int old = fps::select_global_best<kNumWaves>(buf, lane, log2bs, p0x, p0y, p0z, x1, y1, z1);
    if (tid == 0) out[j] = old;
  }
}

// ----------------------------------------------------------------------------------------------
// FPS of a cloud that is ALREADY in furthest-point order (levels 2-4 of the backbone: their input is the
// previous level's samples in selection order, and the reference's own comments at models/backbone_module.py:131-140
// say the indices come out as 0..m-1).  Instead of assuming it, DECIDE it with fully parallel work and run the serial
// kernel only where the answer is no.  With the samples so far being 0..j-1, iteration j of sampling_gpu.cu:100-117
// holds temp[k] = min(1e10, d(0,k), ..., d(j-1,k)) =: r_j(k) for every point that is not skipped (:105-106), and picks
// the maximum of (r_j(k), -key(k)) -- key = the reference's tie order (fps_common.h), a total order on points.  So
// the samples are 0..m-1 exactly when, for every j in 1..m-1, point j is not skipped and no other competing k has
// r_j(k) > r_j(j), or r_j(k) == r_j(j) with a smaller key.  (Already selected points compete with r = 0, as in the
// reference.)  The test is exact, not conservative: verdict 0 <=> the serial kernel would write 0..m-1.
// d is the serial kernel's expression (this TU is built with -ffp-contract=off), symmetric in its two points bit for
// bit.
//   fps_prefix_threshold_kernel: T_j = r_j(j) for j < m, the i-range cut in `parts` pieces (partial minima, folded
//                                when the next kernel stages them): m^2/2 distances per scene, no serial chain
//   fps_prefix_check_kernel    : one thread per point k walks j = 1..m-1 keeping r_j(k) in a register and compares
//                                with T_j from LDS: n*m distances per scene over n/256 workgroups, no barrier in
//                                the loop; a wave that saw a violation leaves the loop
// Workspace = the caller's `temp` (b, n): slot 0 of a scene is the verdict (0 = the samples are 0..m-1; T_0 is never
// needed), slots part*m + j hold the partial thresholds (parts * m <= n).
constexpr int kPrefixMaxM = 2048;      // 32 KB of LDS per workgroup
constexpr int kPrefixMaxN = 8192;      // the register-resident serial kernels' range

__device__ inline float fps_dist(float x2, float y2, float z2, float x1, float y1, float z1) {
  return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
}

__global__ __launch_bounds__(64) void fps_prefix_threshold_kernel(int n, int m, int parts,
                                                                  const float *__restrict__ dataset,
                                                                  float *__restrict__ temp) {
  __shared__ float4 pts_s[kPrefixMaxM];
  const int scene = blockIdx.z, part = blockIdx.y;
  const int j0 = blockIdx.x * 64, lane = threadIdx.x;
  const float *pts = dataset + (size_t)scene * n * 3;
  float *ws = temp + (size_t)scene * n;
  const int j = j0 + lane;
  // this wave's lanes need the points [lo, hi) of the prefix: part p of lane j covers i in [p*j/parts, (p+1)*j/parts)
  const int lo = (int)(((long long)part * j0) / parts);
  int jl = j0 + 63; if (jl > m - 1) jl = m - 1;
  const int hi = (int)(((long long)(part + 1) * jl) / parts);
  for (int i = lo + lane; i < hi; i += 64)
    pts_s[i] = make_float4(pts[i * 3 + 0], pts[i * 3 + 1], pts[i * 3 + 2], 0.f);
  if (blockIdx.x == 0 && part == 0 && lane == 0) reinterpret_cast<int *>(ws)[0] = 0;   // verdict: nothing failed yet
  __syncthreads();
  if (j >= m || j == 0) return;
  const float xj = pts[j * 3 + 0], yj = pts[j * 3 + 1], zj = pts[j * 3 + 2];
  const int a = (int)(((long long)part * j) / parts), e = (int)(((long long)(part + 1) * j) / parts);
  // eight independent distances per trip: the loop is a latency chain otherwise (one wave per SIMD)
  float t = 1e10f;
  int i = a;
  for (; i + 8 <= e; i += 8) {
    float d[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float4 p = pts_s[i + u];
      d[u] = fps_dist(xj, yj, zj, p.x, p.y, p.z);
    }
    t = fminf(t, fminf(fminf(fminf(d[0], d[1]), fminf(d[2], d[3])), fminf(fminf(d[4], d[5]), fminf(d[6], d[7]))));
  }
  for (; i < e; ++i) {
    const float4 p = pts_s[i];
    t = fminf(fps_dist(xj, yj, zj, p.x, p.y, p.z), t);
  }
  ws[(size_t)part * m + j] = t;
}

__global__ __launch_bounds__(256) void fps_prefix_check_kernel(int n, int m, int parts, int log2bs,
                                                               const float *__restrict__ dataset,
                                                               float *__restrict__ temp) {
  __shared__ float4 sel_s[kPrefixMaxM];   // slot j-1: (point j-1, T_j) = what iteration j needs
  const int scene = blockIdx.y, tid = threadIdx.x;
  const float *pts = dataset + (size_t)scene * n * 3;
  float *ws = temp + (size_t)scene * n;
  for (int i = tid; i < m - 1; i += 256) {
    float t = ws[i + 1];
    for (int p = 1; p < parts; ++p) t = fminf(ws[(size_t)p * m + i + 1], t);
    sel_s[i] = make_float4(pts[i * 3 + 0], pts[i * 3 + 1], pts[i * 3 + 2], t);
  }
  __syncthreads();
  const int k = blockIdx.x * 256 + tid;
  const int k0 = k & ~63;                 // first point of this wave
  float x = 0.f, y = 0.f, z = 0.f;
  bool competes = false;
  if (k < n) {
    x = pts[k * 3 + 0]; y = pts[k * 3 + 1]; z = pts[k * 3 + 2];
    competes = !fps::skipped(x, y, z);
  }
  const unsigned mykey = fps::key_of((unsigned)k, log2bs);
  // a skipped point inside the prefix 1..m-1 is never selected
  unsigned long long viol = __ballot(k >= 1 && k < m && !competes);
  float r = 1e10f;
  const unsigned long long cmask = __ballot(competes);
  // one sample: lane (j - k0) owns point j itself (r == T_j there by construction); every other competing lane must
  // stay below, or tie with a larger key (wave-uniform branch, taken on ties only)
  auto one = [&](int j) {
    const float4 s = sel_s[j - 1];
    r = fminf(fps_dist(x, y, z, s.x, s.y, s.z), r);
    const unsigned long long self = (unsigned)(j - k0) < 64u ? (1ull << (j - k0)) : 0ull;
    const unsigned long long ge = __ballot(r >= s.w) & cmask & ~self;
    if (ge != 0ull) viol |= __ballot(r > s.w || mykey < fps::key_of((unsigned)j, log2bs)) & ge;
  };
  int j = 1;
  while (j < m && viol == 0ull) {
    // eight samples per trip with independent distances (the loop is a latency chain otherwise: one wave per SIMD).
    // Not where this wave owns one of the eight samples, and not when a lane reaches a threshold: then one by one.
    const bool own = j + 8 > k0 && j < k0 + 64;
    if (j + 8 <= m && !own) {
      float4 s[8];
      float rr[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] = sel_s[j - 1 + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) rr[u] = fps_dist(x, y, z, s[u].x, s[u].y, s[u].z);
      rr[0] = fminf(rr[0], r);
#pragma unroll
      for (int u = 1; u < 8; ++u) rr[u] = fminf(rr[u], rr[u - 1]);
      bool hit = false;
#pragma unroll
      for (int u = 0; u < 8; ++u) hit |= rr[u] >= s[u].w;
      if ((__ballot(hit) & cmask) == 0ull) {
        r = rr[7];
        j += 8;
        continue;
      }
    }
    const int je = j + 8 < m ? j + 8 : m;
    for (; j < je; ++j) one(j);
  }
  if (viol != 0ull && (tid & 63) == 0) atomicOr(reinterpret_cast<int *>(ws), 1);
}

// Streaming fallback for clouds that fit neither the register-resident kernel nor the pruned path's
// workspace: running min distances live in the caller's `temp` scratch.
constexpr int kFpsThreads = 1024;
__global__ __launch_bounds__(kFpsThreads) void fps_kernel_global(int n, int m, int log2bs,
                                                                 const float *__restrict__ dataset,
                                                                 float *__restrict__ temp,
                                                                 int *__restrict__ idxs) {
  __shared__ fps::Slot slots[2][fps::kMaxWaves];
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float *pts = dataset + (size_t)blockIdx.x * n * 3;
  float *tmp = temp + (size_t)blockIdx.x * n;
  int *out = idxs + (size_t)blockIdx.x * m;
  for (int k = tid; k < n; k += kFpsThreads)
    tmp[k] = fps::skipped(pts[k * 3 + 0], pts[k * 3 + 1], pts[k * 3 + 2]) ? -1.0f : 1e10f;
  const float p0x = pts[0], p0y = pts[1], p0z = pts[2];
  float x1 = p0x, y1 = p0y, z1 = p0z;
  if (tid == 0) out[0] = 0;
  for (int j = 1; j < m; ++j) {
    // 1024 is a multiple of the reference block size, so all of a thread's points share one
    // reference slot and its FIRST strict maximum is the slot's winner.
    float best = -1.0f;
    int besti = 0;
    float bx = 0.f, by = 0.f, bz = 0.f;
    for (int k = tid; k < n; k += kFpsThreads) {
      const float x2 = pts[k * 3 + 0], y2 = pts[k * 3 + 1], z2 = pts[k * 3 + 2];
      const float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
      const float d2 = fminf(d, tmp[k]);
      tmp[k] = d2;
      const bool better = d2 > best;
      besti = better ? k : besti;
      bx = better ? x2 : bx;
      by = better ? y2 : by;
      bz = better ? z2 : bz;
      best = better ? d2 : best;
    }
    fps::Slot *buf = slots[j & 1];
    fps::publish_wave_best(buf, wave, lane, best >= 0.0f, __float_as_uint(best),
                           fps::key_of((unsigned)besti, log2bs), bx, by, bz);
    __syncthreads();
    const int old = fps::select_global_best<kFpsThreads / kWave>(buf, lane, log2bs, p0x, p0y, p0z, x1, y1, z1);
    if (tid == 0) out[j] = old;
  }
}

// ----------------------------------------------------------------------------------------------
// Ball query
// ----------------------------------------------------------------------------------------------
// One wave owns C consecutive centres.  Centre coordinates, hit counters and first-hit indices are
// wave-uniform (SGPRs); the wave walks the cloud 64 points at a time -- lane l tests point k0+l, so
// the v_cmp result IS the ordered hit mask: popcount(mask below my lane) is my output slot and the
// "first nsample in ascending index" rule (ball_query_gpu.cu:32-47) needs no sort.  Each 64-point
// tile is loaded once (coalesced 12 B/lane) and reused for all C centres, which divides the L2
// traffic of the logical M*N*12-byte stream by C.  Rows are finished (padding with the first hit, or
// zeros) by the same wave, so idx needs no memset.
constexpr int kBqThreads = 256;

template <int C>
__global__ __launch_bounds__(kBqThreads) void ball_query_kernel(int n, int m, float radius2,
                                                               int nsample, int waves_per_scene, int total_waves,
                                                               const float *__restrict__ new_xyz,
                                                               const float *__restrict__ xyz,
                                                               int *__restrict__ idx) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * (kBqThreads / kWave) + wave_in_block;  // wave-uniform
  if (gw >= total_waves) return;     // the last workgroup's spare waves: they would be centres of a scene past the batch
  const int scene = gw / waves_per_scene;
  const int j0 = (gw - scene * waves_per_scene) * C;
  if (j0 >= m) return;
  const float *pts = xyz + (size_t)scene * n * 3;
  const float *ctr = new_xyz + (size_t)scene * m * 3;
  int *out = idx + (size_t)scene * m * nsample;

  float cx[C], cy[C], cz[C];
  int cnt[C], first[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int j = (j0 + c) < m ? (j0 + c) : (m - 1);
    cx[c] = ctr[j * 3 + 0];
    cy[c] = ctr[j * 3 + 1];
    cz[c] = ctr[j * 3 + 2];
    cnt[c] = (j0 + c) < m ? 0 : nsample;  // out-of-range centres are born "full"
    first[c] = 0;
  }

  const unsigned long long below = (1ull << lane) - 1ull;
  for (int k0 = 0; k0 < n; k0 += kWave) {
    const int k = k0 + lane;
    const bool in = k < n;
    const int kk = in ? k : (n - 1);
    const float x = pts[kk * 3 + 0];
    const float y = pts[kk * 3 + 1];
    const float z = pts[kk * 3 + 2];
    bool all_full = true;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      if (cnt[c] < nsample) {  // scalar branch
        all_full = false;
        const float d2 = (cx[c] - x) * (cx[c] - x) + (cy[c] - y) * (cy[c] - y) +
                         (cz[c] - z) * (cz[c] - z);
        const bool hit = in && (d2 < radius2);
        const unsigned long long mask = __ballot(hit);
        if (mask != 0ull) {
          if (cnt[c] == 0) first[c] = k0 + __ffsll((long long)mask) - 1;
          const int pos = cnt[c] + __popcll(mask & below);
          if (hit && pos < nsample) out[(size_t)(j0 + c) * nsample + pos] = k;
          cnt[c] += __popcll(mask);
        }
      }
    }
    if (all_full) break;
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int j = j0 + c;
    if (j < m) {
      const int have = cnt[c] < nsample ? cnt[c] : nsample;
      const int fill = cnt[c] > 0 ? first[c] : 0;
      for (int l = have + lane; l < nsample; l += kWave) out[(size_t)j * nsample + l] = fill;
    }
  }
}

// ----------------------------------------------------------------------------------------------
// gather / group / interpolate: bandwidth kernels.  One thread per output position, looping over a
// chunk of channels so the index is read once and every store is coalesced.
// ----------------------------------------------------------------------------------------------
constexpr int kChanChunk = 16;

// points (b,c,n), idx (b,P) -> out (b,c,P); serves gather_points (P = npoints) and group_points
// (P = npoints*nsample): both are out[b,l,p] = points[b,l,idx[b,p]].
__global__ __launch_bounds__(256) void index_select_kernel(int c, int n, int P,
                                                           const float *__restrict__ points,
                                                           const int *__restrict__ idx,
                                                           float *__restrict__ out) {
  const int b = blockIdx.z;
  const int l0 = blockIdx.y * kChanChunk;
  const int l1 = min(l0 + kChanChunk, c);
  const float *src = points + (size_t)b * c * n;
  float *dst = out + (size_t)b * c * P;
  const int *ix = idx + (size_t)b * P;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    const int a = ix[p];
    for (int l = l0; l < l1; ++l) dst[(size_t)l * P + p] = src[(size_t)l * n + a];
  }
}

__global__ __launch_bounds__(256) void index_scatter_add_kernel(int c, int n, int P,
                                                                const float *__restrict__ grad_out,
                                                                const int *__restrict__ idx,
                                                                float *__restrict__ grad_points) {
  const int b = blockIdx.z;
  const int l0 = blockIdx.y * kChanChunk;
  const int l1 = min(l0 + kChanChunk, c);
  const float *src = grad_out + (size_t)b * c * P;
  float *dst = grad_points + (size_t)b * c * n;
  const int *ix = idx + (size_t)b * P;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    const int a = ix[p];
    for (int l = l0; l < l1; ++l) atomicAdd(dst + (size_t)l * n + a, src[(size_t)l * P + p]);
  }
}

// Same scatter-add when the source set is small (SA2..SA4 grouping grads: n <= 4096): a workgroup
// owns kLdsChan channels of one scene, accumulates ALL positions into an LDS image [chan][n] with
// ds_add_f32 (no global atomics, no contention across CUs) and adds the image to the output coalesced.
constexpr int kLdsScatterMaxN = 4096;
constexpr int kLdsScatterFloats = 16384;  // 64 KiB of LDS per workgroup
__global__ __launch_bounds__(1024) void index_scatter_add_lds_kernel(int c, int n, int P, int kLdsChan,
                                                                     const float *__restrict__ grad_out,
                                                                     const int *__restrict__ idx,
                                                                     float *__restrict__ grad_points) {
  extern __shared__ __attribute__((aligned(16))) float acc[];  // [kLdsChan][n]
  const int b = blockIdx.y;
  const int l0 = blockIdx.x * kLdsChan;
  const int nl = min(kLdsChan, c - l0);
  for (int i = threadIdx.x; i < kLdsChan * n; i += blockDim.x) acc[i] = 0.f;
  __syncthreads();
  const float *src = grad_out + ((size_t)b * c + l0) * P;
  const int *ix = idx + (size_t)b * P;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    const int a = ix[p];
    for (int l = 0; l < nl; ++l) atomicAdd(&acc[l * n + a], src[(size_t)l * P + p]);
  }
  __syncthreads();
  float *dst = grad_points + ((size_t)b * c + l0) * n;
  for (int i = threadIdx.x; i < nl * n; i += blockDim.x) dst[i] += acc[i];
}

// three_nn: one thread per unknown point; the (small) known set is staged through LDS in tiles and
// read as wave-uniform broadcasts.  Float compares against a +inf sentinel are equivalent to the
// reference's double best*=1e40 (interpolate_gpu.cu:32): every candidate is an fp32 value.
constexpr int kNnTile = 1024;
__global__ __launch_bounds__(256) void three_nn_kernel(int n, int m, const float *__restrict__ unknown,
                                                       const float *__restrict__ known,
                                                       float *__restrict__ dist2,
                                                       int *__restrict__ idx) {
  __shared__ float kn[kNnTile * 3];
  const int b = blockIdx.y;
  const float *u = unknown + (size_t)b * n * 3;
  const float *kp = known + (size_t)b * m * 3;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = j < n;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (live) {
    ux = u[j * 3 + 0];
    uy = u[j * 3 + 1];
    uz = u[j * 3 + 2];
  }
  float best1 = INFINITY, best2 = INFINITY, best3 = INFINITY;
  int besti1 = 0, besti2 = 0, besti3 = 0;
  for (int k0 = 0; k0 < m; k0 += kNnTile) {
    const int cnt = min(kNnTile, m - k0);
    __syncthreads();
    for (int i = threadIdx.x; i < cnt * 3; i += blockDim.x) kn[i] = kp[(size_t)k0 * 3 + i];
    __syncthreads();
    for (int kk = 0; kk < cnt; ++kk) {
      const float x = kn[kk * 3 + 0], y = kn[kk * 3 + 1], z = kn[kk * 3 + 2];
      const float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
      const int k = k0 + kk;
      if (d < best1) {
        best3 = best2; besti3 = besti2;
        best2 = best1; besti2 = besti1;
        best1 = d; besti1 = k;
      } else if (d < best2) {
        best3 = best2; besti3 = besti2;
        best2 = d; besti2 = k;
      } else if (d < best3) {
        best3 = d; besti3 = k;
      }
    }
  }
  if (live) {
    float *od = dist2 + ((size_t)b * n + j) * 3;
    int *oi = idx + ((size_t)b * n + j) * 3;
    od[0] = best1; od[1] = best2; od[2] = best3;
    oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
  }
}

__global__ __launch_bounds__(256) void three_interpolate_kernel(int c, int m, int n,
                                                                const float *__restrict__ points,
                                                                const int *__restrict__ idx,
                                                                const float *__restrict__ weight,
                                                                float *__restrict__ out) {
  const int b = blockIdx.z;
  const int l0 = blockIdx.y * kChanChunk;
  const int l1 = min(l0 + kChanChunk, c);
  const float *src = points + (size_t)b * c * m;
  float *dst = out + (size_t)b * c * n;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const size_t o = ((size_t)b * n + j) * 3;
    const float w1 = weight[o + 0], w2 = weight[o + 1], w3 = weight[o + 2];
    const int i1 = idx[o + 0], i2 = idx[o + 1], i3 = idx[o + 2];
    for (int l = l0; l < l1; ++l) {
      const float *p = src + (size_t)l * m;
      dst[(size_t)l * n + j] = p[i1] * w1 + p[i2] * w2 + p[i3] * w3;
    }
  }
}

__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(
    int c, int n, int m, const float *__restrict__ grad_out, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ grad_points) {
  const int b = blockIdx.z;
  const int l0 = blockIdx.y * kChanChunk;
  const int l1 = min(l0 + kChanChunk, c);
  const float *src = grad_out + (size_t)b * c * n;
  float *dst = grad_points + (size_t)b * c * m;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const size_t o = ((size_t)b * n + j) * 3;
    const float w1 = weight[o + 0], w2 = weight[o + 1], w3 = weight[o + 2];
    const int i1 = idx[o + 0], i2 = idx[o + 1], i3 = idx[o + 2];
    for (int l = l0; l < l1; ++l) {
      const float g = src[(size_t)l * n + j];
      float *p = dst + (size_t)l * m;
      atomicAdd(p + i1, g * w1);
      atomicAdd(p + i2, g * w2);
      atomicAdd(p + i3, g * w3);
    }
  }
}

// Same with a small known set (the FP modules: m = 256 / 512): a workgroup owns `chan` channels of one scene and
// accumulates into an LDS image [chan][m] (no global atomics; 6.3 M of them cost 105 us per call otherwise).
constexpr int kInterpLdsMaxM = 4096;
constexpr int kInterpLdsChan = 16;
__global__ __launch_bounds__(1024) void three_interpolate_grad_lds_kernel(
    int c, int n, int m, int chan, const float *__restrict__ grad_out, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ grad_points) {
  extern __shared__ __attribute__((aligned(16))) float acc[];  // [chan][m]
  const int b = blockIdx.y;
  const int l0 = blockIdx.x * chan;
  const int nl = min(chan, c - l0);
  for (int i = threadIdx.x; i < chan * m; i += blockDim.x) acc[i] = 0.f;
  __syncthreads();
  const float *src = grad_out + ((size_t)b * c + l0) * n;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const size_t o = ((size_t)b * n + j) * 3;
    const float w1 = weight[o + 0], w2 = weight[o + 1], w3 = weight[o + 2];
    const int i1 = idx[o + 0], i2 = idx[o + 1], i3 = idx[o + 2];
    for (int l = 0; l < nl; ++l) {
      const float g = src[(size_t)l * n + j];
      float *p = acc + l * m;
      atomicAdd(p + i1, g * w1);
      atomicAdd(p + i2, g * w2);
      atomicAdd(p + i3, g * w3);
    }
  }
  __syncthreads();
  float *dst = grad_points + ((size_t)b * c + l0) * m;
  for (int i = threadIdx.x; i < nl * m; i += blockDim.x) dst[i] += acc[i];
}

inline int launch_status() { return (int)hipGetLastError(); }

inline bool fps_prefix_eligible(int n, int m) { return m >= 2 && m <= kPrefixMaxM && m <= n && n <= kPrefixMaxN; }

inline dim3 chan_grid(int P, int c, int b) {
  int gx = (P + 255) / 256;
  if (gx > 4096) gx = 4096;
  if (gx < 1) gx = 1;
  return dim3(gx, (c + kChanChunk - 1) / kChanChunk, b);
}

}  // namespace

extern "C" {

int butd_pointnet2_abi_version(void) { return BUTD_POINTNET2_ABI_VERSION; }

const char *butd_error_string(int err) { return hipGetErrorString((hipError_t)err); }

int butd_opt_n_threads(int work_size) {
  const int pow_2 = (int)(std::log(static_cast<double>(work_size)) / std::log(2.0));
  int t = 1 << pow_2;
  if (t > 512) t = 512;
  if (t < 1) t = 1;
  return t;
}

// butd_fps_workspace_bytes / butd_furthest_point_sampling_ws: see fps_pruned.hip

int butd_fps_prefix_check(int b, int n, int m, const float *dataset, float *temp, butd_stream_t stream) {
  if (b <= 0 || !fps_prefix_eligible(n, m)) return 0;
  if (temp == nullptr) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  int parts = n / m;
  if (parts > 4) parts = 4;
  hipLaunchKernelGGL(fps_prefix_threshold_kernel, dim3((m + 63) / 64, parts, b), dim3(64), 0, s, n, m, parts,
                     dataset, temp);
  hipLaunchKernelGGL(fps_prefix_check_kernel, dim3((n + 255) / 256, b), dim3(256), 0, s, n, m, parts,
                     ilog2_floor((unsigned)butd_opt_n_threads(n)), dataset, temp);
  return launch_status();
}

int butd_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp, int *idxs,
                                 butd_stream_t stream) {
  if (b <= 0 || m <= 0) return 0;
  if (n <= 0) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const int log2bs = ilog2_floor((unsigned)butd_opt_n_threads(n));
  // a cloud that is already in furthest-point order (the backbone's levels 2-4) is recognised by parallel work and
  // the serial kernel below returns at once for it; any other cloud fails the check within its first iterations
  const int *verdict = nullptr;
  if (temp != nullptr && fps_prefix_eligible(n, m) && butd_fps_prefix_check(b, n, m, dataset, temp, stream) == 0)
    verdict = reinterpret_cast<const int *>(temp);
#define FPS_LAUNCH(T, P) \
  hipLaunchKernelGGL((fps_kernel<T, P>), dim3(b), dim3(T), 0, s, n, m, log2bs, dataset, idxs, verdict, n)
  if (n <= 256) FPS_LAUNCH(256, 1);
  else if (n <= 512) FPS_LAUNCH(256, 2);
  else if (n <= 1024) FPS_LAUNCH(256, 4);
  else if (n <= 2048) FPS_LAUNCH(1024, 2);   // measured on MI355X: see DESIGN.md "FPS tuning"
  else if (n <= 4096) FPS_LAUNCH(1024, 4);
  else if (n <= 8192) FPS_LAUNCH(1024, 8);
  else {  // larger clouds: streaming kernel (the pruned path in fps_pruned.hip supersedes it)
    if (temp == nullptr) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(fps_kernel_global, dim3(b), dim3(kFpsThreads), 0, s, n, m, log2bs, dataset,
                       temp, idxs);
  }
#undef FPS_LAUNCH
  return launch_status();
}

int butd_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx,
                       float *out, butd_stream_t stream) {
  if (b <= 0 || c <= 0 || npoints <= 0) return 0;
  hipLaunchKernelGGL(index_select_kernel, chan_grid(npoints, c, b), dim3(256), 0,
                     (hipStream_t)stream, c, n, npoints, points, idx, out);
  return launch_status();
}

int butd_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out, const int *idx,
                            float *grad_points, butd_stream_t stream) {
  if (b <= 0 || c <= 0 || npoints <= 0) return 0;
  hipLaunchKernelGGL(index_scatter_add_kernel, chan_grid(npoints, c, b), dim3(256), 0,
                     (hipStream_t)stream, c, n, npoints, grad_out, idx, grad_points);
  return launch_status();
}

int butd_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                    const float *xyz, int *idx, butd_stream_t stream) {
  if (b <= 0 || m <= 0 || nsample <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (n <= 0) return (int)butd_zero_async(idx, sizeof(int) * (size_t)b * m * nsample, s);
  const float radius2 = radius * radius;  // ball_query_gpu.cu:27 (fp32 product)
  // centres per wave: as many as keep >= ~2 waves per SIMD busy (1024 SIMDs), at most 8
  const long long total = (long long)b * m;
  int C = 8;
  while (C > 1 && total / C < 2048) C >>= 1;
#define BQ_LAUNCH(CC)                                                                          \
  {                                                                                            \
    const int wps = (m + CC - 1) / CC;                                                         \
    const long long waves = (long long)wps * b;                                                \
    const int blocks = (int)((waves + (kBqThreads / kWave) - 1) / (kBqThreads / kWave));       \
    hipLaunchKernelGGL((ball_query_kernel<CC>), dim3(blocks), dim3(kBqThreads), 0, s, n, m,    \
                       radius2, nsample, wps, (int)waves, new_xyz, xyz, idx);                  \
  }
  switch (C) {
    case 8: BQ_LAUNCH(8) break;
    case 4: BQ_LAUNCH(4) break;
    case 2: BQ_LAUNCH(2) break;
    default: BQ_LAUNCH(1) break;
  }
#undef BQ_LAUNCH
  return launch_status();
}

int butd_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                      const int *idx, float *out, butd_stream_t stream) {
  if (b <= 0 || c <= 0 || npoints <= 0 || nsample <= 0) return 0;
  const int P = npoints * nsample;
  hipLaunchKernelGGL(index_select_kernel, chan_grid(P, c, b), dim3(256), 0, (hipStream_t)stream, c,
                     n, P, points, idx, out);
  return launch_status();
}

int butd_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                           const int *idx, float *grad_points, butd_stream_t stream) {
  if (b <= 0 || c <= 0 || npoints <= 0 || nsample <= 0) return 0;
  const int P = npoints * nsample;
  if (n <= kLdsScatterMaxN && P >= 4 * n) {
    const int chan = kLdsScatterFloats / n;  // 8 channels per workgroup at n = 2048
    hipLaunchKernelGGL(index_scatter_add_lds_kernel, dim3((c + chan - 1) / chan, b), dim3(1024),
                       sizeof(float) * chan * n, (hipStream_t)stream, c, n, P, chan, grad_out, idx,
                       grad_points);
    return launch_status();
  }
  hipLaunchKernelGGL(index_scatter_add_kernel, chan_grid(P, c, b), dim3(256), 0,
                     (hipStream_t)stream, c, n, P, grad_out, idx, grad_points);
  return launch_status();
}

int butd_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                  int *idx, butd_stream_t stream) {
  if (b <= 0 || n <= 0) return 0;
  hipLaunchKernelGGL(three_nn_kernel, dim3((n + 255) / 256, b), dim3(256), 0, (hipStream_t)stream, n,
                     m, unknown, known, dist2, idx);
  return launch_status();
}

int butd_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                           const float *weight, float *out, butd_stream_t stream) {
  if (b <= 0 || c <= 0 || n <= 0) return 0;
  hipLaunchKernelGGL(three_interpolate_kernel, chan_grid(n, c, b), dim3(256), 0,
                     (hipStream_t)stream, c, m, n, points, idx, weight, out);
  return launch_status();
}

int butd_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                const float *weight, float *grad_points, butd_stream_t stream) {
  if (b <= 0 || c <= 0 || n <= 0) return 0;
  if (m > 0 && m <= kInterpLdsMaxM && 3L * n >= 2L * m) {
    int chan = 16384 / m;
    if (chan > kInterpLdsChan) chan = kInterpLdsChan;
    hipLaunchKernelGGL(three_interpolate_grad_lds_kernel, dim3((c + chan - 1) / chan, b), dim3(1024),
                       sizeof(float) * chan * m, (hipStream_t)stream, c, n, m, chan, grad_out, idx, weight,
                       grad_points);
    return launch_status();
  }
  hipLaunchKernelGGL(three_interpolate_grad_kernel, chan_grid(n, c, b), dim3(256), 0,
                     (hipStream_t)stream, c, n, m, grad_out, idx, weight, grad_points);
  return launch_status();
}

}  // extern "C"
