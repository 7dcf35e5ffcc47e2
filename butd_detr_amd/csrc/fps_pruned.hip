// fps_pruned.hip -- exact furthest-point sampling for large clouds with spatial pruning (gfx950).
//
// The reference kernel (pointnet2/_ext_src/src/sampling_gpu.cu:74-178) touches all n points in each of
// the m-1 dependent iterations (2047 x 50 000 point updates for SA1).  But an iteration only CHANGES
// the running min-distance temp[k] of points closer to the new sample than their current temp[k], and
// only needs the arg-max over points -- so:
//
//   * points are counting-sorted into a 32x32x32 Morton grid over the scene's bounding box and cut
//     into chunks of 64*PPL consecutive sorted points (one wave owns a chunk: PPL points per lane);
//   * every chunk caches its bounding box, max temp, and best (temp, key) candidate in the VGPRs of
//     an owner lane (chunk c -> wave c % 16, lane c / 16);
//   * per iteration a chunk is re-evaluated only if   lb2(sample, bbox) * (1 - 1e-5) < maxtemp(chunk)
//     where lb2 is the squared distance from the new sample to the bbox.  Otherwise every point of the
//     chunk has fp32 distance >= its temp, fminf leaves temp unchanged and the cached candidate stays
//     valid.  The 1e-5 slack covers fp32 rounding of both sides (< 2^-21 relative), so the skip is
//     conservative and the result is EXACTLY the brute-force one;
//   * the global arg-max runs over the <= 1024 cached chunk candidates with the reference's tie order
//     (value, then bit-reversed slot, then index -- see pointnet2_ops.hip / DESIGN.md).
//
// Work per iteration drops from n points to the handful of chunks near the new sample; the serial
// chain per iteration is one v_cmp (the ballot is the active list), an L2 round trip for the touched
// chunks, four DPP wave reductions and ONE workgroup barrier.  Compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/butd_pointnet2.h"
#include "fps_common.h"

namespace {

using namespace waveops;

constexpr int kWave = fps::kWave;
constexpr int kThreads = 1024;            // setup kernels (bounds / scan)
constexpr int kWaves = kThreads / kWave;
constexpr int kMaxChunk = 1024;
constexpr int kGridBits = 5;                 // 32 cells per axis
constexpr int kCells = 1 << (3 * kGridBits); // 32768

// ---- workspace layout (per scene) ----------------------------------------------------------------
struct WsLayout {
  size_t rec, temp, cnt, off, bounds, total;
  int npad;
};
__host__ __device__ inline WsLayout ws_layout(int n, int chunk_pts) {
  WsLayout L;
  L.npad = ((n + chunk_pts - 1) / chunk_pts) * chunk_pts;
  size_t o = 0;
  L.rec = o;    o += sizeof(float4) * (size_t)L.npad;
  L.temp = o;   o += sizeof(float) * (size_t)L.npad;
  L.cnt = o;    o += sizeof(int) * (size_t)kCells;
  L.off = o;    o += sizeof(int) * (size_t)kCells;
  L.bounds = o; o += 64;
  L.total = (o + 255) & ~(size_t)255;
  return L;
}

__device__ inline unsigned spread5(unsigned v) {  // abcde -> a00b00c00d00e
  v &= 31u;
  return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6) | ((v & 16u) << 8);
}
__device__ inline int cell_of(float x, float y, float z, const float *bounds) {
  // bounds = {minx, miny, minz, invx, invy, invz}
  int qx = (int)((x - bounds[0]) * bounds[3]);
  int qy = (int)((y - bounds[1]) * bounds[4]);
  int qz = (int)((z - bounds[2]) * bounds[5]);
  qx = qx < 0 ? 0 : (qx > 31 ? 31 : qx);
  qy = qy < 0 ? 0 : (qy > 31 ? 31 : qy);
  qz = qz < 0 ? 0 : (qz > 31 ? 31 : qz);
  return (int)(spread5((unsigned)qx) | (spread5((unsigned)qy) << 1) | (spread5((unsigned)qz) << 2));
}

// 1. bounding box of the scene (+ zero the histogram)
__global__ __launch_bounds__(kThreads) void fps_bounds_kernel(int n, size_t ws_stride, int chunk_pts,
                                                              const float *__restrict__ dataset,
                                                              char *__restrict__ ws) {
  __shared__ float red[6][kWaves];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *pts = dataset + (size_t)blockIdx.x * n * 3;
  char *w = ws + (size_t)blockIdx.x * ws_stride;
  const WsLayout L = ws_layout(n, chunk_pts);
  int *cnt = (int *)(w + L.cnt);
  for (int i = tid; i < kCells; i += kThreads) cnt[i] = 0;
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int k = tid; k < n; k += kThreads) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float v = pts[(size_t)k * 3 + a];
      v = (v == v) ? v : 0.f;                 // NaN coordinates do not poison the grid
      v = fminf(fmaxf(v, -1.0e30f), 1.0e30f);
      mn[a] = fminf(mn[a], v);
      mx[a] = fmaxf(mx[a], v);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float lo = wave_min_f32(mn[a]), hi = wave_max_f32(mx[a]);
    if (lane == 0) {
      red[a][wave] = lo;
      red[3 + a][wave] = hi;
    }
  }
  __syncthreads();
  if (tid == 0) {
    float *bounds = (float *)(w + L.bounds);
    for (int a = 0; a < 3; ++a) {
      float lo = red[a][0], hi = red[3 + a][0];
      for (int i = 1; i < kWaves; ++i) {
        lo = fminf(lo, red[a][i]);
        hi = fmaxf(hi, red[3 + a][i]);
      }
      const float ext = hi - lo;
      bounds[a] = lo;
      bounds[3 + a] = ext > 0.f ? 32.0f / ext : 0.f;
    }
  }
}

// 2. histogram of cell occupancy
__global__ __launch_bounds__(256) void fps_hist_kernel(int n, size_t ws_stride, int chunk_pts,
                                                       const float *__restrict__ dataset,
                                                       char *__restrict__ ws) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const float *pts = dataset + (size_t)blockIdx.y * n * 3;
  char *w = ws + (size_t)blockIdx.y * ws_stride;
  const WsLayout L = ws_layout(n, chunk_pts);
  const float *bounds = (const float *)(w + L.bounds);
  const int c = cell_of(pts[(size_t)k * 3], pts[(size_t)k * 3 + 1], pts[(size_t)k * 3 + 2], bounds);
  atomicAdd((int *)(w + L.cnt) + c, 1);
}

// 3. exclusive scan of the histogram -> cell offsets; histogram becomes the scatter cursor (zeroed)
__global__ __launch_bounds__(kThreads) void fps_scan_kernel(int n, size_t ws_stride, int chunk_pts,
                                                            char *__restrict__ ws) {
  __shared__ int part[kThreads];
  char *w = ws + (size_t)blockIdx.x * ws_stride;
  const WsLayout L = ws_layout(n, chunk_pts);
  int *cnt = (int *)(w + L.cnt);
  int *off = (int *)(w + L.off);
  constexpr int per = kCells / kThreads;  // 32
  const int tid = threadIdx.x;
  int local[per];
  int sum = 0;
#pragma unroll
  for (int i = 0; i < per; ++i) {
    local[i] = cnt[tid * per + i];
    sum += local[i];
  }
  part[tid] = sum;
  __syncthreads();
  for (int s = 1; s < kThreads; s <<= 1) {  // Hillis-Steele inclusive scan
    const int v = tid >= s ? part[tid - s] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - sum;
#pragma unroll
  for (int i = 0; i < per; ++i) {
    off[tid * per + i] = run;
    run += local[i];
    cnt[tid * per + i] = 0;
  }
}

// 4. scatter points into cell order: rec = (x, y, z, key bits); pad the tail with invalid records
__global__ __launch_bounds__(256) void fps_scatter_kernel(int n, int log2bs, size_t ws_stride,
                                                          int chunk_pts,
                                                          const float *__restrict__ dataset,
                                                          char *__restrict__ ws) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  char *w = ws + (size_t)blockIdx.y * ws_stride;
  const WsLayout L = ws_layout(n, chunk_pts);
  float4 *rec = (float4 *)(w + L.rec);
  float *temp = (float *)(w + L.temp);
  if (k >= L.npad) return;
  if (k >= n) {  // padding slots [n, npad) are never produced by the scatter below
    rec[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    temp[k] = -1.0f;
    return;
  }
  const float *pts = dataset + (size_t)blockIdx.y * n * 3;
  const float x = pts[(size_t)k * 3], y = pts[(size_t)k * 3 + 1], z = pts[(size_t)k * 3 + 2];
  const float *bounds = (const float *)(w + L.bounds);
  const int c = cell_of(x, y, z, bounds);
  const int pos = ((const int *)(w + L.off))[c] + atomicAdd((int *)(w + L.cnt) + c, 1);
  rec[pos] = make_float4(x, y, z, __uint_as_float(fps::key_of((unsigned)k, log2bs)));
  temp[pos] = fps::skipped(x, y, z) ? -1.0f : 1e10f;  // -1: never competes, never updated
}

using fps::Slot;

// 5. the sampling loop: one 4-wave workgroup per scene (one wave per SIMD), ONE barrier per iteration.
//
// The iteration is a latency chain and every resident wave replays its bookkeeping instructions, so
// the loop runs one wave per SIMD.  Chunk c is owned by wave c % 4; inside the wave, owner lane
// (c / 4) % 64 keeps the chunk's bbox, max temp and best candidate in register set (c / 4) / 64
// (up to 4 sets -> 1024 chunks).  Round-robin ownership spreads spatially adjacent chunks -- which
// become active together -- over the four SIMDs.  A wave tests its chunks with one v_cmp per register
// set (the ballot IS the active list), walks the set bits, and only the final arg-max over the four
// wave winners goes through LDS.
#ifndef FPS_LOOP_WAVES
#define FPS_LOOP_WAVES 16
#endif
constexpr int kLoopWaves = FPS_LOOP_WAVES;
constexpr int kLoopThreads = kLoopWaves * kWave;
constexpr int kSets = kMaxChunk / (kLoopWaves * kWave);

template <int PPL>
__global__ __launch_bounds__(kLoopThreads) void fps_pruned_kernel(int n, int m, int log2bs,
                                                                  size_t ws_stride,
                                                                  const float *__restrict__ dataset,
                                                                  char *__restrict__ ws,
                                                                  int *__restrict__ idxs) {
  constexpr int kChunkPts = kWave * PPL;
  __shared__ Slot slots[2][fps::kMaxWaves];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float *pts = dataset + (size_t)blockIdx.x * n * 3;
  char *w = ws + (size_t)blockIdx.x * ws_stride;
  const WsLayout L = ws_layout(n, kChunkPts);
  const float4 *rec = (const float4 *)(w + L.rec);
  float *temp = (float *)(w + L.temp);
  int *out = idxs + (size_t)blockIdx.x * m;
  const int nchunk = L.npad / kChunkPts;   // <= kMaxChunk

  // owner-lane state of chunk ((set*64 + lane)*4 + wave)
  float bminx[kSets], bminy[kSets], bminz[kSets], bmaxx[kSets], bmaxy[kSets], bmaxz[kSets];
  float mt[kSets];               // max temp of the chunk; -1 = nothing in it can compete
  unsigned bhi[kSets], blo[kSets];  // best candidate: value bits / ~key
  float bx[kSets], by[kSets], bz[kSets];
#pragma unroll
  for (int q = 0; q < kSets; ++q) {
    bminx[q] = bminy[q] = bminz[q] = bmaxx[q] = bmaxy[q] = bmaxz[q] = 0.f;
    mt[q] = -1.0f;
    bhi[q] = blo[q] = 0u;
    bx[q] = by[q] = bz[q] = 0.f;
  }

  // prologue: bounding boxes over the points that can compete
#pragma unroll
  for (int q = 0; q < kSets; ++q) {
    for (int l = 0; l < kWave; ++l) {
      const int c = (q * kWave + l) * kLoopWaves + wave;
      if (c >= nchunk) break;  // wave-uniform
      float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
      bool any = false;
#pragma unroll
      for (int p = 0; p < PPL; ++p) {
        const int i = c * kChunkPts + p * kWave + lane;
        const float4 r = rec[i];
        if (temp[i] >= 0.f) {
          any = true;
          lo[0] = fminf(lo[0], r.x); hi[0] = fmaxf(hi[0], r.x);
          lo[1] = fminf(lo[1], r.y); hi[1] = fmaxf(hi[1], r.y);
          lo[2] = fminf(lo[2], r.z); hi[2] = fmaxf(hi[2], r.z);
        }
      }
      const bool chunk_any = __ballot(any) != 0ull;
      const float l0 = wave_min_f32(lo[0]), l1 = wave_min_f32(lo[1]), l2 = wave_min_f32(lo[2]);
      const float h0 = wave_max_f32(hi[0]), h1 = wave_max_f32(hi[1]), h2 = wave_max_f32(hi[2]);
      if (lane == l) {
        bminx[q] = l0; bminy[q] = l1; bminz[q] = l2;
        bmaxx[q] = h0; bmaxy[q] = h1; bmaxz[q] = h2;
        mt[q] = chunk_any ? 1e10f : -1.0f;
      }
    }
  }
  if (tid == 0) out[0] = 0;
  const float p0x = pts[0], p0y = pts[1], p0z = pts[2];
  float x1 = p0x, y1 = p0y, z1 = p0z;

#ifdef FPS_STATS
  unsigned st_total = 0, st_crit = 0;
#endif
  for (int j = 1; j < m; ++j) {
#ifdef FPS_STATS
    int st_mine = 0;
#endif
#pragma unroll
    for (int q = 0; q < kSets; ++q) {
      if (q * kWave * kLoopWaves >= nchunk) break;  // wave-uniform: unused register sets
      // ---- A: which of my chunks can change?  lb2 = squared distance sample -> chunk bbox.
      // mt = -1 for chunks that do not exist / cannot compete, so they never pass the test.
      const float dx = fmaxf(fmaxf(bminx[q] - x1, x1 - bmaxx[q]), 0.f);
      const float dy = fmaxf(fmaxf(bminy[q] - y1, y1 - bmaxy[q]), 0.f);
      const float dz = fmaxf(fmaxf(bminz[q] - z1, z1 - bmaxz[q]), 0.f);
      const float lb2 = (dx * dx + dy * dy + dz * dz) * 0.99999f;
      unsigned long long todo = __ballot(lb2 < mt[q]);
#ifdef FPS_STATS
      st_mine += __popcll(todo);
#endif
      // ---- B: re-evaluate them (whole wave per chunk, PPL points per lane).  The loads of the NEXT
      // active chunk are issued before the current one is reduced.
      float4 r_nxt[PPL];
      float t_nxt[PPL];
      int l_nxt = -1;
      if (todo != 0ull) {
        l_nxt = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const int c = (q * kWave + l_nxt) * kLoopWaves + wave;
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
          r_nxt[p] = rec[c * kChunkPts + p * kWave + lane];
          t_nxt[p] = temp[c * kChunkPts + p * kWave + lane];
        }
      }
      while (l_nxt >= 0) {
        const int l = l_nxt;
        const int c = (q * kWave + l) * kLoopWaves + wave;
        float4 r_cur[PPL];
        float t_cur[PPL];
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
          r_cur[p] = r_nxt[p];
          t_cur[p] = t_nxt[p];
        }
        l_nxt = -1;
        if (todo != 0ull) {
          l_nxt = __ffsll((long long)todo) - 1;
          todo &= todo - 1ull;
          const int cn = (q * kWave + l_nxt) * kLoopWaves + wave;
#pragma unroll
          for (int p = 0; p < PPL; ++p) {
            r_nxt[p] = rec[cn * kChunkPts + p * kWave + lane];
            t_nxt[p] = temp[cn * kChunkPts + p * kWave + lane];
          }
        }
        unsigned vbits = 0u, vkey = 0xFFFFFFFFu;
        float vx = 0.f, vy = 0.f, vz = 0.f;
        bool have = false;
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
          const float4 r = r_cur[p];
          const float t = t_cur[p];
          const float d = (r.x - x1) * (r.x - x1) + (r.y - y1) * (r.y - y1) + (r.z - z1) * (r.z - z1);
          const float d2 = fminf(d, t);  // t = -1 (skipped / padding) stays -1
          if (d2 < t) temp[c * kChunkPts + p * kWave + lane] = d2;
          if (d2 >= 0.f) {
            const unsigned b = __float_as_uint(d2), key = __float_as_uint(r.w);
            if (!have || b > vbits || (b == vbits && key < vkey)) {  // value desc, then key asc
              vbits = b; vkey = key; vx = r.x; vy = r.y; vz = r.z;
            }
            have = true;
          }
        }
        if (__ballot(have) == 0ull) {
          if (lane == l) { mt[q] = -1.0f; bhi[q] = 0u; blo[q] = 0u; }
        } else {
          const unsigned mx = wave_max_u32(have ? vbits : 0u);
          unsigned long long win = __ballot(have && vbits == mx);
          if (__popcll(win) > 1) {  // tie on the value: smallest key wins (wave-uniform branch)
            const unsigned km = wave_min_u32((have && vbits == mx) ? vkey : 0xFFFFFFFFu);
            win = __ballot(have && vbits == mx && vkey == km);
          }
          const int wl = __ffsll((long long)win) - 1;  // exactly one lane: keys are unique
          const unsigned kmin =
              (unsigned)__builtin_amdgcn_readlane((int)vkey, __builtin_amdgcn_readfirstlane(wl));
          const float wx = bcast_f32(vx, wl), wy = bcast_f32(vy, wl), wz = bcast_f32(vz, wl);
          if (lane == l) {
            mt[q] = __uint_as_float(mx);
            bhi[q] = mx;
            blo[q] = 0xFFFFFFFFu - kmin;
            bx[q] = wx; by[q] = wy; bz[q] = wz;
          }
        }
      }
    }
    // ---- C: arg-max over the cached candidates: per lane over its register sets, in-wave by DPP,
    // across the four waves through LDS slots
    unsigned chi = bhi[0], clo = blo[0];
    float cx = bx[0], cy = by[0], cz = bz[0];
#pragma unroll
    for (int q = 1; q < kSets; ++q) {
      const bool better = bhi[q] > chi || (bhi[q] == chi && blo[q] > clo);
      cx = better ? bx[q] : cx; cy = better ? by[q] : cy; cz = better ? bz[q] : cz;
      clo = better ? blo[q] : clo;
      chi = better ? bhi[q] : chi;
    }
    const unsigned whi = wave_max_u32(chi);
    bool cwin = chi == whi;
    if (__popcll(__ballot(cwin)) > 1) {  // several lanes tie on the value (or the wave has nothing)
      const unsigned wlo = wave_max_u32(cwin ? clo : 0u);
      cwin = cwin && clo == wlo;
      if (wlo == 0u) cwin = lane == 0;  // no candidate in this wave at all: publish (0,0) once
    }
    Slot *buf = slots[j & 1];
    if (cwin) {
      Slot s;
      s.hi = chi; s.lo = clo;
      s.x = cx; s.y = cy; s.z = cz;
      buf[wave] = s;
    }
#ifdef FPS_STATS
    if (lane == 0) buf[wave].pad[0] = __int_as_float(st_mine);
#endif
    __syncthreads();
#ifdef FPS_STATS
    {
      unsigned tot = 0, mxw = 0;
      for (int q = 0; q < kLoopWaves; ++q) {
        const unsigned v = (unsigned)__float_as_int(buf[q].pad[0]);
        tot += v; mxw = v > mxw ? v : mxw;
      }
      st_total += tot; st_crit += mxw;
    }
#endif
    const int old = fps::select_global_best<kLoopWaves>(buf, lane, log2bs, p0x, p0y, p0z, x1, y1, z1);
    if (tid == 0) out[j] = old;
  }
#ifdef FPS_STATS
  if (tid == 0)
    printf("scene %d: chunk evals %u (%.1f/step), critical path chunks %u (%.2f/step)\n", blockIdx.x,
           st_total, st_total / (float)(m - 1), st_crit, st_crit / (float)(m - 1));
#endif
}

inline int ilog2_floor(unsigned v) {
  int r = 0;
  while (v >>= 1) ++r;
  return r;
}

constexpr int kPrunedMinN = 8192;
inline int pruned_ppl(int n) {
  if (n < kPrunedMinN) return 0;
  if (n <= kMaxChunk * kWave) return 1;
  if (n <= kMaxChunk * kWave * 4) return 4;
  return 0;
}

}  // namespace

extern "C" {

size_t butd_fps_workspace_bytes(int b, int n) {
  const int ppl = pruned_ppl(n);
  if (ppl == 0 || b <= 0) return 0;
  return ws_layout(n, kWave * ppl).total * (size_t)b;
}

int butd_furthest_point_sampling_ws(int b, int n, int m, const float *dataset, float *temp,
                                    int *idxs, void *workspace, size_t workspace_bytes,
                                    butd_stream_t stream) {
  const int ppl = pruned_ppl(n);
  if (b <= 0 || m <= 0) return 0;
  if (ppl == 0 || workspace == nullptr || workspace_bytes < butd_fps_workspace_bytes(b, n))
    return butd_furthest_point_sampling(b, n, m, dataset, temp, idxs, stream);
  hipStream_t s = (hipStream_t)stream;
  const int chunk_pts = kWave * ppl;
  const WsLayout L = ws_layout(n, chunk_pts);
  const int log2bs = ilog2_floor((unsigned)butd_opt_n_threads(n));
  char *ws = (char *)workspace;
  hipLaunchKernelGGL(fps_bounds_kernel, dim3(b), dim3(kThreads), 0, s, n, L.total, chunk_pts, dataset, ws);
  hipLaunchKernelGGL(fps_hist_kernel, dim3((n + 255) / 256, b), dim3(256), 0, s, n, L.total,
                     chunk_pts, dataset, ws);
  hipLaunchKernelGGL(fps_scan_kernel, dim3(b), dim3(kThreads), 0, s, n, L.total, chunk_pts, ws);
  hipLaunchKernelGGL(fps_scatter_kernel, dim3((L.npad + 255) / 256, b), dim3(256), 0, s, n, log2bs,
                     L.total, chunk_pts, dataset, ws);
  if (ppl == 1)
    hipLaunchKernelGGL((fps_pruned_kernel<1>), dim3(b), dim3(kLoopThreads), 0, s, n, m, log2bs, L.total,
                       dataset, ws, idxs);
  else
    hipLaunchKernelGGL((fps_pruned_kernel<4>), dim3(b), dim3(kLoopThreads), 0, s, n, m, log2bs, L.total,
                       dataset, ws, idxs);
  return (int)hipGetLastError();
}

}  // extern "C"
