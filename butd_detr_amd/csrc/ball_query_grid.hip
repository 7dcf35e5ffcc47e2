// ball_query_grid.hip -- spatially pruned ball query for large clouds (gfx950).
//
// Same contract as butd_ball_query (include/butd_pointnet2.h; replaces query_ball_point_kernel_wrapper,
// src/ball_query.cpp:9-11 / ball_query_gpu.cu:13-59): for every centre the first `nsample` point indices
// in ASCENDING index order with d2 < radius^2 (fp32, strict, d2 summed x,y,z without FMA), padded with
// the first hit, all-zero rows for centres without a hit.  Bit-exact with the streaming kernel.
//
// The streaming kernel evaluates all M*N pairs (SA1 of the north-star workload: 16384 centres x 50000
// points).  Here each scene's points are binned into a uniform grid whose cells are at least
// 1.001*radius wide, so every hit of a centre lies in the 3x3x3 cells around the centre's cell:
//
//   bq_grid_bbox     bounding box per scene (wave min/max + ordered-key atomicMax)
//   bq_grid_count    one thread per point: grid origin / cell size / dims (<= 32^3) from the box, atomic
//                    histogram over cells; the atomic's return value is the point's rank in its cell
//   bq_grid_scan     one workgroup per scene: exclusive scan -> cell_start
//   bq_grid_scatter  one thread per point: (x, y, z, index) to sorted[cell_start[cell] + rank]
//   bq_grid_query    one wave per centre: the 27 cells are 9 x-contiguous runs of `sorted`; every hit
//                    sets bit `index` of an n-bit bitmap in LDS; an ordered popcount scan of the bitmap
//                    then emits the hits in ascending index order (the order inside a cell -- which the
//                    atomics of the scatter leave arbitrary -- never matters), stopping once every hit
//                    has been seen.
//
// Exactness of the pruning.  cell(p) = trunc(clamp((p - o) * inv, 0, g-1)) is monotone in p and applied
// to points and centres alike.  For a hit |px - cx| < r(1 + 1e-6) (fp32 rounding of d2), so the exact
// difference of the two arguments is < 1/1.001 and the computed one (relative error ~3e-7 on values
// <= 32) stays < 1: the cell coordinates differ by at most 1 on every axis; clamping is monotone and
// 1-Lipschitz and keeps that.  NaN coordinates clamp to cell 0 and never hit (d2 < r2 is false), infinite
// extents or a non-positive / non-finite radius collapse the axis (or the grid) to one cell: slower,
// still exact.
#include <hip/hip_runtime.h>

#include "zero_fill.h"
#include <math.h>
#include <stdint.h>

#include "../../include/butd_pointnet2.h"
#include "wave_ops.h"

namespace {

constexpr int kWave = 64;
constexpr int kGridMax = 32;                                  // cells per axis
constexpr int kCellsMax = kGridMax * kGridMax * kGridMax;     // per scene
constexpr int kSetupThreads = 1024;
constexpr int kScanWordsPerIter = kWave * 4;                  // one uint4 per lane
constexpr int kMinPoints = 8192;                              // below: the streaming kernel wins
constexpr int kMaxPoints = 1 << 18;                           // bitmap = n/8 bytes of LDS <= 32 KB

struct SceneGrid {
  float ox, oy, oz;
  float ix, iy, iz;   // 1 / cell size
  int gx, gy, gz;
  int pad[3];
};

__device__ inline int cell_of(float p, float o, float inv, int g) {
  float u = (p - o) * inv;
  u = fminf(fmaxf(u, 0.0f), (float)(g - 1));  // NaN -> 0
  return (int)u;
}

__device__ inline void axis_setup(float lo, float hi, float h, bool h_ok, float &o, float &inv, int &g) {
  const float ext = hi - lo;
  g = 1;
  o = 0.0f;
  inv = 0.0f;
  if (!h_ok || !(ext >= 0.0f) || !isfinite(ext)) return;
  const float q = ext / h;
  g = q >= (float)(kGridMax - 1) ? kGridMax : (int)q + 1;
  const float cell = g == kGridMax ? fmaxf(h, ext / (float)kGridMax) : h;
  o = lo;
  inv = 1.0f / cell;
}

// monotone float -> uint key (0 is below every key, so a zero-filled slot is the identity of max)
__device__ inline unsigned ordered_key(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float ordered_value(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);  // k = 0 -> NaN: "no point seen"
}

// bounding box: bbox[scene] = max keys of (-x, -y, -z, x, y, z); NaN coordinates are ignored
__global__ __launch_bounds__(256) void bq_grid_bbox(int n, const float *__restrict__ xyz,
                                                    unsigned *__restrict__ bbox) {
  const int scene = blockIdx.y;
  const float *pts = xyz + (size_t)scene * n * 3;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = pts[(size_t)k * 3 + a];
      lo[a] = fminf(lo[a], v);
      hi[a] = fmaxf(hi[a], v);
    }
  }
  __shared__ float red[6][4];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float l = waveops::wave_min_f32(lo[a]);
    const float h = waveops::wave_max_f32(hi[a]);
    if ((threadIdx.x & 63) == 0) {
      red[a][threadIdx.x >> 6] = -l;
      red[3 + a][threadIdx.x >> 6] = h;
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {  // one atomic per block and bound: same-address atomics serialise across the chip
    const float *r = red[threadIdx.x];
    atomicMax(bbox + scene * 8 + threadIdx.x, ordered_key(fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3]))));
  }
}

__device__ inline SceneGrid grid_from_bbox(const unsigned *__restrict__ bb, float radius) {
  const float h = fabsf(radius) * 1.001f;
  const bool h_ok = (h > 0.0f) && isfinite(h);
  SceneGrid g;
  axis_setup(-ordered_value(bb[0]), ordered_value(bb[3]), h, h_ok, g.ox, g.ix, g.gx);
  axis_setup(-ordered_value(bb[1]), ordered_value(bb[4]), h, h_ok, g.oy, g.iy, g.gy);
  axis_setup(-ordered_value(bb[2]), ordered_value(bb[5]), h, h_ok, g.oz, g.iz, g.gz);
  g.pad[0] = g.pad[1] = g.pad[2] = 0;
  return g;
}

__device__ inline int point_cell(const SceneGrid &g, float x, float y, float z) {
  const int cx = cell_of(x, g.ox, g.ix, g.gx);
  const int cy = cell_of(y, g.oy, g.iy, g.gy);
  const int cz = cell_of(z, g.oz, g.iz, g.gz);
  return (cz * g.gy + cy) * g.gx + cx;
}

// histogram over cells; the atomic's return value is the point's rank inside its cell, so the scatter
// needs no second round of atomics.  Block (0, scene) also publishes the scene's grid.
__global__ __launch_bounds__(256) void bq_grid_count(int n, float radius, const float *__restrict__ xyz,
                                                     const unsigned *__restrict__ bbox,
                                                     SceneGrid *__restrict__ grids, int *__restrict__ count,
                                                     int2 *__restrict__ cell_rank) {
  const int scene = blockIdx.y;
  const SceneGrid g = grid_from_bbox(bbox + scene * 8, radius);
  if (blockIdx.x == 0 && threadIdx.x == 0) grids[scene] = g;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const float *p = xyz + ((size_t)scene * n + k) * 3;
  const int cell = point_cell(g, p[0], p[1], p[2]);
  const int rank = atomicAdd(count + (size_t)scene * kCellsMax + cell, 1);
  cell_rank[(size_t)scene * n + k] = make_int2(cell, rank);
}

// inclusive prefix sum over the 64 lanes (DPP: Hillis-Steele inside each 16-lane row, then the row
// totals carried across rows)
__device__ inline int wave_inclusive_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
  return v;
}

// exclusive scan of the (<= 32768) cell counters of one scene: 32 consecutive cells per thread
__global__ __launch_bounds__(kSetupThreads) void bq_grid_scan(int n, const SceneGrid *__restrict__ grids,
                                                              const int *__restrict__ count,
                                                              int *__restrict__ cell_start) {
  __shared__ int wave_total[kSetupThreads / kWave];
  const int scene = blockIdx.x;
  const SceneGrid g = grids[scene];
  const int cells = g.gx * g.gy * g.gz;
  constexpr int kPer = kCellsMax / kSetupThreads;  // 32
  const int4 *cnt = reinterpret_cast<const int4 *>(count + (size_t)scene * kCellsMax);
  int *start = cell_start + (size_t)scene * (kCellsMax + 1);
  const int c0 = threadIdx.x * kPer;
  int v[kPer], sum = 0;
#pragma unroll
  for (int i = 0; i < kPer / 4; ++i) {
    int4 q = make_int4(0, 0, 0, 0);
    if (c0 + i * 4 < cells) q = cnt[threadIdx.x * (kPer / 4) + i];  // counters beyond `cells` stay 0
    v[i * 4 + 0] = q.x;
    v[i * 4 + 1] = q.y;
    v[i * 4 + 2] = q.z;
    v[i * 4 + 3] = q.w;
    sum += v[i * 4] + v[i * 4 + 1] + v[i * 4 + 2] + v[i * 4 + 3];
  }
  const int incl = wave_inclusive_sum(sum);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 63) wave_total[wave] = incl;
  __syncthreads();
  int run = incl - sum;
  for (int w = 0; w < wave; ++w) run += wave_total[w];
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    if ((c0 + i) < cells) start[c0 + i] = run;
    run += v[i];
  }
  if (threadIdx.x == 0) start[cells] = n;
}

__global__ __launch_bounds__(256) void bq_grid_scatter(int n, const float *__restrict__ xyz,
                                                       const int *__restrict__ cell_start,
                                                       const int2 *__restrict__ cell_rank,
                                                       float4 *__restrict__ sorted) {
  const int scene = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const float *p = xyz + ((size_t)scene * n + k) * 3;
  const int2 cr = cell_rank[(size_t)scene * n + k];
  const int pos = cell_start[(size_t)scene * (kCellsMax + 1) + cr.x] + cr.y;
  sorted[(size_t)scene * n + pos] = make_float4(p[0], p[1], p[2], __int_as_float(k));
}

// One wave per centre, up to kQueryWaves independent waves per workgroup (no workgroup barrier: a wave only
// ever touches its own `words` (multiple of 256) bitmap words + first-hit slot of the dynamic LDS, DS
// operations of one wave complete in order, and the wavefront-scope fences keep the compiler from
// reordering across the phases).
constexpr int kQueryWaves = 4;
__device__ inline void wave_phase_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

__global__ __launch_bounds__(kWave * kQueryWaves) void bq_grid_query(int n, int m, int total, float radius2,
                                                       int nsample, int words, int xcd_remap,
                                                       const float *__restrict__ new_xyz,
                                                       const SceneGrid *__restrict__ grids,
                                                       const int *__restrict__ cell_start,
                                                       const float4 *__restrict__ sorted,
                                                       int *__restrict__ idx) {
  extern __shared__ unsigned smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned *bits = smem + wave * (words + 4);
  // workgroups are dealt round-robin to the 8 XCDs: give each XCD a contiguous range of centres, so
  // one L2 serves whole scenes
  const int blk = xcd_remap ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3)
                            : (int)blockIdx.x;
  const int j = blk * (int)(blockDim.x >> 6) + wave;
  if (j >= total) return;
  const int scene = j / m;
  for (int w = lane * 4; w < words; w += kScanWordsPerIter)
    *reinterpret_cast<uint4 *>(bits + w) = make_uint4(0u, 0u, 0u, 0u);

  const SceneGrid g = grids[scene];
  const float cx = new_xyz[(size_t)j * 3 + 0];
  const float cy = new_xyz[(size_t)j * 3 + 1];
  const float cz = new_xyz[(size_t)j * 3 + 2];
  const int icx = cell_of(cx, g.ox, g.ix, g.gx);
  const int icy = cell_of(cy, g.oy, g.iy, g.gy);
  const int icz = cell_of(cz, g.oz, g.iz, g.gz);

  // lanes 0..8: the (dy, dz) rows of the 3x3x3 neighbourhood, each one run of up to 3 x-adjacent cells
  int run_start = 0, run_len = 0;
  if (lane < 9) {
    const int yy = icy + (lane % 3) - 1, zz = icz + (lane / 3) - 1;
    if (yy >= 0 && yy < g.gy && zz >= 0 && zz < g.gz) {
      const int xlo = icx > 0 ? icx - 1 : 0;
      const int xhi = icx + 1 < g.gx ? icx + 1 : g.gx - 1;
      const int *cs = cell_start + (size_t)scene * (kCellsMax + 1) + (zz * g.gy + yy) * g.gx;
      run_start = cs[xlo];
      run_len = cs[xhi + 1] - run_start;
    }
  }
  const float4 *pts = sorted + (size_t)scene * n;
  int hits = 0;  // wave-uniform
  // first 64 candidates of all nine runs: nine independent loads in flight, one memory round trip
  int s0[9], len[9];
  float4 p[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    s0[r] = __builtin_amdgcn_readlane(run_start, r);
    len[r] = __builtin_amdgcn_readlane(run_len, r);
    p[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < len[r]) p[r] = pts[s0[r] + lane];
  }
  wave_phase_fence();  // bitmap cleared
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const float d2 = (cx - p[r].x) * (cx - p[r].x) + (cy - p[r].y) * (cy - p[r].y) +
                     (cz - p[r].z) * (cz - p[r].z);
    const bool hit = lane < len[r] && d2 < radius2;
    const int k = __float_as_int(p[r].w);
    if (hit) atomicOr(bits + (k >> 5), 1u << (k & 31));
    hits += __popcll(__ballot(hit));
  }
#pragma unroll 1
  for (int r = 0; r < 9; ++r) {  // runs longer than one wave
    const int s1 = __builtin_amdgcn_readlane(run_start, r);
    const int l1 = __builtin_amdgcn_readlane(run_len, r);
    for (int o0 = kWave; o0 < l1; o0 += kWave) {
      const int o = o0 + lane;
      bool hit = false;
      int k = 0;
      if (o < l1) {
        const float4 q = pts[s1 + o];
        const float d2 = (cx - q.x) * (cx - q.x) + (cy - q.y) * (cy - q.y) + (cz - q.z) * (cz - q.z);
        hit = d2 < radius2;
        k = __float_as_int(q.w);
      }
      if (hit) atomicOr(bits + (k >> 5), 1u << (k & 31));
      hits += __popcll(__ballot(hit));
    }
  }
  int *out = idx + (size_t)j * nsample;
  if (hits == 0) {
    for (int l = lane; l < nsample; l += kWave) out[l] = 0;
    return;
  }
  wave_phase_fence();  // bits set

  // ordered scan: lane l owns words [w0 + 4l, w0 + 4l + 4) of every 256-word slab
  int base = 0;
  for (int w0 = 0; w0 < words && base < hits; w0 += kScanWordsPerIter) {
    const uint4 v = *reinterpret_cast<const uint4 *>(bits + w0 + lane * 4);
    const int c = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    const int incl = wave_inclusive_sum(c);
    if (c) {
      int pos = base + incl - c;
      const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned word = wv[i];
        while (word) {
          const int k = (w0 + lane * 4 + i) * 32 + (__ffs((int)word) - 1);
          word &= word - 1;
          if (pos == 0) bits[words] = (unsigned)k;
          if (pos < nsample) out[pos] = k;
          ++pos;
        }
      }
    }
    base += __builtin_amdgcn_readlane(incl, 63);
  }
  if (hits < nsample) {
    wave_phase_fence();  // first-hit slot written
    const int first = (int)bits[words];
    for (int l = hits + lane; l < nsample; l += kWave) out[l] = first;
  }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WsLayout {
  size_t sorted, cell_rank, start, count, bbox, grids, total;
};

inline WsLayout ws_layout(int b, int n) {
  WsLayout L;
  size_t off = 0;
  L.sorted = off;
  off = align_up(off + sizeof(float4) * (size_t)b * n, 256);
  L.cell_rank = off;
  off = align_up(off + sizeof(int2) * (size_t)b * n, 256);
  L.start = off;
  off = align_up(off + sizeof(int) * (size_t)b * (kCellsMax + 1), 256);
  L.count = off;  // count and bbox are adjacent: one memset clears both
  off += sizeof(int) * (size_t)b * kCellsMax;
  L.bbox = off;
  off = align_up(off + sizeof(unsigned) * 8 * (size_t)b, 256);
  L.grids = off;
  off = align_up(off + sizeof(SceneGrid) * (size_t)b, 256);
  L.total = off;
  return L;
}

// the grid build costs 4 small launches: worth it when the streaming kernel would test >= ~2^26 pairs
inline bool pruned_pays(int b, int n, int m) {
  return n >= kMinPoints && n <= kMaxPoints && (long long)b * m * n >= (1ll << 26);
}

}  // namespace

extern "C" {

size_t butd_ball_query_workspace_bytes(int b, int n, int m) {
  if (b <= 0 || m <= 0 || !pruned_pays(b, n, m)) return 0;
  return ws_layout(b, n).total;
}

int butd_ball_query_ws(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                       const float *xyz, int *idx, void *workspace, size_t workspace_bytes,
                       butd_stream_t stream) {
  if (b <= 0 || m <= 0 || nsample <= 0) return 0;
  if (workspace == nullptr || n < 1 || n > kMaxPoints || workspace_bytes < ws_layout(b, n).total ||
      ((uintptr_t)workspace & 15u) != 0)
    return butd_ball_query(b, n, m, radius, nsample, new_xyz, xyz, idx, stream);
  hipStream_t s = (hipStream_t)stream;
  const WsLayout L = ws_layout(b, n);
  char *ws = (char *)workspace;
  float4 *sorted = (float4 *)(ws + L.sorted);
  int2 *cell_rank = (int2 *)(ws + L.cell_rank);
  int *start = (int *)(ws + L.start);
  int *count = (int *)(ws + L.count);
  unsigned *bbox = (unsigned *)(ws + L.bbox);
  SceneGrid *grids = (SceneGrid *)(ws + L.grids);
  hipError_t e = butd_zero_async(count, L.grids - L.count, s);
  if (e != hipSuccess) return (int)e;
  const dim3 pgrid((n + 255) / 256, b);
  int bbox_blocks = (n + 255) / 256;
  if (bbox_blocks > 16) bbox_blocks = 16;
  hipLaunchKernelGGL(bq_grid_bbox, dim3(bbox_blocks, b), dim3(256), 0, s, n, xyz, bbox);
  hipLaunchKernelGGL(bq_grid_count, pgrid, dim3(256), 0, s, n, radius, xyz, bbox, grids, count, cell_rank);
  hipLaunchKernelGGL(bq_grid_scan, dim3(b), dim3(kSetupThreads), 0, s, n, grids, count, start);
  hipLaunchKernelGGL(bq_grid_scatter, pgrid, dim3(256), 0, s, n, xyz, start, cell_rank, sorted);
  const int words = (int)align_up((size_t)(n + 31) / 32, kScanWordsPerIter);
  const long long total = (long long)b * m;
  if (total > 0x7fffffffll) return (int)hipErrorInvalidValue;
  const float radius2 = radius * radius;  // ball_query_gpu.cu:27 (fp32 product)
  int qwaves = kQueryWaves;  // as many as fit the default 64 KB of dynamic LDS
  while (qwaves > 1 && sizeof(unsigned) * qwaves * (words + 4) > 65536) qwaves >>= 1;
  const unsigned qblocks = (unsigned)((total + qwaves - 1) / qwaves);
  hipLaunchKernelGGL(bq_grid_query, dim3(qblocks), dim3(kWave * qwaves), sizeof(unsigned) * qwaves * (words + 4),
                     s, n, m, (int)total, radius2, nsample, words, (qblocks % 8 == 0) ? 1 : 0, new_xyz, grids,
                     start, sorted, idx);
  return (int)hipGetLastError();
}

}  // extern "C"
