p='butd_detr_amd/csrc/sa_ops.hip'
s=open(p).read()
# ---- faster colstats: float4 column groups, 4 row-phases per block, more blocks
old=s[s.index('// ------------------------------------------------------------------------- column stats (+ pooling)'):s.index('__global__ void sa_bn_finalize_kernel(')]
new='''// ------------------------------------------------------------------------- column stats (+ pooling)
// Workgroup = 256 threads over a chunk of kChunkRows rows x all C columns.  A thread owns FOUR adjacent
// columns (one float4 per row) and every TPG-th GROUP of the chunk, TPG = 256 / (C/4) (groups =
// pool_ns rows, or single rows when not pooling), so the pooling needs no cross-thread step and every
// load is a coalesced 16-byte access; sums are reduced across the TPG row-phases in LDS and leave the
// block as ONE double atomic per column.
constexpr int kChunkRows = 256;

__global__ __launch_bounds__(kThreads) void sa_colstats_kernel(
    long P, int C, const float *__restrict__ Z, double *__restrict__ sum, double *__restrict__ sumsq,
    int pool_ns, float *__restrict__ zmax, float *__restrict__ zmin, uint8_t *__restrict__ amax,
    uint8_t *__restrict__ amin) {
  __shared__ float red[2][kThreads][4];
  const int c4n = C >> 2;                 // float4 columns (16, 32 or 64)
  const int tpg = kThreads / c4n;         // row phases (16, 8 or 4)
  const int cq = threadIdx.x % c4n, sub = threadIdx.x / c4n;
  const int gs = pool_ns > 0 ? pool_ns : 1;
  const long row0 = (long)blockIdx.x * kChunkRows;
  const long rows = min((long)kChunkRows, P - row0);
  const long ngroups = rows / gs;
  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
  for (long g = sub; g < ngroups; g += tpg) {
    const float *z = Z + (row0 + g * gs) * C + cq * 4;
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    float mn[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    int ax[4] = {0, 0, 0, 0}, an[4] = {0, 0, 0, 0};
    for (int k = 0; k < gs; ++k) {
      const float4 v4 = *reinterpret_cast<const float4 *>(z + (long)k * C);
      const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[e] += v[e];
        q[e] += v[e] * v[e];
        if (v[e] > mx[e]) { mx[e] = v[e]; ax[e] = k; }
        if (v[e] < mn[e]) { mn[e] = v[e]; an[e] = k; }
      }
    }
    if (pool_ns > 0) {
      const long o = ((row0 / gs) + g) * C + cq * 4;
      *reinterpret_cast<float4 *>(zmax + o) = make_float4(mx[0], mx[1], mx[2], mx[3]);
      *reinterpret_cast<float4 *>(zmin + o) = make_float4(mn[0], mn[1], mn[2], mn[3]);
      *reinterpret_cast<uchar4 *>(amax + o) = make_uchar4(ax[0], ax[1], ax[2], ax[3]);
      *reinterpret_cast<uchar4 *>(amin + o) = make_uchar4(an[0], an[1], an[2], an[3]);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[0][threadIdx.x][e] = s[e];
    red[1][threadIdx.x][e] = q[e];
  }
  __syncthreads();
  if (threadIdx.x < C) {
    const int cq2 = threadIdx.x >> 2, e = threadIdx.x & 3;
    double a = 0.0, b = 0.0;
    for (int t = 0; t < tpg; ++t) {
      a += (double)red[0][cq2 + t * c4n][e];
      b += (double)red[1][cq2 + t * c4n][e];
    }
    atomicAdd(sum + threadIdx.x, a);
    atomicAdd(sumsq + threadIdx.x, b);
  }
}

'''
s=s.replace(old,new)
# ---- mask_stats: float4
old=s[s.index('__global__ __launch_bounds__(kThreads) void sa_mask_stats_kernel('):s.index('__global__ __launch_bounds__(kThreads) void sa_dz_mid_kernel(')]
new='''__global__ __launch_bounds__(kThreads) void sa_mask_stats_kernel(
    long P, int C, float *__restrict__ dH, const float *__restrict__ Z, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ mean, const float *__restrict__ rstd,
    double *__restrict__ S1, double *__restrict__ S2) {
  __shared__ float red[2][kThreads][4];
  const int c4n = C >> 2, tpg = kThreads / c4n;
  const int cq = threadIdx.x % c4n, sub = threadIdx.x / c4n;
  const long row0 = (long)blockIdx.x * kChunkRows;
  const long rows = min((long)kChunkRows, P - row0);
  const float4 sc4 = *reinterpret_cast<const float4 *>(scale + cq * 4);
  const float4 sh4 = *reinterpret_cast<const float4 *>(shift + cq * 4);
  const float4 mu4 = *reinterpret_cast<const float4 *>(mean + cq * 4);
  const float4 rs4 = *reinterpret_cast<const float4 *>(rstd + cq * 4);
  const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
  const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, rs[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  for (long r = sub; r < rows; r += tpg) {
    const long o = (row0 + r) * C + cq * 4;
    const float4 z4 = *reinterpret_cast<const float4 *>(Z + o);
    float4 g4 = *reinterpret_cast<const float4 *>(dH + o);
    const float z[4] = {z4.x, z4.y, z4.z, z4.w};
    float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (!(sc[e] * z[e] + sh[e] > 0.f)) g[e] = 0.f;
      s1[e] += g[e];
      s2[e] += g[e] * (z[e] - mu[e]) * rs[e];
    }
    *reinterpret_cast<float4 *>(dH + o) = make_float4(g[0], g[1], g[2], g[3]);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[0][threadIdx.x][e] = s1[e];
    red[1][threadIdx.x][e] = s2[e];
  }
  __syncthreads();
  if (threadIdx.x < C) {
    const int cq2 = threadIdx.x >> 2, e = threadIdx.x & 3;
    double a = 0.0, b = 0.0;
    for (int t = 0; t < tpg; ++t) {
      a += (double)red[0][cq2 + t * c4n][e];
      b += (double)red[1][cq2 + t * c4n][e];
    }
    atomicAdd(S1 + threadIdx.x, a);
    atomicAdd(S2 + threadIdx.x, b);
  }
}

'''
s=s.replace(old,new)
# pool_bwd_stats: 32 groups per block instead of 256 -> 8x more blocks
s=s.replace('''  const long g0 = (long)blockIdx.x * 256;
  const long ng = min((long)256, G - g0);''','''  const long g0 = (long)blockIdx.x * 32;
  const long ng = min((long)32, G - g0);''')
s=s.replace('''  // block = 256 groups x all channels: thread t owns column (t % C), groups sub, sub+tpc, ...''','''  // block = 32 groups x all channels: thread t owns column (t % C), groups sub, sub+tpc, ...''')
s=s.replace('''  hipLaunchKernelGGL(sa_pool_bwd_stats_kernel, dim3((unsigned)((G + 255) / 256)), dim3(kThreads), 0,''','''  hipLaunchKernelGGL(sa_pool_bwd_stats_kernel, dim3((unsigned)((G + 31) / 32)), dim3(kThreads), 0,''')
s=s.replace('inline bool cols_ok(int C) { return C > 0 && C <= kThreads && kThreads % C == 0; }','inline bool cols_ok(int C) { return C >= 16 && C <= kThreads && kThreads % C == 0; }')
open(p,'w').write(s)
