import re
def rep(s,a,b,cnt=1):
    assert s.count(a)==cnt, (s.count(a), a[:70])
    return s.replace(a,b)
p='butd_detr_amd/csrc/sa_ops.hip'
s=open(p).read()
# chunk rows as a kernel parameter
s=rep(s,'''constexpr int kChunkRows = 256;''','''constexpr int kChunkRows = 256;      // tall inputs (>= 2^19 rows): fewest atomics
constexpr int kChunkRowsSmall = 64;  // otherwise: four times the workgroups (a 65 536-row level is 256
                                     // workgroups of 256 rows, and every thread then walks 16-64 rows of
                                     // dependent load -> store: 100 us where the data takes 25)
inline int chunk_rows(long P) { return P >= (1L << 19) ? kChunkRows : kChunkRowsSmall; }''')
# kernels: add int chunk param; replace kChunkRows uses inside kernels
s=rep(s,'''    int pool_ns, float *__restrict__ zmax, float *__restrict__ zmin, uint8_t *__restrict__ amax,
    uint8_t *__restrict__ amin) {''','''    int pool_ns, float *__restrict__ zmax, float *__restrict__ zmin, uint8_t *__restrict__ amax,
    uint8_t *__restrict__ amin, int chunk) {''')
s=s.replace('''  const long row0 = (long)blockIdx.x * kChunkRows;
  const long rows = min((long)kChunkRows, P - row0);''','''  const long row0 = (long)blockIdx.x * chunk;
  const long rows = min((long)chunk, P - row0);''')
assert s.count('blockIdx.x * chunk')==4
s=rep(s,'''    const float *__restrict__ rstd, const double *__restrict__ S1, const double *__restrict__ S2,
    int training) {
  const int c4n = C >> 2, tpg = kThreads / c4n;''','''    const float *__restrict__ rstd, const double *__restrict__ S1, const double *__restrict__ S2,
    int training, int chunk) {
  const int c4n = C >> 2, tpg = kThreads / c4n;''')
s=rep(s,'''    const float *__restrict__ shift, const float *__restrict__ mean, const float *__restrict__ rstd,
    double *__restrict__ S1, double *__restrict__ S2) {
  __shared__ float red[2][kThreads][4];
  const int c4n = C >> 2, tpg = kThreads / c4n;''','''    const float *__restrict__ shift, const float *__restrict__ mean, const float *__restrict__ rstd,
    double *__restrict__ S1, double *__restrict__ S2, int chunk) {
  __shared__ float red[2][kThreads][4];
  const int c4n = C >> 2, tpg = kThreads / c4n;''')
s=rep(s,'''    const double *__restrict__ S1, const double *__restrict__ S2, int training) {
  const int c4n = C >> 2, tpg = kThreads / c4n;''','''    const double *__restrict__ S1, const double *__restrict__ S2, int training, int chunk) {
  const int c4n = C >> 2, tpg = kThreads / c4n;''')
# dz_last / pool_bwd_stats read the gradient position-major
s=rep(s,'''    int np, int ns, int C, long P, float *__restrict__ Z, const float *__restrict__ d_out_cm,''','''    int np, int ns, int C, long P, float *__restrict__ Z, const float *__restrict__ d_out_pm,''')
s=rep(s,'''    const long b = g / np;
    const int j = (int)(g - b * np);
    const long o = p * C + cq * 4;
    const float4 z4 = *reinterpret_cast<const float4 *>(Z + o);''','''    const long o = p * C + cq * 4;
    const float4 z4 = *reinterpret_cast<const float4 *>(Z + o);
    const float4 dy4 = *reinterpret_cast<const float4 *>(d_out_pm + g * C + cq * 4);
    const float dyv[4] = {dy4.x, dy4.y, dy4.z, dy4.w};''')
s=rep(s,'''      if (as[e] == k && sc[e] * zs[e] + sh[e] > 0.f) dy = d_out_cm[(b * C + cq * 4 + e) * np + j];''','''      if (as[e] == k && sc[e] * zs[e] + sh[e] > 0.f) dy = dyv[e];''')
s=rep(s,'''    int np, int C, long G, const float *__restrict__ d_out_cm, const float *__restrict__ zsel,''','''    int np, int C, long G, const float *__restrict__ d_out_pm, const float *__restrict__ zsel,''')
s=rep(s,'''  // block = 32 groups x all channels: thread t owns column (t % C), groups sub, sub+tpc, ...''','''  // block = kPoolGroups groups x all channels: thread t owns column (t % C), groups sub, sub+tpc, ...''')
s=rep(s,'''  const long g0 = (long)blockIdx.x * 32;
  const long ng = min((long)32, G - g0);''','''  const long g0 = (long)blockIdx.x * kPoolGroups;
  const long ng = min((long)kPoolGroups, G - g0);''')
s=rep(s,'''      const long g = g0 + gi;
      const long b = g / np;
      const int j = (int)(g - b * np);
      const float z = zsel[g * C + col];
      if (sc * z + sh > 0.f) {
        const float dy = d_out_cm[(b * C + col) * np + j];''','''      const long g = g0 + gi;
      const float z = zsel[g * C + col];
      if (sc * z + sh > 0.f) {
        const float dy = d_out_pm[g * C + col];''')
s=rep(s,'''__global__ __launch_bounds__(kThreads) void sa_pool_bwd_stats_kernel(''','''constexpr int kPoolGroups = 8;
__global__ __launch_bounds__(kThreads) void sa_pool_bwd_stats_kernel(''')
# host wrappers
s=rep(s,'''  if (!cols_ok(C) || (pool_ns > 0 && (kChunkRows % pool_ns || P % pool_ns)))
    return (int)hipErrorInvalidValue;
  const unsigned blocks = (unsigned)((P + kChunkRows - 1) / kChunkRows);
  hipLaunchKernelGGL(sa_colstats_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, P, C, Z,
                     sum, sumsq, pool_ns, zmax, zmin, amax, amin);''','''  const int chunk = chunk_rows(P);
  if (!cols_ok(C) || (pool_ns > 0 && (chunk % pool_ns || P % pool_ns)))
    return (int)hipErrorInvalidValue;
  const unsigned blocks = (unsigned)((P + chunk - 1) / chunk);
  hipLaunchKernelGGL(sa_colstats_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, P, C, Z,
                     sum, sumsq, pool_ns, zmax, zmin, amax, amin, chunk);''')
s=rep(s,'''int butd_sa_pool_bwd_stats(int B, int np, int C, const float *d_out_cm, const float *zsel,''','''int butd_sa_pool_bwd_stats(int B, int np, int C, const float *d_out_pm, const float *zsel,''')
s=rep(s,'''  hipLaunchKernelGGL(sa_pool_bwd_stats_kernel, dim3((unsigned)((G + 31) / 32)), dim3(kThreads), 0,
                     (hipStream_t)stream, np, C, G, d_out_cm, zsel, scale, shift, mean, rstd, S1, S2);''','''  hipLaunchKernelGGL(sa_pool_bwd_stats_kernel, dim3((unsigned)((G + kPoolGroups - 1) / kPoolGroups)),
                     dim3(kThreads), 0, (hipStream_t)stream, np, C, G, d_out_pm, zsel, scale, shift, mean,
                     rstd, S1, S2);''')
s=rep(s,'''int butd_sa_dz_last(int B, int np, int ns, int C, float *Z, const float *d_out_cm, const float *zsel,''','''int butd_sa_dz_last(int B, int np, int ns, int C, float *Z, const float *d_out_pm, const float *zsel,''')
s=rep(s,'''  hipLaunchKernelGGL(sa_dz_last_kernel, dim3((unsigned)((P + kChunkRows - 1) / kChunkRows)),
                     dim3(kThreads), 0, (hipStream_t)stream, np, ns, C, P, Z, d_out_cm, zsel, asel, gamma,
                     scale, shift, mean, rstd, S1, S2, training);''','''  const int chunk = chunk_rows(P);
  hipLaunchKernelGGL(sa_dz_last_kernel, dim3((unsigned)((P + chunk - 1) / chunk)),
                     dim3(kThreads), 0, (hipStream_t)stream, np, ns, C, P, Z, d_out_pm, zsel, asel, gamma,
                     scale, shift, mean, rstd, S1, S2, training, chunk);''')
s=rep(s,'''  const unsigned blocks = (unsigned)((P + kChunkRows - 1) / kChunkRows);
  hipLaunchKernelGGL(sa_mask_stats_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, P, C,
                     dH, Z, scale, shift, mean, rstd, S1, S2);''','''  const int chunk = chunk_rows(P);
  const unsigned blocks = (unsigned)((P + chunk - 1) / chunk);
  hipLaunchKernelGGL(sa_mask_stats_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, P, C,
                     dH, Z, scale, shift, mean, rstd, S1, S2, chunk);''')
s=rep(s,'''  hipLaunchKernelGGL(sa_dz_mid_kernel, dim3((unsigned)((P + kChunkRows - 1) / kChunkRows)),
                     dim3(kThreads), 0, (hipStream_t)stream, P, C, g, Z, gamma, scale, mean, rstd, S1, S2,
                     training);''','''  const int chunk = chunk_rows(P);
  hipLaunchKernelGGL(sa_dz_mid_kernel, dim3((unsigned)((P + chunk - 1) / chunk)),
                     dim3(kThreads), 0, (hipStream_t)stream, P, C, g, Z, gamma, scale, mean, rstd, S1, S2,
                     training, chunk);''')
open(p,'w').write(s)

p='include/butd_sa.h'
s=open(p).read()
s=s.replace("const float *d_out_cm","const float *d_out_pm")
s=rep(s,''' * (dy = d_out_cm[b,c,j]); S1 = dbeta, S2 = dgamma.  Caller zero-fills S1/S2 (double). */''',''' * (dy = d_out_pm[b,j,c], the gradient of the pooled output POSITION-major (B,np,C): coalesced for the
 * channel-per-thread kernels); S1 = dbeta, S2 = dgamma.  Caller zero-fills S1/S2 (double). */''')
s=s.replace("dy3[p,c] = d_out_cm[b,c,j] if","dy3[p,c] = d_out_pm[b,j,c] if")
open(p,'w').write(s)

p='butd_detr_amd/fused_sa.py'
s=open(p).read()
s=rep(s,'''        if d_cm is None:
            d_out = d_pm.transpose(1, 2).contiguous()
        elif d_pm is None:
            d_out = d_cm.contiguous()
        else:
            d_out = d_cm + d_pm.transpose(1, 2)''','''        # gradient of the pooled output, position-major (B, np, C) like everything else here
        if d_cm is None:
            d_out = d_pm.contiguous()
        elif d_pm is None:
            d_out = d_cm.transpose(1, 2).contiguous()
        else:
            d_out = d_pm + d_cm.transpose(1, 2)''')
open(p,'w').write(s)
