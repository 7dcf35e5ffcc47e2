p='butd_detr_amd/csrc/sa_ops.hip'
s=open(p).read()
old=s[s.index('__global__ __launch_bounds__(kThreads) void sa_dz_last_kernel('):s.index('__global__ __launch_bounds__(kThreads) void sa_mask_stats_kernel(')]
new='''// dZ of the last layer, in place on Z: workgroup = kChunkRows rows x all columns, a thread owns one
// float4 column group and every TPG-th row (same decomposition as the statistics kernels).
__global__ __launch_bounds__(kThreads) void sa_dz_last_kernel(
    int np, int ns, int C, long P, float *__restrict__ Z, const float *__restrict__ d_out_cm,
    const float *__restrict__ zsel, const uint8_t *__restrict__ asel, const float *__restrict__ gamma,
    const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ mean,
    const float *__restrict__ rstd, const double *__restrict__ S1, const double *__restrict__ S2,
    int training) {
  const int c4n = C >> 2, tpg = kThreads / c4n;
  const int cq = threadIdx.x % c4n, sub = threadIdx.x / c4n;
  const long row0 = (long)blockIdx.x * kChunkRows;
  const long rows = min((long)kChunkRows, P - row0);
  const double invP = 1.0 / (double)P;
  float sc[4], sh[4], mu[4], rs[4], ga[4], a1[4], a2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = cq * 4 + e;
    sc[e] = scale[c]; sh[e] = shift[c]; mu[e] = mean[c]; rs[e] = rstd[c]; ga[e] = gamma[c];
    a1[e] = (float)(S1[c] * invP);
    a2[e] = (float)(S2[c] * invP);
  }
  for (long r = sub; r < rows; r += tpg) {
    const long p = row0 + r;
    const long g = p / ns;
    const int k = (int)(p - g * ns);
    const long b = g / np;
    const int j = (int)(g - b * np);
    const long o = p * C + cq * 4;
    const float4 z4 = *reinterpret_cast<const float4 *>(Z + o);
    const float4 zs4 = *reinterpret_cast<const float4 *>(zsel + g * C + cq * 4);
    const uchar4 as4 = *reinterpret_cast<const uchar4 *>(asel + g * C + cq * 4);
    const float z[4] = {z4.x, z4.y, z4.z, z4.w}, zs[4] = {zs4.x, zs4.y, zs4.z, zs4.w};
    const int as[4] = {as4.x, as4.y, as4.z, as4.w};
    float out[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float dy = 0.f;
      if (as[e] == k && sc[e] * zs[e] + sh[e] > 0.f) dy = d_out_cm[(b * C + cq * 4 + e) * np + j];
      out[e] = training ? ga[e] * rs[e] * (dy - a1[e] - (z[e] - mu[e]) * rs[e] * a2[e]) : sc[e] * dy;
    }
    *reinterpret_cast<float4 *>(Z + o) = make_float4(out[0], out[1], out[2], out[3]);
  }
}

'''
s=s.replace(old,new)
old=s[s.index('__global__ __launch_bounds__(kThreads) void sa_dz_mid_kernel('):s.index('__global__ __launch_bounds__(kThreads) void sa_scatter_rows_kernel(')]
new='''__global__ __launch_bounds__(kThreads) void sa_dz_mid_kernel(
    long P, int C, float *__restrict__ g, const float *__restrict__ Z, const float *__restrict__ gamma,
    const float *__restrict__ scale, const float *__restrict__ mean, const float *__restrict__ rstd,
    const double *__restrict__ S1, const double *__restrict__ S2, int training) {
  const int c4n = C >> 2, tpg = kThreads / c4n;
  const int cq = threadIdx.x % c4n, sub = threadIdx.x / c4n;
  const long row0 = (long)blockIdx.x * kChunkRows;
  const long rows = min((long)kChunkRows, P - row0);
  const double invP = 1.0 / (double)P;
  float sc[4], mu[4], rs[4], ga[4], a1[4], a2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = cq * 4 + e;
    sc[e] = scale[c]; mu[e] = mean[c]; rs[e] = rstd[c]; ga[e] = gamma[c];
    a1[e] = (float)(S1[c] * invP);
    a2[e] = (float)(S2[c] * invP);
  }
  for (long r = sub; r < rows; r += tpg) {
    const long o = (row0 + r) * C + cq * 4;
    const float4 z4 = *reinterpret_cast<const float4 *>(Z + o);
    const float4 g4 = *reinterpret_cast<const float4 *>(g + o);
    const float z[4] = {z4.x, z4.y, z4.z, z4.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w};
    float out[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      out[e] = training ? ga[e] * rs[e] * (gv[e] - a1[e] - (z[e] - mu[e]) * rs[e] * a2[e]) : sc[e] * gv[e];
    *reinterpret_cast<float4 *>(g + o) = make_float4(out[0], out[1], out[2], out[3]);
  }
}

'''
s=s.replace(old,new)
s=s.replace('''  const long total = (long)B * np * ns * C;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(sa_dz_last_kernel, dim3(blocks_for(total, 65536)), dim3(kThreads), 0,
                     (hipStream_t)stream, np, ns, C, total, Z, d_out_cm, zsel, asel, gamma, scale, shift,
                     mean, rstd, S1, S2, training);''','''  const long P = (long)B * np * ns;
  if (P <= 0) return 0;
  if (!cols_ok(C)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(sa_dz_last_kernel, dim3((unsigned)((P + kChunkRows - 1) / kChunkRows)),
                     dim3(kThreads), 0, (hipStream_t)stream, np, ns, C, P, Z, d_out_cm, zsel, asel, gamma,
                     scale, shift, mean, rstd, S1, S2, training);''')
s=s.replace('''  const long total = P * C;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(sa_dz_mid_kernel, dim3(blocks_for(total, 65536)), dim3(kThreads), 0,
                     (hipStream_t)stream, total, C, g, Z, gamma, scale, mean, rstd, S1, S2, training);''','''  if (P <= 0) return 0;
  if (!cols_ok(C)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(sa_dz_mid_kernel, dim3((unsigned)((P + kChunkRows - 1) / kChunkRows)),
                     dim3(kThreads), 0, (hipStream_t)stream, P, C, g, Z, gamma, scale, mean, rstd, S1, S2,
                     training);''')
open(p,'w').write(s)
