def rep(s,a,b,cnt=1):
    assert s.count(a)==cnt, (s.count(a), a)
    return s.replace(a,b)
p='butd_detr_amd/csrc/sa_ops.hip'
s=open(p).read()
s=rep(s,'''                                                            int normalize, float *__restrict__ X,
                                                            long total) {
  const int Cin = 3 + C;
  for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < total; e += (long)gridDim.x * kThreads) {
    const long p = e / Cin;
    const int c = (int)(e - p * Cin);''','''                                                            int normalize, float *__restrict__ X,
                                                            int ldx, long total) {
  const int Cin = 3 + C;
  for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < total; e += (long)gridDim.x * kThreads) {
    const long p = e / ldx;
    const int c = (int)(e - p * ldx);
    if (c >= Cin) {   // padding columns (row stride rounded up for 16-byte rows)
      X[e] = 0.f;
      continue;
    }''')
s=rep(s,'''                                                                   float *__restrict__ d_feats,
                                                                   long total) {
  const int Cin = 3 + C;
  for (long e''','''                                                                   float *__restrict__ d_feats,
                                                                   int ldx, long total) {
  for (long e''')
s=rep(s,'''dX[p * Cin + 3 + c]);''','''dX[p * ldx + 3 + c]);''')
s=rep(s,'''                  float *X, butd_stream_t stream) {
  const long total = (long)B * np * ns * (3 + C);
  if (total <= 0) return 0;
  hipLaunchKernelGGL(sa_group_kernel, dim3(blocks_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                     N, np, ns, C, xyz, new_xyz, feats, feat_stride, idx, radius, normalize, X, total);''','''                  float *X, int ldx, butd_stream_t stream) {
  if (ldx < 3 + C) return (int)hipErrorInvalidValue;
  const long total = (long)B * np * ns * ldx;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(sa_group_kernel, dim3(blocks_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                     N, np, ns, C, xyz, new_xyz, feats, feat_stride, idx, radius, normalize, X, ldx, total);''')
s=rep(s,'''int butd_sa_scatter_rows(int B, int N, int np, int ns, int C, const float *dX, const int *idx,
                         float *d_feats_pm, butd_stream_t stream) {''','''int butd_sa_scatter_rows(int B, int N, int np, int ns, int C, const float *dX, int ldx, const int *idx,
                         float *d_feats_pm, butd_stream_t stream) {
  if (ldx < 3 + C) return (int)hipErrorInvalidValue;''')
s=rep(s,'''(hipStream_t)stream, N, np, ns, C, dX, idx, d_feats_pm, total);''','''(hipStream_t)stream, N, np, ns, C, dX, idx, d_feats_pm, ldx, total);''')
open(p,'w').write(s)

p='include/butd_sa.h'
s=open(p).read()
s=rep(s,''' * C = 0); idx (B,np,ns) int32; X (B*np*ns, 3+C). */
int butd_sa_group(int B, int N, int np, int ns, int C, const float *xyz, const float *new_xyz,
                  const float *feats, long feat_stride, const int *idx, float radius, int normalize,
                  float *X, butd_stream_t stream);''',''' * C = 0); idx (B,np,ns) int32; X (B*np*ns rows of ldx >= 3+C floats; columns 3+C..ldx-1 are zero-filled:
 * ldx = 3+C rounded up to a multiple of 4 keeps every row 16-byte aligned for the GEMM's float4 path). */
int butd_sa_group(int B, int N, int np, int ns, int C, const float *xyz, const float *new_xyz,
                  const float *feats, long feat_stride, const int *idx, float radius, int normalize,
                  float *X, int ldx, butd_stream_t stream);''')
s=rep(s,'''/* d_feats_pm[b, idx[p], c] += dX[p, 3 + c]  (dX (P, 3+C) row-major, d_feats_pm (B,N,C) point-major,
 * caller zero-fills). */
int butd_sa_scatter_rows(int B, int N, int np, int ns, int C, const float *dX, const int *idx,
                         float *d_feats_pm, butd_stream_t stream);''','''/* d_feats_pm[b, idx[p], c] += dX[p, 3 + c]  (dX: P rows of ldx >= 3+C floats, d_feats_pm (B,N,C)
 * point-major, caller zero-fills). */
int butd_sa_scatter_rows(int B, int N, int np, int ns, int C, const float *dX, int ldx, const int *idx,
                         float *d_feats_pm, butd_stream_t stream);''')
open(p,'w').write(s)

p='butd_detr_amd/_hiplib.py'
s=open(p).read()
s=rep(s,'''    "butd_sa_group": (_c_int, [_c_int] * 5 + [_P, _P, _P, _c_long, _P, _c_float, _c_int, _P, _P]),''','''    "butd_sa_group": (_c_int, [_c_int] * 5 + [_P, _P, _P, _c_long, _P, _c_float, _c_int, _P, _c_int, _P]),''')
s=rep(s,'''    "butd_sa_scatter_rows": (_c_int, [_c_int] * 5 + [_P] * 3 + [_P]),''','''    "butd_sa_scatter_rows": (_c_int, [_c_int] * 5 + [_P, _c_int, _P, _P] + [_P]),''')
open(p,'w').write(s)

# ---- GEMM: partial last slab in the fast path (K % 4 == 0)
p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
s=rep(s,'''  return p.a2 == nullptr && p.K > 0 && (p.K % kBK) == 0 &&''','''  return p.a2 == nullptr && p.K > 0 && (p.K & 3) == 0 &&   // a ragged LAST slab is predicated per float4''')
s=rep(s,'''    auto fetch_fast = [&](int slab) {
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        ra[u] = ldg4(pa + (long)(slab * kSub + u) * sa16);
        rb[u] = ldg4(pb + (long)(slab * kSub + u) * sb16);
      }
    };''','''    // k position (inside a staging step) of this thread's float4, per operand; a float4 whose k is
    // beyond the slice (ragged last slab, K % 4 == 0) is zero
    const int krange = kend - kbeg;
    const int a_k = a_kc ? a_fast : a_slow, b_k = b_kc ? b_fast : b_slow;
    bool a_in[kSub], b_in[kSub];
#pragma unroll
    for (int u = 0; u < kSub; ++u) a_in[u] = b_in[u] = true;
    auto fetch_fast = [&](int slab) {
      const bool whole = (slab + 1) * kBK <= krange;   // uniform
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        if (whole) {
          ra[u] = ldg4(pa + (long)(slab * kSub + u) * sa16);
          rb[u] = ldg4(pb + (long)(slab * kSub + u) * sb16);
        } else {
          const int k0 = slab * kBK + u * kSW;
          a_in[u] = k0 + a_k < krange;
          b_in[u] = k0 + b_k < krange;
          ra[u] = a_in[u] ? ldg4(pa + (long)(slab * kSub + u) * sa16) : zero4;
          rb[u] = b_in[u] ? ldg4(pb + (long)(slab * kSub + u) * sb16) : zero4;
        }
      }
    };''')
s=rep(s,'''        float4 va = a_ok ? ra[u] : zero4, vb = b_ok ? rb[u] : zero4;
        if (a_aff && a_ok) {''','''        const bool a_live = a_ok && a_in[u], b_live = b_ok && b_in[u];
        float4 va = a_live ? ra[u] : zero4, vb = b_live ? rb[u] : zero4;
        if (a_aff && a_live) {''')
s=rep(s,'''        if (b_aff && b_ok) {
          vb.x = fmaxf(vb.x * bsc4.x''','''        if (b_aff && b_live) {
          vb.x = fmaxf(vb.x * bsc4.x''')
s=rep(s,'''        if (a_dropout && a_ok)
          va = drop4(''','''        if (a_dropout && a_live)
          va = drop4(''')
s=rep(s,'''        if (b_dropout && b_ok)
          vb = drop4(''','''        if (b_dropout && b_live)
          vb = drop4(''')
s=rep(s,'''    const int nslab = (kend - kbeg) / kBK;''','''    const int nslab = (krange + kBK - 1) / kBK;''')
open(p,'w').write(s)
