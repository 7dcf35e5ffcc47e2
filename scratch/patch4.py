p='include/butd_attention.h'
s=open(p).read()
s=s.replace(''' * A(m,k) = a[m*lda_m + k*lda_k] (+ a2[...] with the same strides when a2 != NULL),''',''' * A(m,k) = a[m*lda_m + k*lda_k], combined with a2[...] (same strides) when a2 != NULL:
 *          a2_mode 0: a + a2 (e.g. src + pos);  a2_mode 1: a * (a2 > 0 ? a2_scale : 0)  (ReLU/dropout gate),''')
s=s.replace('''  float scale;
  int relu, accumulate, ones_col, split_k;''','''  float scale;
  int a2_mode;
  float a2_scale;
  int relu, accumulate, ones_col, split_k;''')
open(p,'w').write(s)
p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
s=s.replace('''__device__ inline void stage_tile(float (*tile)[kLd], const float *__restrict__ src,
                                  const float *__restrict__ src2, long ld_row, long ld_k, int row0,
                                  int nrows, int k0, int kend, bool ones, int tid) {''','''__device__ inline float combine(float a, float a2, int mode, float gate_scale) {
  return mode == 0 ? a + a2 : a * (a2 > 0.f ? gate_scale : 0.f);
}

__device__ inline void stage_tile(float (*tile)[kLd], const float *__restrict__ src,
                                  const float *__restrict__ src2, int mode2, float scale2,
                                  long ld_row, long ld_k, int row0, int nrows, int k0, int kend,
                                  bool ones, int tid) {''')
s=s.replace('''          const float4 q2 = *reinterpret_cast<const float4 *>(src2 + (long)gr * ld_row + gk);
          v[0] += q2.x; v[1] += q2.y; v[2] += q2.z; v[3] += q2.w;''','''          const float4 q2 = *reinterpret_cast<const float4 *>(src2 + (long)gr * ld_row + gk);
          v[0] = combine(v[0], q2.x, mode2, scale2); v[1] = combine(v[1], q2.y, mode2, scale2);
          v[2] = combine(v[2], q2.z, mode2, scale2); v[3] = combine(v[3], q2.w, mode2, scale2);''')
s=s.replace('''          if (gk + i < kend) v[i] = p[i] + (src2 ? src2[(long)gr * ld_row + gk + i] : 0.f);''','''          if (gk + i < kend)
            v[i] = src2 ? combine(p[i], src2[(long)gr * ld_row + gk + i], mode2, scale2) : p[i];''')
s=s.replace('''          const float4 q2 = *reinterpret_cast<const float4 *>(src2 + (long)gk * ld_k + gr);
          v[0] += q2.x; v[1] += q2.y; v[2] += q2.z; v[3] += q2.w;''','''          const float4 q2 = *reinterpret_cast<const float4 *>(src2 + (long)gk * ld_k + gr);
          v[0] = combine(v[0], q2.x, mode2, scale2); v[1] = combine(v[1], q2.y, mode2, scale2);
          v[2] = combine(v[2], q2.z, mode2, scale2); v[3] = combine(v[3], q2.w, mode2, scale2);''')
s=s.replace('''          if (gr + i < nrows) v[i] = p[i] + (src2 ? src2[(long)gk * ld_k + gr + i] : 0.f);''','''          if (gr + i < nrows)
            v[i] = src2 ? combine(p[i], src2[(long)gk * ld_k + gr + i], mode2, scale2) : p[i];''')
s=s.replace('''    stage_tile(As, P.a, P.a2, P.lda_m, P.lda_k, m0, P.M, k0, kend, false, tid);
    stage_tile(Bs, P.b, nullptr, P.ldb_n, P.ldb_k, n0, P.N, k0, kend, P.ones_col != 0, tid);''','''    stage_tile(As, P.a, P.a2, P.a2_mode, P.a2_scale, P.lda_m, P.lda_k, m0, P.M, k0, kend, false, tid);
    stage_tile(Bs, P.b, nullptr, 0, 0.f, P.ldb_n, P.ldb_k, n0, P.N, k0, kend, P.ones_col != 0, tid);''')
s=s.replace('''          atomicAdd(P.bias_grad + m, v);''','''          atomicAdd(P.bias_grad + m, v * P.scale);''')
open(p,'w').write(s)
