p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
s=s.replace('constexpr int kBM = 64, kBN = 64, kBK = 16, kLd = kBK + 4;  // LDS row stride 20 floats = 80 B',
 'constexpr int kBM = 64, kBN = 64, kBK = 64, kLd = kBK + 4;  // LDS row stride 68 floats = 17 x 16 B\nconstexpr int kSub = kBK / 16;  // 16-wide sub-slabs per staged slab')
s=s.replace('struct Frag4 { float v[4]; };','struct Frag4 { float v[4]; };\nstruct FragSlab { Frag4 s[kSub]; };')
# gemm main loop: fetch/commit per sub-slab
old=s[s.index('  const bool ones = P.ones_col != 0;\n  Frag4 fa = fetch_tile('):s.index('  // epilogue: lane holds C[row = fg*4 + r][col = fr] of each 16x16 tile')]
new='''  const bool ones = P.ones_col != 0;
  FragSlab fa, fb;
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < kSub; ++u) {
      fa.s[u] = fetch_tile(P.a, P.a2, P.a2_mode, P.a2_scale, P.lda_m, P.lda_k, m0, P.M, k0 + u * 16, kend, false, tid);
      fb.s[u] = fetch_tile(P.b, nullptr, 0, 0.f, P.ldb_n, P.ldb_k, n0, P.N, k0 + u * 16, kend, ones, tid);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int u = 0; u < kSub; ++u) {
      commit_tile(As[buf], fa.s[u], P.lda_k, u * 16, tid);
      commit_tile(Bs[buf], fb.s[u], P.ldb_k, u * 16, tid);
    }
  };
  fetch(kbeg);
  commit(0);
  __syncthreads();
  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += kBK) {
    const bool more = k0 + kBK < kend;
    if (more) fetch(k0 + kBK);
#pragma unroll
    for (int u = 0; u < kSub; ++u) {
      f32x4 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i] = *reinterpret_cast<const f32x4 *>(&As[cur][wr * 32 + i * 16 + fr][u * 16 + fg * 4]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bf[j] = *reinterpret_cast<const f32x4 *>(&Bs[cur][wc * 32 + j * 16 + fr][u * 16 + fg * 4]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
    if (more) commit(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

'''
s=s.replace(old,new)
s=s.replace('''__device__ inline void commit_tile(float (*tile)[kLd], const Frag4 &f, long ld_k, int tid) {
  if (ld_k == 1) {
    const int r = tid >> 2, kq = (tid & 3) * 4;
    *reinterpret_cast<float4 *>(&tile[r][kq]) = make_float4(f.v[0], f.v[1], f.v[2], f.v[3]);
  } else {  // transpose into the K-contiguous LDS image
    const int k = tid >> 4, r4 = (tid & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[r4 + i][k] = f.v[i];
  }
}''','''__device__ inline void commit_tile(float (*tile)[kLd], const Frag4 &f, long ld_k, int koff, int tid) {
  if (ld_k == 1) {
    const int r = tid >> 2, kq = (tid & 3) * 4;
    *reinterpret_cast<float4 *>(&tile[r][koff + kq]) = make_float4(f.v[0], f.v[1], f.v[2], f.v[3]);
  } else {  // transpose into the K-contiguous LDS image
    const int k = tid >> 4, r4 = (tid & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[r4 + i][koff + k] = f.v[i];
  }
}''')
s=s.replace('''//                        64x64 output tile per 4-wave workgroup, 32x32 per wave (2x2 MFMA tiles), BK=16,''','''//                        64x64 output tile per 4-wave workgroup, 32x32 per wave (2x2 MFMA tiles), BK=64
//                        (64 MFMAs per wave between barriers: enough work to cover the next slab's loads),''')
open(p,'w').write(s)
