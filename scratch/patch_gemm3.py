p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    assert s.count(a)>=1, a
    if cnt==1: assert s.count(a)==1, (s.count(a), a)
    s=s.replace(a,b)
rep('''constexpr int kSub = kBK / 16;  // 16-wide sub-slabs per staged slab
constexpr int kGemmThreads = 256;
''','''// The kernel is instantiated for 256 threads (4 waves, wave tile 32x32) and 512 threads (8 waves, wave
// tile 32x16: half the MFMA chain per wave and twice the waves per SIMD for small grids).  One staging
// step moves ONE float4 per thread and operand: a (64 rows x SW k) sub-slab, SW = THREADS/16.
''')
# tile_idx
rep('''__device__ inline TileIdx tile_idx(long ld_k, int tid) {
  TileIdx t;
  t.kc = ld_k == 1;
  t.slow = t.kc ? (tid >> 2) : (tid >> 4);
  t.fast = t.kc ? (tid & 3) * 4 : (tid & 15) * 4;
  return t;
}''','''template <int SW>
__device__ inline TileIdx tile_idx(long ld_k, int tid) {
  TileIdx t;
  t.kc = ld_k == 1;
  t.slow = t.kc ? (tid / (SW / 4)) : (tid >> 4);
  t.fast = t.kc ? (tid % (SW / 4)) * 4 : (tid & 15) * 4;
  return t;
}''')
rep('''template <bool WITH_A2 = true>
__device__ inline Frag4 fetch_tile(''','''template <int SW, bool WITH_A2 = true>
__device__ inline Frag4 fetch_tile(''')
rep('''  const TileIdx t = tile_idx(ld_k, tid);
  const long ld_slow''','''  const TileIdx t = tile_idx<SW>(ld_k, tid);
  const long ld_slow''')
rep('''__device__ inline void commit_tile(float (*tile)[kLd], const Frag4 &f, const OperandFx &fx, long ld_k,
                                   int row0, int nrows, int k0, int kend, bool ones, int koff, int tid) {
  const TileIdx t = tile_idx(ld_k, tid);''','''template <int SW>
__device__ inline void commit_tile(float (*tile)[kLd], const Frag4 &f, const OperandFx &fx, long ld_k,
                                   int row0, int nrows, int k0, int kend, bool ones, int koff, int tid) {
  const TileIdx t = tile_idx<SW>(ld_k, tid);''')
rep('''__global__ __launch_bounds__(kGemmThreads) void gemm_kernel(GemmBatch batch,
                                                            const uint64_t *__restrict__ rng_counter) {''','''template <int THREADS>
__global__ __launch_bounds__(THREADS) void gemm_kernel(GemmBatch batch,
                                                       const uint64_t *__restrict__ rng_counter) {
  constexpr int kSW = THREADS / 16;        // k-width of one staging step
  constexpr int kSub = kBK / kSW;          // staging steps per slab
  constexpr int kWavesN = THREADS / 128;   // wave grid 2 x kWavesN
  constexpr int kNJ = 4 / kWavesN;         // 16-column fragments per wave
  constexpr int kRowPhases = THREADS / 16; // rows written per epilogue pass''')
rep('''  const int wr = wave >> 1, wc = wave & 1;''','''  const int wr = wave / kWavesN, wc = wave % kWavesN;''')
rep('''  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};''','''  f32x4 acc[2][kNJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < kNJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};''')
old=s[s.index('  auto mfma_slab = [&](int buf) {'):s.index('  // Fast path (interior tiles')]
new='''  auto mfma_slab = [&](int buf) {
#pragma unroll
    for (int u = 0; u < kBK / 16; ++u) {
      f32x4 af[2], bf[kNJ];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i] = *reinterpret_cast<const f32x4 *>(&As[buf][wr * 32 + i * 16 + fr][u * 16 + fg * 4]);
#pragma unroll
      for (int j = 0; j < kNJ; ++j)
        bf[j] = *reinterpret_cast<const f32x4 *>(&Bs[buf][wc * (16 * kNJ) + j * 16 + fr][u * 16 + fg * 4]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < kNJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
  };
'''
s=s.replace(old,new)
rep('''    const int a_slow = a_kc ? (tid >> 2) : (tid >> 4), a_fast = a_kc ? (tid & 3) * 4 : (tid & 15) * 4;
    const int b_slow = b_kc ? (tid >> 2) : (tid >> 4), b_fast = b_kc ? (tid & 3) * 4 : (tid & 15) * 4;''','''    const int a_slow = a_kc ? (tid / (kSW / 4)) : (tid >> 4);
    const int a_fast = a_kc ? (tid % (kSW / 4)) * 4 : (tid & 15) * 4;
    const int b_slow = b_kc ? (tid / (kSW / 4)) : (tid >> 4);
    const int b_fast = b_kc ? (tid % (kSW / 4)) * 4 : (tid & 15) * 4;''')
rep('''    const long sa16 = !a_ok ? 0 : (a_kc ? 16 : 16 * P.lda_k);   // per 16 k
    const long sb16 = !b_ok ? 0 : (b_kc ? 16 : 16 * P.ldb_k);''','''    const long sa16 = !a_ok ? 0 : (a_kc ? kSW : kSW * P.lda_k);   // per staging step (kSW k)
    const long sb16 = !b_ok ? 0 : (b_kc ? kSW : kSW * P.ldb_k);''')
rep('''      for (int k = tid; k < kend - kbeg; k += kGemmThreads) {''','''      for (int k = tid; k < kend - kbeg; k += THREADS) {''')
rep('''            sc = *reinterpret_cast<const float4 *>(&Asc[kslab0 + u * 16 + a_fast]);
            sh = *reinterpret_cast<const float4 *>(&Ash[kslab0 + u * 16 + a_fast]);
          } else {
            const float s1 = Asc[kslab0 + u * 16 + a_slow], h1 = Ash[kslab0 + u * 16 + a_slow];''','''            sc = *reinterpret_cast<const float4 *>(&Asc[kslab0 + u * kSW + a_fast]);
            sh = *reinterpret_cast<const float4 *>(&Ash[kslab0 + u * kSW + a_fast]);
          } else {
            const float s1 = Asc[kslab0 + u * kSW + a_slow], h1 = Ash[kslab0 + u * kSW + a_slow];''')
rep('''          va = drop4(va, a_key, (uint32_t)(offa0 + (long)(kslab0 / 16 + u) * sa16), P.a_drop_p, a_inv);
        if (b_dropout && b_ok)
          vb = drop4(vb, b_key, (uint32_t)(offb0 + (long)(kslab0 / 16 + u) * sb16), P.b_drop_p, b_inv);
        put(As[buf], a_kc, a_slow, a_fast, u * 16, va);
        put(Bs[buf], b_kc, b_slow, b_fast, u * 16, vb);''','''          va = drop4(va, a_key, (uint32_t)(offa0 + (long)(kslab0 / kSW + u) * sa16), P.a_drop_p, a_inv);
        if (b_dropout && b_ok)
          vb = drop4(vb, b_key, (uint32_t)(offb0 + (long)(kslab0 / kSW + u) * sb16), P.b_drop_p, b_inv);
        put(As[buf], a_kc, a_slow, a_fast, u * kSW, va);
        put(Bs[buf], b_kc, b_slow, b_fast, u * kSW, vb);''')
rep('''        fa[u] = fetch_tile(P.a, P.a2, P.lda_m, P.lda_k, m0, P.M, k0 + u * 16, kend, tid);
        fb[u] = fetch_tile<false>(P.b, nullptr, P.ldb_n, P.ldb_k, n0, P.N, k0 + u * 16, kend, tid);''','''        fa[u] = fetch_tile<kSW>(P.a, P.a2, P.lda_m, P.lda_k, m0, P.M, k0 + u * kSW, kend, tid);
        fb[u] = fetch_tile<kSW, false>(P.b, nullptr, P.ldb_n, P.ldb_k, n0, P.N, k0 + u * kSW, kend, tid);''')
rep('''        commit_tile(As[buf], fa[u], fxa, P.lda_k, m0, P.M, kfetched + u * 16, kend, false, u * 16, tid);
        commit_tile(Bs[buf], fb[u], fxb, P.ldb_k, n0, P.N, kfetched + u * 16, kend, ones, u * 16, tid);''','''        commit_tile<kSW>(As[buf], fa[u], fxa, P.lda_k, m0, P.M, kfetched + u * kSW, kend, false, u * kSW, tid);
        commit_tile<kSW>(Bs[buf], fb[u], fxb, P.ldb_k, n0, P.N, kfetched + u * kSW, kend, ones, u * kSW, tid);''')
# epilogue
rep('''#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Cs[wr * 32 + i * 16 + fg * 4 + r][wc * 32 + j * 16 + fr] = acc[i][j][r];''','''#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < kNJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Cs[wr * 32 + i * 16 + fg * 4 + r][wc * (16 * kNJ) + j * 16 + fr] = acc[i][j][r];''')
rep('''    for (int qq = 0; qq < 4; ++qq) {
      const int row = (tid >> 4) + qq * 16;''','''    for (int qq = 0; qq < kBM / kRowPhases; ++qq) {
      const int row = (tid >> 4) + qq * kRowPhases;''')
rep('''      static_assert(sizeof(Bs) >= sizeof(float) * 2 * 16 * kBN, "statistics scratch must fit the B buffers");
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[(0 * 16 + (tid >> 4)) * kBN + c4 + e] = cs[e];
        red[(1 * 16 + (tid >> 4)) * kBN + c4 + e] = cq[e];
      }
      __syncthreads();
      if (tid < 2 * kBN) {
        const int which = tid >> 6, col = tid & 63;
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc += (double)red[(which * 16 + r) * kBN + col];''','''      static_assert(sizeof(Bs) >= sizeof(float) * 2 * kRowPhases * kBN, "statistics scratch must fit the B buffers");
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[(0 * kRowPhases + (tid >> 4)) * kBN + c4 + e] = cs[e];
        red[(1 * kRowPhases + (tid >> 4)) * kBN + c4 + e] = cq[e];
      }
      __syncthreads();
      if (tid < 2 * kBN) {
        const int which = tid >> 6, col = tid & 63;
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < kRowPhases; ++r) acc += (double)red[(which * kRowPhases + r) * kBN + col];''')
# atomic path
old=s[s.index('  float bias_v[2];'):s.index('// ------------------------------------------------------------------------------------------------\n// y = LayerNorm(residual + dropout(x))')]
new='''  float bias_v[kNJ];
#pragma unroll
  for (int j = 0; j < kNJ; ++j) {
    const int n = n0 + wc * (16 * kNJ) + j * 16 + fr;
    bias_v[j] = (P.bias && slice == 0 && n < pN) ? P.bias[n] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < kNJ; ++j) {
      const int n = n0 + wc * (16 * kNJ) + j * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wr * 32 + i * 16 + fg * 4 + r;
        if (m >= pM) continue;
        float v = acc[i][j][r];
        if (n < pN) {
          v = (v + bias_v[j]) * scale;
          if (relu) v = fmaxf(v, 0.f);
          if (drop)
            v = rng::keep(ctr, site, (uint32_t)((long)m * pN + n), p_drop) ? v * inv_keep : 0.f;
          atomicAdd(cptr + (long)m * ldc + n, v);
        } else if (ones_col && n == pN) {
          atomicAdd(bgrad + m, v * scale);
        }
      }
    }
}

'''
s=s.replace(old,new)
rep('''  hipLaunchKernelGGL(gemm_kernel, dim3((unsigned)total), dim3(kGemmThreads), 0, (hipStream_t)stream,''','''  static const int forced = getenv("BUTD_GEMM_THREADS") ? atoi(getenv("BUTD_GEMM_THREADS")) : 0;
  const int threads = forced ? forced : 512;
  if (threads == 256)
    hipLaunchKernelGGL(gemm_kernel<256>, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, batch,
                       rng_counter);
  else
  hipLaunchKernelGGL(gemm_kernel<512>, dim3((unsigned)total), dim3(512), 0, (hipStream_t)stream,''')
open(p,'w').write(s)
