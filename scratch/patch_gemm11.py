import re
p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    assert s.count(a)==cnt, (s.count(a), a[:80])
    s=s.replace(a,b)
rep('constexpr int kBM = 64, kBN = 64, kBK = 32, kLd = kBK + 4;  // LDS row stride 68 floats = 17 x 16 B',
    'constexpr int kBK = 32, kLd = kBK + 4;  // slab depth; LDS row stride 36 floats = 9 x 16 B\n// Output tiles are TILE x TILE with TILE = 64 (large grids) or 32 (grids that would leave CUs idle).')
# tile_idx: needs RQ = float4 per k-row of a row-contiguous operand = TILE/4
rep('''template <int SW>
__device__ inline TileIdx tile_idx(long ld_k, int tid) {
  TileIdx t;
  t.kc = ld_k == 1;
  t.slow = t.kc ? (tid / (SW / 4)) : (tid >> 4);
  t.fast = t.kc ? (tid % (SW / 4)) * 4 : (tid & 15) * 4;
  return t;
}''','''template <int SW, int TILE>
__device__ inline TileIdx tile_idx(long ld_k, int tid) {
  TileIdx t;
  t.kc = ld_k == 1;
  t.slow = t.kc ? (tid / (SW / 4)) : (tid / (TILE / 4));
  t.fast = t.kc ? (tid % (SW / 4)) * 4 : (tid % (TILE / 4)) * 4;
  return t;
}''')
rep('''template <int SW, bool WITH_A2 = true>
__device__ inline Frag4 fetch_tile(''','''template <int SW, int TILE, bool WITH_A2 = true>
__device__ inline Frag4 fetch_tile(''')
rep('''  const TileIdx t = tile_idx<SW>(ld_k, tid);
  const long ld_slow''','''  const TileIdx t = tile_idx<SW, TILE>(ld_k, tid);
  const long ld_slow''')
rep('''template <int SW>
__device__ inline void commit_tile(float (*tile)[kLd], const Frag4 &f, const OperandFx &fx, long ld_k,
                                   int row0, int nrows, int k0, int kend, bool ones, int koff, int tid) {
  const TileIdx t = tile_idx<SW>(ld_k, tid);''','''template <int SW, int TILE>
__device__ inline void commit_tile(float (*tile)[kLd], const Frag4 &f, const OperandFx &fx, long ld_k,
                                   int row0, int nrows, int k0, int kend, bool ones, int koff, int tid) {
  const TileIdx t = tile_idx<SW, TILE>(ld_k, tid);''')
rep('''template <int THREADS, bool FAST>
__global__ __launch_bounds__(THREADS) void gemm_kernel(GemmBatch batch,
                                                       const uint64_t *__restrict__ rng_counter) {
  constexpr int kSW = THREADS / 16;        // k-width of one staging step
  constexpr int kSub = kBK / kSW;          // staging steps per slab
  constexpr int kWavesN = THREADS / 128;   // wave grid 2 x kWavesN
  constexpr int kNJ = 4 / kWavesN;         // 16-column fragments per wave
  constexpr int kRowPhases = THREADS / 16; // rows written per epilogue pass
  __shared__ __attribute__((aligned(16))) float As[2][kBM][kLd];
  __shared__ __attribute__((aligned(16))) float Bs[2][kBN][kLd];''','''template <int THREADS, int TILE, bool FAST>
__global__ __launch_bounds__(THREADS) void gemm_kernel(GemmBatch batch,
                                                       const uint64_t *__restrict__ rng_counter) {
  constexpr int kBM = TILE, kBN = TILE;
  constexpr int kSW = 4 * THREADS / TILE;  // k-width of one staging step (one float4 per thread)
  constexpr int kSub = kBK / kSW;          // staging steps per slab
  constexpr int kWavesN = THREADS / 128;   // wave grid 2 x kWavesN
  constexpr int kMI = TILE / 32;           // 16-row fragments per wave
  constexpr int kNJ = TILE / (16 * kWavesN);   // 16-column fragments per wave
  constexpr int kRQ = TILE / 4;            // float4 per tile row / per k-row of a row-contiguous operand
  constexpr int kRowPhases = THREADS / kRQ;    // rows written per epilogue pass
  static_assert(kSub >= 1 && kNJ >= 1 && kMI >= 1 && kRowPhases <= TILE, "unsupported THREADS x TILE");
  __shared__ __attribute__((aligned(16))) float As[2][kBM][kLd];
  __shared__ __attribute__((aligned(16))) float Bs[2][kBN][kLd];''')
rep('''  f32x4 acc[2][kNJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < kNJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};''','''  f32x4 acc[kMI][kNJ];
#pragma unroll
  for (int i = 0; i < kMI; ++i)
#pragma unroll
    for (int j = 0; j < kNJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};''')
# generic mfma_slab
rep('''      f32x4 af[2], bf[kNJ];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i] = *reinterpret_cast<const f32x4 *>(&As[buf][wr * 32 + i * 16 + fr][u * 16 + fg * 4]);''','''      f32x4 af[kMI], bf[kNJ];
#pragma unroll
      for (int i = 0; i < kMI; ++i)
        af[i] = *reinterpret_cast<const f32x4 *>(&As[buf][wr * (16 * kMI) + i * 16 + fr][u * 16 + fg * 4]);''')
rep('''      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < kNJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
  };''','''      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < kMI; ++i)
#pragma unroll
          for (int j = 0; j < kNJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
  };''')
rep('''    const int a_slow = a_kc ? (tid / (kSW / 4)) : (tid >> 4);
    const int a_fast = a_kc ? (tid % (kSW / 4)) * 4 : (tid & 15) * 4;
    const int b_slow = b_kc ? (tid / (kSW / 4)) : (tid >> 4);
    const int b_fast = b_kc ? (tid % (kSW / 4)) * 4 : (tid & 15) * 4;''','''    const int a_slow = a_kc ? (tid / (kSW / 4)) : (tid / kRQ);
    const int a_fast = a_kc ? (tid % (kSW / 4)) * 4 : (tid % kRQ) * 4;
    const int b_slow = b_kc ? (tid / (kSW / 4)) : (tid / kRQ);
    const int b_fast = b_kc ? (tid % (kSW / 4)) * 4 : (tid % kRQ) * 4;''')
rep('''        f32x4 af[2], bf[kNJ];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = frag(As[buf], a_kc, wr * 32 + i * 16 + fr, u * 16 + fg * 4);''','''        f32x4 af[kMI], bf[kNJ];
#pragma unroll
        for (int i = 0; i < kMI; ++i)
          af[i] = frag(As[buf], a_kc, wr * (16 * kMI) + i * 16 + fr, u * 16 + fg * 4);''')
rep('''          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < kNJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);''','''          for (int i = 0; i < kMI; ++i)
#pragma unroll
            for (int j = 0; j < kNJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);''')
# generic path calls
rep('''        fa[u] = fetch_tile<kSW>(P.a, P.a2, P.lda_m, P.lda_k, m0, P.M, k0 + u * kSW, kend, tid);
        fb[u] = fetch_tile<kSW, false>(P.b, nullptr, P.ldb_n, P.ldb_k, n0, P.N, k0 + u * kSW, kend, tid);''','''        fa[u] = fetch_tile<kSW, TILE>(P.a, P.a2, P.lda_m, P.lda_k, m0, P.M, k0 + u * kSW, kend, tid);
        fb[u] = fetch_tile<kSW, TILE, false>(P.b, nullptr, P.ldb_n, P.ldb_k, n0, P.N, k0 + u * kSW, kend, tid);''')
rep('''        commit_tile<kSW>(As[buf], fa[u], fxa, P.lda_k, m0, P.M, kfetched + u * kSW, kend, false, u * kSW, tid);
        commit_tile<kSW>(Bs[buf], fb[u], fxb, P.ldb_k, n0, P.N, kfetched + u * kSW, kend, ones, u * kSW, tid);''','''        commit_tile<kSW, TILE>(As[buf], fa[u], fxa, P.lda_k, m0, P.M, kfetched + u * kSW, kend, false, u * kSW, tid);
        commit_tile<kSW, TILE>(Bs[buf], fb[u], fxb, P.ldb_k, n0, P.N, kfetched + u * kSW, kend, ones, u * kSW, tid);''')
# epilogue
rep('''#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < kNJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Cs[wr * 32 + i * 16 + fg * 4 + r][wc * (16 * kNJ) + j * 16 + fr] = acc[i][j][r];''','''#pragma unroll
    for (int i = 0; i < kMI; ++i)
#pragma unroll
      for (int j = 0; j < kNJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Cs[wr * (16 * kMI) + i * 16 + fg * 4 + r][wc * (16 * kNJ) + j * 16 + fr] = acc[i][j][r];''')
rep('''    const int c4 = (tid & 15) * 4;''','''    const int c4 = (tid % kRQ) * 4, rphase = tid / kRQ;''')
rep('''      const int row = (tid >> 4) + qq * kRowPhases;''','''      const int row = rphase + qq * kRowPhases;''')
rep('''        red[(0 * kRowPhases + (tid >> 4)) * kBN + c4 + e] = cs[e];
        red[(1 * kRowPhases + (tid >> 4)) * kBN + c4 + e] = cq[e];''','''        red[(0 * kRowPhases + rphase) * kBN + c4 + e] = cs[e];
        red[(1 * kRowPhases + rphase) * kBN + c4 + e] = cq[e];''')
rep('''        const int which = tid >> 6, col = tid & 63;''','''        const int which = tid / kBN, col = tid % kBN;''')
rep('''#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < kNJ; ++j) {
      const int n = n0 + wc * (16 * kNJ) + j * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wr * 32 + i * 16 + fg * 4 + r;''','''#pragma unroll
  for (int i = 0; i < kMI; ++i)
#pragma unroll
    for (int j = 0; j < kNJ; ++j) {
      const int n = n0 + wc * (16 * kNJ) + j * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wr * (16 * kMI) + i * 16 + fg * 4 + r;''')
# host
a=s.index('static int launch_group(')
b=s.index('int butd_gemm_grouped(const butd_gemm_problem *problems, int count')
host='''template <int TILE>
static long fill_batch(GemmBatch &batch, const butd_gemm_problem *problems, const int *index, int count) {
  long total = 0;
  batch.count = 0;
  for (int i = 0; i < count; ++i) {
    butd_gemm_problem p = problems[index[i]];
    if (p.split_k < 1) p.split_k = 1;
    const int ncols = p.N + (p.ones_col ? 1 : 0);
    const int tn = (ncols + TILE - 1) / TILE, tm = (p.M + TILE - 1) / TILE;
    batch.blk_begin[batch.count] = (int)total;
    batch.tiles_n[batch.count] = tn;
    batch.tiles_m[batch.count] = tm;
    batch.p[batch.count++] = p;
    total += (long)tn * tm * p.split_k;
    if (total > 0x7fffffffL) return -1;
  }
  for (int i = batch.count; i <= kMaxProblems; ++i) batch.blk_begin[i] = (int)total;
  return total;
}

static int launch_group(const butd_gemm_problem *problems, const int *index, int count, bool fast,
                        const uint64_t *rng_counter, hipStream_t stream) {
  if (count == 0) return 0;
  GemmBatch batch;
  long total = fill_batch<64>(batch, problems, index, count);
  if (total < 0) return (int)hipErrorInvalidValue;
  if (total == 0) return 0;
  // Configuration by grid size (measured, graph replay): 64x64 tiles / 4 waves for large grids;
  // 32x32 tiles when 64x64 would leave fewer than ~3 workgroups per CU (four times the workgroups, a
  // quarter of the matrix phase each, phases of co-resident workgroups overlap).
  static const int forced = getenv("BUTD_GEMM_CFG") ? atoi(getenv("BUTD_GEMM_CFG")) : 0;
  const int cfg = forced ? forced : (total <= 768 ? 32 : 64);
  if (cfg == 32) {
    total = fill_batch<32>(batch, problems, index, count);
    if (total < 0) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)total);
    if (fast) hipLaunchKernelGGL((gemm_kernel<256, 32, true>), grid, dim3(256), 0, stream, batch, rng_counter);
    else hipLaunchKernelGGL((gemm_kernel<256, 32, false>), grid, dim3(256), 0, stream, batch, rng_counter);
  } else if (cfg == 512) {
    const dim3 grid((unsigned)total);
    if (fast) hipLaunchKernelGGL((gemm_kernel<512, 64, true>), grid, dim3(512), 0, stream, batch, rng_counter);
    else hipLaunchKernelGGL((gemm_kernel<512, 64, false>), grid, dim3(512), 0, stream, batch, rng_counter);
  } else {
    const dim3 grid((unsigned)total);
    if (fast) hipLaunchKernelGGL((gemm_kernel<256, 64, true>), grid, dim3(256), 0, stream, batch, rng_counter);
    else hipLaunchKernelGGL((gemm_kernel<256, 64, false>), grid, dim3(256), 0, stream, batch, rng_counter);
  }
  return (int)hipGetLastError();
}

'''
s=s[:a]+host+s[b:]
open(p,'w').write(s)
