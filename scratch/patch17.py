p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
# 1. template fetch_tile on WITH_A2
s=s.replace('''__device__ inline Frag4 fetch_tile(const float *__restrict__ src, const float *__restrict__ src2,
                                   long ld_row, long ld_k, int row0, int nrows, int k0, int kend,
                                   int tid) {''','''template <bool WITH_A2 = true>
__device__ inline Frag4 fetch_tile(const float *__restrict__ src, const float *__restrict__ src2,
                                   long ld_row, long ld_k, int row0, int nrows, int k0, int kend,
                                   int tid) {''')
s=s.replace('''    if (src2) {
      if (vec && ((((uintptr_t)src2) & 15) == 0)) {''','''    if (WITH_A2 && src2) {
      if (vec && ((((uintptr_t)src2) & 15) == 0)) {''')
# 2. kernel template + PF main loop
s=s.replace('''__global__ __launch_bounds__(kGemmThreads) void gemm_kernel(GemmBatch batch,
                                                            const uint64_t *__restrict__ rng_counter) {''','''// PF ("prefetch-all") variant: when the contraction range of a workgroup is at most kPfSlabs slabs and
// there is no companion operand, ALL of its global loads are issued before the first MFMA, so the
// HBM/L2 latency is paid once per workgroup instead of once per slab -- these GEMMs are small and a
// workgroup's life is a latency chain, not a bandwidth stream.
constexpr int kPfSlabs = 9;  // 9 x 32 = 288 = d_model
template <bool PF>
__global__ __launch_bounds__(kGemmThreads) void gemm_kernel(GemmBatch batch,
                                                            const uint64_t *__restrict__ rng_counter) {''')
old=s[s.index('  // double-buffered LDS, one barrier per slab: slab i+1 travels global -> registers while slab i is'):s.index('  // epilogue: lane holds C[row = fg*4 + r][col = fr] of each 16x16 tile.')]
new='''  const bool ones = P.ones_col != 0;
  auto mfma_slab = [&](int buf) {
#pragma unroll
    for (int u = 0; u < kSub; ++u) {
      f32x4 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i] = *reinterpret_cast<const f32x4 *>(&As[buf][wr * 32 + i * 16 + fr][u * 16 + fg * 4]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bf[j] = *reinterpret_cast<const f32x4 *>(&Bs[buf][wc * 32 + j * 16 + fr][u * 16 + fg * 4]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
  };
  if (PF) {
    // every load of this workgroup up front (<= 36 float4 per thread), then commit/multiply slab by slab
    const int nslab = (kend - kbeg + kBK - 1) / kBK;
    float4 ra[kPfSlabs][kSub], rb[kPfSlabs][kSub];
#pragma unroll
    for (int sl = 0; sl < kPfSlabs; ++sl)
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        ra[sl][u] = make_float4(0.f, 0.f, 0.f, 0.f);
        rb[sl][u] = ra[sl][u];
        if (sl < nslab) {
          ra[sl][u] = fetch_tile<false>(P.a, nullptr, P.lda_m, P.lda_k, m0, P.M, kbeg + sl * kBK + u * 16, kend, tid).a;
          rb[sl][u] = fetch_tile<false>(P.b, nullptr, P.ldb_n, P.ldb_k, n0, P.N, kbeg + sl * kBK + u * 16, kend, tid).a;
        }
      }
    auto commit_pf = [&](int sl, int buf) {
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        Frag4 f;
        f.a = ra[sl][u];
        commit_tile(As[buf], f, false, 0, 0.f, P.a_chan_scale, P.a_chan_shift, true, P.lda_k, m0, P.M,
                    kbeg + sl * kBK + u * 16, kend, false, u * 16, tid);
        f.a = rb[sl][u];
        commit_tile(Bs[buf], f, false, 0, 0.f, P.b_chan_scale, P.b_chan_shift, false, P.ldb_k, n0, P.N,
                    kbeg + sl * kBK + u * 16, kend, ones, u * 16, tid);
      }
    };
    commit_pf(0, 0);
    __syncthreads();
#pragma unroll
    for (int sl = 0; sl < kPfSlabs; ++sl) {
      if (sl < nslab) {
        if (sl + 1 < nslab) commit_pf(sl + 1 < kPfSlabs ? sl + 1 : 0, (sl + 1) & 1);
        mfma_slab(sl & 1);
        __syncthreads();
      }
    }
  } else {
    // streaming: double-buffered LDS, one barrier per slab: slab i+1 travels global -> registers while
    // slab i is multiplied, then lands in the other buffer
    Frag4 fa[kSub], fb[kSub];
    int kfetched = kbeg;
    auto fetch = [&](int k0) {
      kfetched = k0;
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        fa[u] = fetch_tile(P.a, P.a2, P.lda_m, P.lda_k, m0, P.M, k0 + u * 16, kend, tid);
        fb[u] = fetch_tile<false>(P.b, nullptr, P.ldb_n, P.ldb_k, n0, P.N, k0 + u * 16, kend, tid);
      }
    };
    auto commit = [&](int buf) {
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        commit_tile(As[buf], fa[u], P.a2 != nullptr, P.a2_mode, P.a2_scale, P.a_chan_scale,
                    P.a_chan_shift, true, P.lda_k, m0, P.M, kfetched + u * 16, kend, false, u * 16, tid);
        commit_tile(Bs[buf], fb[u], false, 0, 0.f, P.b_chan_scale, P.b_chan_shift, false, P.ldb_k, n0,
                    P.N, kfetched + u * 16, kend, ones, u * 16, tid);
      }
    };
    fetch(kbeg);
    commit(0);
    __syncthreads();
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += kBK) {
      const bool more = k0 + kBK < kend;
      if (more) fetch(k0 + kBK);
      mfma_slab(cur);
      if (more) commit(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }

'''
s=s.replace(old,new)
# 3. host: choose PF
s=s.replace('''  hipLaunchKernelGGL(gemm_kernel, dim3((unsigned)total), dim3(kGemmThreads), 0, (hipStream_t)stream,
                     batch, rng_counter);''','''  bool pf = true;  // every problem: no companion operand, <= kPfSlabs slabs per workgroup
  for (int i = 0; i < batch.count; ++i) {
    const butd_gemm_problem &p = batch.p[i];
    const int kslab = (p.K + kBK - 1) / kBK;
    const int per = (kslab + p.split_k - 1) / p.split_k;
    if (p.a2 != nullptr || per > kPfSlabs) pf = false;
  }
  if (pf)
    hipLaunchKernelGGL(gemm_kernel<true>, dim3((unsigned)total), dim3(kGemmThreads), 0,
                       (hipStream_t)stream, batch, rng_counter);
  else
    hipLaunchKernelGGL(gemm_kernel<false>, dim3((unsigned)total), dim3(kGemmThreads), 0,
                       (hipStream_t)stream, batch, rng_counter);''')
open(p,'w').write(s)
