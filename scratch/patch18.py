p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
# remove PF template: make kernel non-template again but keep mfma_slab lambda
s=s.replace('''// PF ("prefetch-all") variant: when the contraction range of a workgroup is at most kPfSlabs slabs and
// there is no companion operand, ALL of its global loads are issued before the first MFMA, so the
// HBM/L2 latency is paid once per workgroup instead of once per slab -- these GEMMs are small and a
// workgroup's life is a latency chain, not a bandwidth stream.
constexpr int kPfSlabs = 9;  // 9 x 32 = 288 = d_model
template <bool PF>
__global__''','''__global__''')
a=s.index('  if (PF) {\n    // every load of this workgroup up front')
b=s.index('  } else {\n    // streaming: double-buffered LDS, one barrier per slab')
s=s[:a]+'''  // Fast path (interior tiles, the common case): every address is  base + slab * step  with the
  // per-thread bases computed once; a slab costs each thread 2*kSub float4 loads, 2*kSub LDS writes and
  // the MFMAs -- no bounds checks, no index arithmetic.  Edge tiles take the generic path below.
  const bool a_kc = P.lda_k == 1, b_kc = P.ldb_k == 1;
  const bool fast =
      P.a2 == nullptr && !ones && ((kend - kbeg) % kBK) == 0 && m0 + kBM <= P.M && n0 + kBN <= P.N &&
      ((a_kc ? P.lda_m : P.lda_k) & 3) == 0 && ((b_kc ? P.ldb_n : P.ldb_k) & 3) == 0 &&
      ((((uintptr_t)P.a) | ((uintptr_t)P.b)) & 15) == 0 && (kbeg & 3) == 0;
  if (fast) {
    const int a_slow = a_kc ? (tid >> 2) : (tid >> 4), a_fast = a_kc ? (tid & 3) * 4 : (tid & 15) * 4;
    const int b_slow = b_kc ? (tid >> 2) : (tid >> 4), b_fast = b_kc ? (tid & 3) * 4 : (tid & 15) * 4;
    const float *pa = a_kc ? P.a + (long)(m0 + a_slow) * P.lda_m + kbeg + a_fast
                           : P.a + (long)(kbeg + a_slow) * P.lda_k + m0 + a_fast;
    const float *pb = b_kc ? P.b + (long)(n0 + b_slow) * P.ldb_n + kbeg + b_fast
                           : P.b + (long)(kbeg + b_slow) * P.ldb_k + n0 + b_fast;
    const long sa16 = a_kc ? 16 : 16 * P.lda_k, sb16 = b_kc ? 16 : 16 * P.ldb_k;  // per 16 k
    const float *asc = P.a_chan_scale, *ash = P.a_chan_shift;   // channel = k (varies per slab)
    float4 bsc4 = make_float4(1.f, 1.f, 1.f, 1.f), bsh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool b_aff = P.b_chan_scale != nullptr;
    if (b_aff && !b_kc) {  // channel = B row = 4 consecutive rows of this thread: loop-invariant
      bsc4 = *reinterpret_cast<const float4 *>(P.b_chan_scale + n0 + b_fast);
      bsh4 = *reinterpret_cast<const float4 *>(P.b_chan_shift + n0 + b_fast);
    } else if (b_aff) {
      const float sc = P.b_chan_scale[n0 + b_slow], sh = P.b_chan_shift[n0 + b_slow];
      bsc4 = make_float4(sc, sc, sc, sc);
      bsh4 = make_float4(sh, sh, sh, sh);
    }
    float4 ra[kSub], rb[kSub], rsc[kSub], rsh[kSub];
    auto fetch_fast = [&](int slab) {
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        ra[u] = *reinterpret_cast<const float4 *>(pa + (long)(slab * kSub + u) * sa16);
        rb[u] = *reinterpret_cast<const float4 *>(pb + (long)(slab * kSub + u) * sb16);
        if (asc) {
          const int kk = kbeg + (slab * kSub + u) * 16 + (a_kc ? a_fast : a_slow);
          if (a_kc) {
            rsc[u] = *reinterpret_cast<const float4 *>(asc + kk);
            rsh[u] = *reinterpret_cast<const float4 *>(ash + kk);
          } else {
            const float sc = asc[kk], sh = ash[kk];
            rsc[u] = make_float4(sc, sc, sc, sc);
            rsh[u] = make_float4(sh, sh, sh, sh);
          }
        }
      }
    };
    auto put = [&](float (*tile)[kLd], bool kc, int slow, int fst, int koff, float4 v) {
      if (kc) {
        *reinterpret_cast<float4 *>(&tile[slow][koff + fst]) = v;
      } else {
        tile[fst + 0][koff + slow] = v.x; tile[fst + 1][koff + slow] = v.y;
        tile[fst + 2][koff + slow] = v.z; tile[fst + 3][koff + slow] = v.w;
      }
    };
    auto commit_fast = [&](int buf) {
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        float4 va = ra[u], vb = rb[u];
        if (asc) {
          va.x = fmaxf(va.x * rsc[u].x + rsh[u].x, 0.f); va.y = fmaxf(va.y * rsc[u].y + rsh[u].y, 0.f);
          va.z = fmaxf(va.z * rsc[u].z + rsh[u].z, 0.f); va.w = fmaxf(va.w * rsc[u].w + rsh[u].w, 0.f);
        }
        if (b_aff) {
          vb.x = fmaxf(vb.x * bsc4.x + bsh4.x, 0.f); vb.y = fmaxf(vb.y * bsc4.y + bsh4.y, 0.f);
          vb.z = fmaxf(vb.z * bsc4.z + bsh4.z, 0.f); vb.w = fmaxf(vb.w * bsc4.w + bsh4.w, 0.f);
        }
        put(As[buf], a_kc, a_slow, a_fast, u * 16, va);
        put(Bs[buf], b_kc, b_slow, b_fast, u * 16, vb);
      }
    };
    const int nslab = (kend - kbeg) / kBK;
    fetch_fast(0);
    commit_fast(0);
    __syncthreads();
    for (int sl = 0; sl < nslab; ++sl) {
      const bool more = sl + 1 < nslab;
      if (more) fetch_fast(sl + 1);
      mfma_slab(sl & 1);
      if (more) commit_fast((sl + 1) & 1);
      __syncthreads();
    }
'''+s[b:]
s=s.replace('''  bool pf = true;  // every problem: no companion operand, <= kPfSlabs slabs per workgroup
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
                       (hipStream_t)stream, batch, rng_counter);''','''  hipLaunchKernelGGL(gemm_kernel, dim3((unsigned)total), dim3(kGemmThreads), 0, (hipStream_t)stream,
                     batch, rng_counter);''')
open(p,'w').write(s)
