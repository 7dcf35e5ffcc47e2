p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
def rep(a,b):
    global s
    assert s.count(a)==1, (s.count(a), a)
    s=s.replace(a,b)
rep('''template <int THREADS>
__global__ __launch_bounds__(THREADS) void gemm_kernel(GemmBatch batch,''','''// FAST: every problem of the launch satisfies fast_eligible() (host side): interior tiles stream whole
// float4s with addresses  base + slab * step  and no bounds checks; the generic instantiation handles
// ragged K, unaligned operands and the a2 companion.  Two kernels instead of one runtime branch: with
// both paths in one body the compiler merged their MFMA blocks and serialized loads behind them.
template <int THREADS, bool FAST>
__global__ __launch_bounds__(THREADS) void gemm_kernel(GemmBatch batch,''')
a=s.index('  const bool fast =\n      P.a2 == nullptr')
b=s.index('  if (fast) {')
s=s[:a]+s[b:]
rep('  if (fast) {','  if constexpr (FAST) {')
# ones in fast path
rep('''    if (a_aff) __syncthreads();
    // Two register sets''','''    if (a_aff) __syncthreads();
    // virtual ones-row of B (row index N): which of this thread's elements is it, if any
    const int ones_e = !ones ? -1 : (b_kc ? (n0 + b_slow == P.N ? 4 : -1)
                                          : ((n0 + b_fast <= P.N && P.N < n0 + b_fast + 4) ? P.N - (n0 + b_fast) : -1));
    // Two register sets''')
rep('''        put(As[buf], a_kc, a_slow, a_fast, u * kSW, va);
        put(Bs[buf], b_kc, b_slow, b_fast, u * kSW, vb);
      }
    };
    const int nslab''','''        if (ones_e == 4) vb = make_float4(1.f, 1.f, 1.f, 1.f);
        else if (ones_e == 0) vb.x = 1.f;
        else if (ones_e == 1) vb.y = 1.f;
        else if (ones_e == 2) vb.z = 1.f;
        else if (ones_e == 3) vb.w = 1.f;
        put(As[buf], a_kc, a_slow, a_fast, u * kSW, va);
        put(Bs[buf], b_kc, b_slow, b_fast, u * kSW, vb);
      }
    };
    const int nslab''')
# host
a=s.index('int butd_gemm_grouped(const butd_gemm_problem *problems, int count')
b=s.index('#define LN_DISPATCH_T')
host='''static bool fast_eligible(const butd_gemm_problem &p) {
  const bool a_kc = p.lda_k == 1, b_kc = p.ldb_k == 1;
  const int kslab = (p.K + kBK - 1) / kBK, split = p.split_k < 1 ? 1 : p.split_k;
  const long per = (long)((kslab + split - 1) / split) * kBK;   // contraction range of one slice
  return p.a2 == nullptr && p.K > 0 && (p.K % kBK) == 0 &&
         (a_kc || (p.M & 3) == 0) && (b_kc || (p.N & 3) == 0) &&   // partial tiles: whole float4 in or out
         ((a_kc ? p.lda_m : p.lda_k) & 3) == 0 && ((b_kc ? p.ldb_n : p.ldb_k) & 3) == 0 &&
         ((((uintptr_t)p.a) | ((uintptr_t)p.b)) & 15) == 0 &&
         (p.a_chan_scale == nullptr || per <= kAffK);
}

static int launch_group(const butd_gemm_problem *problems, const int *index, int count, bool fast,
                        const uint64_t *rng_counter, hipStream_t stream) {
  GemmBatch batch;
  long total = 0;
  batch.count = 0;
  for (int i = 0; i < count; ++i) {
    butd_gemm_problem p = problems[index[i]];
    if (p.split_k < 1) p.split_k = 1;
    const int ncols = p.N + (p.ones_col ? 1 : 0);
    const int tn = (ncols + kBN - 1) / kBN, tm = (p.M + kBM - 1) / kBM;
    batch.blk_begin[batch.count] = (int)total;
    batch.tiles_n[batch.count] = tn;
    batch.tiles_m[batch.count] = tm;
    batch.p[batch.count++] = p;
    total += (long)tn * tm * p.split_k;
    if (total > 0x7fffffffL) return (int)hipErrorInvalidValue;
  }
  if (batch.count == 0) return 0;
  for (int i = batch.count; i <= kMaxProblems; ++i) batch.blk_begin[i] = (int)total;
  static const int forced = getenv("BUTD_GEMM_THREADS") ? atoi(getenv("BUTD_GEMM_THREADS")) : 0;
  const int threads = forced ? forced : 512;
  const dim3 grid((unsigned)total);
  if (fast && threads == 256)
    hipLaunchKernelGGL((gemm_kernel<256, true>), grid, dim3(256), 0, stream, batch, rng_counter);
  else if (fast)
    hipLaunchKernelGGL((gemm_kernel<512, true>), grid, dim3(512), 0, stream, batch, rng_counter);
  else if (threads == 256)
    hipLaunchKernelGGL((gemm_kernel<256, false>), grid, dim3(256), 0, stream, batch, rng_counter);
  else
    hipLaunchKernelGGL((gemm_kernel<512, false>), grid, dim3(512), 0, stream, batch, rng_counter);
  return (int)hipGetLastError();
}

int butd_gemm_grouped(const butd_gemm_problem *problems, int count, const uint64_t *rng_counter,
                      butd_stream_t stream) {
  if (count <= 0) return 0;
  if (count > kMaxProblems) return (int)hipErrorInvalidValue;
  // the problems of a group are independent: the fast-eligible ones and the rest run as two launches
  int fast_idx[kMaxProblems], slow_idx[kMaxProblems], nf = 0, ns = 0;
  for (int i = 0; i < count; ++i) {
    const butd_gemm_problem &p = problems[i];
    if (p.M <= 0 || p.N <= 0) continue;
    if (p.split_k > 1 && !p.accumulate) return (int)hipErrorInvalidValue;
    if ((p.col_sum != nullptr) && (p.accumulate || p.split_k > 1)) return (int)hipErrorInvalidValue;
    if (fast_eligible(p)) fast_idx[nf++] = i; else slow_idx[ns++] = i;
  }
  int err = launch_group(problems, fast_idx, nf, true, rng_counter, (hipStream_t)stream);
  if (err) return err;
  return launch_group(problems, slow_idx, ns, false, rng_counter, (hipStream_t)stream);
}

'''
s=s[:a]+host+s[b:]
open(p,'w').write(s)
