p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
s=s.replace('''struct GemmBatch {
  butd_gemm_problem p[kMaxProblems];
  int z_begin[kMaxProblems + 1];  // blockIdx.z range of each problem (split_k slices)
  int count;
};''','''struct GemmBatch {
  butd_gemm_problem p[kMaxProblems];
  int blk_begin[kMaxProblems + 1];  // linear workgroup range of each problem
  int tiles_n[kMaxProblems], tiles_m[kMaxProblems];
  int count;
};''')
s=s.replace('''  int pi = 0;
  while (pi + 1 < batch.count && (int)blockIdx.z >= batch.z_begin[pi + 1]) ++pi;
  const butd_gemm_problem &P = batch.p[pi];
  const int slice = blockIdx.z - batch.z_begin[pi];
  const int m0 = blockIdx.y * kBM, n0 = blockIdx.x * kBN;
  const int ncols = P.N + (P.ones_col ? 1 : 0);
  if (m0 >= P.M || n0 >= ncols) return;
''','''  // 1-D grid: every problem owns exactly tiles_n x tiles_m x split_k consecutive workgroups
  int pi = 0;
  while (pi + 1 < batch.count && (int)blockIdx.x >= batch.blk_begin[pi + 1]) ++pi;
  const butd_gemm_problem &P = batch.p[pi];
  int rel = blockIdx.x - batch.blk_begin[pi];
  const int tn = batch.tiles_n[pi], tm = batch.tiles_m[pi];
  const int bx = rel % tn;
  rel /= tn;
  const int by = rel % tm;
  const int slice = rel / tm;
  const int m0 = by * kBM, n0 = bx * kBN;
''')
old=s[s.index('  GemmBatch batch;\n  int gx = 0, gy = 0, z = 0;'):s.index('  return (int)hipGetLastError();\n}\n\n#define LN_DISPATCH_T')]
new='''  GemmBatch batch;
  long total = 0;
  batch.count = 0;
  for (int i = 0; i < count; ++i) {
    butd_gemm_problem p = problems[i];
    if (p.M <= 0 || p.N <= 0) continue;
    if (p.split_k < 1) p.split_k = 1;
    if (p.split_k > 1 && !p.accumulate) return (int)hipErrorInvalidValue;
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
  hipLaunchKernelGGL(gemm_kernel, dim3((unsigned)total), dim3(kGemmThreads), 0, (hipStream_t)stream,
                     batch, rng_counter);
'''
s=s.replace(old,new)
open(p,'w').write(s)
