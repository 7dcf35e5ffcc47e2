p='butd_detr_amd/train_step.py'
s=open(p).read()
s=s.replace('''    return torch.optim.AdamW(groups, lr=lr, weight_decay=weight_decay, capturable=capturable,
                             foreach=True if capturable else None)''','''    # capturable + fused: ONE multi-tensor kernel per parameter group; the capturable *foreach* path
    # degenerates into ~1 200 single-tensor divisions per step on this stack (rocprof, round 1)
    return torch.optim.AdamW(groups, lr=lr, weight_decay=weight_decay, capturable=capturable,
                             fused=True if capturable else None)''')
open(p,'w').write(s)

p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
old=s[s.index('__device__ inline float wave_sum(float v) {'):s.index('template <int PER>\n__global__ __launch_bounds__(kLnThreads) void ln_fwd_kernel')]
new='''// wave-wide sum on the DPP crossbar: running sums inside each 16-lane row (row_shr 1/2/4/8), then
// row_bcast:15 / row_bcast:31 carry the row totals; lane 63 holds the total, v_readlane broadcasts it.
// (__shfl_xor would be six ds_bpermute round trips, ~100 cycles each.)
__device__ inline float wave_sum(float v) {
#define BUTD_ADD_DPP(CTRL, RMASK)                                                                  \\
  asm volatile("s_nop 1\\n\\tv_add_f32_dpp %0, %0, %0 " CTRL " row_mask:" RMASK " bank_mask:0xf" : "+v"(v))
  BUTD_ADD_DPP("row_shr:1", "0xf");
  BUTD_ADD_DPP("row_shr:2", "0xf");
  BUTD_ADD_DPP("row_shr:4", "0xf");
  BUTD_ADD_DPP("row_shr:8", "0xf");
  BUTD_ADD_DPP("row_bcast:15", "0xa");
  BUTD_ADD_DPP("row_bcast:31", "0xc");
#undef BUTD_ADD_DPP
  asm volatile("s_nop 1" ::: "memory");
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}

'''
s=s.replace(old,new)
# ln bwd: 16 waves x 4 rows per block
s=s.replace('''// Each workgroup walks kLnRowsPerBlock rows (4 waves x 16 rows), keeps dgamma/dbeta partials of its
// columns in registers, reduces them across its waves through LDS and issues ONE atomic per column.
constexpr int kLnRowsPerWave = 16;
template <int PER>
__global__ __launch_bounds__(kLnThreads) void ln_bwd_kernel(''','''// Each workgroup covers 64 rows (16 waves x 4 rows: short serial chains), keeps dgamma/dbeta partials
// of its columns in registers, reduces them across its waves through LDS and issues ONE atomic per
// column.
constexpr int kLnRowsPerWave = 4;
constexpr int kLnBwdThreads = 1024;
template <int PER>
__global__ __launch_bounds__(kLnBwdThreads) void ln_bwd_kernel(''')
a=s.index('__global__ __launch_bounds__(kLnBwdThreads) void ln_bwd_kernel(')
b=s.index('}  // namespace', a)
body=s[a:b]
body=body.replace('red[2][kLnThreads / 64][64 * PER]','red[2][kLnBwdThreads / 64][64 * PER]')
body=body.replace('(blockIdx.x * (kLnThreads / 64) + wave) * kLnRowsPerWave','(blockIdx.x * (kLnBwdThreads / 64) + wave) * kLnRowsPerWave')
body=body.replace('for (int c = threadIdx.x; c < cols; c += kLnThreads) {','for (int c = threadIdx.x; c < cols; c += kLnBwdThreads) {')
body=body.replace('for (int w = 0; w < kLnThreads / 64; ++w) {','for (int w = 0; w < kLnBwdThreads / 64; ++w) {')
s=s[:a]+body+s[b:]
s=s.replace('''  const int rows_per_block = (kLnThreads / 64) * kLnRowsPerWave;
  const dim3 grid((rows + rows_per_block - 1) / rows_per_block);
  LN_DISPATCH(ln_bwd_kernel, rows, cols, dy,''','''  const int rows_per_block = (kLnBwdThreads / 64) * kLnRowsPerWave;
  const dim3 grid((rows + rows_per_block - 1) / rows_per_block);
  LN_DISPATCH_T(kLnBwdThreads, ln_bwd_kernel, rows, cols, dy,''')
s=s.replace('''#define LN_DISPATCH(KERNEL, ...)                                                                   \\
  do {                                                                                             \\
    const int per = (cols + 63) / 64;                                                              \\
    if (per <= 4) hipLaunchKernelGGL((KERNEL<4>), grid, dim3(kLnThreads), 0, s, __VA_ARGS__);      \\
    else if (per <= 5) hipLaunchKernelGGL((KERNEL<5>), grid, dim3(kLnThreads), 0, s, __VA_ARGS__); \\
    else if (per <= 8) hipLaunchKernelGGL((KERNEL<8>), grid, dim3(kLnThreads), 0, s, __VA_ARGS__); \\
    else hipLaunchKernelGGL((KERNEL<16>), grid, dim3(kLnThreads), 0, s, __VA_ARGS__);              \\
  } while (0)''','''#define LN_DISPATCH_T(THREADS, KERNEL, ...)                                                     \\
  do {                                                                                          \\
    const int per = (cols + 63) / 64;                                                           \\
    if (per <= 4) hipLaunchKernelGGL((KERNEL<4>), grid, dim3(THREADS), 0, s, __VA_ARGS__);      \\
    else if (per <= 5) hipLaunchKernelGGL((KERNEL<5>), grid, dim3(THREADS), 0, s, __VA_ARGS__); \\
    else if (per <= 8) hipLaunchKernelGGL((KERNEL<8>), grid, dim3(THREADS), 0, s, __VA_ARGS__); \\
    else hipLaunchKernelGGL((KERNEL<16>), grid, dim3(THREADS), 0, s, __VA_ARGS__);              \\
  } while (0)
#define LN_DISPATCH(KERNEL, ...) LN_DISPATCH_T(kLnThreads, KERNEL, __VA_ARGS__)''')
open(p,'w').write(s)
