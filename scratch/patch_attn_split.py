def rep(s,a,b,cnt=1):
    assert s.count(a)==cnt, (s.count(a), a[:80])
    return s.replace(a,b)
p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()

# ---------------- forward
s=rep(s,'''template <int NS, int NT>
__global__ __launch_bounds__(kAttnThreads) void attn_fwd_kernel(
    int H, int Lq, int Lk, int D, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const uint8_t *__restrict__ mask, float *__restrict__ out,
    float *__restrict__ lse, float p_drop, uint32_t site, const uint64_t *__restrict__ rng_counter) {
  using I = Img<NS>;
  __shared__ __attribute__((aligned(16))) float Kimg[2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float Vimg[2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float Bias[2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;''','''// NG = 2 (small grids, e.g. the decoder's 256 queries: 256 workgroups = ONE wave per SIMD, nothing to
// hide a stall behind): two wave groups of a 512-thread workgroup walk the even / odd key tiles of the
// same 64 queries with their own LDS images and merge their (o, m, l) states through LDS at the end.
template <int NS, int NT, int NG>
__global__ __launch_bounds__(kAttnThreads * NG) void attn_fwd_kernel(
    int H, int Lq, int Lk, int D, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const uint8_t *__restrict__ mask, float *__restrict__ out,
    float *__restrict__ lse, float p_drop, uint32_t site, const uint64_t *__restrict__ rng_counter) {
  using I = Img<NS>;
  __shared__ __attribute__((aligned(16))) float KimgG[NG][2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float VimgG[NG][2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float BiasG[NG][2][64];
  const int grp = threadIdx.x / kAttnThreads;
  float(*Kimg)[64][I::LD] = KimgG[grp];
  float(*Vimg)[64][I::LD] = VimgG[grp];
  float(*Bias)[64] = BiasG[grp];
  const int tid = threadIdx.x % kAttnThreads, lane = tid & 63, wave = tid >> 6;''')
a=s.index('  typename I::Regs kr, vr;\n  float br = 0.f;\n  I::fetch(kr, kb, E, D, 0, Lk, tid);\n  I::fetch(vr, vb, E, D, 0, Lk, tid);\n  if (tid < 64) br = key_bias(mb, tid, Lk);\n  I::commit(Kimg[0], kr, D, tid);\n  I::commit(Vimg[0], vr, D, tid);\n  if (tid < 64) Bias[0][tid] = br;\n  __syncthreads();\n  int cur = 0;\n  for (int key0 = 0; key0 < Lk; key0 += 64) {\n    const bool more = key0 + 64 < Lk;\n    if (more) {\n      I::fetch(kr, kb, E, D, key0 + 64, Lk, tid);\n      I::fetch(vr, vb, E, D, key0 + 64, Lk, tid);\n      if (tid < 64) br = key_bias(mb, key0 + 64 + tid, Lk);\n    }\n    if (live) {\n      f32x4 st[4];')
pro_old='''  typename I::Regs kr, vr;
  float br = 0.f;
  I::fetch(kr, kb, E, D, 0, Lk, tid);
  I::fetch(vr, vb, E, D, 0, Lk, tid);
  if (tid < 64) br = key_bias(mb, tid, Lk);
  I::commit(Kimg[0], kr, D, tid);
  I::commit(Vimg[0], vr, D, tid);
  if (tid < 64) Bias[0][tid] = br;
  __syncthreads();
  int cur = 0;
  for (int key0 = 0; key0 < Lk; key0 += 64) {
    const bool more = key0 + 64 < Lk;
    if (more) {
      I::fetch(kr, kb, E, D, key0 + 64, Lk, tid);
      I::fetch(vr, vb, E, D, key0 + 64, Lk, tid);
      if (tid < 64) br = key_bias(mb, key0 + 64 + tid, Lk);
    }
'''
pro_new='''  // key tiles of this wave group: grp, grp + NG, ... (a tile past Lk stages zeros with -inf bias and
  // contributes nothing, so both groups run the same number of iterations and barriers)
  const int iters = ((Lk + 63) / 64 + NG - 1) / NG;
  typename I::Regs kr, vr;
  float br = 0.f;
  I::fetch(kr, kb, E, D, grp * 64, Lk, tid);
  I::fetch(vr, vb, E, D, grp * 64, Lk, tid);
  if (tid < 64) br = key_bias(mb, grp * 64 + tid, Lk);
  I::commit(Kimg[0], kr, D, tid);
  I::commit(Vimg[0], vr, D, tid);
  if (tid < 64) Bias[0][tid] = br;
  __syncthreads();
  int cur = 0;
  for (int it = 0; it < iters; ++it) {
    const int key0 = (it * NG + grp) * 64;
    const bool more = it + 1 < iters;
    if (more) {
      I::fetch(kr, kb, E, D, key0 + NG * 64, Lk, tid);
      I::fetch(vr, vb, E, D, key0 + NG * 64, Lk, tid);
      if (tid < 64) br = key_bias(mb, key0 + NG * 64 + tid, Lk);
    }
'''
assert s.count(pro_old)==2   # fwd and dq share this text
s=s.replace(pro_old, pro_new)
# fwd epilogue: merge
s=rep(s,'''  if (live) {
    l = quad_sum(l);
    if (qi < Lq) {
      const float inv_l = 1.f / l;''','''  if constexpr (NG == 2) {
    constexpr int kX = 4 * NT + 2;
    float *xch = &KimgG[0][0][0][0];   // free after the loop's last barrier
    static_assert(sizeof(float) * 256 * kX <= sizeof(KimgG[0]), "exchange area");
    if (grp == 1 && live) {
      float *px = xch + (wave * 64 + lane) * kX;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) px[nt * 4 + i] = o[nt][i];
      px[4 * NT] = m;
      px[4 * NT + 1] = l;
    }
    __syncthreads();
    if (grp == 1) return;
    if (live) {
      const float *px = xch + (wave * 64 + lane) * kX;
      const float m1 = px[4 * NT], l1 = px[4 * NT + 1];
      const float m_new = fmaxf(m, m1);
      const bool dead = m_new == kNegInf;
      const float a0 = dead ? 1.f : __expf(m - m_new), a1 = dead ? 1.f : __expf(m1 - m_new);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) o[nt][i] = o[nt][i] * a0 + px[nt * 4 + i] * a1;
      l = l * a0 + l1 * a1;
      m = m_new;
    }
  }
  if (live) {
    l = quad_sum(l);
    if (qi < Lq) {
      const float inv_l = 1.f / l;''')

# ---------------- dq
s=rep(s,'''template <int NS, int NT>
__global__ __launch_bounds__(kAttnThreads) void attn_bwd_dq_kernel(''','''template <int NS, int NT, int NG>
__global__ __launch_bounds__(kAttnThreads * NG) void attn_bwd_dq_kernel(''')
s=rep(s,'''    float p_drop, uint32_t site, const uint64_t *__restrict__ rng_counter) {
  using I = Img<NS>;
  __shared__ __attribute__((aligned(16))) float Kimg[2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float Vimg[2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float Bias[2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const long E = (long)H * D;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const bool live = q0 < Lq;
  const float *qb = q + (long)b * Lq * E + h * D;
  const float *gb = dout + (long)b * Lq * E + h * D;''','''    float p_drop, uint32_t site, const uint64_t *__restrict__ rng_counter) {
  using I = Img<NS>;
  __shared__ __attribute__((aligned(16))) float KimgG[NG][2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float VimgG[NG][2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float BiasG[NG][2][64];
  const int grp = threadIdx.x / kAttnThreads;   // key-tile group, see attn_fwd_kernel
  float(*Kimg)[64][I::LD] = KimgG[grp];
  float(*Vimg)[64][I::LD] = VimgG[grp];
  float(*Bias)[64] = BiasG[grp];
  const int tid = threadIdx.x % kAttnThreads, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const long E = (long)H * D;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const bool live = q0 < Lq;
  const float *qb = q + (long)b * Lq * E + h * D;
  const float *gb = dout + (long)b * Lq * E + h * D;''')
s=rep(s,'''    if (fg == 0 && qi < Lq) delta[((long)b * H + h) * Lq + qi] = my_delta;''','''    if (grp == 0 && fg == 0 && qi < Lq) delta[((long)b * H + h) * Lq + qi] = my_delta;''')
s=rep(s,'''  if (live && qi < Lq) {
    float *ob = dq + ((long)b * Lq + qi) * ldo + h * D;''','''  if constexpr (NG == 2) {   // dQ is a plain sum over the key tiles: add the second group's share
    float *xch = &KimgG[0][0][0][0];
    static_assert(sizeof(float) * 256 * 4 * NT <= sizeof(KimgG[0]), "exchange area");
    if (grp == 1 && live) {
      float *px = xch + (wave * 64 + lane) * (4 * NT);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) px[nt * 4 + i] = acc[nt][i];
    }
    __syncthreads();
    if (grp == 1) return;
    if (live) {
      const float *px = xch + (wave * 64 + lane) * (4 * NT);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[nt][i] += px[nt * 4 + i];
    }
  }
  if (live && qi < Lq) {
    float *ob = dq + ((long)b * Lq + qi) * ldo + h * D;''')

# ---------------- dispatch
s=rep(s,'''#define ATTN_DISPATCH(KERNEL, grid, ...)                                                            \\
  do {                                                                                              \\
    if (D <= 16) hipLaunchKernelGGL((KERNEL<4, 1>), grid, dim3(kAttnThreads), 0, s, __VA_ARGS__);   \\
    else if (D <= 32) hipLaunchKernelGGL((KERNEL<8, 2>), grid, dim3(kAttnThreads), 0, s, __VA_ARGS__); \\
    else if (D <= 36) hipLaunchKernelGGL((KERNEL<9, 3>), grid, dim3(kAttnThreads), 0, s, __VA_ARGS__); \\
    else hipLaunchKernelGGL((KERNEL<12, 3>), grid, dim3(kAttnThreads), 0, s, __VA_ARGS__);          \\
  } while (0)''','''#define ATTN_DISPATCH(KERNEL, grid, ...)                                                            \\
  do {                                                                                              \\
    if (D <= 16) hipLaunchKernelGGL((KERNEL<4, 1>), grid, dim3(kAttnThreads), 0, s, __VA_ARGS__);   \\
    else if (D <= 32) hipLaunchKernelGGL((KERNEL<8, 2>), grid, dim3(kAttnThreads), 0, s, __VA_ARGS__); \\
    else if (D <= 36) hipLaunchKernelGGL((KERNEL<9, 3>), grid, dim3(kAttnThreads), 0, s, __VA_ARGS__); \\
    else hipLaunchKernelGGL((KERNEL<12, 3>), grid, dim3(kAttnThreads), 0, s, __VA_ARGS__);          \\
  } while (0)
// kernels with the key-group parameter: NG = 2 for grids that leave the SIMDs a single wave each
#define ATTN_DISPATCH_G(KERNEL, split, grid, ...)                                                   \\
  do {                                                                                              \\
    if (split) {                                                                                    \\
      const dim3 blk(kAttnThreads * 2);                                                             \\
      if (D <= 16) hipLaunchKernelGGL((KERNEL<4, 1, 2>), grid, blk, 0, s, __VA_ARGS__);             \\
      else if (D <= 32) hipLaunchKernelGGL((KERNEL<8, 2, 2>), grid, blk, 0, s, __VA_ARGS__);        \\
      else if (D <= 36) hipLaunchKernelGGL((KERNEL<9, 3, 2>), grid, blk, 0, s, __VA_ARGS__);        \\
      else hipLaunchKernelGGL((KERNEL<12, 3, 2>), grid, blk, 0, s, __VA_ARGS__);                    \\
    } else {                                                                                        \\
      const dim3 blk(kAttnThreads);                                                                 \\
      if (D <= 16) hipLaunchKernelGGL((KERNEL<4, 1, 1>), grid, blk, 0, s, __VA_ARGS__);             \\
      else if (D <= 32) hipLaunchKernelGGL((KERNEL<8, 2, 1>), grid, blk, 0, s, __VA_ARGS__);        \\
      else if (D <= 36) hipLaunchKernelGGL((KERNEL<9, 3, 1>), grid, blk, 0, s, __VA_ARGS__);        \\
      else hipLaunchKernelGGL((KERNEL<12, 3, 1>), grid, blk, 0, s, __VA_ARGS__);                    \\
    }                                                                                               \\
  } while (0)
static bool split_keys(const dim3 &g, int Lk) {
  static const int forced = getenv("BUTD_ATTN_SPLIT") ? atoi(getenv("BUTD_ATTN_SPLIT")) : -1;
  if (forced >= 0) return forced != 0 && Lk > 64;
  return (long)g.x * g.y * g.z <= 512 && Lk >= 128;
}''')
open(p,'w').write(s)
