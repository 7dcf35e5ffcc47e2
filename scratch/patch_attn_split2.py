def rep(s,a,b,cnt=1):
    assert s.count(a)==cnt, (s.count(a), a[:80])
    return s.replace(a,b)
p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
s=rep(s,'''template <int NS, int NT>
__global__ __launch_bounds__(kAttnThreads) void attn_bwd_dkv_kernel(''','''// NG = 2 (few key tiles: cross-attention to 80 tokens / 132 boxes is 128-192 workgroups): two wave
// groups walk the even / odd QUERY tiles and add their dK / dV shares through LDS at the end.
template <int NS, int NT, int NG>
__global__ __launch_bounds__(kAttnThreads * NG) void attn_bwd_dkv_kernel(''')
s=rep(s,'''  using I = Img<NS>;
  __shared__ __attribute__((aligned(16))) float Qimg[2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float Gimg[2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float Lse[2][64];
  __shared__ __attribute__((aligned(16))) float Del[2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const long E = (long)H * D;
  const int k0 = blockIdx.x * 64 + wave * 16;''','''  using I = Img<NS>;
  __shared__ __attribute__((aligned(16))) float QimgG[NG][2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float GimgG[NG][2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float LseG[NG][2][64];
  __shared__ __attribute__((aligned(16))) float DelG[NG][2][64];
  const int grp = threadIdx.x / kAttnThreads;
  float(*Qimg)[64][I::LD] = QimgG[grp];
  float(*Gimg)[64][I::LD] = GimgG[grp];
  float(*Lse)[64] = LseG[grp];
  float(*Del)[64] = DelG[grp];
  const int tid = threadIdx.x % kAttnThreads, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const long E = (long)H * D;
  const int k0 = blockIdx.x * 64 + wave * 16;''')
s=rep(s,'''  I::fetch(qr, qb, E, D, 0, Lq, tid);
  I::fetch(gr, gb, E, D, 0, Lq, tid);
  fetch_stats(0);
  I::commit(Qimg[0], qr, D, tid);
  I::commit(Gimg[0], gr, D, tid);
  commit_stats(0);
  __syncthreads();
  int cur = 0;
  for (int qs = 0; qs < Lq; qs += 64) {
    const bool more = qs + 64 < Lq;
    if (more) {
      I::fetch(qr, qb, E, D, qs + 64, Lq, tid);
      I::fetch(gr, gb, E, D, qs + 64, Lq, tid);
      fetch_stats(qs + 64);
    }''','''  // query tiles of this wave group: grp, grp + NG, ... (a tile past Lq stages zeros with lse = +inf:
  // every probability is 0, so both groups run the same number of iterations and barriers)
  const int iters = ((Lq + 63) / 64 + NG - 1) / NG;
  I::fetch(qr, qb, E, D, grp * 64, Lq, tid);
  I::fetch(gr, gb, E, D, grp * 64, Lq, tid);
  fetch_stats(grp * 64);
  I::commit(Qimg[0], qr, D, tid);
  I::commit(Gimg[0], gr, D, tid);
  commit_stats(0);
  __syncthreads();
  int cur = 0;
  for (int it = 0; it < iters; ++it) {
    const int qs = (it * NG + grp) * 64;
    const bool more = it + 1 < iters;
    if (more) {
      I::fetch(qr, qb, E, D, qs + NG * 64, Lq, tid);
      I::fetch(gr, gb, E, D, qs + NG * 64, Lq, tid);
      fetch_stats(qs + NG * 64);
    }''')
s=rep(s,'''  if (live && ki < Lk) {
    float *okp = dk + ((long)b * Lk + ki) * ldo + h * D;''','''  if constexpr (NG == 2) {
    float *xch = &QimgG[0][0][0][0];
    static_assert(sizeof(float) * 256 * 8 * NT <= sizeof(QimgG[0]), "exchange area");
    if (grp == 1 && live) {
      float *px = xch + (wave * 64 + lane) * (8 * NT);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          px[nt * 8 + i] = ak[nt][i];
          px[nt * 8 + 4 + i] = av[nt][i];
        }
    }
    __syncthreads();
    if (grp == 1) return;
    if (live) {
      const float *px = xch + (wave * 64 + lane) * (8 * NT);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ak[nt][i] += px[nt * 8 + i];
          av[nt][i] += px[nt * 8 + 4 + i];
        }
    }
  }
  if (live && ki < Lk) {
    float *okp = dk + ((long)b * Lk + ki) * ldo + h * D;''')
s=rep(s,'''  ATTN_DISPATCH(attn_bwd_dkv_kernel, gk, H, Lq, Lk, D,''','''  ATTN_DISPATCH_G(attn_bwd_dkv_kernel, split_keys(gk, Lq), gk, H, Lq, Lk, D,''')
open(p,'w').write(s)
