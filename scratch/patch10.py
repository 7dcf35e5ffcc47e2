p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
a=s.index('// ================================================================================================\n// Attention core')
b=s.index('extern "C" {\n\n#define ATTN_DISPATCH')
new = r'''// ================================================================================================
// Attention core (flash-style, fp32 MFMA 16x16x4, head_dim <= 48, head_dim % 4 == 0)
// ================================================================================================
// All score tiles are computed TRANSPOSED so that every per-query quantity (running max, running sum,
// rescale factor, delta) is lane-local:   S^T[key][q] = sum_d K[key][d] Q[q][d]   has C-layout
// lane (c = lane&15, g = lane>>4) -> S^T[key = 4g+i][q = c], i = 0..3, and that register quartet is
// exactly the B-operand fragment (k = 4g+s, col = q) of the next product  O^T[n][q] += V^T[n][key] P^T.
// The four lanes {c, c+16, c+32, c+48} that share a query combine their partial max / sum with
// v_permlane16_swap / v_permlane32_swap (VALU, no LDS).
//
// A workgroup = 4 waves x 16 queries.  The 64-key K and V tiles are staged once per workgroup into
// LDS (coalesced float4 global loads, double-buffered, one barrier per tile: tile i+1 is in flight in
// registers while tile i is multiplied) in a "fragment" image: row = key, element d stored at
// [d / NS][d % NS] with each of the four d-groups padded to 16 bytes, so an MFMA operand fragment
// (NS consecutive d of one group) is two ds_read_b128 + one ds_read_b32, and the row stride of 52
// floats keeps both access patterns (whole fragments for S^T, single elements for V^T / K^T operands)
// essentially bank-conflict free.
namespace {

constexpr int kAttnThreads = 256;
constexpr float kNegInf = -INFINITY;

__device__ inline float quad_max(float v) {  // over lanes c, c^16, c^32, c^48 ; result in all four
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ inline float quad_sum(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// additive score bias of key `kk`: 0 when it takes part, -inf when padded (mask byte != 0) or beyond Lk.
// Branch-free on purpose: selects on MFMA accumulators behind short-circuit branches were miscompiled
// into wrong results by hipcc 7.2 (see DESIGN.md); the mask byte is always loaded (index clamped).
__device__ inline float key_bias(const uint8_t *__restrict__ mb, int kk, int Lk) {
  const int kc = kk < Lk ? kk : Lk - 1;
  const unsigned mv = mb ? (unsigned)mb[kc] : 0u;
  return (kk < Lk && mv == 0u) ? 0.f : kNegInf;
}

// row `r` of a (rows x D) head slice in global memory, elements d = g*NS + s (zero outside)
template <int NS>
__device__ inline void load_row_frag(float (&f)[NS], const float *__restrict__ base, long stride, int r,
                                     int nrows, int g, int D) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int d = g * NS + s;
    f[s] = (r < nrows && d < D) ? base[(long)r * stride + d] : 0.f;
  }
}

// ---- LDS fragment image of a 64-row head slice --------------------------------------------------
template <int NS>
struct Img {
  static constexpr int NSP = (NS + 3) / 4 * 4;  // slots per d-group, 16-byte multiple
  static constexpr int LD = 4 * NSP + 4;        // row stride in floats (52 for head_dim 36)
  static constexpr int kVecPerThread = (64 * NS + kAttnThreads - 1) / kAttnThreads;  // float4 per thread
  struct Regs { float4 v[kVecPerThread]; };

  // global -> registers: rows [row0, row0+64) of a (L x D) slice with row stride E; D = 4*NS' <= 4*NS
  static __device__ inline void fetch(Regs &r, const float *__restrict__ base, long E, int D, int row0,
                                      int L, int tid) {
    const int vpr = D >> 2;  // float4 per row
#pragma unroll
    for (int j = 0; j < kVecPerThread; ++j) {
      const int f = tid + j * kAttnThreads;
      const int row = f / vpr, c4 = f - row * vpr;
      const int gr = row0 + row;
      r.v[j] = (row < 64 && gr < L) ? *reinterpret_cast<const float4 *>(base + (long)gr * E + c4 * 4)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // registers -> LDS image
  static __device__ inline void commit(float (*img)[LD], const Regs &r, int D, int tid) {
    const int vpr = D >> 2;
#pragma unroll
    for (int j = 0; j < kVecPerThread; ++j) {
      const int f = tid + j * kAttnThreads;
      const int row = f / vpr, c4 = f - row * vpr;
      if (row < 64) {
        const float e[4] = {r.v[j].x, r.v[j].y, r.v[j].z, r.v[j].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int d = c4 * 4 + i;
          const int g = d / NS, sl = d - g * NS;
          img[row][g * NSP + sl] = e[i];
        }
      }
    }
  }
  // operand fragment of row `row`, group g: d = g*NS + s, s = 0..NS-1
  static __device__ inline void frag(float (&f)[NS], const float (*img)[LD], int row, int g) {
    const float *p = &img[row][g * NSP];
#pragma unroll
    for (int q = 0; q + 4 <= NS; q += 4) {
      const float4 v = *reinterpret_cast<const float4 *>(p + q);
      f[q] = v.x; f[q + 1] = v.y; f[q + 2] = v.z; f[q + 3] = v.w;
    }
#pragma unroll
    for (int q = NS / 4 * 4; q < NS; ++q) f[q] = p[q];
  }
  // column offset of element d inside a row of the image
  static __device__ inline int col(int d) {
    const int g = d / NS;
    return g * NSP + (d - g * NS);
  }
};

template <int NS, int NT>
__global__ __launch_bounds__(kAttnThreads) void attn_fwd_kernel(
    int H, int Lq, int Lk, int D, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const uint8_t *__restrict__ mask, float *__restrict__ out,
    float *__restrict__ lse, float p_drop, uint32_t site, const uint64_t *__restrict__ rng_counter) {
  using I = Img<NS>;
  __shared__ __attribute__((aligned(16))) float Kimg[2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float Vimg[2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float Bias[2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const long E = (long)H * D;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const bool live = q0 < Lq;  // whole wave beyond Lq: still stages tiles and hits the barriers
  const float *qb = q + (long)b * Lq * E + h * D;
  const float *kb = k + (long)b * Lk * E + h * D;
  const float *vb = v + (long)b * Lk * E + h * D;
  const uint8_t *mb = mask ? mask + (long)b * Lk : nullptr;
  const bool drop = p_drop > 0.f;
  const float inv_keep = drop ? 1.f / (1.f - p_drop) : 1.f;
  const uint64_t ctr = (drop && rng_counter) ? *rng_counter : 0ull;
  const int qi = q0 + fr;  // this lane's query (column of every transposed tile)

  float qf[NS];
  load_row_frag<NS>(qf, qb, E, qi, Lq, fg, D);
  int vcol[NT];
  bool vok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = nt * 16 + fr;
    vok[nt] = n < D;
    vcol[nt] = I::col(vok[nt] ? n : 0);
  }
  float m = kNegInf, l = 0.f;
  f32x4 o[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  typename I::Regs kr, vr;
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
    if (live) {
      f32x4 st[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float kf[NS];
        I::frag(kf, Kimg[cur], t * 16 + fr, fg);
        st[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s)
          st[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s], qf[s], st[t], 0, 0, 0);
      }
      float tmax = kNegInf;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 bb = *reinterpret_cast<const float4 *>(&Bias[cur][t * 16 + fg * 4]);
        st[t][0] += bb.x; st[t][1] += bb.y; st[t][2] += bb.z; st[t][3] += bb.w;
        tmax = fmaxf(fmaxf(fmaxf(tmax, st[t][0]), fmaxf(st[t][1], st[t][2])), st[t][3]);
      }
      tmax = quad_max(tmax);
      const float m_new = fmaxf(m, tmax);
      const bool dead = m_new == kNegInf;  // nothing but masked keys so far
      const float alpha = dead ? 1.f : __expf(m - m_new);
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p = dead ? 0.f : __expf(st[t][i] - m_new);
          psum += p;
          float pd = p;
          if (drop) {
            const int kk = key0 + t * 16 + fg * 4 + i;
            const uint32_t idx = (uint32_t)((((long)b * H + h) * Lq + qi) * Lk + kk);
            pd = rng::keep(ctr, site, idx, p_drop) ? p * inv_keep : 0.f;
          }
          st[t][i] = pd;
        }
      l = l * alpha + psum;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) o[nt] *= alpha;
      // O^T[n][q] += V^T[n][key] P^T[key][q]
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float *vrow = Vimg[cur][t * 16 + fg * 4 + s];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float a = vok[nt] ? vrow[vcol[nt]] : 0.f;
            o[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, st[t][s], o[nt], 0, 0, 0);
          }
        }
      m = m_new;
    }
    if (more) {
      I::commit(Kimg[cur ^ 1], kr, D, tid);
      I::commit(Vimg[cur ^ 1], vr, D, tid);
      if (tid < 64) Bias[cur ^ 1][tid] = br;
    }
    __syncthreads();
    cur ^= 1;
  }
  if (live) {
    l = quad_sum(l);
    if (qi < Lq) {
      const float inv_l = 1.f / l;  // l == 0 (every key masked) -> inf * 0 = NaN like torch's softmax
      float *ob = out + ((long)b * Lq + qi) * E + h * D;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int n = nt * 16 + fg * 4 + i;
          if (n < D) ob[n] = o[nt][i] * inv_l;
        }
      if (fg == 0) lse[((long)b * H + h) * Lq + qi] = m + __logf(l);
    }
  }
}

// delta[b,h,q] = sum_n dO[q][n] * O[q][n]
__global__ __launch_bounds__(256) void attn_delta_kernel(int H, int Lq, int D, long total,
                                                         const float *__restrict__ out,
                                                         const float *__restrict__ dout,
                                                         float *__restrict__ delta) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over (b, q, h)
  if (i >= total) return;
  const int h = (int)(i % H);
  const long bq = i / H;
  const int qq = (int)(bq % Lq);
  const long b = bq / Lq;
  const float *o = out + bq * (long)H * D + (long)h * D;
  const float *g = dout + bq * (long)H * D + (long)h * D;
  float s = 0.f;
  for (int n = 0; n < D; ++n) s += o[n] * g[n];
  delta[(b * H + h) * Lq + qq] = s;
}

// dQ: same walk as the forward with K and V swapping roles.
template <int NS, int NT>
__global__ __launch_bounds__(kAttnThreads) void attn_bwd_dq_kernel(
    int H, int Lq, int Lk, int D, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const uint8_t *__restrict__ mask, const float *__restrict__ dout,
    const float *__restrict__ lse, const float *__restrict__ delta, float *__restrict__ dq,
    float p_drop, uint32_t site, const uint64_t *__restrict__ rng_counter) {
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
  const float *gb = dout + (long)b * Lq * E + h * D;
  const float *kb = k + (long)b * Lk * E + h * D;
  const float *vb = v + (long)b * Lk * E + h * D;
  const uint8_t *mb = mask ? mask + (long)b * Lk : nullptr;
  const bool drop = p_drop > 0.f;
  const float inv_keep = drop ? 1.f / (1.f - p_drop) : 1.f;
  const uint64_t ctr = (drop && rng_counter) ? *rng_counter : 0ull;
  const int qi = q0 + fr;

  float qf[NS], gf[NS];
  load_row_frag<NS>(qf, qb, E, qi, Lq, fg, D);
  load_row_frag<NS>(gf, gb, E, qi, Lq, fg, D);
  // rows beyond Lq: lse = +inf makes every probability exp(s - inf) = 0
  const float my_lse = qi < Lq ? lse[((long)b * H + h) * Lq + qi] : INFINITY;
  const float my_delta = qi < Lq ? delta[((long)b * H + h) * Lq + qi] : 0.f;
  int kcol[NT];
  bool kok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int d = nt * 16 + fr;
    kok[nt] = d < D;
    kcol[nt] = I::col(kok[nt] ? d : 0);
  }
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  typename I::Regs kr, vr;
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
    if (live) {
      f32x4 ds[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float kf[NS], vf[NS];
        I::frag(kf, Kimg[cur], t * 16 + fr, fg);
        I::frag(vf, Vimg[cur], t * 16 + fr, fg);
        f32x4 st = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s], qf[s], st, 0, 0, 0);  // S^T
          dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[s], gf[s], dp, 0, 0, 0);  // dP^T
        }
        const float4 bb = *reinterpret_cast<const float4 *>(&Bias[cur][t * 16 + fg * 4]);
        const float bias4[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p = __expf(st[i] + bias4[i] - my_lse);
          float dpe = dp[i];
          if (drop) {
            const int kk = key0 + t * 16 + fg * 4 + i;
            const uint32_t idx = (uint32_t)((((long)b * H + h) * Lq + qi) * Lk + kk);
            dpe = rng::keep(ctr, site, idx, p_drop) ? dpe * inv_keep : 0.f;
          }
          ds[t][i] = p * (dpe - my_delta);
        }
      }
      // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float *krow = Kimg[cur][t * 16 + fg * 4 + s];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float a = kok[nt] ? krow[kcol[nt]] : 0.f;
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ds[t][s], acc[nt], 0, 0, 0);
          }
        }
    }
    if (more) {
      I::commit(Kimg[cur ^ 1], kr, D, tid);
      I::commit(Vimg[cur ^ 1], vr, D, tid);
      if (tid < 64) Bias[cur ^ 1][tid] = br;
    }
    __syncthreads();
    cur ^= 1;
  }
  if (live && qi < Lq) {
    float *ob = dq + ((long)b * Lq + qi) * E + h * D;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int d = nt * 16 + fg * 4 + i;
        if (d < D) ob[d] = acc[nt][i];
      }
  }
}

// dK, dV: a workgroup owns 64 keys (a wave 16 of them, as columns) and walks the queries 64 at a time
// through LDS images of Q and dO; tiles are NOT transposed here:  S[q][key] has C-layout
// lane (c = key, g) -> q = 4g+i, which is the B fragment of
// dV^T[n][key] += dO^T[n][q] P[q][key]  and  dK^T[d][key] += Q^T[d][q] dS[q][key].
template <int NS, int NT>
__global__ __launch_bounds__(kAttnThreads) void attn_bwd_dkv_kernel(
    int H, int Lq, int Lk, int D, const float *__restrict__ q, const float *__restrict__ k,
    const float *__restrict__ v, const uint8_t *__restrict__ mask, const float *__restrict__ dout,
    const float *__restrict__ lse, const float *__restrict__ delta, float *__restrict__ dk,
    float *__restrict__ dv, float p_drop, uint32_t site, const uint64_t *__restrict__ rng_counter) {
  using I = Img<NS>;
  __shared__ __attribute__((aligned(16))) float Qimg[2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float Gimg[2][64][I::LD];
  __shared__ __attribute__((aligned(16))) float Lse[2][64];
  __shared__ __attribute__((aligned(16))) float Del[2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const long E = (long)H * D;
  const int k0 = blockIdx.x * 64 + wave * 16;
  const bool live = k0 < Lk;
  const float *qb = q + (long)b * Lq * E + h * D;
  const float *gb = dout + (long)b * Lq * E + h * D;
  const float *kb = k + (long)b * Lk * E + h * D;
  const float *vb = v + (long)b * Lk * E + h * D;
  const float *lb = lse + ((long)b * H + h) * Lq;
  const float *db = delta + ((long)b * H + h) * Lq;
  const bool drop = p_drop > 0.f;
  const float inv_keep = drop ? 1.f / (1.f - p_drop) : 1.f;
  const uint64_t ctr = (drop && rng_counter) ? *rng_counter : 0ull;
  const int ki = k0 + fr;  // this lane's key (column)
  const float my_bias = key_bias(mask ? mask + (long)b * Lk : nullptr, ki, Lk);

  float kf[NS], vf[NS];
  load_row_frag<NS>(kf, kb, E, ki, Lk, fg, D);
  load_row_frag<NS>(vf, vb, E, ki, Lk, fg, D);
  int ncol[NT];
  bool nok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = nt * 16 + fr;
    nok[nt] = n < D;
    ncol[nt] = I::col(nok[nt] ? n : 0);
  }
  f32x4 ak[NT], av[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    ak[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    av[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  typename I::Regs qr, gr;
  float sr = 0.f;  // threads 0..63: lse of row tid ; threads 64..127: delta of row tid-64
  auto fetch_stats = [&](int qs) {
    if (tid < 64) sr = (qs + tid < Lq) ? lb[qs + tid] : INFINITY;       // +inf -> probability 0
    else if (tid < 128) sr = (qs + tid - 64 < Lq) ? db[qs + tid - 64] : 0.f;
  };
  auto commit_stats = [&](int buf) {
    if (tid < 64) Lse[buf][tid] = sr;
    else if (tid < 128) Del[buf][tid - 64] = sr;
  };
  I::fetch(qr, qb, E, D, 0, Lq, tid);
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
    }
    if (live) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float qf[NS], gf[NS];
        I::frag(qf, Qimg[cur], t * 16 + fr, fg);
        I::frag(gf, Gimg[cur], t * 16 + fr, fg);
        f32x4 st = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          st = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[s], kf[s], st, 0, 0, 0);  // S[q][key]
          dp = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[s], vf[s], dp, 0, 0, 0);  // dP[q][key]
        }
        const float4 l4 = *reinterpret_cast<const float4 *>(&Lse[cur][t * 16 + fg * 4]);
        const float4 d4 = *reinterpret_cast<const float4 *>(&Del[cur][t * 16 + fg * 4]);
        const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq4[4] = {d4.x, d4.y, d4.z, d4.w};
        f32x4 pd, ds;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p = __expf(st[i] + my_bias - lq[i]);
          float keepf = 1.f;
          if (drop) {
            const int qq = qs + t * 16 + fg * 4 + i;
            const uint32_t idx = (uint32_t)((((long)b * H + h) * Lq + qq) * Lk + ki);
            keepf = rng::keep(ctr, site, idx, p_drop) ? inv_keep : 0.f;
          }
          pd[i] = p * keepf;
          ds[i] = p * (dp[i] * keepf - dq4[i]);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float *grow = Gimg[cur][t * 16 + fg * 4 + s];
          const float *qrow = Qimg[cur][t * 16 + fg * 4 + s];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float ag = nok[nt] ? grow[ncol[nt]] : 0.f;
            const float aq = nok[nt] ? qrow[ncol[nt]] : 0.f;
            av[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ag, pd[s], av[nt], 0, 0, 0);
            ak[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, ds[s], ak[nt], 0, 0, 0);
          }
        }
      }
    }
    if (more) {
      I::commit(Qimg[cur ^ 1], qr, D, tid);
      I::commit(Gimg[cur ^ 1], gr, D, tid);
      commit_stats(cur ^ 1);
    }
    __syncthreads();
    cur ^= 1;
  }
  if (live && ki < Lk) {
    float *okp = dk + ((long)b * Lk + ki) * E + h * D;
    float *ovp = dv + ((long)b * Lk + ki) * E + h * D;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = nt * 16 + fg * 4 + i;
        if (n < D) {
          okp[n] = ak[nt][i];
          ovp[n] = av[nt][i];
        }
      }
  }
}

}  // namespace

'''
s=s[:a]+new+s[b:]
s=s.replace('''  if (D <= 0 || D > 48 || Lk <= 0) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((Lq + 63) / 64, H, B);''','''  if (D <= 0 || D > 48 || (D & 3) || Lk <= 0) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((Lq + 63) / 64, H, B);''')
s=s.replace('''  if (D <= 0 || D > 48 || Lk <= 0) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)B * Lq * H;''','''  if (D <= 0 || D > 48 || (D & 3) || Lk <= 0) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)B * Lq * H;''')
open(p,'w').write(s)
