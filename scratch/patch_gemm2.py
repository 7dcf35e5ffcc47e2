p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
s=s.replace('constexpr int kMaxProblems = 4;','constexpr int kMaxProblems = 8;')
s=s.replace('constexpr int kAffK = 256;','constexpr int kAffK = 320;')

# ---- generic path: operand effects struct
old=s[s.index('// chan_is_k: the affine\'s channel index'):s.index('__global__ __launch_bounds__(kGemmThreads) void gemm_kernel')]
new='''// Everything commit_tile() applies to a staged operand besides the plain copy.
// chan_is_k: the affine's channel index is the contraction index (A operand) or the row index (B)
struct OperandFx {
  bool has2; int mode2; float scale2;          // companion operand a2
  const float *csc, *csh; bool chan_is_k;      // per-channel affine + ReLU
  float drop_p, drop_inv; uint32_t drop_key;   // dropout keyed by the element's memory offset
  long ld_row;
};

__device__ inline void commit_tile(float (*tile)[kLd], const Frag4 &f, const OperandFx &fx, long ld_k,
                                   int row0, int nrows, int k0, int kend, bool ones, int koff, int tid) {
  const TileIdx t = tile_idx(ld_k, tid);
  float v[4] = {f.a.x, f.a.y, f.a.z, f.a.w};
  if (fx.has2) {
    v[0] = combine(v[0], f.a2.x, fx.mode2, fx.scale2); v[1] = combine(v[1], f.a2.y, fx.mode2, fx.scale2);
    v[2] = combine(v[2], f.a2.z, fx.mode2, fx.scale2); v[3] = combine(v[3], f.a2.w, fx.mode2, fx.scale2);
  }
  if (fx.csc || fx.drop_p > 0.f) {  // relu(v * scale[chan] + shift[chan]), dropout; out-of-range stays 0
    const int rbase = row0 + (t.kc ? t.slow : t.fast), kbase = k0 + (t.kc ? t.fast : t.slow);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = rbase + (t.kc ? 0 : i), k = kbase + (t.kc ? i : 0);
      if (r < nrows && k < kend) {
        if (fx.csc) {
          const int ch = fx.chan_is_k ? k : r;
          v[i] = fmaxf(v[i] * fx.csc[ch] + fx.csh[ch], 0.f);
        }
        if (fx.drop_p > 0.f) {
          const uint32_t off = (uint32_t)((long)r * fx.ld_row + (long)k * ld_k);
          v[i] = rng::keep_keyed(fx.drop_key, off, fx.drop_p) ? v[i] * fx.drop_inv : 0.f;
        }
      }
    }
  }
  if (t.kc) {
    if (ones && row0 + t.slow == nrows) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = (k0 + t.fast + i < kend) ? 1.f : 0.f;
    }
    *reinterpret_cast<float4 *>(&tile[t.slow][koff + t.fast]) = make_float4(v[0], v[1], v[2], v[3]);
  } else {  // transpose into the K-contiguous LDS image
    if (ones && k0 + t.slow < kend) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (row0 + t.fast + i == nrows) v[i] = 1.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[t.fast + i][koff + t.slow] = v[i];
  }
}

'''
s=s.replace(old,new)

# ---- kernel: rng keys up front
s=s.replace('''  const bool ones = P.ones_col != 0;
  auto mfma_slab''','''  const bool ones = P.ones_col != 0;
  // operand dropout: the (step, site) halves of the hash are kernel-invariant
  const bool a_dropout = P.a_drop_p > 0.f, b_dropout = P.b_drop_p > 0.f;
  const uint64_t step_ctr = ((a_dropout || b_dropout || P.dropout_p > 0.f) && rng_counter) ? *rng_counter : 0ull;
  const uint32_t a_key = rng::site_key(step_ctr, P.a_drop_site), b_key = rng::site_key(step_ctr, P.b_drop_site);
  const float a_inv = a_dropout ? 1.f / (1.f - P.a_drop_p) : 1.f, b_inv = b_dropout ? 1.f / (1.f - P.b_drop_p) : 1.f;
  auto mfma_slab''')

# ---- fast path commit: dropout by memory offset
s=s.replace('''    float4 ra[kSub], rb[kSub];
    int kslab0 = 0;   // k offset (relative to kbeg) of the slab held in ra/rb
    auto fetch_fast = [&](int slab) {
      kslab0 = slab * kBK;''','''    float4 ra[kSub], rb[kSub];
    int kslab0 = 0;   // k offset (relative to kbeg) of the slab held in ra/rb
    const long offa0 = pa - P.a, offb0 = pb - P.b;   // element offsets of this thread's first float4
    auto drop4 = [](float4 v, uint32_t key, uint32_t off, float p, float inv) {
      v.x = rng::keep_keyed(key, off + 0, p) ? v.x * inv : 0.f;
      v.y = rng::keep_keyed(key, off + 1, p) ? v.y * inv : 0.f;
      v.z = rng::keep_keyed(key, off + 2, p) ? v.z * inv : 0.f;
      v.w = rng::keep_keyed(key, off + 3, p) ? v.w * inv : 0.f;
      return v;
    };
    auto fetch_fast = [&](int slab) {
      kslab0 = slab * kBK;''')
s=s.replace('''        put(As[buf], a_kc, a_slow, a_fast, u * 16, va);
        put(Bs[buf], b_kc, b_slow, b_fast, u * 16, vb);''','''        if (a_dropout && a_ok)
          va = drop4(va, a_key, (uint32_t)(offa0 + (long)(kslab0 / 16 + u) * sa16), P.a_drop_p, a_inv);
        if (b_dropout && b_ok)
          vb = drop4(vb, b_key, (uint32_t)(offb0 + (long)(kslab0 / 16 + u) * sb16), P.b_drop_p, b_inv);
        put(As[buf], a_kc, a_slow, a_fast, u * 16, va);
        put(Bs[buf], b_kc, b_slow, b_fast, u * 16, vb);''')

# ---- generic path call sites
s=s.replace('''    Frag4 fa[kSub], fb[kSub];
    int kfetched = kbeg;''','''    Frag4 fa[kSub], fb[kSub];
    int kfetched = kbeg;
    const OperandFx fxa = {P.a2 != nullptr, P.a2_mode, P.a2_scale, P.a_chan_scale, P.a_chan_shift, true,
                           P.a_drop_p, a_inv, a_key, P.lda_m};
    const OperandFx fxb = {false, 0, 0.f, P.b_chan_scale, P.b_chan_shift, false,
                           P.b_drop_p, b_inv, b_key, P.ldb_n};''')
s=s.replace('''        commit_tile(As[buf], fa[u], P.a2 != nullptr, P.a2_mode, P.a2_scale, P.a_chan_scale,
                    P.a_chan_shift, true, P.lda_k, m0, P.M, kfetched + u * 16, kend, false, u * 16, tid);
        commit_tile(Bs[buf], fb[u], false, 0, 0.f, P.b_chan_scale, P.b_chan_shift, false, P.ldb_k, n0,
                    P.N, kfetched + u * 16, kend, ones, u * 16, tid);''','''        commit_tile(As[buf], fa[u], fxa, P.lda_k, m0, P.M, kfetched + u * 16, kend, false, u * 16, tid);
        commit_tile(Bs[buf], fb[u], fxb, P.ldb_k, n0, P.N, kfetched + u * 16, kend, ones, u * 16, tid);''')

# ---- epilogue
s=s.replace('''  const uint64_t ctr = (drop && rng_counter) ? *rng_counter : 0ull;''','''  const uint64_t ctr = step_ctr;''')
old=s[s.index('''    const bool vec_ok = (n + 3 < pN) && ((ldc & 3) == 0) && ((((uintptr_t)cptr) & 15) == 0);'''):s.index('''  // accumulate / bias-gradient path''')]
new='''    const bool vec_ok = (n + 3 < pN) && ((ldc & 3) == 0) && ((((uintptr_t)cptr) & 15) == 0);
    double *const col_sum = P.col_sum, *const col_sumsq = P.col_sumsq;
    float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const int row = (tid >> 4) + qq * 16;
      const int m = m0 + row;
      if (m >= pM || n >= pN) continue;
      const float4 cv = *reinterpret_cast<const float4 *>(&Cs[row][c4]);
      float v[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = (v[e] + bv[e]) * scale;
        if (relu) v[e] = fmaxf(v[e], 0.f);
        if (drop)
          v[e] = rng::keep(ctr, site, (uint32_t)((long)m * pN + n + e), p_drop) ? v[e] * inv_keep : 0.f;
        if (n + e < pN) {
          cs[e] += v[e];
          cq[e] += v[e] * v[e];
        }
      }
      float *dst = cptr + (long)m * ldc + n;
      if (vec_ok) {
        *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < pN) dst[e] = v[e];
      }
    }
    if (col_sum) {
      // column sums of the tile: 16 row-phase partials per column through LDS (the B buffers are free
      // after the last barrier of the K loop), then one double atomic per column and statistic
      float *red = &Bs[0][0][0];
      static_assert(sizeof(Bs) >= sizeof(float) * 2 * 16 * kBN, "statistics scratch must fit the B buffers");
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
        for (int r = 0; r < 16; ++r) acc += (double)red[(which * 16 + r) * kBN + col];
        if (n0 + col < pN) atomicAdd((which ? col_sumsq : col_sum) + n0 + col, acc);
      }
    }
    return;
  }
'''
s=s.replace(old,new)
open(p,'w').write(s)
