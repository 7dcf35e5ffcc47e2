p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
a=s.index('// Staging of a (rows x 16) operand slab, split in two')
b=s.index('__global__ __launch_bounds__(kGemmThreads) void gemm_kernel(')
new=r'''// Staging of a (rows x 16) operand slab, split in two so the global loads of slab i+1 are in flight
// while the MFMAs of slab i run:  fetch_tile() only ISSUES loads (raw values of the operand and of its
// optional companion a2 land in registers, nothing consumes them), commit_tile() combines and writes
// the LDS image tile[row][k].   element(row, k) = src[row*ld_row + k*ld_k]; exactly one of the two
// strides is 1 and each thread moves the float4 that is contiguous in memory: 4 consecutive k of one
// row (contraction-contiguous operand) or 4 consecutive rows of one k (row-contiguous operand, which
// commit_tile transposes).  Rows >= nrows and k >= kend read as 0, except the virtual ones-row.
struct Frag4 {
  float4 a, a2;
};

struct TileIdx {
  int slow, fast;   // position along the strided / contiguous dimension inside the slab
  bool kc;          // contraction-contiguous?
};
__device__ inline TileIdx tile_idx(long ld_k, int tid) {
  TileIdx t;
  t.kc = ld_k == 1;
  t.slow = t.kc ? (tid >> 2) : (tid >> 4);
  t.fast = t.kc ? (tid & 3) * 4 : (tid & 15) * 4;
  return t;
}

__device__ inline Frag4 fetch_tile(const float *__restrict__ src, const float *__restrict__ src2,
                                   long ld_row, long ld_k, int row0, int nrows, int k0, int kend,
                                   int tid) {
  const TileIdx t = tile_idx(ld_k, tid);
  const long ld_slow = t.kc ? ld_row : ld_k;
  const int slow_g = (t.kc ? row0 : k0) + t.slow, fast_g = (t.kc ? k0 : row0) + t.fast;
  const int slow_lim = t.kc ? nrows : kend, fast_lim = t.kc ? kend : nrows;
  Frag4 f;
  f.a = make_float4(0.f, 0.f, 0.f, 0.f);
  f.a2 = f.a;
  if (slow_g < slow_lim && fast_g < fast_lim) {
    const long o = (long)slow_g * ld_slow + fast_g;
    const bool vec = (fast_g + 3 < fast_lim) && ((ld_slow & 3) == 0);
    if (vec && ((((uintptr_t)src) & 15) == 0)) {
      f.a = *reinterpret_cast<const float4 *>(src + o);
    } else {
      f.a.x = src[o];
      if (fast_g + 1 < fast_lim) f.a.y = src[o + 1];
      if (fast_g + 2 < fast_lim) f.a.z = src[o + 2];
      if (fast_g + 3 < fast_lim) f.a.w = src[o + 3];
    }
    if (src2) {
      if (vec && ((((uintptr_t)src2) & 15) == 0)) {
        f.a2 = *reinterpret_cast<const float4 *>(src2 + o);
      } else {
        f.a2.x = src2[o];
        if (fast_g + 1 < fast_lim) f.a2.y = src2[o + 1];
        if (fast_g + 2 < fast_lim) f.a2.z = src2[o + 2];
        if (fast_g + 3 < fast_lim) f.a2.w = src2[o + 3];
      }
    }
  }
  return f;
}

__device__ inline float combine(float a, float a2, int mode, float gate_scale) {
  return mode == 0 ? a + a2 : a * (a2 > 0.f ? gate_scale : 0.f);
}

__device__ inline void commit_tile(float (*tile)[kLd], const Frag4 &f, bool has2, int mode2,
                                   float scale2, long ld_k, int row0, int nrows, int k0, int kend,
                                   bool ones, int koff, int tid) {
  const TileIdx t = tile_idx(ld_k, tid);
  float v[4] = {f.a.x, f.a.y, f.a.z, f.a.w};
  if (has2) {
    v[0] = combine(v[0], f.a2.x, mode2, scale2); v[1] = combine(v[1], f.a2.y, mode2, scale2);
    v[2] = combine(v[2], f.a2.z, mode2, scale2); v[3] = combine(v[3], f.a2.w, mode2, scale2);
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
s=s[:a]+new+s[b:]
old=s[s.index('  FragSlab fa, fb;'):s.index('  fetch(kbeg);\n  commit(0);')]
new='''  Frag4 fa[kSub], fb[kSub];
  int kfetched = kbeg;
  auto fetch = [&](int k0) {
    kfetched = k0;
#pragma unroll
    for (int u = 0; u < kSub; ++u) {
      fa[u] = fetch_tile(P.a, P.a2, P.lda_m, P.lda_k, m0, P.M, k0 + u * 16, kend, tid);
      fb[u] = fetch_tile(P.b, nullptr, P.ldb_n, P.ldb_k, n0, P.N, k0 + u * 16, kend, tid);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int u = 0; u < kSub; ++u) {
      commit_tile(As[buf], fa[u], P.a2 != nullptr, P.a2_mode, P.a2_scale, P.lda_k, m0, P.M,
                  kfetched + u * 16, kend, false, u * 16, tid);
      commit_tile(Bs[buf], fb[u], false, 0, 0.f, P.ldb_k, n0, P.N, kfetched + u * 16, kend, ones,
                  u * 16, tid);
    }
  };
'''
s=s.replace(old,new)
s=s.replace('struct Frag4 { float v[4]; };\nstruct FragSlab { Frag4 s[kSub]; };\n','')
open(p,'w').write(s)
