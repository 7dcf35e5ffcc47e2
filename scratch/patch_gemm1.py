import re
p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
# 1. global-address-space float4 load helper
s=s.replace('''struct GemmBatch {''','''// Loads that must be emitted as global_load_*: a FLAT load also counts against lgkmcnt, so the
// s_waitcnt lgkmcnt(0) in front of the MFMAs (for the LDS fragment reads) would wait for the prefetch
// of the NEXT slab as well and serialize HBM latency with the matrix pipe.
typedef const __attribute__((address_space(1))) float4 *global_f4_ptr;
__device__ inline float4 ldg4(const float *p) {
  return *reinterpret_cast<global_f4_ptr>(reinterpret_cast<uintptr_t>(p));
}
constexpr int kAffK = 256;  // contraction range whose A-operand affine is staged in LDS (fast path)

struct GemmBatch {''')
s=s.replace('''  __shared__ __attribute__((aligned(16))) float Bs[2][kBN][kLd];
''','''  __shared__ __attribute__((aligned(16))) float Bs[2][kBN][kLd];
  __shared__ __attribute__((aligned(16))) float Asc[kAffK], Ash[kAffK];
''')
s=s.replace('''      ((((uintptr_t)P.a) | ((uintptr_t)P.b)) & 15) == 0 && (kbeg & 3) == 0;''','''      ((((uintptr_t)P.a) | ((uintptr_t)P.b)) & 15) == 0 && (kbeg & 3) == 0 &&
      (P.a_chan_scale == nullptr || kend - kbeg <= kAffK);''')
old=s[s.index('    const long sa16 = a_kc ? 16 : 16 * P.lda_k'):s.index('    auto put = [&]')]
new='''    // rows of this thread inside the matrix?  (loop-invariant; rows outside a partial tile read 0:
    // their loads are redirected to the operand base with stride 0 and discarded at commit time, so
    // every load stays an unconditional global_load)
    const bool a_ok = m0 + (a_kc ? a_slow : a_fast) < P.M;
    const bool b_ok = n0 + (b_kc ? b_slow : b_fast) < P.N;
    const long sa16 = !a_ok ? 0 : (a_kc ? 16 : 16 * P.lda_k);   // per 16 k
    const long sb16 = !b_ok ? 0 : (b_kc ? 16 : 16 * P.ldb_k);
    if (!a_ok) pa = P.a;
    if (!b_ok) pb = P.b;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool a_aff = P.a_chan_scale != nullptr;   // channel = k (varies per slab): staged in LDS
    if (a_aff) {
      for (int k = tid; k < kend - kbeg; k += kGemmThreads) {
        Asc[k] = P.a_chan_scale[kbeg + k];
        Ash[k] = P.a_chan_shift[kbeg + k];
      }
    }
    float4 bsc4 = make_float4(1.f, 1.f, 1.f, 1.f), bsh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool b_aff = P.b_chan_scale != nullptr;
    if (b_aff && !b_kc) {  // channel = B row = 4 consecutive rows of this thread: loop-invariant
      if (b_ok) {
        bsc4 = *reinterpret_cast<const float4 *>(P.b_chan_scale + n0 + b_fast);
        bsh4 = *reinterpret_cast<const float4 *>(P.b_chan_shift + n0 + b_fast);
      }
    } else if (b_aff && b_ok) {
      const float sc = P.b_chan_scale[n0 + b_slow], sh = P.b_chan_shift[n0 + b_slow];
      bsc4 = make_float4(sc, sc, sc, sc);
      bsh4 = make_float4(sh, sh, sh, sh);
    }
    if (a_aff) __syncthreads();
    float4 ra[kSub], rb[kSub];
    int kslab0 = 0;   // k offset (relative to kbeg) of the slab held in ra/rb
    auto fetch_fast = [&](int slab) {
      kslab0 = slab * kBK;
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        ra[u] = ldg4(pa + (long)(slab * kSub + u) * sa16);
        rb[u] = ldg4(pb + (long)(slab * kSub + u) * sb16);
      }
    };
'''
s=s.replace(old,new)
# remove the old a_ok/b_ok/zero4 defs that preceded (they were inside 'old')? check duplicates later
old2=s[s.index('    auto commit_fast = [&](int buf) {'):s.index('    const int nslab = (kend - kbeg) / kBK;')]
new2='''    auto commit_fast = [&](int buf) {
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        float4 va = a_ok ? ra[u] : zero4, vb = b_ok ? rb[u] : zero4;
        if (a_aff && a_ok) {
          float4 sc, sh;
          if (a_kc) {
            sc = *reinterpret_cast<const float4 *>(&Asc[kslab0 + u * 16 + a_fast]);
            sh = *reinterpret_cast<const float4 *>(&Ash[kslab0 + u * 16 + a_fast]);
          } else {
            const float s1 = Asc[kslab0 + u * 16 + a_slow], h1 = Ash[kslab0 + u * 16 + a_slow];
            sc = make_float4(s1, s1, s1, s1);
            sh = make_float4(h1, h1, h1, h1);
          }
          va.x = fmaxf(va.x * sc.x + sh.x, 0.f); va.y = fmaxf(va.y * sc.y + sh.y, 0.f);
          va.z = fmaxf(va.z * sc.z + sh.z, 0.f); va.w = fmaxf(va.w * sc.w + sh.w, 0.f);
        }
        if (b_aff && b_ok) {
          vb.x = fmaxf(vb.x * bsc4.x + bsh4.x, 0.f); vb.y = fmaxf(vb.y * bsc4.y + bsh4.y, 0.f);
          vb.z = fmaxf(vb.z * bsc4.z + bsh4.z, 0.f); vb.w = fmaxf(vb.w * bsc4.w + bsh4.w, 0.f);
        }
        put(As[buf], a_kc, a_slow, a_fast, u * 16, va);
        put(Bs[buf], b_kc, b_slow, b_fast, u * 16, vb);
      }
    };
'''
s=s.replace(old2,new2)
open(p,'w').write(s)
