p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
a=s.index('    float4 ra[kSub], rb[kSub];\n    int kslab0 = 0;')
b=s.index('  } else {\n    // streaming: double-buffered LDS, one barrier per slab')
new='''    // Two register sets: the loads of slab i+2 are issued before slab i is multiplied, so a load has two
    // MFMA phases (not one) to come back -- with a single set the loop ran at one L2 round trip per slab.
    float4 ra0[kSub], rb0[kSub], ra1[kSub], rb1[kSub];
    const long offa0 = pa - P.a, offb0 = pb - P.b;   // element offsets of this thread's first float4
    auto drop4 = [](float4 v, uint32_t key, uint32_t off, float p, float inv) {
      v.x = rng::keep_keyed(key, off + 0, p) ? v.x * inv : 0.f;
      v.y = rng::keep_keyed(key, off + 1, p) ? v.y * inv : 0.f;
      v.z = rng::keep_keyed(key, off + 2, p) ? v.z * inv : 0.f;
      v.w = rng::keep_keyed(key, off + 3, p) ? v.w * inv : 0.f;
      return v;
    };
    auto fetch_fast = [&](int slab, float4 (&ra)[kSub], float4 (&rb)[kSub]) {
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        ra[u] = ldg4(pa + (long)(slab * kSub + u) * sa16);
        rb[u] = ldg4(pb + (long)(slab * kSub + u) * sb16);
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
    auto commit_fast = [&](int slab, int buf, const float4 (&ra)[kSub], const float4 (&rb)[kSub]) {
      const int kslab0 = slab * kBK;   // k offset (relative to kbeg) of the slab held in ra/rb
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        float4 va = a_ok ? ra[u] : zero4, vb = b_ok ? rb[u] : zero4;
        if (a_aff && a_ok) {
          float4 sc, sh;
          if (a_kc) {
            sc = *reinterpret_cast<const float4 *>(&Asc[kslab0 + u * kSW + a_fast]);
            sh = *reinterpret_cast<const float4 *>(&Ash[kslab0 + u * kSW + a_fast]);
          } else {
            const float s1 = Asc[kslab0 + u * kSW + a_slow], h1 = Ash[kslab0 + u * kSW + a_slow];
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
        if (a_dropout && a_ok)
          va = drop4(va, a_key, (uint32_t)(offa0 + (long)(slab * kSub + u) * sa16), P.a_drop_p, a_inv);
        if (b_dropout && b_ok)
          vb = drop4(vb, b_key, (uint32_t)(offb0 + (long)(slab * kSub + u) * sb16), P.b_drop_p, b_inv);
        put(As[buf], a_kc, a_slow, a_fast, u * kSW, va);
        put(Bs[buf], b_kc, b_slow, b_fast, u * kSW, vb);
      }
    };
    const int nslab = (kend - kbeg) / kBK;
    fetch_fast(0, ra0, rb0);
    if (nslab > 1) fetch_fast(1, ra1, rb1);
    commit_fast(0, 0, ra0, rb0);
    __syncthreads();
    for (int sl = 0; sl < nslab; sl += 2) {
      // even slab sl sits in LDS buffer 0, set 1 holds slab sl+1 (in flight), set 0 is free
      if (sl + 2 < nslab) fetch_fast(sl + 2, ra0, rb0);
      mfma_slab(0);
      if (sl + 1 < nslab) commit_fast(sl + 1, 1, ra1, rb1);
      __syncthreads();
      if (sl + 1 >= nslab) break;
      if (sl + 3 < nslab) fetch_fast(sl + 3, ra1, rb1);
      mfma_slab(1);
      if (sl + 2 < nslab) commit_fast(sl + 2, 0, ra0, rb0);
      __syncthreads();
    }
'''
s=s[:a]+new+s[b:]
open(p,'w').write(s)
