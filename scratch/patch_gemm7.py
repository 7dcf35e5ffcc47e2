p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
# remove async helpers
a=s.index('// Asynchronous variant for the software-pipelined K loop')
b=s.index('constexpr int kAffK = 320;')
s=s[:a]+s[b:]
# remove pin asm
a=s.index('    // the loop below runs with loads in flight under manual wait counts')
b=s.index('    if (a_aff) __syncthreads();\n    // virtual ones-row of B')
s=s[:a]+s[b:]
a=s.index('    // Two register sets: the loads of slab i+2')
b=s.index('    auto put = [&](float (*tile)[kLd], bool kc, int slow, int fst, int koff, float4 v) {')
s=s[:a]+'''    float4 ra[kSub], rb[kSub];
    const long offa0 = pa - P.a, offb0 = pb - P.b;   // element offsets of this thread's first float4
    auto drop4 = [](float4 v, uint32_t key, uint32_t off, float p, float inv) {
      v.x = rng::keep_keyed(key, off + 0, p) ? v.x * inv : 0.f;
      v.y = rng::keep_keyed(key, off + 1, p) ? v.y * inv : 0.f;
      v.z = rng::keep_keyed(key, off + 2, p) ? v.z * inv : 0.f;
      v.w = rng::keep_keyed(key, off + 3, p) ? v.w * inv : 0.f;
      return v;
    };
    auto fetch_fast = [&](int slab) {
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        ra[u] = ldg4(pa + (long)(slab * kSub + u) * sa16);
        rb[u] = ldg4(pb + (long)(slab * kSub + u) * sb16);
      }
    };
'''+s[b:]
s=s.replace('''    auto commit_fast = [&](int slab, int buf, const f32x4_t (&ra)[kSub], const f32x4_t (&rb)[kSub]) {''','''    auto commit_fast = [&](int slab, int buf) {''')
s=s.replace('''        float4 va = a_ok ? make_float4(ra[u][0], ra[u][1], ra[u][2], ra[u][3]) : zero4;
        float4 vb = b_ok ? make_float4(rb[u][0], rb[u][1], rb[u][2], rb[u][3]) : zero4;''','''        float4 va = a_ok ? ra[u] : zero4, vb = b_ok ? rb[u] : zero4;''')
a=s.index('    fetch_fast(0, ra0, rb0);')
b=s.index('  } else {\n    // streaming: double-buffered LDS, one barrier per slab')
s=s[:a]+'''    // (a two-slab-deep register prefetch was measured: no gain -- the loop is not bound by the L2 round
    // trip -- so one register set it is)
    fetch_fast(0);
    commit_fast(0, 0);
    __syncthreads();
    for (int sl = 0; sl < nslab; ++sl) {
      const bool more = sl + 1 < nslab;
      if (more) fetch_fast(sl + 1);
      mfma_slab(sl & 1);
      if (more) commit_fast(sl + 1, (sl + 1) & 1);
      __syncthreads();
    }
'''+s[b:]
open(p,'w').write(s)
