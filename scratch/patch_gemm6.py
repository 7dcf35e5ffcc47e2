p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
def rep(a,b):
    global s
    assert s.count(a)==1, (s.count(a), a)
    s=s.replace(a,b)
rep('''constexpr int kAffK = 320;''','''// Asynchronous variant for the software-pipelined K loop: the compiler's own s_waitcnt placement drains
// EVERY outstanding load at each loop back-edge, which collapses a two-slab-deep prefetch to depth one.
// Loads issued through this asm are invisible to that pass; wait_loads<N>() is the matching explicit
// wait ("at most N loads still in flight"; loads retire in issue order) and threads the destination
// registers through itself so no use can be scheduled above it.
__device__ inline void ldg4_async(f32x4_t &dst, const float *p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ inline void wait_loads(f32x4_t &a, f32x4_t &b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N>
__device__ inline void wait_loads(f32x4_t &a, f32x4_t &b, f32x4_t &c, f32x4_t &d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
constexpr int kAffK = 320;''')
a=s.index('    float4 ra0[kSub], rb0[kSub], ra1[kSub], rb1[kSub];')
b=s.index('    auto put = [&](float (*tile)[kLd], bool kc, int slow, int fst, int koff, float4 v) {')
new='''    f32x4_t ra0[kSub], rb0[kSub], ra1[kSub], rb1[kSub];
    constexpr int kLoadsPerSlab = 2 * kSub;
    const long offa0 = pa - P.a, offb0 = pb - P.b;   // element offsets of this thread's first float4
    auto drop4 = [](float4 v, uint32_t key, uint32_t off, float p, float inv) {
      v.x = rng::keep_keyed(key, off + 0, p) ? v.x * inv : 0.f;
      v.y = rng::keep_keyed(key, off + 1, p) ? v.y * inv : 0.f;
      v.z = rng::keep_keyed(key, off + 2, p) ? v.z * inv : 0.f;
      v.w = rng::keep_keyed(key, off + 3, p) ? v.w * inv : 0.f;
      return v;
    };
    auto fetch_fast = [&](int slab, f32x4_t (&ra)[kSub], f32x4_t (&rb)[kSub]) {
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        ldg4_async(ra[u], pa + (long)(slab * kSub + u) * sa16);
        ldg4_async(rb[u], pb + (long)(slab * kSub + u) * sb16);
      }
    };
    // newer = true: one younger slab (kLoadsPerSlab loads) may stay in flight
    auto arrive = [&](bool newer, f32x4_t (&ra)[kSub], f32x4_t (&rb)[kSub]) {
      if constexpr (kSub == 1) {
        if (newer) wait_loads<kLoadsPerSlab>(ra[0], rb[0]); else wait_loads<0>(ra[0], rb[0]);
      } else {
        if (newer) wait_loads<kLoadsPerSlab>(ra[0], rb[0], ra[kSub - 1], rb[kSub - 1]);
        else wait_loads<0>(ra[0], rb[0], ra[kSub - 1], rb[kSub - 1]);
      }
    };
'''
s=s[:a]+new+s[b:]
rep('''    auto commit_fast = [&](int slab, int buf, const float4 (&ra)[kSub], const float4 (&rb)[kSub]) {''','''    auto commit_fast = [&](int slab, int buf, const f32x4_t (&ra)[kSub], const f32x4_t (&rb)[kSub]) {''')
rep('''        float4 va = a_ok ? ra[u] : zero4, vb = b_ok ? rb[u] : zero4;
        if (a_aff && a_ok) {
          float4 sc, sh;
          if (a_kc) {
            sc = *reinterpret_cast<const float4 *>(&Asc[kslab0 + u * kSW + a_fast]);''','''        float4 va = a_ok ? make_float4(ra[u][0], ra[u][1], ra[u][2], ra[u][3]) : zero4;
        float4 vb = b_ok ? make_float4(rb[u][0], rb[u][1], rb[u][2], rb[u][3]) : zero4;
        if (a_aff && a_ok) {
          float4 sc, sh;
          if (a_kc) {
            sc = *reinterpret_cast<const float4 *>(&Asc[kslab0 + u * kSW + a_fast]);''')
a=s.index('    fetch_fast(0, ra0, rb0);\n    if (nslab > 1) fetch_fast(1, ra1, rb1);')
b=s.index('  } else {\n    // streaming: double-buffered LDS, one barrier per slab')
new='''    fetch_fast(0, ra0, rb0);
    if (nslab > 1) fetch_fast(1, ra1, rb1);
    arrive(nslab > 1, ra0, rb0);
    commit_fast(0, 0, ra0, rb0);
    __syncthreads();
    for (int sl = 0; sl < nslab; sl += 2) {
      // even slab sl sits in LDS buffer 0, set 1 holds slab sl+1 (in flight), set 0 is free
      if (sl + 2 < nslab) fetch_fast(sl + 2, ra0, rb0);
      mfma_slab(0);
      if (sl + 1 < nslab) {
        arrive(sl + 2 < nslab, ra1, rb1);
        commit_fast(sl + 1, 1, ra1, rb1);
      }
      __syncthreads();
      if (sl + 1 >= nslab) break;
      if (sl + 3 < nslab) fetch_fast(sl + 3, ra1, rb1);
      mfma_slab(1);
      if (sl + 2 < nslab) {
        arrive(sl + 3 < nslab, ra0, rb0);
        commit_fast(sl + 2, 0, ra0, rb0);
      }
      __syncthreads();
    }
'''
s=s[:a]+new+s[b:]
open(p,'w').write(s)
