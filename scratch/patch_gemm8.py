p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
def rep(a,b):
    global s
    assert s.count(a)==1, (s.count(a), a)
    s=s.replace(a,b)
rep('''    bool a_in[kSub], b_in[kSub];
#pragma unroll
    for (int u = 0; u < kSub; ++u) a_in[u] = b_in[u] = true;
    auto fetch_fast = [&](int slab) {
#ifdef GEMM_NO_RAGGED
      const bool whole = true;
#else
      const bool whole = (slab + 1) * kBK <= krange;   // uniform
#endif
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        if (whole) {
          ra[u] = ldg4(pa + (long)(slab * kSub + u) * sa16);
          rb[u] = ldg4(pb + (long)(slab * kSub + u) * sb16);
        } else {
          const int k0 = slab * kBK + u * kSW;
          a_in[u] = k0 + a_k < krange;
          b_in[u] = k0 + b_k < krange;
          ra[u] = a_in[u] ? ldg4(pa + (long)(slab * kSub + u) * sa16) : zero4;
          rb[u] = b_in[u] ? ldg4(pb + (long)(slab * kSub + u) * sb16) : zero4;
        }
      }
    };''','''    // fetch / commit come in two flavours selected at compile time: whole slabs (the steady state: no
    // predicates at all) and the ragged last slab (its predicates cost ~8 % when left in the main loop)
    struct Whole { static constexpr bool ragged = false; };
    struct Ragged { static constexpr bool ragged = true; };
    auto fetch_fast = [&](int slab, auto kind) {
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        if constexpr (!decltype(kind)::ragged) {
          ra[u] = ldg4(pa + (long)(slab * kSub + u) * sa16);
          rb[u] = ldg4(pb + (long)(slab * kSub + u) * sb16);
        } else {
          const int k0 = slab * kBK + u * kSW;
          ra[u] = (k0 + a_k < krange) ? ldg4(pa + (long)(slab * kSub + u) * sa16) : zero4;
          rb[u] = (k0 + b_k < krange) ? ldg4(pb + (long)(slab * kSub + u) * sb16) : zero4;
        }
      }
    };''')
rep('''    auto commit_fast = [&](int slab, int buf) {
      const int kslab0 = slab * kBK;   // k offset (relative to kbeg) of the slab held in ra/rb
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        const bool a_live = a_ok && a_in[u], b_live = b_ok && b_in[u];''','''    auto commit_fast = [&](int slab, int buf, auto kind) {
      const int kslab0 = slab * kBK;   // k offset (relative to kbeg) of the slab held in ra/rb
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        bool a_in = true, b_in = true;
        if constexpr (decltype(kind)::ragged) {
          a_in = kslab0 + u * kSW + a_k < krange;
          b_in = kslab0 + u * kSW + b_k < krange;
        }
        const bool a_live = a_ok && a_in, b_live = b_ok && b_in;''')
rep('''        if (b_in[u]) {   // the ones-row is 1 for every k inside the slice''','''        if (b_in) {   // the ones-row is 1 for every k inside the slice''')
rep('''    fetch_fast(0);
    commit_fast(0, 0);
    __syncthreads();
    for (int sl = 0; sl < nslab; ++sl) {
      const bool more = sl + 1 < nslab;
      if (more) fetch_fast(sl + 1);
      mfma_slab(sl & 1);
      if (more) commit_fast(sl + 1, (sl + 1) & 1);
      __syncthreads();
    }''','''    const int nwhole = krange / kBK;
    if (nwhole > 0) {
      fetch_fast(0, Whole());
      commit_fast(0, 0, Whole());
    } else {
      fetch_fast(0, Ragged());
      commit_fast(0, 0, Ragged());
    }
    __syncthreads();
    for (int sl = 0; sl < nslab; ++sl) {
      const int nx = sl + 1;
      if (nx < nwhole) {
        fetch_fast(nx, Whole());
        mfma_slab(sl & 1);
        commit_fast(nx, nx & 1, Whole());
      } else if (nx < nslab) {
        fetch_fast(nx, Ragged());
        mfma_slab(sl & 1);
        commit_fast(nx, nx & 1, Ragged());
      } else {
        mfma_slab(sl & 1);
      }
      __syncthreads();
    }''')
open(p,'w').write(s)
