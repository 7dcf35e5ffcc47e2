def rep(s,a,b,cnt=1):
    assert s.count(a)==cnt, (s.count(a), a[:70])
    return s.replace(a,b)
p='include/butd_attention.h'
s=open(p).read()
s=rep(s,'''  double *col_sum, *col_sumsq;
} butd_gemm_problem;''','''  double *col_sum, *col_sumsq;
  /* Optional accumulation into existing data by the plain-store epilogue (accumulate == 0,
   * split_k == 1; every element has exactly one writer, so no atomics):
   *   c_add != 0:  C <- C + epilogue(v)            (e.g. the residual-path gradient already in C)
   *   c2 != NULL:  C2 <- C2 + epilogue(v) as well   (same ldc; a second consumer of the same product)
   * They let a block return  d_res + dq*Wq  and  dq*Wq  from ONE product instead of autograd adding
   * tensors afterwards.  No other problem of the same launch may write C (c_add) or C2. */
  int c_add;
  float *c2;
} butd_gemm_problem;''')
open(p,'w').write(s)

p='butd_detr_amd/_hiplib.py'
s=open(p).read()
s=rep(s,'''                ("col_sum", _c_void_p), ("col_sumsq", _c_void_p)]''','''                ("col_sum", _c_void_p), ("col_sumsq", _c_void_p),
                ("c_add", _c_int), ("c2", _c_void_p)]''')
open(p,'w').write(s)

p='butd_detr_amd/fused_attention.py'
s=open(p).read()
s=rep(s,'''             col_stats=None):''','''             col_stats=None, c_add=False, c2=None):''')
s=rep(s,'''                       _ptr(col_stats[1]) if col_stats is not None else None)''','''                       _ptr(col_stats[1]) if col_stats is not None else None,
                       int(c_add), _ptr(c2))''')
open(p,'w').write(s)

p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
# fast path a2
s=rep(s,'''    const bool rt_fx = P.a_chan_scale != nullptr || P.b_chan_scale != nullptr || a_dropout || b_dropout ||''','''    const bool rt_fx = P.a2 != nullptr || P.a_chan_scale != nullptr || P.b_chan_scale != nullptr || a_dropout || b_dropout ||''')
s=rep(s,'''    float4 ra[kSub], rb[kSub];
    const long offa0 = pa - P.a, offb0 = pb - P.b;   // element offsets of this thread's first float4''','''    float4 ra[kSub], rb[kSub], ra2[kSub];
    const long offa0 = pa - P.a, offb0 = pb - P.b;   // element offsets of this thread's first float4
    const bool f_a2 = FX && P.a2 != nullptr;         // companion operand: same strides, same offsets
    const float *pa2 = f_a2 ? P.a2 + offa0 : pa;
    const int a2_mode = P.a2_mode;
    const float a2_scale = P.a2_scale;''')
s=rep(s,'''        if constexpr (!decltype(kind)::ragged) {
          ra[u] = ldg4(pa + (long)(slab * kSub + u) * sa16);
          rb[u] = ldg4(pb + (long)(slab * kSub + u) * sb16);
        } else {
          const int k0 = slab * kBK + u * kSW;
          ra[u] = (k0 + a_k < krange) ? ldg4(pa + (long)(slab * kSub + u) * sa16) : zero4;
          rb[u] = (k0 + b_k < krange) ? ldg4(pb + (long)(slab * kSub + u) * sb16) : zero4;
        }''','''        if constexpr (!decltype(kind)::ragged) {
          ra[u] = ldg4(pa + (long)(slab * kSub + u) * sa16);
          rb[u] = ldg4(pb + (long)(slab * kSub + u) * sb16);
          if (f_a2) ra2[u] = ldg4(pa2 + (long)(slab * kSub + u) * sa16);
        } else {
          const int k0 = slab * kBK + u * kSW;
          ra[u] = (k0 + a_k < krange) ? ldg4(pa + (long)(slab * kSub + u) * sa16) : zero4;
          rb[u] = (k0 + b_k < krange) ? ldg4(pb + (long)(slab * kSub + u) * sb16) : zero4;
          if (f_a2) ra2[u] = (k0 + a_k < krange) ? ldg4(pa2 + (long)(slab * kSub + u) * sa16) : zero4;
        }''')
s=rep(s,'''        float4 va = a_live ? ra[u] : zero4, vb = b_live ? rb[u] : zero4;
        if (a_aff && a_live) {''','''        float4 va = a_live ? ra[u] : zero4, vb = b_live ? rb[u] : zero4;
        if (f_a2 && a_live) {
          va.x = combine(va.x, ra2[u].x, a2_mode, a2_scale); va.y = combine(va.y, ra2[u].y, a2_mode, a2_scale);
          va.z = combine(va.z, ra2[u].z, a2_mode, a2_scale); va.w = combine(va.w, ra2[u].w, a2_mode, a2_scale);
        }
        if (a_aff && a_live) {''')
# fast eligibility
s=rep(s,'''  return p.a2 == nullptr && p.K > 0 && (p.K & 3) == 0 &&   // a ragged LAST slab is predicated per float4''','''  return (p.a2 == nullptr || (((uintptr_t)p.a2) & 15) == 0) && p.K > 0 &&
         (p.K & 3) == 0 &&   // a ragged LAST slab is predicated per float4''')
# epilogue c_add / c2
s=rep(s,'''    double *const col_sum = P.col_sum, *const col_sumsq = P.col_sumsq;''','''    double *const col_sum = P.col_sum, *const col_sumsq = P.col_sumsq;
    const bool c_add = P.c_add != 0;
    float *const c2ptr = P.c2;
    const bool vec2_ok = vec_ok && ((((uintptr_t)c2ptr) & 15) == 0);''')
s=rep(s,'''      float *dst = cptr + (long)m * ldc + n;
      if (vec_ok) {
        *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < pN) dst[e] = v[e];
      }
    }
    if (col_sum) {''','''      float *dst = cptr + (long)m * ldc + n;
      if (c2ptr) {   // second destination accumulates the same values
        float *d2 = c2ptr + (long)m * ldc + n;
        if (vec2_ok) {
          const float4 o = *reinterpret_cast<const float4 *>(d2);
          *reinterpret_cast<float4 *>(d2) = make_float4(o.x + v[0], o.y + v[1], o.z + v[2], o.w + v[3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < pN) d2[e] += v[e];
        }
      }
      if (vec_ok) {
        if (c_add) {
          const float4 o = *reinterpret_cast<const float4 *>(dst);
          v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
        }
        *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < pN) dst[e] = c_add ? dst[e] + v[e] : v[e];
      }
    }
    if (col_sum) {''')
# host validation
s=rep(s,'''    if ((p.col_sum != nullptr) && (p.accumulate || p.split_k > 1)) return (int)hipErrorInvalidValue;''','''    if ((p.col_sum != nullptr || p.c_add || p.c2 != nullptr) && (p.accumulate || p.ones_col || p.split_k > 1))
      return (int)hipErrorInvalidValue;''')
open(p,'w').write(s)
