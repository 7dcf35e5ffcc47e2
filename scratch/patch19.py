p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
s=s.replace('''      P.a2 == nullptr && !ones && ((kend - kbeg) % kBK) == 0 && m0 + kBM <= P.M && n0 + kBN <= P.N &&
      ((a_kc ? P.lda_m : P.lda_k) & 3) == 0 && ((b_kc ? P.ldb_n : P.ldb_k) & 3) == 0 &&
      ((((uintptr_t)P.a) | ((uintptr_t)P.b)) & 15) == 0 && (kbeg & 3) == 0;''','''      P.a2 == nullptr && !ones && ((kend - kbeg) % kBK) == 0 &&
      (a_kc || (P.M & 3) == 0) && (b_kc || (P.N & 3) == 0) &&   // partial tiles: whole float4 in or out
      ((a_kc ? P.lda_m : P.lda_k) & 3) == 0 && ((b_kc ? P.ldb_n : P.ldb_k) & 3) == 0 &&
      ((((uintptr_t)P.a) | ((uintptr_t)P.b)) & 15) == 0 && (kbeg & 3) == 0;''')
s=s.replace('''    const long sa16 = a_kc ? 16 : 16 * P.lda_k, sb16 = b_kc ? 16 : 16 * P.ldb_k;  // per 16 k''','''    const long sa16 = a_kc ? 16 : 16 * P.lda_k, sb16 = b_kc ? 16 : 16 * P.ldb_k;  // per 16 k
    // rows of this thread inside the matrix?  (loop-invariant; rows outside a partial tile read 0)
    const bool a_ok = m0 + (a_kc ? a_slow : a_fast) < P.M;
    const bool b_ok = n0 + (b_kc ? b_slow : b_fast) < P.N;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);''')
s=s.replace('''    if (b_aff && !b_kc) {  // channel = B row = 4 consecutive rows of this thread: loop-invariant
      bsc4 = *reinterpret_cast<const float4 *>(P.b_chan_scale + n0 + b_fast);
      bsh4 = *reinterpret_cast<const float4 *>(P.b_chan_shift + n0 + b_fast);
    } else if (b_aff) {''','''    if (b_aff && !b_kc) {  // channel = B row = 4 consecutive rows of this thread: loop-invariant
      if (b_ok) {
        bsc4 = *reinterpret_cast<const float4 *>(P.b_chan_scale + n0 + b_fast);
        bsh4 = *reinterpret_cast<const float4 *>(P.b_chan_shift + n0 + b_fast);
      }
    } else if (b_aff && b_ok) {''')
s=s.replace('''        ra[u] = *reinterpret_cast<const float4 *>(pa + (long)(slab * kSub + u) * sa16);
        rb[u] = *reinterpret_cast<const float4 *>(pb + (long)(slab * kSub + u) * sb16);''','''        ra[u] = a_ok ? *reinterpret_cast<const float4 *>(pa + (long)(slab * kSub + u) * sa16) : zero4;
        rb[u] = b_ok ? *reinterpret_cast<const float4 *>(pb + (long)(slab * kSub + u) * sb16) : zero4;''')
s=s.replace('''        if (asc) {
          va.x = fmaxf(va.x * rsc[u].x + rsh[u].x, 0.f); va.y = fmaxf(va.y * rsc[u].y + rsh[u].y, 0.f);
          va.z = fmaxf(va.z * rsc[u].z + rsh[u].z, 0.f); va.w = fmaxf(va.w * rsc[u].w + rsh[u].w, 0.f);
        }
        if (b_aff) {''','''        if (asc && a_ok) {
          va.x = fmaxf(va.x * rsc[u].x + rsh[u].x, 0.f); va.y = fmaxf(va.y * rsc[u].y + rsh[u].y, 0.f);
          va.z = fmaxf(va.z * rsc[u].z + rsh[u].z, 0.f); va.w = fmaxf(va.w * rsc[u].w + rsh[u].w, 0.f);
        }
        if (b_aff && b_ok) {''')
open(p,'w').write(s)
