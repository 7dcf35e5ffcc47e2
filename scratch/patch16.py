p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
old=s[s.index('  float bias_v[2];\n#pragma unroll\n  for (int j = 0; j < 2; ++j) {'):s.index('// ------------------------------------------------------------------------------------------------\n// y = LayerNorm(residual + dropout(x))')]
new='''  if (!accumulate && !ones_col) {
    // Plain stores: stage the 64x64 tile through LDS (the operand buffers are free after the last
    // barrier) so every thread writes whole float4 row segments -- the MFMA C-layout would otherwise
    // emit sixteen 4-byte stores per lane, 64 contiguous bytes per wave-instruction.
    float(*Cs)[kBN + 4] = reinterpret_cast<float(*)[kBN + 4]>(&As[0][0][0]);
    static_assert(sizeof(As) >= sizeof(float) * kBM * (kBN + 4), "C tile must fit the A buffers");
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Cs[wr * 32 + i * 16 + fg * 4 + r][wc * 32 + j * 16 + fr] = acc[i][j][r];
    __syncthreads();
    const int c4 = (tid & 15) * 4;
    const int n = n0 + c4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (P.bias && slice == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < pN) bv[e] = P.bias[n + e];
    }
    const bool vec_ok = (n + 3 < pN) && ((ldc & 3) == 0) && ((((uintptr_t)cptr) & 15) == 0);
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
    return;
  }
  // accumulate / bias-gradient path: element-wise atomics straight from the accumulators
  float bias_v[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wc * 32 + j * 16 + fr;
    bias_v[j] = (P.bias && slice == 0 && n < pN) ? P.bias[n] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wc * 32 + j * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wr * 32 + i * 16 + fg * 4 + r;
        if (m >= pM) continue;
        float v = acc[i][j][r];
        if (n < pN) {
          v = (v + bias_v[j]) * scale;
          if (relu) v = fmaxf(v, 0.f);
          if (drop)
            v = rng::keep(ctr, site, (uint32_t)((long)m * pN + n), p_drop) ? v * inv_keep : 0.f;
          atomicAdd(cptr + (long)m * ldc + n, v);
        } else if (ones_col && n == pN) {
          atomicAdd(bgrad + m, v * scale);
        }
      }
    }
}

'''
s=s.replace(old,new)
open(p,'w').write(s)
