p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
s=s.replace('kBK = 64, kLd','kBK = 32, kLd')
s=s.replace('''//                        64x64 output tile per 4-wave workgroup, 32x32 per wave (2x2 MFMA tiles), BK=64
//                        (64 MFMAs per wave between barriers: enough work to cover the next slab's loads),''','''//                        64x64 output tile per 4-wave workgroup, 32x32 per wave (2x2 MFMA tiles), BK=32,''')
old=s[s.index('  // epilogue: lane holds C[row = fg*4 + r][col = fr] of each 16x16 tile'):s.index('// ------------------------------------------------------------------------------------------------\n// y = LayerNorm(residual + dropout(x))')]
new='''  // epilogue: lane holds C[row = fg*4 + r][col = fr] of each 16x16 tile.  Everything that is LOADED
  // (bias, RNG counter) is fetched before the first store: the output may alias nothing here, but the
  // compiler cannot know, and a load issued after a store waits for it (16 serialized L2 round trips
  // made the epilogue cost more than the whole K loop).
  const bool drop = P.dropout_p > 0.f;
  const float inv_keep = drop ? 1.f / (1.f - P.dropout_p) : 1.f;
  const uint64_t ctr = (drop && rng_counter) ? *rng_counter : 0ull;
  float *const cptr = P.c;
  float *const bgrad = P.bias_grad;
  const int pM = P.M, pN = P.N, relu = P.relu, accumulate = P.accumulate, ones_col = P.ones_col;
  const long ldc = P.ldc;
  const float scale = P.scale, p_drop = P.dropout_p;
  const uint32_t site = P.dropout_site;
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
          float *dst = cptr + (long)m * ldc + n;
          if (accumulate) atomicAdd(dst, v);
          else *dst = v;
        } else if (ones_col && n == pN) {
          atomicAdd(bgrad + m, v * scale);
        }
      }
    }
}

'''
s=s.replace(old,new)
open(p,'w').write(s)
