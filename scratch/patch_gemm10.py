p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    assert s.count(a)==cnt, (s.count(a), a[:70])
    s=s.replace(a,b)
rep('''    auto put = [&](float (*tile)[kLd], bool kc, int slow, int fst, int koff, float4 v) {
      if (kc) {
        *reinterpret_cast<float4 *>(&tile[slow][koff + fst]) = v;
      } else {
        tile[fst + 0][koff + slow] = v.x; tile[fst + 1][koff + slow] = v.y;
        tile[fst + 2][koff + slow] = v.z; tile[fst + 3][koff + slow] = v.w;
      }
    };''','''    // LDS images.  A contraction-contiguous operand is stored [row][k] (row stride kLd) and a fragment
    // (4 consecutive k of one row) is one ds_read_b128.  A row-contiguous operand (both operands of a
    // weight-gradient product) is stored AS IT ARRIVES, [k][row] with row stride kLdT: the float4 write is
    // conflict-free and the fragment becomes four conflict-free ds_read_b32 -- transposing on the way
    // in cost sixteen 4-way-conflicting scalar writes per thread and slab and made these products run
    // at half the per-slab rate of the forward ones.
    constexpr int kLdT = kBM + 4;
    static_assert(kBK * kLdT <= kBM * kLd && kBM == kBN, "[k][row] image must fit the [row][k] buffer");
    auto put = [&](float (*tile)[kLd], bool kc, int slow, int fst, int koff, float4 v) {
      if (kc) {
        *reinterpret_cast<float4 *>(&tile[slow][koff + fst]) = v;
      } else {
        float(*t)[kLdT] = reinterpret_cast<float(*)[kLdT]>(&tile[0][0]);
        *reinterpret_cast<float4 *>(&t[koff + slow][fst]) = v;
      }
    };
    auto frag = [&](float (*tile)[kLd], bool kc, int row, int k0) -> f32x4 {
      if (kc) return *reinterpret_cast<const f32x4 *>(&tile[row][k0]);
      const float(*t)[kLdT] = reinterpret_cast<const float(*)[kLdT]>(&tile[0][0]);
      return (f32x4){t[k0 + 0][row], t[k0 + 1][row], t[k0 + 2][row], t[k0 + 3][row]};
    };
    auto mfma_fast = [&](int buf) {
#pragma unroll
      for (int u = 0; u < kBK / 16; ++u) {
        f32x4 af[2], bf[kNJ];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = frag(As[buf], a_kc, wr * 32 + i * 16 + fr, u * 16 + fg * 4);
#pragma unroll
        for (int j = 0; j < kNJ; ++j)
          bf[j] = frag(Bs[buf], b_kc, wc * (16 * kNJ) + j * 16 + fr, u * 16 + fg * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < kNJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
      }
    };''')
# replace the three mfma_slab calls inside the fast loop
a=s.index('    const int nwhole = krange / kBK;')
b=s.index('    };   // run_fast')
seg=s[a:b]
assert seg.count('mfma_slab(sl & 1);')==3
seg=seg.replace('mfma_slab(sl & 1);','mfma_fast(sl & 1);')
s=s[:a]+seg+s[b:]
open(p,'w').write(s)
