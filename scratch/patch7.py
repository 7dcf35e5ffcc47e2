p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
old='''      for (int i = 0; i < 4; ++i) {
        const int kk = key0 + t * 16 + fg * 4 + i;
        const bool valid = kk < Lk && !(mb && mb[kk]);
        st[t][i] = valid ? st[t][i] : kNegInf;
        tmax = fmaxf(tmax, st[t][i]);
      }'''
new='''      for (int i = 0; i < 4; ++i) {
        const int kk = key0 + t * 16 + fg * 4 + i;
        const float sc = st[t][i] + key_bias(mb, kk, Lk);  // -inf for masked / out-of-range keys
        st[t][i] = sc;
        tmax = fmaxf(tmax, sc);
      }'''
assert old in s
s=s.replace(old,new)
s=s.replace('''// row `r` of a (rows x D) head slice''','''// additive score bias of key `kk`: 0 when it takes part, -inf when padded (mask byte != 0) or beyond Lk.
// Branch-free on purpose: the mask byte is always loaded (index clamped).
__device__ inline float key_bias(const uint8_t *__restrict__ mb, int kk, int Lk) {
  const int kc = kk < Lk ? kk : Lk - 1;
  const unsigned mv = mb ? (unsigned)mb[kc] : 0u;
  return (kk < Lk && mv == 0u) ? 0.f : kNegInf;
}

// row `r` of a (rows x D) head slice''')
old='''        const int kk = key0 + t * 16 + fg * 4 + i;
        const bool valid = kk < Lk && !(mb && mb[kk]) && qi < Lq;
        const float p = valid ? __expf(st[i] - my_lse) : 0.f;'''
assert old in s
s=s.replace(old,'''        const int kk = key0 + t * 16 + fg * 4 + i;
        const float p = qi < Lq ? __expf(st[i] + key_bias(mb, kk, Lk) - my_lse) : 0.f;''')
old='''  const bool key_ok = ki < Lk && !(mask && mask[(long)b * Lk + ki]);'''
assert old in s
s=s.replace(old,'''  const float my_bias = key_bias(mask ? mask + (long)b * Lk : nullptr, ki, Lk);
  const bool key_ok = my_bias == 0.f;''')
open(p,'w').write(s)
