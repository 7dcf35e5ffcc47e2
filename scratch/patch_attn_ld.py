def rep(s,a,b,cnt=1):
    assert s.count(a)==cnt, (s.count(a), a[:70])
    return s.replace(a,b)
p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
# dq kernel: add ldo param
s=rep(s,'''    const float *__restrict__ dout, const float *__restrict__ lse, float *__restrict__ delta,
    float *__restrict__ dq,''','''    const float *__restrict__ dout, const float *__restrict__ lse, float *__restrict__ delta,
    float *__restrict__ dq, long ldo,''')
s=rep(s,'''    float *ob = dq + ((long)b * Lq + qi) * E + h * D;''','''    float *ob = dq + ((long)b * Lq + qi) * ldo + h * D;''')
s=rep(s,'''    const float *__restrict__ lse, const float *__restrict__ delta, float *__restrict__ dk,
    float *__restrict__ dv, float p_drop, uint32_t site, const uint64_t *__restrict__ rng_counter) {''','''    const float *__restrict__ lse, const float *__restrict__ delta, float *__restrict__ dk,
    float *__restrict__ dv, long ldo, float p_drop, uint32_t site,
    const uint64_t *__restrict__ rng_counter) {''')
s=rep(s,'''    float *okp = dk + ((long)b * Lk + ki) * E + h * D;
    float *ovp = dv + ((long)b * Lk + ki) * E + h * D;''','''    float *okp = dk + ((long)b * Lk + ki) * ldo + h * D;
    float *ovp = dv + ((long)b * Lk + ki) * ldo + h * D;''')
open(p,'w').write(s)
