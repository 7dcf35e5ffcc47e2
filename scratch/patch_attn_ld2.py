def rep(s,a,b,cnt=1):
    assert s.count(a)==cnt, (s.count(a), a[:70])
    return s.replace(a,b)
p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
s=rep(s,'''    float *__restrict__ dq, long ldo,''','''    float *__restrict__ dq, long ldo, float dq_scale,''')
s=rep(s,'''        if (d < D) ob[d] = acc[nt][i];''','''        if (d < D) ob[d] = acc[nt][i] * dq_scale;''')
s=rep(s,'''                       float *dv, long ld_grad, float dropout_p, uint32_t dropout_site,
                       const uint64_t *rng_counter, butd_stream_t stream) {
  if (B <= 0 || H <= 0 || Lq <= 0) return 0;
  if (D <= 0 || D > 48 || (D & 3) || Lk <= 0) return (int)hipErrorInvalidValue;
  if (ld_grad == 0) ld_grad = (long)H * D;
  if (ld_grad < (long)H * D) return (int)hipErrorInvalidValue;''','''                       float *dv, long ld_dq, long ld_dkv, float dq_scale, float dropout_p,
                       uint32_t dropout_site, const uint64_t *rng_counter, butd_stream_t stream) {
  if (B <= 0 || H <= 0 || Lq <= 0) return 0;
  if (D <= 0 || D > 48 || (D & 3) || Lk <= 0) return (int)hipErrorInvalidValue;
  if (ld_dq == 0) ld_dq = (long)H * D;
  if (ld_dkv == 0) ld_dkv = (long)H * D;
  if (ld_dq < (long)H * D || ld_dkv < (long)H * D) return (int)hipErrorInvalidValue;''')
s=rep(s,'''                ld_grad, dropout_p, dropout_site, rng_counter);
  ATTN_DISPATCH(attn_bwd_dkv_kernel, gk, H, Lq, Lk, D, q, k, v, key_padding_mask, dout, lse, delta,
                dk, dv, ld_grad, dropout_p, dropout_site, rng_counter);''','''                ld_dq, dq_scale, dropout_p, dropout_site, rng_counter);
  ATTN_DISPATCH(attn_bwd_dkv_kernel, gk, H, Lq, Lk, D, q, k, v, key_padding_mask, dout, lse, delta,
                dk, dv, ld_dkv, dropout_p, dropout_site, rng_counter);''')
open(p,'w').write(s)
p='include/butd_attention.h'
s=open(p).read()
s=rep(s,'''/* Backward of the above.  delta (B,H,Lq) scratch; dq (B,Lq,H*D), dk, dv (B,Lk,H*D) are overwritten. */
int butd_attention_bwd(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                       const float *v, const uint8_t *key_padding_mask, const float *out,
                       const float *dout, const float *lse, float *delta, float *dq, float *dk,
                       float *dv, float dropout_p, uint32_t dropout_site,
                       const uint64_t *rng_counter, butd_stream_t stream);''','''/* Backward of the above.  delta (B,H,Lq) scratch (written here); dq (B,Lq,.), dk, dv (B,Lk,.) are
 * overwritten.  The gradient rows may be wider than H*D: ld_dq / ld_dkv are their row strides in floats
 * (0 = H*D), so dq|dk|dv (or dk|dv) can sit side by side in one matrix and the input-projection
 * gradients become ONE product over the concatenated contraction.  dq is multiplied by dq_scale on the
 * way out (the 1/sqrt(D) the forward projection applied to q). */
int butd_attention_bwd(int B, int H, int Lq, int Lk, int D, const float *q, const float *k,
                       const float *v, const uint8_t *key_padding_mask, const float *out,
                       const float *dout, const float *lse, float *delta, float *dq, float *dk,
                       float *dv, long ld_dq, long ld_dkv, float dq_scale, float dropout_p,
                       uint32_t dropout_site, const uint64_t *rng_counter, butd_stream_t stream);''')
open(p,'w').write(s)
p='butd_detr_amd/_hiplib.py'
s=open(p).read()
s=rep(s,'''    "butd_attention_bwd": (_c_int, [_c_int] * 5 + [_c_void_p] * 11 + [_c_float, _c_u32, _c_void_p, _c_void_p]),''','''    "butd_attention_bwd": (_c_int, [_c_int] * 5 + [_c_void_p] * 11 + [_c_long, _c_long, _c_float]
                           + [_c_float, _c_u32, _c_void_p, _c_void_p]),''')
open(p,'w').write(s)
p='butd_detr_amd/fused_attention.py'
s=open(p).read()
s=rep(s,'''                                          delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                                          p_attn, site_attn, rng_counter(dev).data_ptr(), _stream(xq))''','''                                          delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                                          0, 0, 1.0, p_attn, site_attn, rng_counter(dev).data_ptr(),
                                          _stream(xq))''')
open(p,'w').write(s)
