def rep(s,a,b,cnt=1):
    assert s.count(a)==cnt, (s.count(a), a[:80])
    return s.replace(a,b)
p='include/butd_attention.h'
s=open(p).read()
s=rep(s,'''  int c_add;
  float *c2;
} butd_gemm_problem;''','''  int c_add;
  float *c2;
  /* Column statistics spread over col_slots (a power of two, 0/1 = one) copies of the sum arrays,
   * slot s at col_sum + s * col_slot_stride: a workgroup adds into slot (its linear index mod
   * col_slots), so 10^4..10^5 row tiles do not serialise on the same 2*N addresses; the consumer sums
   * the slots (butd_sa_bn_finalize). */
  int col_slots;
  long col_slot_stride;
} butd_gemm_problem;''')
open(p,'w').write(s)
p='butd_detr_amd/_hiplib.py'
s=open(p).read()
s=rep(s,'''                ("c_add", _c_int), ("c2", _c_void_p)]''','''                ("c_add", _c_int), ("c2", _c_void_p), ("col_slots", _c_int), ("col_slot_stride", _c_long)]''')
s=rep(s,'''    "butd_sa_bn_finalize": (_c_int, [_c_int, _c_long, _P, _P, _P, _P, _c_float, _c_float, _c_int]
                            + [_P] * 7 + [_P]),''','''    "butd_sa_bn_finalize": (_c_int, [_c_int, _c_long, _P, _P, _c_int, _c_long, _P, _P, _c_float, _c_float, _c_int]
                            + [_P] * 7 + [_P]),''')
open(p,'w').write(s)
p='butd_detr_amd/fused_attention.py'
s=open(p).read()
s=rep(s,'''             col_stats=None, c_add=False, c2=None):''','''             col_stats=None, c_add=False, c2=None, col_slots=(0, 0)):''')
s=rep(s,'''                       int(c_add), _ptr(c2))''','''                       int(c_add), _ptr(c2), int(col_slots[0]), int(col_slots[1]))''')
open(p,'w').write(s)
p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
s=rep(s,'''        if (n0 + col < pN) atomicAdd((which ? col_sumsq : col_sum) + n0 + col, acc);''','''        const long slot_off = P.col_slots > 1 ? (long)(blockIdx.x & (P.col_slots - 1)) * P.col_slot_stride : 0;
        if (n0 + col < pN) atomicAdd((which ? col_sumsq : col_sum) + slot_off + n0 + col, acc);''')
s=rep(s,'''    if ((p.col_sum != nullptr || p.c_add || p.c2 != nullptr) && (p.accumulate || p.ones_col || p.split_k > 1))
      return (int)hipErrorInvalidValue;''','''    if ((p.col_sum != nullptr || p.c_add || p.c2 != nullptr) && (p.accumulate || p.ones_col || p.split_k > 1))
      return (int)hipErrorInvalidValue;
    if (p.col_slots > 1 && (p.col_slots & (p.col_slots - 1))) return (int)hipErrorInvalidValue;''')
open(p,'w').write(s)

p='include/butd_sa.h'
s=open(p).read()
s=rep(s,'''int butd_sa_bn_finalize(int C, long count, const double *sum, const double *sumsq, const float *gamma,''','''int butd_sa_bn_finalize(int C, long count, const double *sum, const double *sumsq, int slots,
                        long slot_stride, const float *gamma,''')
s=rep(s,'''/* BatchNorm bookkeeping of one layer (training): from sum/sumsq over `count` rows ->''','''/* BatchNorm bookkeeping of one layer (training): from sum/sumsq over `count` rows (given as `slots`
 * partial copies `slot_stride` doubles apart, slots <= 1: one copy) ->''')
open(p,'w').write(s)
p='butd_detr_amd/csrc/sa_ops.hip'
s=open(p).read()
s=rep(s,'''__global__ void sa_bn_finalize_kernel(int C, long count, const double *__restrict__ sum,
                                      const double *__restrict__ sumsq, const float *__restrict__ gamma,''','''__global__ void sa_bn_finalize_kernel(int C, long count, const double *__restrict__ sum,
                                      const double *__restrict__ sumsq, int slots, long slot_stride,
                                      const float *__restrict__ gamma,''')
s=rep(s,'''    const double m = sum[c] / (double)count;
    double v = sumsq[c] / (double)count - m * m;''','''    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < (slots > 1 ? slots : 1); ++k) {
      s1 += sum[c + k * slot_stride];
      s2 += sumsq[c + k * slot_stride];
    }
    const double m = s1 / (double)count;
    double v = s2 / (double)count - m * m;''')
open(p,'w').write(s)
