def rep(s,a,b,cnt=1):
    assert s.count(a)==cnt, (s.count(a), a[:70])
    return s.replace(a,b)
# ---- kernel + ABI
p='include/butd_mlp.h'
s=open(p).read()
s=rep(s,'''#ifdef __cplusplus
}
#endif
#endif /* BUTD_MLP_H */''','''/* out[p, c] = relu(scale[c] * Z[p, c] + shift[c])  (P x C, row stride ld for both): the materialised
 * output of a chain that ENDS in BatchNorm + ReLU (the SharedMLP of PointnetFPModule,
 * pointnet2_modules.py:371-416).  C % 4 == 0. */
int butd_mlp_bn_relu_apply(long P, int C, long ld, const float *Z, const float *scale,
                           const float *shift, float *out, butd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BUTD_MLP_H */''')
open(p,'w').write(s)
p='butd_detr_amd/csrc/mlp_ops.hip'
s=open(p).read()
s=rep(s,'''inline int launch_status() { return (int)hipGetLastError(); }''','''__global__ __launch_bounds__(256) void mlp_bn_relu_apply_kernel(long P, int C, long ld,
                                                                const float *__restrict__ Z,
                                                                const float *__restrict__ scale,
                                                                const float *__restrict__ shift,
                                                                float *__restrict__ out) {
  const int cq = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int c = blockIdx.y * 256 + cq * 4;
  if (c >= C) return;
  const float4 sc = *reinterpret_cast<const float4 *>(scale + c);
  const float4 sh = *reinterpret_cast<const float4 *>(shift + c);
  const long r0 = (long)blockIdx.x * kRows, r1 = min(P, r0 + kRows);
  for (long r = r0 + ph; r < r1; r += 4) {
    const float4 z = *reinterpret_cast<const float4 *>(Z + r * ld + c);
    *reinterpret_cast<float4 *>(out + r * ld + c) =
        make_float4(fmaxf(sc.x * z.x + sh.x, 0.f), fmaxf(sc.y * z.y + sh.y, 0.f),
                    fmaxf(sc.z * z.z + sh.z, 0.f), fmaxf(sc.w * z.w + sh.w, 0.f));
  }
}

inline int launch_status() { return (int)hipGetLastError(); }''')
s=rep(s,'''}  // extern "C"''','''int butd_mlp_bn_relu_apply(long P, int C, long ld, const float *Z, const float *scale,
                           const float *shift, float *out, butd_stream_t stream) {
  if (P < 1 || C < 4 || (C & 3) || (ld & 3)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mlp_bn_relu_apply_kernel, dim3((unsigned)((P + kRows - 1) / kRows), (C + 255) / 256),
                     dim3(256), 0, (hipStream_t)stream, P, C, ld, Z, scale, shift, out);
  return launch_status();
}

}  // extern "C"''')
open(p,'w').write(s)
p='butd_detr_amd/_hiplib.py'
s=open(p).read()
s=rep(s,'''    "butd_mlp_dz": (_c_int, [_c_long, _c_int, _c_long] + [_P] * 7 + [_c_int, _P]),''','''    "butd_mlp_dz": (_c_int, [_c_long, _c_int, _c_long] + [_P] * 7 + [_c_int, _P]),
    "butd_mlp_bn_relu_apply": (_c_int, [_c_long, _c_int, _c_long, _P, _P, _P, _P, _P]),''')
open(p,'w').write(s)
