p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
a=s.index('// Stage a (rows x 16) slab of an operand into LDS as tile[row][k].')
b=s.index('__global__ __launch_bounds__(kGemmThreads) void gemm_kernel(')
new_stage = r'''// Staging of a (rows x 16) operand slab, split in two so the global loads of slab i+1 are in flight
// while the MFMAs of slab i run:  fetch_tile() -> 4 floats in registers,  commit_tile() -> LDS as
// tile[row][k].   element(row, k) = src[row*ld_row + k*ld_k] combined with src2 (see butd_gemm_problem);
// rows >= nrows and k >= kend read 0, except the virtual ones-row (row == nrows && ones): 1.0.
__device__ inline float combine(float a, float a2, int mode, float gate_scale) {
  return mode == 0 ? a + a2 : a * (a2 > 0.f ? gate_scale : 0.f);
}

struct Frag4 { float v[4]; };

__device__ inline Frag4 fetch_tile(const float *__restrict__ src, const float *__restrict__ src2,
                                   int mode2, float scale2, long ld_row, long ld_k, int row0,
                                   int nrows, int k0, int kend, bool ones, int tid) {
  Frag4 f;
  f.v[0] = f.v[1] = f.v[2] = f.v[3] = 0.f;
  if (ld_k == 1) {  // contraction-contiguous: 4 consecutive k of one row
    const int r = tid >> 2, kq = (tid & 3) * 4;
    const int gr = row0 + r, gk = k0 + kq;
    if (gr < nrows) {
      const long o = (long)gr * ld_row + gk;
      const bool vec = (gk + 3 < kend) && ((ld_row & 3) == 0) && ((((uintptr_t)src) & 15) == 0);
      if (vec) {
        const float4 q = *reinterpret_cast<const float4 *>(src + o);
        f.v[0] = q.x; f.v[1] = q.y; f.v[2] = q.z; f.v[3] = q.w;
        if (src2) {
          const float4 q2 = *reinterpret_cast<const float4 *>(src2 + o);
          f.v[0] = combine(f.v[0], q2.x, mode2, scale2); f.v[1] = combine(f.v[1], q2.y, mode2, scale2);
          f.v[2] = combine(f.v[2], q2.z, mode2, scale2); f.v[3] = combine(f.v[3], q2.w, mode2, scale2);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (gk + i < kend)
            f.v[i] = src2 ? combine(src[o + i], src2[o + i], mode2, scale2) : src[o + i];
      }
    } else if (ones && gr == nrows) {
#pragma unroll
      for (int i = 0; i < 4; ++i) f.v[i] = (gk + i < kend) ? 1.f : 0.f;
    }
  } else {  // row-contiguous: 4 consecutive rows of one k
    const int k = tid >> 4, r4 = (tid & 15) * 4;
    const int gk = k0 + k, gr = row0 + r4;
    if (gk < kend) {
      const long o = (long)gk * ld_k + gr;
      const bool vec = (gr + 3 < nrows) && ((ld_k & 3) == 0) && ((((uintptr_t)src) & 15) == 0);
      if (vec) {
        const float4 q = *reinterpret_cast<const float4 *>(src + o);
        f.v[0] = q.x; f.v[1] = q.y; f.v[2] = q.z; f.v[3] = q.w;
        if (src2) {
          const float4 q2 = *reinterpret_cast<const float4 *>(src2 + o);
          f.v[0] = combine(f.v[0], q2.x, mode2, scale2); f.v[1] = combine(f.v[1], q2.y, mode2, scale2);
          f.v[2] = combine(f.v[2], q2.z, mode2, scale2); f.v[3] = combine(f.v[3], q2.w, mode2, scale2);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (gr + i < nrows)
            f.v[i] = src2 ? combine(src[o + i], src2[o + i], mode2, scale2) : src[o + i];
          else if (ones && gr + i == nrows) f.v[i] = 1.f;
        }
      }
    }
  }
  return f;
}

__device__ inline void commit_tile(float (*tile)[kLd], const Frag4 &f, long ld_k, int tid) {
  if (ld_k == 1) {
    const int r = tid >> 2, kq = (tid & 3) * 4;
    *reinterpret_cast<float4 *>(&tile[r][kq]) = make_float4(f.v[0], f.v[1], f.v[2], f.v[3]);
  } else {  // transpose into the K-contiguous LDS image
    const int k = tid >> 4, r4 = (tid & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[r4 + i][k] = f.v[i];
  }
}

'''
s=s[:a]+new_stage+s[b:]
old=s[s.index('  __shared__ __attribute__((aligned(16))) float As[kBM][kLd];'):s.index('  int pi = 0;')]
s=s.replace(old,'''  __shared__ __attribute__((aligned(16))) float As[2][kBM][kLd];
  __shared__ __attribute__((aligned(16))) float Bs[2][kBN][kLd];

''')
old=s[s.index('  for (int k0 = kbeg; k0 < kend; k0 += kBK) {\n    stage_tile(As'):s.index('  // epilogue: lane holds C[row = fg*4 + r][col = fr] of each 16x16 tile')]
new='''  // double-buffered LDS, one barrier per slab: slab i+1 travels global -> registers while slab i is
  // multiplied, then lands in the other buffer
  const bool ones = P.ones_col != 0;
  Frag4 fa = fetch_tile(P.a, P.a2, P.a2_mode, P.a2_scale, P.lda_m, P.lda_k, m0, P.M, kbeg, kend, false, tid);
  Frag4 fb = fetch_tile(P.b, nullptr, 0, 0.f, P.ldb_n, P.ldb_k, n0, P.N, kbeg, kend, ones, tid);
  commit_tile(As[0], fa, P.lda_k, tid);
  commit_tile(Bs[0], fb, P.ldb_k, tid);
  __syncthreads();
  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += kBK) {
    const bool more = k0 + kBK < kend;
    if (more) {
      fa = fetch_tile(P.a, P.a2, P.a2_mode, P.a2_scale, P.lda_m, P.lda_k, m0, P.M, k0 + kBK, kend, false, tid);
      fb = fetch_tile(P.b, nullptr, 0, 0.f, P.ldb_n, P.ldb_k, n0, P.N, k0 + kBK, kend, ones, tid);
    }
    f32x4 af[2], bf[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
      af[i] = *reinterpret_cast<const f32x4 *>(&As[cur][wr * 32 + i * 16 + fr][fg * 4]);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      bf[j] = *reinterpret_cast<const f32x4 *>(&Bs[cur][wc * 32 + j * 16 + fr][fg * 4]);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    if (more) {
      commit_tile(As[cur ^ 1], fa, P.lda_k, tid);
      commit_tile(Bs[cur ^ 1], fb, P.ldb_k, tid);
    }
    __syncthreads();
    cur ^= 1;
  }

'''
s=s.replace(old,new)
open(p,'w').write(s)
