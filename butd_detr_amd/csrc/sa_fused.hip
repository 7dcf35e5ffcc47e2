// sa_fused.hip -- the set-abstraction level as ONE kernel for inference (BatchNorm folded): ball-query
// neighbourhoods are gathered into LDS tiles, the three 1x1 convolutions run LDS -> MFMA registers -> LDS, and
// only the max-pooled (B, npoint, C3) features ever reach HBM (include/butd_sa.h: butd_sa_fused_eval).
//
// Reference path it replaces (nickgkan/butd_detr): QueryAndGroup (pointnet2_utils.py:317-376) ->
// SharedMLP = 3 x [Conv2d 1x1 -> BatchNorm2d -> ReLU] (pytorch_utils.py:11-36) -> F.max_pool2d over nsample
// (pointnet2_modules.py:243-257), which materialises the (B, 3+C, npoint, nsample) grouped tensor and three
// (B, C_l, npoint, nsample) activations (SA1 at 8 x 50 000 points: 34 + 268 + 268 + 537 MB).
//
// Workgroup = 256 threads = 2 x 2 waves over ROWS (64 or 32) consecutive grouped rows = ROWS / nsample centres.
//   gather   X[ROWS][3+C] -> LDS          (xyz - centre) / radius | features, zero padded to a multiple of 16
//   layer l  H_l = relu(scale_l * (H_{l-1} W_l^T) + shift_l): the activation tile is the A operand straight from
//            LDS (one ds_read_b128 per fragment, k-permuted like gemm_ops.hip), W_l streams through a (<=128 x 32)
//            LDS slab, v_mfma_f32_16x16x4_f32 (exact fp32), outputs <= 128 columns per pass
//   pool     max over the rows of a centre: registers -> lane groups (v_permlane swaps) -> the two row-halves
#include <hip/hip_runtime.h>
// (-DBUTD_MAIN_PRIO=n raises the wave priority of this unit's kernels: the captured step's prefetch branches share CUs
// with them; profiles/r06_side_branches.txt)
#ifdef BUTD_MAIN_PRIO
#define BUTD_MAIN_PRIO_SET() __builtin_amdgcn_s_setprio(BUTD_MAIN_PRIO)
#else
#define BUTD_MAIN_PRIO_SET()
#endif

#include <stdint.h>

#include "../../include/butd_sa.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 256;
constexpr int kWLd = 36;         // row stride of the weight slab (32 k + 4)
constexpr int kPassCols = 128;   // output columns per pass

struct Args {
  int N, np, ns, C;
  const float *xyz, *new_xyz, *feats;
  long feat_stride;
  const int *idx;
  float radius;
  int normalize;
  int c_out[3];
  const float *w[3];
  long ldw[3];
  const float *scale[3], *shift[3];
  float *out_pm, *out_cm;
  int ldx, ld1, region0;          // LDS strides / size of the X|H2 region (floats)
};

__device__ inline float quad_max(float v) {  // over lanes c, c^16, c^32, c^48 ; result in all four
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// one layer on the workgroup's tile.  ain[ROWS][lda] (columns K..round16(K) zero), W (c_out x kw, row stride ldw).
// aout != nullptr: H = relu(scale * Z + shift) -> aout[ROWS][ldo];  otherwise the pooled maxima go to `pool`.
template <int ROWS>
__device__ inline void layer(const float *ain, int lda, int K, const float *__restrict__ W, long ldw, int kw,
                             int c_out, const float *__restrict__ sc, const float *__restrict__ sh, float *aout,
                             int ldo, float *wbuf, float *pool /* [2][c_out] */) {
  constexpr int kMI = ROWS / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fg = lane >> 4;
  const bool w_vec = ((ldw & 3) == 0) && ((((uintptr_t)W) & 15) == 0);
  for (int n0 = 0; n0 < c_out; n0 += kPassCols) {
    const int pw = min(kPassCols, c_out - n0);        // 32, 64 or 128 columns in this pass
    const int nj = pw / 32;                           // 16-column tiles per wave
    f32x4 acc[kMI][4];
#pragma unroll
    for (int i = 0; i < kMI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 32) {
      __syncthreads();                                // slab free (and, first time, the input tile written)
      for (int f = tid; f < pw * 8; f += kThreads) {
        const int row = f >> 3, kq = (f & 7) * 4;
        const float *src = W + (long)(n0 + row) * ldw + k0 + kq;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (w_vec && k0 + kq + 3 < kw) {
          v = *reinterpret_cast<const float4 *>(src);
        } else {
          if (k0 + kq + 0 < kw) v.x = src[0];
          if (k0 + kq + 1 < kw) v.y = src[1];
          if (k0 + kq + 2 < kw) v.z = src[2];
          if (k0 + kq + 3 < kw) v.w = src[3];
        }
        *reinterpret_cast<float4 *>(wbuf + row * kWLd + kq) = v;
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (k0 + u * 16 >= K) break;
        f32x4 af[kMI], bf[4];
#pragma unroll
        for (int i = 0; i < kMI; ++i)
          af[i] = *reinterpret_cast<const f32x4 *>(ain + (wr * (ROWS / 2) + i * 16 + fr) * lda + k0 + u * 16 + fg * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < nj)
            bf[j] = *reinterpret_cast<const f32x4 *>(wbuf + (wc * (pw / 2) + j * 16 + fr) * kWLd + u * 16 + fg * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < kMI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (j < nj) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
      }
    }
    // epilogue: lane holds rows 4*fg + r of tile i, column fr of tile j
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j >= nj) continue;
      const int n = n0 + wc * (pw / 2) + j * 16 + fr;
      const float s1 = sc[n], h1 = sh[n];
      float best = 0.f;                               // ReLU outputs are >= 0
#pragma unroll
      for (int i = 0; i < kMI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = fmaxf(acc[i][j][r] * s1 + h1, 0.f);
          if (aout) aout[(wr * (ROWS / 2) + i * 16 + fg * 4 + r) * ldo + n] = v;
          best = fmaxf(best, v);
        }
      if (!aout) {
        best = quad_max(best);                        // over the four lane groups: all ROWS/2 rows of this wave
        if (fg == 0) pool[wr * c_out + n] = best;
      }
    }
  }
}

template <int ROWS>
__global__ __launch_bounds__(kThreads) void sa_fused_eval_kernel(Args a) {
  BUTD_MAIN_PRIO_SET();
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *reg0 = smem;                                  // X, later H2
  float *h1 = smem + a.region0;
  float *wbuf = h1 + ROWS * a.ld1;
  float *pool = wbuf + kPassCols * kWLd;               // [2][C3]
  const int tid = threadIdx.x;
  const long row0 = (long)blockIdx.x * ROWS;           // first grouped row (b, j, k) of this workgroup
  const int K0 = 3 + a.C;
  const int K0r = (K0 + 15) / 16 * 16;                 // tile columns (zero padded)
  // ---- gather the neighbourhood rows
  for (int e = tid; e < ROWS * K0r; e += kThreads) {
    const int r = e / K0r, c = e - r * K0r;
    const long p = row0 + r;
    const long g = p / a.ns;                           // (b, j)
    const long b = g / a.np;
    float v = 0.f;
    if (c < K0) {
      const int src = a.idx[p];
      if (c < 3) {
        v = a.xyz[(b * a.N + src) * 3 + c] - a.new_xyz[g * 3 + c];
        if (a.normalize) v = v / a.radius;
      } else {
        v = a.feats[(b * a.N + src) * a.feat_stride + (c - 3)];
      }
    }
    reg0[r * a.ldx + c] = v;
  }
  // (layer() starts with a barrier)
  const int c1 = a.c_out[0], c2 = a.c_out[1], c3 = a.c_out[2];
  layer<ROWS>(reg0, a.ldx, K0r, a.w[0], a.ldw[0], K0, c1, a.scale[0], a.shift[0], h1, a.ld1, wbuf, pool);
  const int ld2 = c2 + 4;
  layer<ROWS>(h1, a.ld1, c1, a.w[1], a.ldw[1], c1, c2, a.scale[1], a.shift[1], reg0, ld2, wbuf, pool);
  layer<ROWS>(reg0, ld2, c2, a.w[2], a.ldw[2], c2, c3, a.scale[2], a.shift[2], nullptr, 0, wbuf, pool);
  __syncthreads();
  // ---- pooled rows out: ROWS / ns centres (1 or 2) per workgroup
  const int centres = ROWS / a.ns;
  const long g0 = row0 / a.ns;
  for (int e = tid; e < centres * c3; e += kThreads) {
    const int cen = e / c3, n = e - cen * c3;
    const float v = centres == 2 ? pool[cen * c3 + n] : fmaxf(pool[n], pool[c3 + n]);
    const long g = g0 + cen;
    if (a.out_pm) a.out_pm[g * c3 + n] = v;
    if (a.out_cm) {
      const long b = g / a.np, j = g - b * a.np;
      a.out_cm[(b * c3 + n) * a.np + j] = v;
    }
  }
}

}  // namespace

extern "C" int butd_sa_fused_eval(int B, int N, int np, int ns, int C, const float *xyz, const float *new_xyz,
                                  const float *feats, long feat_stride, const int *idx, float radius, int normalize,
                                  const int *c_out, const float *const *w, const long *ldw,
                                  const float *const *scale, const float *const *shift, float *out_pm,
                                  float *out_cm, butd_stream_t stream) {
  if (B <= 0 || np <= 0) return 0;
  if (!(ns == 16 || ns == 32 || ns == 64) || C < 0 || (C > 0 && feats == nullptr)) return (int)hipErrorInvalidValue;
  for (int l = 0; l < 3; ++l)
    if (c_out[l] < 32 || c_out[l] > 256 || (c_out[l] % 32) || (l < 2 && c_out[l] > 128)) return (int)hipErrorInvalidValue;
  const int rows = ns == 64 ? 64 : 32;
  const long P = (long)B * np * ns;
  if (P % rows) return (int)hipErrorInvalidValue;
  Args a;
  a.N = N; a.np = np; a.ns = ns; a.C = C;
  a.xyz = xyz; a.new_xyz = new_xyz; a.feats = feats; a.feat_stride = feat_stride; a.idx = idx;
  a.radius = radius; a.normalize = normalize;
  for (int l = 0; l < 3; ++l) {
    a.c_out[l] = c_out[l]; a.w[l] = w[l]; a.ldw[l] = ldw[l]; a.scale[l] = scale[l]; a.shift[l] = shift[l];
  }
  a.out_pm = out_pm; a.out_cm = out_cm;
  const int k0r = (3 + C + 15) / 16 * 16;
  a.ldx = k0r + 4;
  a.ld1 = c_out[0] + 4;
  const int x_floats = rows * a.ldx, h2_floats = rows * (c_out[1] + 4);
  a.region0 = (x_floats > h2_floats ? x_floats : h2_floats);
  a.region0 = (a.region0 + 3) / 4 * 4;
  const size_t lds = sizeof(float) * ((size_t)a.region0 + (size_t)rows * a.ld1 + kPassCols * kWLd + 2 * c_out[2]);
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  const dim3 grid((unsigned)(P / rows));
  static const hipError_t attr = [] {   // dynamic LDS beyond 64 KiB has to be allowed once per kernel
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sa_fused_eval_kernel<64>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(sa_fused_eval_kernel<32>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }();
  if (attr != hipSuccess) return (int)attr;
  if (rows == 64) hipLaunchKernelGGL(sa_fused_eval_kernel<64>, grid, dim3(kThreads), lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(sa_fused_eval_kernel<32>, grid, dim3(kThreads), lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
