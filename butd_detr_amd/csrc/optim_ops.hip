// optim_ops.hip -- flat AdamW update (include/butd_optim.h): one float4 stream over p, g, m, v.
#include <hip/hip_runtime.h>
// (-DBUTD_MAIN_PRIO=n raises the wave priority of this unit's kernels: the captured step's prefetch branches share CUs
// with them; profiles/r06_side_branches.txt)
#ifdef BUTD_MAIN_PRIO
#define BUTD_MAIN_PRIO_SET() __builtin_amdgcn_s_setprio(BUTD_MAIN_PRIO)
#else
#define BUTD_MAIN_PRIO_SET()
#endif

#include <stdint.h>
#include <math.h>

#include "../../include/butd_optim.h"

namespace {
__global__ __launch_bounds__(256) void adamw_flat_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                         float *__restrict__ m, float *__restrict__ v,
                                                         long begin, long end, float lr, float b1, float b2,
                                                         float eps, float wd, const float *__restrict__ step,
                                                         const float *__restrict__ grad_scale,
                                                         const float *__restrict__ hyper) {
  BUTD_MAIN_PRIO_SET();
  if (hyper) {   // {lr, weight_decay} of this group, device-resident: a captured graph follows the scheduler
    lr = hyper[0];
    wd = hyper[1];
  }
  const float t = *step;
  const float gs = grad_scale ? *grad_scale : 1.f;
  const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
  const float step_size = lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2), decay = 1.f - lr * wd;
  const long n4 = (end - begin) >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long o = begin + i * 4;
    float4 pp = *reinterpret_cast<float4 *>(p + o), mm = *reinterpret_cast<float4 *>(m + o),
           vv = *reinterpret_cast<float4 *>(v + o);
    const float4 gg = *reinterpret_cast<const float4 *>(g + o);
    float pe[4] = {pp.x, pp.y, pp.z, pp.w}, me[4] = {mm.x, mm.y, mm.z, mm.w},
          ve[4] = {vv.x, vv.y, vv.z, vv.w};
    const float ge[4] = {gg.x * gs, gg.y * gs, gg.z * gs, gg.w * gs};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pe[e] *= decay;
      me[e] = b1 * me[e] + (1.f - b1) * ge[e];
      ve[e] = b2 * ve[e] + (1.f - b2) * ge[e] * ge[e];
      pe[e] -= step_size * me[e] / (sqrtf(ve[e]) * inv_sqrt_bc2 + eps);
    }
    *reinterpret_cast<float4 *>(p + o) = make_float4(pe[0], pe[1], pe[2], pe[3]);
    *reinterpret_cast<float4 *>(m + o) = make_float4(me[0], me[1], me[2], me[3]);
    *reinterpret_cast<float4 *>(v + o) = make_float4(ve[0], ve[1], ve[2], ve[3]);
  }
}
// one workgroup = one 4096-float chunk of one segment (binary search over the segments' first workgroups)
__global__ __launch_bounds__(256) void gather_segments_kernel(int n, const int64_t *__restrict__ table,
                                                              float *__restrict__ dst) {
  BUTD_MAIN_PRIO_SET();
  const int64_t *src_ptr = table, *dst_off = table + n, *numel = table + 2 * n, *blk = table + 3 * n;
  int lo = 0, hi = n - 1;
  const long b = blockIdx.x;
  while (lo < hi) {  // last segment whose first workgroup <= b
    const int mid = (lo + hi + 1) >> 1;
    if (blk[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const long first = (b - blk[lo]) * BUTD_GATHER_CHUNK;
  const long count = min((long)BUTD_GATHER_CHUNK, numel[lo] - first);
  const float *src = reinterpret_cast<const float *>(src_ptr[lo]) + first;
  float *out = dst + dst_off[lo] + first;
  if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0) {
    const long n4 = count >> 2;
    for (long i = threadIdx.x; i < n4; i += 256)
      reinterpret_cast<float4 *>(out)[i] = reinterpret_cast<const float4 *>(src)[i];
    for (long i = (n4 << 2) + threadIdx.x; i < count; i += 256) out[i] = src[i];
  } else {
    for (long i = threadIdx.x; i < count; i += 256) out[i] = src[i];
  }
}
}  // namespace

// clip_grad_norm_ coefficient of the packed gradient buffer in two launches, without a zero-initialised
// semaphore or atomics: per-workgroup sums of squares (fp64) into a workspace, then one workgroup folds them.
namespace {
constexpr int kClipBlocks = 1024;
__device__ inline double block_sum_256(double v, double *lds) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  return lds[0] + lds[1] + lds[2] + lds[3];
}
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float *__restrict__ g, long n,
                                                            double *__restrict__ partial) {
  BUTD_MAIN_PRIO_SET();
  __shared__ double lds[4];
  const long n4 = n >> 2;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;     // four independent fp32 chains per lane, folded in fp64
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4 *>(g)[i];
    a0 = fmaf(v.x, v.x, a0); a1 = fmaf(v.y, v.y, a1); a2 = fmaf(v.z, v.z, a2); a3 = fmaf(v.w, v.w, a3);
  }
  double acc = ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float t = g[(n & ~3L) + threadIdx.x];
    acc += (double)t * (double)t;
  }
  const double s = block_sum_256(acc, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void clip_finalize_kernel(const double *__restrict__ partial, int blocks,
                                                            float max_norm, float grad_div,
                                                            float *__restrict__ grad_scale,
                                                            float *__restrict__ norm_out) {
  BUTD_MAIN_PRIO_SET();
  __shared__ double lds[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < blocks; i += 256) acc += partial[i];
  const double s = block_sum_256(acc, lds);
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(s) / grad_div;
    float c = max_norm > 0.f ? fminf(max_norm / (norm + 1e-6f), 1.f) : 1.f;
    if (!(norm == norm)) c = norm;                 // NaN gradients stay visible (torch.clamp propagates NaN)
    *grad_scale = c / grad_div;
    if (norm_out) *norm_out = norm;
  }
}
}  // namespace

extern "C" size_t butd_clip_workspace_bytes(void) { return kClipBlocks * sizeof(double); }

extern "C" int butd_clip_coefficient(const float *g, long n, float max_norm, float grad_div, void *workspace,
                                     float *grad_scale, float *norm_out, butd_stream_t stream) {
  if (n < 0 || !workspace || !grad_scale || !(grad_div > 0.f)) return (int)hipErrorInvalidValue;
  if ((uintptr_t)g & 15) return (int)hipErrorInvalidValue;
  long blocks = ((n >> 2) + 255) / 256;
  if (blocks > kClipBlocks) blocks = kClipBlocks;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, n,
                     (double *)workspace);
  hipLaunchKernelGGL(clip_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double *)workspace,
                     (int)blocks, max_norm, grad_div, grad_scale, norm_out);
  return (int)hipGetLastError();
}

extern "C" int butd_gather_segments(int n, const int64_t *table, float *dst, butd_stream_t stream,
                                    long total_blocks) {
  if (n <= 0 || total_blocks <= 0) return 0;
  if (total_blocks > 0x7fffffffL) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(gather_segments_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, n,
                     table, dst);
  return (int)hipGetLastError();
}

extern "C" int butd_adamw_flat(float *p, const float *g, float *m, float *v, long begin, long end, float lr,
                               float beta1, float beta2, float eps, float weight_decay, const float *step,
                               const float *grad_scale, const float *hyper, butd_stream_t stream) {
  if (end <= begin) return 0;
  if ((begin & 3) || (end & 3)) return (int)hipErrorInvalidValue;  // segments are padded to 4 floats
  const long n4 = (end - begin) >> 2;
  long blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     begin, end, lr, beta1, beta2, eps, weight_decay, step, grad_scale, hyper);
  return (int)hipGetLastError();
}


// out = s0 + s1 + ... (n <= 8 tensors of `numel` floats): the gradient fan-in of a tensor that feeds several blocks
// (autograd sums the incoming gradients pairwise: one launch and one extra pass per addend).
namespace {
struct SumSrc { const float *p[8]; };
__global__ __launch_bounds__(256) void sum_n_kernel(SumSrc src, int n, long n4, long numel, float *__restrict__ out) {
  BUTD_MAIN_PRIO_SET();
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 a = reinterpret_cast<const float4 *>(src.p[0])[i];
    for (int k = 1; k < n; ++k) {
      const float4 b = reinterpret_cast<const float4 *>(src.p[k])[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4 *>(out)[i] = a;
  }
  if (blockIdx.x == 0 && threadIdx.x < (numel & 3)) {   // tail
    const long i = (numel & ~3L) + threadIdx.x;
    float a = src.p[0][i];
    for (int k = 1; k < n; ++k) a += src.p[k][i];
    out[i] = a;
  }
}
}  // namespace

extern "C" int butd_sum_tensors(int n, const float *const *srcs, long numel, float *out, butd_stream_t stream) {
  if (n < 1 || n > 8 || numel < 0) return (int)hipErrorInvalidValue;
  if (numel == 0) return 0;
  SumSrc s;
  for (int k = 0; k < 8; ++k) s.p[k] = srcs[k < n ? k : 0];
  for (int k = 0; k < n; ++k)
    if (((uintptr_t)s.p[k] | (uintptr_t)out) & 15) return (int)hipErrorInvalidValue;   // float4 path: 16-byte aligned
  const long n4 = numel >> 2;
  long blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sum_n_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, s, n, n4, numel, out);
  return (int)hipGetLastError();
}
