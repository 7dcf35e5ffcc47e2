// Do MFMA and VALU instructions of DIFFERENT waves on one SIMD overlap?  8 waves per workgroup = 2 per SIMD, one workgroup
// per CU.  Mode 0: every wave issues MFMAs; 1: every wave VALU FMAs; 2: even waves MFMA, odd waves VALU (each wave does the
// same amount as in modes 0 / 1).  If the pipes were independent, t(2) ~ max(t(0), t(1)) / ... ; if they share the issue
// port / datapath, t(2) ~ (t(0) + t(1)) / 2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>   // 0: f32 16x16x4, 1: bf16 16x16x32
__device__ inline void mfma_loop(int iters, float seed, float *out) {
  f32x4 acc[4] = {{seed, 0, 0, 0}, {0, seed, 0, 0}, {0, 0, seed, 0}, {0, 0, 0, seed}};
  bf16x8 hb;
  for (int i = 0; i < 8; ++i) hb[i] = (__bf16)seed;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if constexpr (KIND == 0) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, acc[u], 0, 0, 0);
      else acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hb, hb, acc[u], 0, 0, 0);
    }
  }
  out[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
__device__ inline void valu_loop(int iters, float seed, float *out) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = __builtin_fmaf(a[u], seed, 1.0f);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[threadIdx.x] = s;
}
template <int KIND>
__global__ __launch_bounds__(512) void k(int mode, int mi, int vi, float seed, float *out) {
  const int wave = threadIdx.x >> 6;
  float *o = out + (long)blockIdx.x * 512;
  const bool do_mfma = mode == 0 || (mode == 2 && (wave & 4) == 0);   // waves 0-3 / 4-7: one of each per SIMD
  if (mode == 3) { if ((wave & 4) == 0) mfma_loop<KIND>(mi, seed, o); return; }   // only half the waves exist
  if (mode == 4) { if ((wave & 4) != 0) valu_loop(vi, seed, o); return; }
  if (do_mfma) mfma_loop<KIND>(mi, seed, o); else valu_loop(vi, seed, o);
}
template <int KIND>
void run(const char *name, int mi, int vi) {
  float *out; hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *modes[] = {"all waves MFMA (2/SIMD)", "all waves VALU (2/SIMD)", "one MFMA + one VALU wave per SIMD",
                         "one MFMA wave per SIMD alone", "one VALU wave per SIMD alone"};
  for (int mode = 0; mode < 5; ++mode) {
    k<KIND><<<256, 512>>>(mode, mi, vi, 1.0f, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<KIND><<<256, 512>>>(mode, mi, vi, 1.0f, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %-36s %8.1f us\n", name, modes[mode], ms / 5 * 1e3);
  }
  hipFree(out);
}
int main() {
  // per wave: mi*4 MFMAs, vi*8 VALU FMAs.  16x16x4 f32 = 32 cycles -> 20000*4*32 = 2.56 M cycles; VALU 4 cycles each
  run<0>("v_mfma_f32_16x16x4_f32", 20000, 80000);
  run<1>("v_mfma_f32_16x16x32_bf16", 20000, 40000);
  return 0;
}
