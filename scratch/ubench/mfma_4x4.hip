// How long does v_mfma_f32_4x4x1_16b_f32 occupy the matrix pipe next to v_mfma_f32_16x16x4_f32?  (round 5: the head
// dimension 36 = 2 full 16-row tiles + 4 rows; the third tile of the P.V / dV / dK / dQ products is 3/4 padding.)
// One wave per SIMD (1024 waves), CHAINS independent accumulators, N instructions each; cycles from wall time x clock.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int KIND, int CHAINS>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
  f4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = f4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if (KIND == 0) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
        else if (KIND == 1) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
        else {   // 3 : 1 mix as the products would issue them (two full tiles + the thin one ... per contraction step)
          if (c % 3 == 2) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
          else acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
        }
      }
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND, int CHAINS>
void run(const char *name, float *out) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(256), dim3(256), 0, 0, out, 10, 1.f, 1.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(256), dim3(256), 0, 0, out, iters, 1.f, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 8 * CHAINS;
  printf("%-34s chains %d: %8.3f ms  %6.2f ns per instruction (%.1f cycles at 2.4 GHz)\n", name, CHAINS, ms, ms * 1e6 / n, ms * 1e6 / n * 2.4);
}
int main() {
  float *out; hipMalloc(&out, 256 * 256 * 4);
  run<0, 1>("16x16x4 f32", out); run<0, 3>("16x16x4 f32", out); run<0, 6>("16x16x4 f32", out);
  run<1, 1>("4x4x1 f32", out); run<1, 3>("4x4x1 f32", out); run<1, 6>("4x4x1 f32", out);
  run<2, 3>("2 x 16x16x4 + 1 x 4x4x1", out); run<2, 6>("2 x 16x16x4 + 1 x 4x4x1", out);
  return 0;
}
