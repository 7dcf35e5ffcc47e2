// v_mfma_f32_16x16x4_f32 issue interval as a function of how many independent accumulator chains a wave interleaves
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(256) void k(int iters, float seed, float *out) {
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = (f32x4){seed, 0, 0, (float)c};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8 / CH; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, acc[c], 0, 0, 0);
  }
  float s = 0;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH> void run() {
  float *out; (void)hipMalloc(&out, 256 * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  k<CH><<<256, 256>>>(iters, 1.0f, out); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<CH><<<256, 256>>>(iters, 1.0f, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms / 5 * 1e6 / ((double)iters * 8);
  printf("%d chain(s): %.2f ns per MFMA = %.1f cycles at 2.4 GHz\n", CH, ns, ns * 2.4);
  (void)hipFree(out);
}
int main() { run<1>(); run<2>(); run<4>(); run<8>(); return 0; }
