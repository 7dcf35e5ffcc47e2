// issue cost of VALU instruction kinds on gfx950: one wave per SIMD (256 threads / CU), 8 independent chains per lane
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int KIND>
__global__ __launch_bounds__(256) void k(int iters, uint32_t seed, uint32_t *out) {
  uint32_t a[8]; float f[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + i * 977 + threadIdx.x; f[i] = (float)a[i] * 1e-9f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if constexpr (KIND == 0) f[u] = __builtin_fmaf(f[u], 1.0001f, 0.5f);
      else if constexpr (KIND == 1) a[u] = a[u] * 0x7feb352dU;                 // v_mul_lo_u32
      else if constexpr (KIND == 2) a[u] = __umul24(a[u], 0x7feb35U);          // v_mul_u32_u24
      else if constexpr (KIND == 3) f[u] = __builtin_amdgcn_exp2f(f[u]);       // v_exp_f32
      else if constexpr (KIND == 4) a[u] = a[u] ^ (a[u] >> 15);                // shift + xor
      else if constexpr (KIND == 5) f[u] = fmaxf(f[u], 0.25f) ;                // v_max
      else if constexpr (KIND == 6) a[u] = (a[u] >> 16) >= seed ? a[u] + 3u : 0u;   // shift, cmp, cndmask/add
    }
  }
  uint32_t s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + __float_as_uint(f[i]);
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND> void run(const char *name, int per_iter) {
  uint32_t *out; (void)hipMalloc(&out, 256 * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  k<KIND><<<256, 256>>>(iters, 12345u, out); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<KIND><<<256, 256>>>(iters, 12345u, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double ns_per = ms / 5 * 1e6 / ((double)iters * 8 * per_iter);
  printf("%-34s %6.2f ns per wave instruction = %5.1f cycles at 2.4 GHz\n", name, ns_per, ns_per * 2.4);
  (void)hipFree(out);
}
int main() {
  run<0>("v_fma_f32", 1); run<1>("v_mul_lo_u32", 1); run<2>("v_mul_u32_u24", 1); run<3>("v_exp_f32", 1);
  run<4>("v_lshrrev + v_xor (2 instr)", 2); run<5>("v_max_f32", 1); run<6>("shift+cmp+cndmask+add (~4)", 4);
  return 0;
}
