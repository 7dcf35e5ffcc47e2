// register layout of v_mfma_f32_4x4x1_16B_f32 and v_mfma_f32_4x4x4_16B_bf16 (hypothesis: block = lane/4; A row / B col =
// lane%4; D[row = vgpr][col = lane%4]) and their issue interval
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void layout(float *out, float *outb) {
  const int l = threadIdx.x;
  const float a = 1.f + l;            // A[block l/4][row l%4]
  const float b = 100.f * (1 + l);    // B[block l/4][col l%4]
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
  bf16x4 ha, hb;
  for (int k = 0; k < 4; ++k) { ha[k] = (__bf16)(float)(1 + (l % 4) + 4 * k); hb[k] = (__bf16)(float)((k == (l / 4) % 4) ? (1 + l % 4) : 0); }
  f32x4 d = {0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ha, hb, d, 0, 0, 0);
  for (int r = 0; r < 4; ++r) outb[l * 4 + r] = d[r];
}
template <int KIND>
__global__ __launch_bounds__(256) void rate(int iters, float seed, float *out) {
  f32x4 acc[4];
  for (int c = 0; c < 4; ++c) acc[c] = (f32x4){seed, 0, 0, (float)c};
  bf16x4 h = {(__bf16)seed, (__bf16)seed, (__bf16)seed, (__bf16)seed};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if constexpr (KIND == 0) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(seed, seed, acc[c], 0, 0, 0);
        else acc[c] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(h, h, acc[c], 0, 0, 0);
      }
  out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
int main() {
  float *out, *outb; (void)hipMalloc(&out, 64 * 16 * 4 * 64); (void)hipMalloc(&outb, 64 * 16);
  layout<<<1, 64>>>(out, outb); (void)hipDeviceSynchronize();
  float h[256], hb[256]; (void)hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost); (void)hipMemcpy(hb, outb, 1024, hipMemcpyDeviceToHost);
  int bad = 0, badb = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int blk = l / 4, col = l % 4;
      const float want = (1.f + blk * 4 + r) * 100.f * (1 + blk * 4 + col);   // A[blk][r] * B[blk][col]
      if (h[l * 4 + r] != want) ++bad;
      // bf16 K=4: A[row][k] = 1 + row + 4k, B[k][col] = (k == blk%4) ? 1 + col : 0  ->  D[r][col] = (1 + r + 4*(blk%4)) * (1 + col)
      const float wb = (1.f + r + 4 * (blk % 4)) * (1 + col);
      if (hb[l * 4 + r] != wb) ++badb;
    }
  printf("4x4x1 f32 layout hypothesis: %s (%d mismatches); 4x4x4 bf16 (a[k] = A[lane%%4][k], b[k] = B[k][lane%%4]): %s (%d)\n",
         bad ? "WRONG" : "confirmed", bad, badb ? "WRONG" : "confirmed", badb);
  if (bad) for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int kind = 0; kind < 2; ++kind) {
    const int iters = 40000;
    if (kind == 0) rate<0><<<256, 256>>>(iters, 1.f, out); else rate<1><<<256, 256>>>(iters, 1.f, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) { if (kind == 0) rate<0><<<256, 256>>>(iters, 1.f, out); else rate<1><<<256, 256>>>(iters, 1.f, out); }
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms / 5 * 1e6 / ((double)iters * 8);
    printf("%s: %.2f ns = %.1f cycles at 2.4 GHz per instruction\n", kind ? "v_mfma_f32_4x4x4_16B_bf16" : "v_mfma_f32_4x4x1_16B_f32", ns, ns * 2.4);
  }
  return 0;
}
