#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__global__ void k_fma(float *out, long long *cyc, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) a = a * b + c;  // dependent chain
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x * blockDim.x] = a;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_dpp(unsigned *out, long long *cyc, int iters) {
  unsigned v = threadIdx.x * 2654435761u;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1" : "+v"(v));
    v = (unsigned)__builtin_amdgcn_readlane((int)v, 63) + i + threadIdx.x;
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x * blockDim.x] = v;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_bperm(unsigned *out, long long *cyc, int iters) {
  unsigned v = threadIdx.x * 2654435761u;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    for (int s = 32; s >= 1; s >>= 1) { unsigned o = __shfl_xor(v, s, 64); v = o > v ? o : v; }
    v += i + threadIdx.x;
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x * blockDim.x] = v;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_lds(unsigned *out, long long *cyc, int iters) {
  __shared__ unsigned buf[1024];
  buf[threadIdx.x] = (threadIdx.x * 7 + 1) & 1023;
  __syncthreads();
  unsigned v = threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) v = buf[v];   // dependent LDS read chain
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x * blockDim.x] = v;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_barrier(unsigned *out, long long *cyc, int iters) {
  __shared__ unsigned buf[1024];
  unsigned v = threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) { buf[threadIdx.x] = v; __syncthreads(); v = buf[(threadIdx.x + 64) % blockDim.x] + 1; }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x * blockDim.x] = v;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_gload(const unsigned *chain, unsigned *out, long long *cyc, int iters) {
  unsigned v = threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) v = chain[v];   // dependent global (L2) read chain
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x * blockDim.x] = v;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <typename F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  float *fo; unsigned *uo; long long *cyc; unsigned *chain;
  CHECK(hipMalloc(&fo, 1 << 22)); CHECK(hipMalloc(&uo, 1 << 22)); CHECK(hipMalloc(&cyc, 8 * 4096));
  const int CN = 1 << 18;  // 1 MB chain (L2 resident)
  std::vector<unsigned> h(CN); for (int i = 0; i < CN; ++i) h[i] = (unsigned)((i * 9973ull + 12345) % CN);
  CHECK(hipMalloc(&chain, CN * 4)); CHECK(hipMemcpy(chain, h.data(), CN * 4, hipMemcpyHostToDevice));
  long long c;
  for (int blocks : {1, 8, 256}) {
    for (int threads : {64, 256, 1024}) {
      int it = 20000;
      float ms = timeit([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(threads), 0, 0, fo, cyc, it); });
      hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("fma   blocks=%3d thr=%4d: %.3f ms, %lld cyc -> %.2f cyc/fma, clock %.0f MHz\n", blocks, threads, ms, c, c / (16.0 * it), c / (ms * 1e3));
    }
  }
  for (int threads : {64, 256, 1024}) {
    int it = 20000;
    float ms = timeit([&] { hipLaunchKernelGGL(k_dpp, dim3(8), dim3(threads), 0, 0, uo, cyc, it); });
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("dpp-reduce   thr=%4d: %lld cyc/reduce (%.1f ns), clock %.0f MHz\n", threads, c / it, ms * 1e6 / it, c / (ms * 1e3));
    ms = timeit([&] { hipLaunchKernelGGL(k_bperm, dim3(8), dim3(threads), 0, 0, uo, cyc, it); });
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("bperm-reduce thr=%4d: %lld cyc/reduce (%.1f ns)\n", threads, c / it, ms * 1e6 / it);
    ms = timeit([&] { hipLaunchKernelGGL(k_lds, dim3(8), dim3(threads), 0, 0, uo, cyc, it); });
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("lds chain    thr=%4d: %lld cyc/read (%.1f ns)\n", threads, c / it, ms * 1e6 / it);
    ms = timeit([&] { hipLaunchKernelGGL(k_barrier, dim3(8), dim3(threads), 0, 0, uo, cyc, it); });
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("lds+barrier  thr=%4d: %lld cyc/iter (%.1f ns)\n", threads, c / it, ms * 1e6 / it);
    ms = timeit([&] { hipLaunchKernelGGL(k_gload, dim3(8), dim3(threads), 0, 0, chain, uo, cyc, it); });
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("L2 chain     thr=%4d: %lld cyc/load (%.1f ns)\n", threads, c / it, ms * 1e6 / it);
  }
  return 0;
}
