// round 6: how large may a by-value kernel argument be on this runtime?  (the grouped GEMM's 8-problem cap assumes 4 KB)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Big { int v[N]; };
template <int N> __global__ void k(Big<N> b, int *out) { out[0] = b.v[N - 1] + b.v[0]; }
template <int N> void run(int *d) {
  Big<N> b; for (int i = 0; i < N; ++i) b.v[i] = i;
  hipMemset(d, 0, 4);
  hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, b, d);
  hipError_t e = hipDeviceSynchronize(); hipError_t e2 = hipGetLastError();
  int h = -1; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
  printf("%7zu bytes: sync=%d last=%d result=%d (want %d)\n", sizeof(b), (int)e, (int)e2, h, N - 1);
}
int main() { int *d; hipMalloc(&d, 4); run<512>(d); run<1024>(d); run<2048>(d); run<4096>(d); run<8192>(d); run<16384>(d); return 0; }
