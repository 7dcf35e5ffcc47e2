// zero_fill.h -- zeroing of a device range as an ordinary KERNEL launch.
//
// hipMemsetAsync must not be used on the path: captured into a hipGraph it becomes a MEMSET node, and on ROCm 7.2
// (AQL packet capture of graphs, the default) a memset node of a graph that is replayed behind a still-running
// graph -- or, for multi-megabyte ranges, simply replayed a second time -- writes a garbage pattern instead of its
// value (scratch/graph_node_order.py).  Kernel and memcpy nodes are not affected.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
__global__ __launch_bounds__(256) void butd_zero_fill_kernel(unsigned char *__restrict__ p, size_t bytes) {
  const size_t head = (16 - ((uintptr_t)p & 15)) & 15;          // bytes before the first 16-byte boundary
  const size_t h = head < bytes ? head : bytes;
  const size_t n16 = (bytes - h) >> 4;
  uint4 *q = reinterpret_cast<uint4 *>(p + h);
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) q[i] = z;
  if (blockIdx.x == 0) {
    if (threadIdx.x < h) p[threadIdx.x] = 0;
    const size_t tail = h + (n16 << 4);
    if (tail + threadIdx.x < bytes) p[tail + threadIdx.x] = 0;  // < 16 bytes
  }
}
}  // namespace

static inline hipError_t butd_zero_async(void *p, size_t bytes, hipStream_t s) {
  if (bytes == 0) return hipSuccess;
  size_t blocks = ((bytes >> 4) + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(butd_zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (unsigned char *)p, bytes);
  return hipGetLastError();
}
