// graph_ops.hip -- hipGraph hygiene (include/butd_graph.h): memset nodes -> kernel nodes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/butd_graph.h"

namespace {
// value replicated over `width` elements of `esize` bytes in each of `height` rows `pitch` bytes apart
__global__ __launch_bounds__(256) void memset_node_kernel(unsigned char *dst, unsigned value, unsigned esize,
                                                          size_t width, size_t height, size_t pitch) {
  for (size_t row = blockIdx.y; row < height; row += gridDim.y) {
    unsigned char *p = dst + row * pitch;
    const size_t bytes = width * esize;
    unsigned word = value;                                     // the 32-bit pattern of four / two / one element(s)
    if (esize == 1) word = (value & 0xffu) * 0x01010101u;
    else if (esize == 2) word = (value & 0xffffu) * 0x00010001u;
    const size_t head = (16 - ((uintptr_t)p & 15)) & 15;       // (element sizes divide 16: patterns stay in phase)
    const size_t h = head < bytes ? head : bytes;
    const size_t n16 = (bytes - h) >> 4;
    uint4 *q = reinterpret_cast<uint4 *>(p + h);
    const uint4 w4 = make_uint4(word, word, word, word);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) q[i] = w4;
    if (blockIdx.x == 0) {
      const size_t tail = h + (n16 << 4);
      for (size_t i = threadIdx.x; i < h; i += 256) p[i] = (unsigned char)(word >> (8 * ((((uintptr_t)(p + i)) & 3))));
      for (size_t i = tail + threadIdx.x; i < bytes; i += 256)
        p[i] = (unsigned char)(word >> (8 * ((((uintptr_t)(p + i)) & 3))));
    }
  }
}
}  // namespace

namespace {
int count_nodes(hipGraph_t graph, int counts[16]) {
  size_t n = 0;
  hipError_t e = hipGraphGetNodes(graph, nullptr, &n);
  if (e != hipSuccess) return (int)e;
  std::vector<hipGraphNode_t> nodes(n);
  if (n && (e = hipGraphGetNodes(graph, nodes.data(), &n)) != hipSuccess) return (int)e;
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType t;
    if ((e = hipGraphNodeGetType(nodes[i], &t)) != hipSuccess) return (int)e;
    if ((int)t >= 0 && (int)t < 16) counts[(int)t]++;
    if (t == hipGraphNodeTypeGraph) {   // the nodes of an embedded child graph are replayed too: count them as well
      hipGraph_t child;
      if ((e = hipGraphChildGraphNodeGetGraph(nodes[i], &child)) != hipSuccess) return (int)e;
      if (int err = count_nodes(child, counts)) return err;
    }
  }
  return 0;
}

int replace_memsets(hipGraph_t graph, int *replaced) {
  size_t n = 0;
  hipError_t e = hipGraphGetNodes(graph, nullptr, &n);
  if (e != hipSuccess) return (int)e;
  std::vector<hipGraphNode_t> nodes(n);
  if (n && (e = hipGraphGetNodes(graph, nodes.data(), &n)) != hipSuccess) return (int)e;
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType t;
    if ((e = hipGraphNodeGetType(nodes[i], &t)) != hipSuccess) return (int)e;
    if (t == hipGraphNodeTypeGraph) {   // recurse: a memset inside a child graph is as unsafe as a top-level one
      hipGraph_t child;
      if ((e = hipGraphChildGraphNodeGetGraph(nodes[i], &child)) != hipSuccess) return (int)e;
      if (int err = replace_memsets(child, replaced)) return err;
      continue;
    }
    if (t != hipGraphNodeTypeMemset) continue;
    hipMemsetParams mp;
    if ((e = hipGraphMemsetNodeGetParams(nodes[i], &mp)) != hipSuccess) return (int)e;
    if (mp.elementSize != 1 && mp.elementSize != 2 && mp.elementSize != 4) return (int)hipErrorInvalidValue;
    size_t nd = 0, nn = 0;
    if ((e = hipGraphNodeGetDependencies(nodes[i], nullptr, &nd)) != hipSuccess) return (int)e;
    std::vector<hipGraphNode_t> deps(nd);
    if (nd && (e = hipGraphNodeGetDependencies(nodes[i], deps.data(), &nd)) != hipSuccess) return (int)e;
    if ((e = hipGraphNodeGetDependentNodes(nodes[i], nullptr, &nn)) != hipSuccess) return (int)e;
    std::vector<hipGraphNode_t> next(nn);
    if (nn && (e = hipGraphNodeGetDependentNodes(nodes[i], next.data(), &nn)) != hipSuccess) return (int)e;

    unsigned char *dst = (unsigned char *)mp.dst;
    unsigned value = mp.value, esize = mp.elementSize;
    size_t width = mp.width, height = mp.height ? mp.height : 1, pitch = mp.pitch;
    void *args[] = {&dst, &value, &esize, &width, &height, &pitch};
    const size_t bytes = width * esize;
    size_t bx = ((bytes >> 4) + 255) / 256;
    if (bx > 1024) bx = 1024;
    if (bx < 1) bx = 1;
    hipKernelNodeParams kp = {};
    kp.func = (void *)memset_node_kernel;
    kp.gridDim = dim3((unsigned)bx, (unsigned)(height < 1024 ? height : 1024), 1);
    kp.blockDim = dim3(256, 1, 1);
    kp.sharedMemBytes = 0;
    kp.kernelParams = args;
    kp.extra = nullptr;
    hipGraphNode_t fill;
    if ((e = hipGraphAddKernelNode(&fill, graph, deps.data(), deps.size(), &kp)) != hipSuccess) return (int)e;
    for (size_t k = 0; k < nn; ++k)
      if ((e = hipGraphAddDependencies(graph, &fill, &next[k], 1)) != hipSuccess) return (int)e;
    if ((e = hipGraphDestroyNode(nodes[i])) != hipSuccess) return (int)e;
    if (replaced) ++*replaced;
  }
  return 0;
}
}  // namespace

extern "C" int butd_graph_node_counts(void *graph, int counts[16]) {
  for (int i = 0; i < 16; ++i) counts[i] = 0;
  return count_nodes((hipGraph_t)graph, counts);
}

extern "C" int butd_graph_replace_memset_nodes(void *graph, int *replaced) {
  if (replaced) *replaced = 0;
  return replace_memsets((hipGraph_t)graph, replaced);
}

namespace {
__global__ void timeline_mark_kernel(unsigned long long *slots, int slot) {
  if (threadIdx.x == 0) slots[slot] = wall_clock64();   // s_memrealtime: constant-rate counter (100 MHz)
}
}  // namespace

extern "C" int butd_timeline_mark(unsigned long long *slots, int slot, void *stream) {
  hipLaunchKernelGGL(timeline_mark_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, slots, slot);
  return (int)hipGetLastError();
}

extern "C" int butd_runtime_versions(int *runtime, int *driver) {
  hipError_t e = hipRuntimeGetVersion(runtime);
  if (e != hipSuccess) return (int)e;
  return (int)hipDriverGetVersion(driver);
}

extern "C" int butd_stream_create(int priority, void **stream) {
  if (!stream) return (int)hipErrorInvalidValue;
  int lo = 0, hi = 0;
  hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);     // lo = least priority (numerically largest)
  if (e != hipSuccess) return (int)e;
  if (priority < hi) priority = hi;
  if (priority > lo) priority = lo;
  hipStream_t s = nullptr;
  e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority);
  *stream = (void *)s;
  return (int)e;
}

extern "C" int butd_stream_destroy(void *stream) {
  return stream ? (int)hipStreamDestroy((hipStream_t)stream) : 0;
}
