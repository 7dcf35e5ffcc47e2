// fps_common.h -- pieces shared by the two FPS kernels (pointnet2_ops.hip, fps_pruned.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "wave_ops.h"

namespace fps {

constexpr int kWave = 64;
constexpr int kMaxWaves = 16;   // LDS slot arrays are sized for up to 1024-thread workgroups

// The reference block (sampling_gpu.cu:74-178) has bs = opt_n_threads(n) threads; thread t owns the
// points k == t (mod bs), keeps its FIRST strict maximum, and the shared-memory tree (:119-173) then
// prefers the LOWER slot on ties, level by level from stride bs/2 down to 1.  Net effect: among equal
// maxima the winner has the smallest bit-reversed slot (k mod bs), then the smallest k.  That is a
// total order on points, so any reduction shape reproduces the reference as long as it maximises
//        (d2, -bitrev(k mod bs), -(k div bs)).
// key(k) packs the last two terms so that a SMALLER key wins; candidates travel as
// (hi = float bits of d2 >= 0, lo = ~key) and are compared as unsigned pairs.
__device__ inline unsigned key_of(unsigned k, int log2bs) {
  const unsigned slot = k & ((1u << log2bs) - 1u);
  const unsigned rev = log2bs ? (__brev(slot) >> (32 - log2bs)) : 0u;
  return (rev << 23) | (k >> log2bs);
}
__device__ inline unsigned index_of(unsigned key, int log2bs) {
  const unsigned rev = key >> 23;
  const unsigned slot = log2bs ? (__brev(rev) >> (32 - log2bs)) : 0u;
  return ((key & 0x7FFFFFu) << log2bs) | slot;
}

// mag <= 1e-3 is a DOUBLE compare in the reference (sampling_gpu.cu:106); float32(1e-3) > 1e-3.
__device__ inline bool skipped(float x, float y, float z) {
  const float mag = (x * x) + (y * y) + (z * z);
  return (double)mag <= 1e-3;
}

struct alignas(16) Slot {
  unsigned hi, lo;
  float x, y, z;
  float pad[3];
};

// In-wave arg-max of (have, bits, key) candidates: one DPP max-reduction of the value bits; the
// key reduction runs only when two lanes tie on the maximum (rare).  The single winning lane (keys are
// unique) publishes the wave's candidate and its coordinates to slot[wave].  A wave without any
// candidate publishes (0,0) from lane 0.
__device__ inline void publish_wave_best(Slot *buf, int wave, int lane, bool have, unsigned bits,
                                         unsigned key, float x, float y, float z) {
  using namespace waveops;
  const unsigned long long any = __ballot(have);
  if (any == 0ull) {
    if (lane == 0) {
      Slot s;
      s.hi = 0u; s.lo = 0u; s.x = 0.f; s.y = 0.f; s.z = 0.f;
      buf[wave] = s;
    }
    return;
  }
  const unsigned mx = wave_max_u32(have ? bits : 0u);
  bool win = have && bits == mx;
  const unsigned long long tied = __ballot(win);
  if (__popcll(tied) > 1) {  // wave-uniform branch
    const unsigned kmin = wave_min_u32(win ? key : 0xFFFFFFFFu);
    win = win && key == kmin;
  }
  if (win) {
    Slot s;
    s.hi = mx; s.lo = 0xFFFFFFFFu - key; s.x = x; s.y = y; s.z = z;
    buf[wave] = s;
  }
}

// After the barrier: every wave redundantly picks the best of the NWAVES published candidates (no
// second barrier for the broadcast): straight LDS broadcast reads + a compare chain, which is shorter
// than two more DPP reductions for the workgroup sizes used here.  Returns the selected index (0 when
// nothing can compete, like the reference's besti = 0) and updates the sample coordinates.
template <int NWAVES>
__device__ inline int select_global_best(const Slot *buf, int lane, int log2bs, float p0x, float p0y,
                                         float p0z, float &x1, float &y1, float &z1) {
  unsigned ghi, glo;
  int wsel = 0;
  if (NWAVES <= 8) {  // few slots: broadcast reads + compare chain
    ghi = buf[0].hi;
    glo = buf[0].lo;
#pragma unroll
    for (int q = 1; q < NWAVES; ++q) {
      const unsigned h = buf[q].hi, l = buf[q].lo;
      const bool better = h > ghi || (h == ghi && l > glo);
      ghi = better ? h : ghi;
      glo = better ? l : glo;
      wsel = better ? q : wsel;
    }
    wsel = __builtin_amdgcn_readfirstlane(wsel);
  } else {            // 16 slots: one per lane of DPP row 0, two row reductions
    using namespace waveops;
    const unsigned shi = lane < NWAVES ? buf[lane].hi : 0u;
    const unsigned slo = lane < NWAVES ? buf[lane].lo : 0u;
    ghi = row0_max_u32(shi);
    glo = row0_max_u32(shi == ghi ? slo : 0u);
    const unsigned long long hit = __ballot(lane < NWAVES && shi == ghi && slo == glo);
    wsel = hit ? __ffsll((long long)hit) - 1 : 0;
  }
  if ((ghi | glo) == 0u) {  // lo = ~key >= 1 for every real candidate
    x1 = p0x; y1 = p0y; z1 = p0z;
    return 0;
  }
  x1 = buf[wsel].x;
  y1 = buf[wsel].y;
  z1 = buf[wsel].z;
  return (int)index_of(0xFFFFFFFFu - glo, log2bs);
}

}  // namespace fps
