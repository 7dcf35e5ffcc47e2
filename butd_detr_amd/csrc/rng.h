// rng.h -- counter-based dropout masks (stateless, replayable).
//
// keep(counter, site, index, p): one Bernoulli(1-p) decision per (training step, dropout site,
// element).  `counter` is a device-resident step counter bumped once per forward (so a captured
// hipGraph draws fresh masks on every replay), `site` distinguishes the dropout layers of the model,
// `index` is the element's linear index.  Forward and backward recompute the same mask instead of
// storing it.  Statistically equivalent to, not bit-identical with, torch's Philox dropout.
#pragma once
#include <stdint.h>

namespace rng {

__host__ __device__ inline uint32_t mix32(uint32_t x) {  // "lowbias32" finaliser
  x ^= x >> 16; x *= 0x7feb352dU;
  x ^= x >> 15; x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

// the (step, site) half of the hash: loop-invariant for a kernel that draws many elements of one site
__host__ __device__ inline uint32_t site_key(uint64_t counter, uint32_t site) {
  return mix32((uint32_t)counter * 0x9E3779B9u + site * 0x85EBCA6Bu + (uint32_t)(counter >> 32));
}

__host__ __device__ inline bool keep_keyed(uint32_t key, uint32_t index, float p) {
  const uint32_t h = mix32(index ^ key);
  return (float)(h >> 8) * (1.0f / 16777216.0f) >= p;
}

__host__ __device__ inline bool keep(uint64_t counter, uint32_t site, uint32_t index, float p) {
  return keep_keyed(site_key(counter, site), index, p);
}

}  // namespace rng
