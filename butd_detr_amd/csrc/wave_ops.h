// wave_ops.h -- 64-lane wavefront reductions on the DPP crossbar (gfx950).
//
// __shfl_xor lowers to ds_bpermute_b32: an LDS-crossbar round trip (~100+ cycles) per step.  The DPP
// forms below stay in the VALU: row_shr:1/2/4/8 build a running max/min inside each 16-lane row,
// row_bcast:15 / row_bcast:31 carry the row results across rows, lane 63 ends up with the wave result
// and v_readlane makes it wave-uniform (an SGPR).  `old = src = v` keeps a lane's own value wherever
// the DPP source is invalid or the row is masked off, which is neutral for idempotent max/min.
#pragma once
#include <hip/hip_runtime.h>

namespace waveops {

template <int CTRL, int ROW_MASK>
__device__ inline unsigned dpp_u32(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ inline unsigned umax(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ inline unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }

// Fused DPP ALU steps: `v_max_u32_dpp v, v, v row_shr:N` = one instruction per butterfly level
// (dst = max(dpp(src0), src1) on lanes whose DPP source is valid; other lanes are not written, i.e. keep
// v).  hipcc lowers __builtin_amdgcn_update_dpp + max to mov / nop / mov_dpp / max (4 issues per
// level); spelling the fused form in asm quarters the dependent chain.  `s_nop 1` = the two wait
// states a VALU write needs before a DPP read of the same VGPR (gfx9 hazard; hipcc does not pad
// inside asm).
#define BUTD_DPP_STEP(OPC, v, CTRL, RMASK) \
  asm volatile("s_nop 1\n\t" OPC " %0, %0, %0 " CTRL " row_mask:" RMASK " bank_mask:0xf" : "+v"(v))

#define BUTD_WAVE_REDUCE_ASM(OPC, v)              \
  BUTD_DPP_STEP(OPC, v, "row_shr:1", "0xf");      \
  BUTD_DPP_STEP(OPC, v, "row_shr:2", "0xf");      \
  BUTD_DPP_STEP(OPC, v, "row_shr:4", "0xf");      \
  BUTD_DPP_STEP(OPC, v, "row_shr:8", "0xf");      \
  BUTD_DPP_STEP(OPC, v, "row_bcast:15", "0xa");   \
  BUTD_DPP_STEP(OPC, v, "row_bcast:31", "0xc");   \
  asm volatile("s_nop 1" ::: "memory")

// wave-uniform max / min over all 64 lanes
__device__ inline unsigned wave_max_u32(unsigned v) {
  BUTD_WAVE_REDUCE_ASM("v_max_u32_dpp", v);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ inline unsigned wave_min_u32(unsigned v) {
  BUTD_WAVE_REDUCE_ASM("v_min_u32_dpp", v);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// max over lanes 0..15 only (one DPP row); lanes 16..63 may hold anything
__device__ inline unsigned row0_max_u32(unsigned v) {
  BUTD_DPP_STEP("v_max_u32_dpp", v, "row_shr:1", "0xf");
  BUTD_DPP_STEP("v_max_u32_dpp", v, "row_shr:2", "0xf");
  BUTD_DPP_STEP("v_max_u32_dpp", v, "row_shr:4", "0xf");
  BUTD_DPP_STEP("v_max_u32_dpp", v, "row_shr:8", "0xf");
  asm volatile("s_nop 1" ::: "memory");
  return (unsigned)__builtin_amdgcn_readlane((int)v, 15);
}

#define BUTD_WAVE_REDUCE(v, OP)                 \
  v = OP(v, dpp_u32<0x111, 0xf>(v)); /* row_shr:1 */   \
  v = OP(v, dpp_u32<0x112, 0xf>(v)); /* row_shr:2 */   \
  v = OP(v, dpp_u32<0x114, 0xf>(v)); /* row_shr:4 */   \
  v = OP(v, dpp_u32<0x118, 0xf>(v)); /* row_shr:8 */   \
  v = OP(v, dpp_u32<0x142, 0xa>(v)); /* row_bcast:15 -> rows 1,3 */ \
  v = OP(v, dpp_u32<0x143, 0xc>(v)); /* row_bcast:31 -> rows 2,3 */

__device__ inline unsigned fmin_bits(unsigned a, unsigned b) {
  return __float_as_uint(fminf(__uint_as_float(a), __uint_as_float(b)));
}
__device__ inline unsigned fmax_bits(unsigned a, unsigned b) {
  return __float_as_uint(fmaxf(__uint_as_float(a), __uint_as_float(b)));
}
__device__ inline float wave_min_f32(float f) {
  unsigned v = __float_as_uint(f);
  BUTD_WAVE_REDUCE(v, fmin_bits)
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)v, 63));
}
__device__ inline float wave_max_f32(float f) {
  unsigned v = __float_as_uint(f);
  BUTD_WAVE_REDUCE(v, fmax_bits)
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)v, 63));
}
#undef BUTD_WAVE_REDUCE

// wave-uniform broadcast of lane `src`'s value (src must be wave-uniform)
__device__ inline float bcast_f32(float v, int src) {
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v),
                                                             __builtin_amdgcn_readfirstlane(src)));
}

}  // namespace waveops
