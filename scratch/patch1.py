p='butd_detr_amd/csrc/fps_common.h'
s=open(p).read()
a=s.index('// In-wave arg-max of (have, bits, key) candidates')
new = r'''// In-wave arg-max of (have, bits, key) candidates: one DPP max-reduction of the value bits; the
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
__device__ inline int select_global_best(const Slot *buf, int log2bs, float p0x, float p0y, float p0z,
                                         float &x1, float &y1, float &z1) {
  unsigned ghi = buf[0].hi, glo = buf[0].lo;
  int wsel = 0;
#pragma unroll
  for (int q = 1; q < NWAVES; ++q) {
    const unsigned h = buf[q].hi, l = buf[q].lo;
    const bool better = h > ghi || (h == ghi && l > glo);
    ghi = better ? h : ghi;
    glo = better ? l : glo;
    wsel = better ? q : wsel;
  }
  if ((ghi | glo) == 0u) {  // lo = ~key >= 1 for every real candidate
    x1 = p0x; y1 = p0y; z1 = p0z;
    return 0;
  }
  wsel = __builtin_amdgcn_readfirstlane(wsel);
  x1 = buf[wsel].x;
  y1 = buf[wsel].y;
  z1 = buf[wsel].z;
  return (int)index_of(0xFFFFFFFFu - glo, log2bs);
}

}  // namespace fps
'''
s=s[:a]+new
open(p,'w').write(s)

p='butd_detr_amd/csrc/pointnet2_ops.hip'
s=open(p).read()
s=s.replace('fps::select_global_best(buf, kNumWaves, lane, log2bs, p0x, p0y, p0z, x1, y1, z1)','fps::select_global_best<kNumWaves>(buf, log2bs, p0x, p0y, p0z, x1, y1, z1)')
s=s.replace('''fps::select_global_best(buf, kFpsThreads / kWave, lane, log2bs, p0x, p0y, p0z,
                                            x1, y1, z1)''','''fps::select_global_best<kFpsThreads / kWave>(buf, log2bs, p0x, p0y, p0z, x1, y1, z1)''')
assert 'fps::select_global_best(' not in s
s=s.replace('''  if (n <= 256) FPS_LAUNCH(256, 1);
  else if (n <= 512) FPS_LAUNCH(256, 2);
  else if (n <= 1024) FPS_LAUNCH(256, 4);
  else if (n <= 2048) FPS_LAUNCH(256, 8);
  else if (n <= 4096) FPS_LAUNCH(256, 16);
  else if (n <= 8192) FPS_LAUNCH(1024, 8);''','''  const char *cfg = getenv("BUTD_FPS_CFG");  // tuning hook: "threads,ppt"
  int ct = 0, cp = 0;
  if (cfg && sscanf(cfg, "%d,%d", &ct, &cp) == 2 && (long long)ct * cp >= n) {
    if (ct == 256 && cp == 2) FPS_LAUNCH(256, 2);
    else if (ct == 256 && cp == 4) FPS_LAUNCH(256, 4);
    else if (ct == 256 && cp == 8) FPS_LAUNCH(256, 8);
    else if (ct == 512 && cp == 1) FPS_LAUNCH(512, 1);
    else if (ct == 512 && cp == 2) FPS_LAUNCH(512, 2);
    else if (ct == 512 && cp == 4) FPS_LAUNCH(512, 4);
    else if (ct == 1024 && cp == 1) FPS_LAUNCH(1024, 1);
    else if (ct == 1024 && cp == 2) FPS_LAUNCH(1024, 2);
    else return (int)hipErrorInvalidValue;
    return launch_status();
  }
  if (n <= 256) FPS_LAUNCH(256, 1);
  else if (n <= 512) FPS_LAUNCH(256, 2);
  else if (n <= 1024) FPS_LAUNCH(256, 4);
  else if (n <= 2048) FPS_LAUNCH(256, 8);
  else if (n <= 4096) FPS_LAUNCH(256, 16);
  else if (n <= 8192) FPS_LAUNCH(1024, 8);''')
s=s.replace('#include <math.h>\n#include <stdint.h>\n','#include <math.h>\n#include <stdint.h>\n#include <stdio.h>\n#include <stdlib.h>\n',1)
open(p,'w').write(s)

p='butd_detr_amd/csrc/fps_pruned.hip'
s=open(p).read()
s=s.replace('fps::select_global_best(buf, kLoopWaves, lane, log2bs, p0x, p0y, p0z, x1, y1, z1)','fps::select_global_best<kLoopWaves>(buf, log2bs, p0x, p0y, p0z, x1, y1, z1)')
old='''          const unsigned mx = wave_max_u32(have ? vbits : 0u);
          const unsigned kmin = wave_min_u32((have && vbits == mx) ? vkey : 0xFFFFFFFFu);
          const unsigned long long win = __ballot(have && vbits == mx && vkey == kmin);
          const int wl = __ffsll((long long)win) - 1;  // exactly one lane: keys are unique'''
assert old in s
s=s.replace(old,'''          const unsigned mx = wave_max_u32(have ? vbits : 0u);
          unsigned long long win = __ballot(have && vbits == mx);
          if (__popcll(win) > 1) {  // tie on the value: smallest key wins (wave-uniform branch)
            const unsigned km = wave_min_u32((have && vbits == mx) ? vkey : 0xFFFFFFFFu);
            win = __ballot(have && vbits == mx && vkey == km);
          }
          const int wl = __ffsll((long long)win) - 1;  // exactly one lane: keys are unique
          const unsigned kmin =
              (unsigned)__builtin_amdgcn_readlane((int)vkey, __builtin_amdgcn_readfirstlane(wl));''')
old='''    const unsigned whi = wave_max_u32(chi);
    const unsigned wlo = wave_max_u32(chi == whi ? clo : 0u);
    Slot *buf = slots[j & 1];
    if (chi == whi && clo == wlo && ((whi | wlo) != 0u || lane == 0)) {
      Slot s;
      s.hi = whi; s.lo = wlo;
      s.x = cx; s.y = cy; s.z = cz;
      buf[wave] = s;
    }'''
assert old in s
s=s.replace(old,'''    const unsigned whi = wave_max_u32(chi);
    bool cwin = chi == whi;
    if (__popcll(__ballot(cwin)) > 1) {  // several lanes tie on the value (or the wave has nothing)
      const unsigned wlo = wave_max_u32(cwin ? clo : 0u);
      cwin = cwin && clo == wlo;
      if (wlo == 0u) cwin = lane == 0;  // no candidate in this wave at all: publish (0,0) once
    }
    Slot *buf = slots[j & 1];
    if (cwin) {
      Slot s;
      s.hi = chi; s.lo = clo;
      s.x = cx; s.y = cy; s.z = cz;
      buf[wave] = s;
    }''')
old='''constexpr int kLoopWaves = 4;
constexpr int kLoopThreads = kLoopWaves * kWave;
constexpr int kSets = kMaxChunk / (kLoopWaves * kWave);  // 4'''
assert old in s
s=s.replace(old,'''#ifndef FPS_LOOP_WAVES
#define FPS_LOOP_WAVES 8
#endif
constexpr int kLoopWaves = FPS_LOOP_WAVES;
constexpr int kLoopThreads = kLoopWaves * kWave;
constexpr int kSets = kMaxChunk / (kLoopWaves * kWave);''')
open(p,'w').write(s)
