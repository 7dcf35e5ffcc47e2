p='butd_detr_amd/csrc/fps_common.h'
s=open(p).read()
old=s[s.index('template <int NWAVES>\n__device__ inline int select_global_best'):s.index('}  // namespace fps')]
new='''template <int NWAVES>
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

'''
s=s.replace(old,new)
open(p,'w').write(s)
for p in ('butd_detr_amd/csrc/pointnet2_ops.hip','butd_detr_amd/csrc/fps_pruned.hip'):
    s=open(p).read()
    s=s.replace('>(buf, log2bs, p0x, p0y, p0z, x1, y1, z1)','>(buf, lane, log2bs, p0x, p0y, p0z, x1, y1, z1)')
    s=s.replace('#define FPS_LOOP_WAVES 8','#define FPS_LOOP_WAVES 16')
    s=s.replace('''  else if (n <= 2048) FPS_LAUNCH(256, 8);
  else if (n <= 4096) FPS_LAUNCH(256, 16);
  else if (n <= 8192) FPS_LAUNCH(1024, 8);''','''  else if (n <= 2048) FPS_LAUNCH(1024, 2);   // measured on MI355X: see DESIGN.md "FPS tuning"
  else if (n <= 4096) FPS_LAUNCH(1024, 4);
  else if (n <= 8192) FPS_LAUNCH(1024, 8);''')
    open(p,'w').write(s)
