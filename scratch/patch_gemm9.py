p='butd_detr_amd/csrc/attention_ops.hip'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    assert s.count(a)==cnt, (s.count(a), a)
    s=s.replace(a,b)
a=s.index('  const bool a_kc = P.lda_k == 1, b_kc = P.ldb_k == 1;\n  if constexpr (FAST) {')
b=s.index('  } else {\n    // streaming: double-buffered LDS, one barrier per slab')
body=s[a:b]
head='''  if constexpr (FAST) {
    // The loop is specialised at compile time on the operand layouts and on "plain" vs "with effects"
    // (affine / dropout / ones-row / ragged last slab) and selected by one switch per workgroup: with
    // every mode behind run-time branches in one loop body the kernel was ~8 % slower (the path taken
    // was a few hundred instructions scattered over a 30 KB body).
    const bool rt_a_kc = P.lda_k == 1, rt_b_kc = P.ldb_k == 1;
    const bool rt_fx = P.a_chan_scale != nullptr || P.b_chan_scale != nullptr || a_dropout || b_dropout ||
                       ones || ((kend - kbeg) % kBK) != 0;
    auto run_fast = [&](auto a_kc_t, auto b_kc_t, auto fx_t) {
    constexpr bool a_kc = decltype(a_kc_t)::value, b_kc = decltype(b_kc_t)::value;
    constexpr bool FX = decltype(fx_t)::value;
    const bool f_ones = FX && ones, f_adrop = FX && a_dropout, f_bdrop = FX && b_dropout;
'''
inner=body[len('  const bool a_kc = P.lda_k == 1, b_kc = P.ldb_k == 1;\n  if constexpr (FAST) {\n'):]
def irep(a,b,cnt=1):
    global inner
    assert inner.count(a)==cnt, (inner.count(a), a)
    inner=inner.replace(a,b)
irep('''    const bool a_aff = P.a_chan_scale != nullptr;   // channel = k (varies per slab): staged in LDS''','''    const bool a_aff = FX && P.a_chan_scale != nullptr;   // channel = k (varies per slab): staged in LDS''')
irep('''    const bool b_aff = P.b_chan_scale != nullptr;''','''    const bool b_aff = FX && P.b_chan_scale != nullptr;''')
irep('''    const int ones_e = !ones ? -1 :''','''    const int ones_e = !f_ones ? -1 :''')
irep('''        if (a_dropout && a_live)''','''        if (f_adrop && a_live)''')
irep('''        if (b_dropout && b_live)''','''        if (f_bdrop && b_live)''')
irep('''      } else if (nx < nslab) {
        fetch_fast(nx, Ragged());
        mfma_slab(sl & 1);
        commit_fast(nx, nx & 1, Ragged());
      } else {''','''      } else if (FX && nx < nslab) {
        fetch_fast(nx, Ragged());
        mfma_slab(sl & 1);
        commit_fast(nx, nx & 1, Ragged());
      } else {''')
irep('''    if (nwhole > 0) {
      fetch_fast(0, Whole());
      commit_fast(0, 0, Whole());
    } else {''','''    if (!FX || nwhole > 0) {
      fetch_fast(0, Whole());
      commit_fast(0, 0, Whole());
    } else {''')
tail='''    };   // run_fast
    typedef std::true_type T_;
    typedef std::false_type F_;
    switch ((rt_a_kc ? 1 : 0) | (rt_b_kc ? 2 : 0) | (rt_fx ? 4 : 0)) {
      case 0: run_fast(F_(), F_(), F_()); break;
      case 1: run_fast(T_(), F_(), F_()); break;
      case 2: run_fast(F_(), T_(), F_()); break;
      case 3: run_fast(T_(), T_(), F_()); break;
      case 4: run_fast(F_(), F_(), T_()); break;
      case 5: run_fast(T_(), F_(), T_()); break;
      case 6: run_fast(F_(), T_(), T_()); break;
      default: run_fast(T_(), T_(), T_()); break;
    }
'''
s=s[:a]+head+inner+tail+s[b:]
rep('#include <stdlib.h>\n','#include <stdlib.h>\n\n#include <type_traits>\n')
open(p,'w').write(s)
