"""static instruction-class counts per basic block of one kernel in an ISA listing (hipcc -S); loop blocks marked"""
import re, collections, sys
path, needle = sys.argv[1], sys.argv[2]
txt = open(path).read()
names = [m.group(1) for m in re.finditer(r"^(_Z\w+):", txt, re.M) if needle in m.group(1)]
name = names[0]
start = txt.index(name + ":"); end = txt.index(".Lfunc_end", start)
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith(("ds_read", "ds_load")): return "dsr"
    if op.startswith(("ds_write", "ds_store")): return "dsw"
    if op.startswith(("global_load", "buffer_load")): return "vld"
    if op.startswith(("global_store", "global_atomic")): return "vst"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "bar"
    if op.startswith(("s_cbranch", "s_branch")): return "br"
    if op.startswith("s_"): return "salu"
    return "other"
tot = collections.Counter(); cur = None; quiet = len(sys.argv) > 3
for line in txt[start:end].splitlines():
    l = line.strip()
    m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
    if m:
        if cur and cur[2] and not quiet: print(cur[0], "LOOP" if cur[3] else "", cur[2], dict(cur[1]))
        cur = [m.group(1), collections.Counter(), 0, "Loop" in l]; continue
    if not l or l.startswith((";", ".")): continue
    if cur is None: cur = ["entry", collections.Counter(), 0, False]
    op = l.split()[0]; cur[1][cls(op)] += 1; cur[2] += 1
    if cur[3]: tot[cls(op)] += 1
print(name[:80]); print("loop totals (static)", dict(tot))
m = re.search(re.escape(name) + r".*?\.vgpr_count:\s+(\d+)", txt[txt.index(".amdgpu_metadata"):], re.S)
