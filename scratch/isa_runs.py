"""Run-length sequence of instruction classes of a kernel's basic blocks in an ISA listing (hipcc -S): where matrix and
vector instructions alternate.  On gfx950 an fp32 matrix instruction and the vector ALU never overlap and every
matrix -> vector -> matrix alternation costs ~12 cycles (profiles/r05_mfma_valu_interleave.txt).
    python scratch/isa_runs.py file.s kernel_substring [min_block_instructions]"""
import re, sys, collections
path, needle = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 40
txt = open(path).read()
names = [m.group(1) for m in re.finditer(r"^(_Z\w+):", txt, re.M) if needle in m.group(1)]
name = names[0]
start = txt.index(name + ":"); end = txt.index(".Lfunc_end", start)
def cls(op):
    if op.startswith("v_mfma"): return "M"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")): return "T"
    if op.startswith("v_pk_"): return "P"
    if op.startswith("v_accvgpr"): return "a"
    if op.startswith("v_"): return "V"
    if op.startswith(("ds_read", "ds_load")): return "L"
    if op.startswith(("ds_write", "ds_store")): return "W"
    if op.startswith(("global_load", "buffer_load")): return "G"
    if op.startswith(("global_store", "global_atomic")): return "g"
    if op.startswith("s_waitcnt"): return "w"
    if op.startswith("s_barrier"): return "B"
    if op.startswith("s_nop"): return "n"
    if op.startswith("s_"): return "s"
    return "?"
blocks = []; cur = ["entry", [], False]
for line in txt[start:end].splitlines():
    l = line.strip()
    m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
    if m:
        blocks.append(cur); cur = [m.group(1), [], "Loop" in l or "Inner" in l]; continue
    if not l or l.startswith((";", ".")): continue
    cur[1].append(cls(l.split()[0]))
blocks.append(cur)
print(name[:100])
for label, seq, loop in blocks:
    if len(seq) < minlen: continue
    c = collections.Counter(seq)
    # alternations: matrix run followed (ignoring LDS / scalar / waits) by vector work followed by matrix again
    core = [x for x in seq if x in "MVTPa"]
    alt = sum(1 for i in range(1, len(core)) if (core[i] == "M") != (core[i - 1] == "M"))
    rl = re.sub(r"(.)\1*", lambda m: f"{m.group(1)}{len(m.group(0))} " if len(m.group(0)) > 1 else m.group(1) + " ", "".join(seq))
    print(f"{label} {'LOOP' if loop else ''} n={len(seq)} {dict(c)} matrix<->vector switches={alt}")
    print("   ", rl[:1800])
