"""which problems of a training step are NOT eligible for the fast GEMM path (python mirror of fast_eligible())"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from butd_detr_amd import fused_attention as fa, fused_mlp, fused_sa
from butd_detr_amd.train_step import synthetic_batch
args = bench.parse()
dev = torch.device("cuda", 0)
model, _ = bench.build_model(args, dev)
crit = bench.make_criterion(args)
inputs, targets = synthetic_batch(args.batch, dev, n_points=args.points, tokens=args.tokens)
targets = crit.prepare(targets)
log = collections.Counter()
orig = fa._gemm
def why(p):
    a_kc, b_kc = p.lda_k == 1, p.ldb_k == 1
    r = []
    if p.a2: r.append("a2")
    if p.K & 3: r.append("K%4")
    if not a_kc and p.M & 3: r.append("M%4")
    if not b_kc and p.N & 3: r.append("N%4")
    if (p.lda_m if a_kc else p.lda_k) & 3: r.append("lda%4")
    if (p.ldb_n if b_kc else p.ldb_k) & 3: r.append("ldb%4")
    if (p.a or 0) & 15: r.append("a unaligned")
    if (p.b or 0) & 15: r.append("b unaligned")
    split = max(p.split_k, 1); per = ((p.K + 31) // 32 + split - 1) // split * 32
    if p.a_chan_scale and per > 320: r.append("a_aff K>320")
    return r
def spy(problems, ref):
    reasons = [why(p) for p in problems]
    if any(reasons):
        site = "?"
        for fr in reversed(traceback.extract_stack(limit=12)):
            if "/butd_detr_amd/" in fr.filename and "fused_attention.py" not in fr.filename.split("/")[-1] or "fused_attention.py" in fr.filename and fr.name != "_gemm" and fr.name != "spy":
                site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"; break
        log[(site, tuple((p.M, p.N, p.K, ",".join(w)) for p, w in zip(problems, reasons)))] += 1
    return orig(problems, ref)
for mod in (fa, fused_mlp, fused_sa):
    if hasattr(mod, "_gemm"): mod._gemm = spy
loss = crit(model(inputs), targets); loss.backward()
for (site, probs), n in sorted(log.items(), key=lambda kv: -kv[1]): print(n, site, probs)
print("launches with a non-fast problem:", sum(log.values()))
