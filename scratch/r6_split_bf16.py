"""round 6, review item 3: split-bf16 ("bf16 x 3") products with fp32 accumulation for the attention core -- ONE
microbenchmark, then decide.  butd_attention_fwd_split_bf16 (csrc/attention_ops.hip, attn_fwd_h_kernel<9, 3, 1, SPLIT>):
a = a_hi + a_lo in bf16 while staging, S = a_hi b_hi + a_lo b_hi + a_hi b_lo on v_mfma_f32_16x16x32_bf16, the same for P V.
  (i)  time per launch (graph replay, 20 launches per replay) of the fp32 / bf16 / split forward kernels at the step's sites;
  (ii) error of each against a float64 evaluation of the same attention (and of split vs the fp32 kernel);
  (iii) the encoder golden (tests/golden/encoder_small.npz, the REFERENCE's vectors) with the forward core on split-bf16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from butd_detr_amd import _hiplib, attention_blocks, fused_attention as fa
lib = _hiplib.load()
dev = torch.device("cuda", 0)
H, D = 8, 36
E = H * D
ctr = fa.rng_counter(dev).data_ptr()


def run(fn, B, Lq, Lk, q, k, v, out, lse, p):
    return fn(B, H, Lq, Lk, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), None, out.data_ptr(), lse.data_ptr(), p, 7, ctr,
              torch.cuda.current_stream().cuda_stream)


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g):
            for _ in range(reps): fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


def truth(q, k, v):
    qd, kd, vd = (t.double().view(t.shape[0], t.shape[1], H, D).transpose(1, 2) for t in (q, k, v))
    p = torch.softmax(qd @ kd.transpose(-1, -2), -1)
    return (p @ vd).transpose(1, 2).reshape(q.shape[0], q.shape[1], E)


print("# (i) + (ii): forward core, B = 8, 8 heads x 36, q pre-scaled by 1/6, N(0,1) operands; us per launch (graph replay), dropout 0.1;")
print("#     max / mean error of the dropout-free output against float64, in units of max|truth|")
print("#   Lq x Lk      fp32 us   bf16 us  split us   fp32/split   err fp32 (max mean)    err bf16 (max mean)     err split (max mean)    split vs fp32 kernel (max)")
kern = {"fp32": lib.butd_attention_fwd, "bf16": lib.butd_attention_fwd_bf16, "split": lib.butd_attention_fwd_split_bf16}
for Lq, Lk in ((1024, 1024), (256, 1024), (1024, 132), (1024, 80), (256, 256), (256, 132), (256, 80)):
    B = 8
    torch.manual_seed(Lq + Lk)
    q = torch.randn(B, Lq, E, device=dev) / 6.0
    k, v = torch.randn(B, Lk, E, device=dev), torch.randn(B, Lk, E, device=dev)
    out, lse = torch.empty(B, Lq, E, device=dev), torch.empty(B, H, Lq, device=dev)
    t = {n: timed(lambda f=f: run(f, B, Lq, Lk, q, k, v, out, lse, 0.1)) for n, f in kern.items()}
    ref = truth(q, k, v)
    scale = float(ref.abs().max())
    errs, outs = {}, {}
    for n, f in kern.items():
        assert run(f, B, Lq, Lk, q, k, v, out, lse, 0.0) == 0
        torch.cuda.synchronize()
        outs[n] = out.clone()
        e = (out.double() - ref).abs() / scale
        errs[n] = (float(e.max()), float(e.mean()))
    sv = float((outs["split"] - outs["fp32"]).abs().max()) / scale
    print(f"  {Lq:4d} x {Lk:4d}  {t['fp32']:8.1f}  {t['bf16']:8.1f}  {t['split']:8.1f}   {t['fp32'] / t['split']:8.2f}x     "
          f"{errs['fp32'][0]:.2e} {errs['fp32'][1]:.2e}   {errs['bf16'][0]:.2e} {errs['bf16'][1]:.2e}    "
          f"{errs['split'][0]:.2e} {errs['split'][1]:.2e}     {sv:.2e}")

# (iii) the encoder golden with the forward core on split-bf16 operands (backward: the fp32 kernels, from the saved q, k, v)
from tests.golden import weights
from tests.golden.cases import encoder_inputs
from butd_detr_amd.encoder_decoder_layers import BiEncoder, BiEncoderLayer
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "encoder_small.npz"))
attention_blocks.set_backend("hip")
layer = BiEncoderLayer(288, dropout=0.1, activation="relu", n_heads=8, dim_feedforward=256, self_attend_lang=True,
                       self_attend_vis=True, use_butd_enc_attn=True)
model = weights.fill_(BiEncoder(layer, 3), seed=11).cuda().eval()
inp = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in encoder_inputs().items()}
print("# (iii) encoder_small.npz (3 BiEncoder layers, the reference's vectors): max / mean error of the outputs in units of max|reference|")
for name, flag in (("fp32 core", False), ("split-bf16 core", True)):
    fa._split_fwd[0] = flag
    with torch.no_grad():
        vis_out, text_out = model(inp["vis"], inp["pos"], inp["vis_mask"], inp["text"], inp["text_mask"], {},
                                  detected_feats=inp["boxes"], detected_mask=inp["box_mask"])
    fa._split_fwd[0] = False
    for key, t in (("vis_out", vis_out), ("text_out", text_out)):
        e = np.abs(t.float().cpu().numpy() - g[key]) / np.abs(g[key]).max()
        print(f"    {name:16s} {key:9s} {e.max():.2e} max  {e.mean():.2e} mean")
