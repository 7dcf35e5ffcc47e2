"""round 6: the frozen language model alone (8 utterances x 80 tokens, train mode): stock transformers forward vs
text_stream.fast_language_model -- launches and time per pass under a graph replay"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from butd_detr_amd import attention_blocks, text_stream
from butd_detr_amd.offline_text import HashTokenizer, random_roberta_base, synthetic_utterances
attention_blocks.set_backend("hip")
enc = random_roberta_base(0).cuda().train()
for p in enc.parameters():
    p.requires_grad_(False)
tok = HashTokenizer().batch_encode_plus(synthetic_utterances(8, tokens=80, seed=5)).to("cuda")
for name, fn in (("stock", lambda: enc(**tok).last_hidden_state), ("fast", lambda: text_stream.fast_language_model(enc, tok))):
    with torch.no_grad():
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                out = fn()
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): g.replay()
        b.record(); torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b) / 50 * 1e3:.0f} us per pass")
