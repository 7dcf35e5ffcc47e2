"""node kinds of the captured forward+backward graph (hipGraph debug dump), to find memcpy nodes that read HOST memory"""
import os, sys, re, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, synthetic_batch
args = bench.parse()
dev = torch.device("cuda", 0)
model, _ = bench.build_model(args, dev)
crit = bench.make_criterion(args)
orig = torch.cuda.CUDAGraph
made = []
class G(orig):
    def __new__(cls, *a, **k):
        g = orig.__new__(cls, *a, **k); made.append(g); return g
torch.cuda.CUDAGraph = G
step = GraphedTrainStep(model, FlatAdamW(model), criterion=crit, overlap_exchange=os.environ.get("OVERLAP") == "1")
for g in made: pass
b = synthetic_batch(args.batch, dev, n_points=args.points, tokens=args.tokens)
# enable debug mode before capture: patch capture_begin
beg = orig.capture_begin
def capture_begin(self, *a, **k):
    self.enable_debug_mode(); return beg(self, *a, **k)
orig.capture_begin = capture_begin
step(*b, next_inputs=b[0]); torch.cuda.synchronize()
for i, g in enumerate(made):
    path = f"/tmp/graph_{i}.dot"
    try:
        g.debug_dump(path)
        txt = open(path).read()
        kinds = collections.Counter(re.findall(r'label="[^"]*?(MEMCPY|MEMSET|KERNEL|EMPTY|HOST|EVENT|memcpy|memset|kernel)', txt))
        print(f"graph {i}: {len(txt)} bytes, kinds {dict(kinds)}")
        for m in re.finditer(r'label="([^"]*(?:MEMCPY|memcpy|Memcpy)[^"]*)"', txt):
            print("   ", m.group(1)[:200].replace("\\n", " | "))
    except Exception as e:
        print("dump failed:", repr(e)[:200])
