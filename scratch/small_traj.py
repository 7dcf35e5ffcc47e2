"""small-scale check: does the training trajectory of the graphed step equal the eager one (packed optimizer both)?"""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from tests.test_gpu_graph_step import _model
from butd_detr_amd import attention_blocks
from butd_detr_amd.train_step import FlatAdamW, GraphedTrainStep, HungarianCriterion, synthetic_batch
import torch.distributed as dist
dev = torch.device("cuda", 0)
if os.environ.get("SPLIT") == "1":
    os.environ["BUTD_FORCE_COLLECTIVE"] = "1"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
    dist.init_process_group("nccl", init_method="env://", rank=0, world_size=1)
batches = [synthetic_batch(2, dev, seed=900 + 7 * i, n_points=4096, tokens=24) for i in range(3)]
base = _model()                     # every dropout p = 0
N = int(os.environ.get("STEPS", "40"))
def eager():
    m = copy.deepcopy(base); opt = FlatAdamW(m); crit = HungarianCriterion(num_decoder_layers=2)
    out = []
    for it in range(N):
        inp, tgt = batches[it % 3]
        loss = crit(m(inp), crit.prepare(tgt)); opt.zero_grad(); loss.backward(); opt.collect_grads(); opt.clip_(0.1); opt.step(packed=True)
        out.append(float(loss))
    return out
def graph(**kw):
    m = copy.deepcopy(base); step = GraphedTrainStep(m, FlatAdamW(m), criterion=HungarianCriterion(num_decoder_layers=2), warmup=1, **kw)
    out = []
    for it in range(N):
        out.append(step(*batches[it % 3], next_inputs=batches[(it + 1) % 3][0]))
    torch.cuda.synchronize()
    return [float(x) for x in out[-1:]]     # (static loss tensor: only the last value is meaningful without syncing)
e = eager()
print("eager   last losses", [round(x, 4) for x in e[-3:]])
print("graph   default    ", graph())
if os.environ.get("SPLIT") == "1":
    os.environ["BUTD_STEP_SYNC"] = "0"; print("two-piece free-run ", graph(overlap_exchange=True))
    os.environ["BUTD_STEP_SYNC"] = "1"; print("two-piece with wait", graph(overlap_exchange=True))
attention_blocks.set_backend("torch")
