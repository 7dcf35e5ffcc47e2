import sys; sys.path.insert(0,'.')
import copy, torch
from butd_detr_amd.train_step import FlatAdamW
torch.manual_seed(0)
class M(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone_net = torch.nn.Linear(37, 19)
        self.head = torch.nn.Sequential(torch.nn.Linear(19, 7), torch.nn.LayerNorm(7))
    def forward(self, x): return self.head(torch.relu(self.backbone_net(x)))
ref = M().cuda(); mine = copy.deepcopy(ref)
named = list(ref.named_parameters())
opt_ref = torch.optim.AdamW([{"params": [p for n, p in named if "backbone_net" not in n]}, {"params": [p for n, p in named if "backbone_net" in n], "lr": 1e-2}], lr=1e-3, weight_decay=5e-4)
opt = FlatAdamW(mine, lr=1e-3, lr_backbone=1e-2, weight_decay=5e-4)
for step in range(3):
    x = torch.randn(32, 37, device="cuda")
    for m, o in ((ref, opt_ref), (mine, opt)):
        o.zero_grad(set_to_none=True); m(x).pow(2).sum().backward()
    gd = max(float((a.grad-b.grad).abs().max()) for a,b in zip(ref.parameters(), mine.parameters()))
    n1 = torch.nn.utils.clip_grad_norm_([p for g in opt_ref.param_groups for p in g["params"]], 0.1)
    opt_ref.step()
    torch._foreach_copy_(opt.grad_views, [p.grad for p in opt.params])
    n2 = opt.clip_(0.1); opt.step()
    print("step", step, "grad diff before", gd, "norms", float(n1), float(n2), "scale", float(opt.grad_scale))
    for (n, a), (_, b) in zip(ref.named_parameters(), mine.named_parameters()):
        print("   ", n, float((a-b).abs().max()), float(a.abs().max()))
