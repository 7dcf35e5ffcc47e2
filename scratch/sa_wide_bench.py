"""SA2-4 training backward with / without butd_sa_mid_wide_bwd: time of forward + backward of one level (B = 8)."""
import sys, time, torch
sys.path.insert(0, ".")
from butd_detr_amd import attention_blocks, fused_sa
from butd_detr_amd.pointnet2_modules import PointnetSAModuleVotes
LEVELS = [("SA2", dict(N=2048, C=128, npoint=1024, radius=0.4, nsample=32, mlp=[128, 128, 128, 256])),
          ("SA3", dict(N=1024, C=256, npoint=512, radius=0.8, nsample=16, mlp=[256, 128, 128, 256])),
          ("SA4", dict(N=512, C=256, npoint=256, radius=1.2, nsample=16, mlp=[256, 128, 128, 256]))]
attention_blocks.set_backend("hip")
B = 8
for name, cfg in LEVELS:
    torch.manual_seed(0)
    m = PointnetSAModuleVotes(npoint=cfg["npoint"], radius=cfg["radius"], nsample=cfg["nsample"], mlp=list(cfg["mlp"]),
                              use_xyz=True, normalize_xyz=True).cuda().train()
    xyz = torch.rand(B, cfg["N"], 3, device="cuda") * 2 - 1
    feats = torch.randn(B, cfg["C"], cfg["N"], device="cuda", requires_grad=True)
    probe = torch.randn(B, cfg["mlp"][-1], cfg["npoint"], device="cuda")
    for wide in (False, True, False, True):
        fused_sa._MID_WIDE[0] = wide
        def step():
            y = m(xyz, feats)[1]
            (y * probe).sum().backward()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(30):
            step()
        torch.cuda.synchronize()
        print(f"{name} wide={int(wide)}: {(time.perf_counter() - t) / 30 * 1e3:.3f} ms fwd+bwd", flush=True)
