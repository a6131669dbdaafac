import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import raft_oracle as O
from ptlflow_amd.raft import RAFT, GMA
dev = torch.device("cuda", 0)
for cls, small, B, H, W, it in ((RAFT, False, 2, 375, 1242, 8), (RAFT, False, 16, 436, 1024, 4), (RAFT, True, 3, 200, 328, 8), (GMA, False, 1, 184, 320, 8), (RAFT, False, 1, 64, 96, 4), (RAFT, False, 5, 368, 496, 6)):
    kw = dict(iters=it) if cls is GMA else dict(iters=it, small=small)
    a = cls(**kw).load_synthetic(3).eval()
    P = a.state_dict()
    b = cls(conv_precision="bf16", **kw).eval(); b.load_state_dict(P)
    x = {"images": O.smooth_pair(B, H, W, 7).to(dev)}
    fa = a.to(dev)(x)["flows"][:, 0].float().cpu(); fb = b.to(dev)(x)["flows"][:, 0].float().cpu()
    m, mx = O.epe(fb, fa)
    print(cls.__name__, "small" if small else "", B, H, W, it, f"EPE bf16 vs fp32 mean {m:.3e} max {mx:.3e} finite {bool(torch.isfinite(fb).all())} flow scale {fa.abs().mean():.2f}", flush=True)
