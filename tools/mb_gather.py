"""project_gather fwd/bwd timing at the bench shape (B=32, 2048 points, small pyramid C=992)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hoisdf_amd import ops, testing as T
dev = "cuda"
B, P = 32, 2048
pyr = T.synthetic_pyramid(B, seed=1)
levels = [v.to(dev).permute(0, 2, 3, 1).contiguous().requires_grad_(True) for v in pyr.values()]
ins, _, meta = T.synthetic_batch(B, P // 2, P // 2, seed=2)
pts = torch.cat([ins["hand_sdf_points"], ins["obj_sdf_points"]], 1).to(dev)
ctr, K = meta["mano_root"].to(dev), meta["cam_intr"].to(dev)
def run():
    for l in levels: l.grad = None
    feat, _ = ops.project_gather(ops.PyramidNHWC(levels), pts, ctr, K, 3.1)
    return feat
feat = run(); g = torch.randn_like(feat)
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters
tf = timeit(run)
def fb():
    f = run(); f.backward(g)
tfb = timeit(fb)
print(f"gather fwd {tf*1e3:.0f} us, fwd+bwd {tfb*1e3:.0f} us (bwd incl. zero-init of level grads ~ {(tfb - tf)*1e3:.0f} us)")
