#!/usr/bin/env python3
"""MANO head kernels (csrc/mano.hip): time of the ground-truth launch, the prediction launch with the fused losses, and the backward"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hoisdf_amd import ops as O
from hoisdf_amd.nets import mano as MANO
dev = "cuda"
layer = MANO.ManoLayer(MANO.synthetic_assets(0)).to(dev)
assets = layer.kernel_assets()
L, B = 3, 32
g = torch.Generator().manual_seed(0)
pose = torch.randn(L * B, 16, 6, generator=g).to(dev).requires_grad_(True)
betas = torch.randn(L * B, 10, generator=g).to(dev).requires_grad_(True)
mp = torch.cat([0.4 * torch.randn(B, 48, generator=g), torch.randn(B, 10, generator=g)], 1).to(dev)
def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
gv, gj, gr = O.mano_gt(mp, assets)
pack = (gv, gj, gr, mp[:, 48:])
print(f"ground truth ({B} hands): {timeit(lambda: O.mano_gt(mp, assets)):.1f} us")
print(f"predictions + losses ({L * B} hands): {timeit(lambda: O.mano_head(pose, betas, assets, pack)):.1f} us")
def fb():
    v, j, r, s = O.mano_head(pose, betas, assets, pack)
    s.sum().backward()
t_fb = timeit(fb)
print(f"forward + backward: {t_fb:.1f} us")
