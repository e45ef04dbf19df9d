"""attention fwd/bwd TF vs (B, S): is the S = 2048 shortfall a per-workgroup fixed cost or a cache effect?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hoisdf_amd import ops
dev = "cuda"
def timeit(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters * 1e-3
E, H = 256, 4
for B, S in [(32, 2048), (32, 2048), (8, 2048), (128, 2048), (32, 1024), (128, 1024), (16, 4096), (8, 8192), (2, 16384), (32, 512), (256, 512)]:
    qkv = torch.randn(B, S, 3 * E, device=dev); do = torch.randn(B, S, E, device=dev)
    fl = 4.0 * B * H * S * S * 64
    o, lse = ops._attn_fwd(qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:], H, S, 0.0, 1)
    d = torch.empty_like(qkv)
    t1 = timeit(lambda: ops._attn_fwd(qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:], H, S, 0.0, 1))
    t2 = timeit(lambda: ops._attn_bwd(qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:], o, lse, do, d[:, :, :E], d[:, :, E:2 * E], d[:, :, 2 * E:], H, S, 0.0, 1))
    print(f"B={B:4d} S={S:6d}: fwd {fl/t1/1e12:6.1f} TF ({t1*1e3:7.2f} ms)  bwd {2.5*fl/t2/1e12:6.1f} TF ({t2*1e3:7.2f} ms)")
