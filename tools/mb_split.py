"""split-precision attention at the bench shape (for rocprofv3 --kernel-trace --stats)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hoisdf_amd import ops
B, S, E, H, p = 32, 2048, 256, 4, 0.1
qkv = torch.randn(B, S, 3 * E, device="cuda"); do = torch.randn(B, S, E, device="cuda"); d = torch.empty_like(qkv)
q, k, v = qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:]
for _ in range(5):
    o, lse = ops._attn_fwd_split(q, k, v, H, S, p, 1234)
    ops._attn_bwd_split(q, k, v, o, lse, do, d[:, :, :E], d[:, :, E:2 * E], d[:, :, 2 * E:], H, S, p, 1234)
torch.cuda.synchronize()
