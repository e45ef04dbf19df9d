"""Driver for PMC collection on the dominant kernel family (attention backward at the bench shape)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hoisdf_amd import ops
dev = "cuda"
B, S, E, H, p = 32, 2048, 256, 4, 0.1
qkv = torch.randn(B, S, 3 * E, device=dev)
do = torch.randn(B, S, E, device=dev)
o, lse = ops._attn_fwd(qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:], H, S, p, 1234)
d = torch.empty_like(qkv)
for _ in range(3):
    ops._attn_fwd(qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:], H, S, p, 1234)
    ops._attn_bwd(qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:], o, lse, do, d[:, :, :E], d[:, :, E:2 * E], d[:, :, 2 * E:], H, S, p, 1234)
torch.cuda.synchronize()
