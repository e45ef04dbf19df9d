#!/usr/bin/env python3
"""the emulated attention backward alone (kept planes), for a rocprofv3 kernel trace: B = 32, S = 2048, dropout 0.1"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hoisdf_amd import ops
dev = "cuda"
B, Lq, Lk, p, E, H = 32, 2048, 2048, 0.1, 256, 4
q = torch.randn(B, Lq, E, device=dev); kv = torch.randn(B, Lk, 2 * E, device=dev); do = torch.randn(B, Lq, E, device=dev)
k, v = kv[:, :, :E], kv[:, :, E:]
dq = torch.empty_like(q); dkv = torch.empty_like(kv)
for _ in range(int(__import__("os").environ.get("ITERS", "6"))):
    oe, lsee = ops._attn_fwd_emu(q, k, v, H, Lk, p, 1234, keep=True)
    ops._attn_bwd_emu(q, k, v, oe, lsee, do, dq, dkv[:, :, :E], dkv[:, :, E:], H, Lk, p, 1234)
torch.cuda.synchronize()
