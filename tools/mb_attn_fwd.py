#!/usr/bin/env python3
"""emulated attention forward only: time at the step's shape, per library variant (HOISDF_LIB) / form (HOISDF_EMU_ATTN_FWD)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hoisdf_amd import ops
dev = "cuda"
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
out = []
for B, Lq, Lk, p in [(32, 2048, 2048, 0.1), (32, 2048, 2048, 0.0), (16, 4096, 4096, 0.0)]:
    E, H = 256, 4
    q = torch.randn(B, Lq, E, device=dev); kv = torch.randn(B, Lk, 2 * E, device=dev)
    k, v = kv[:, :, :E], kv[:, :, E:]
    fl = 4.0 * B * H * Lq * Lk * 64
    t = min(timeit(lambda: ops._attn_fwd_emu(q, k, v, H, Lk, p, 1234, keep=True)) for _ in range(3))
    out.append(f"{t*1e3:.3f} ms ({fl/t/1e12:.0f} TF)")
print(os.environ.get("LABEL", ""), " | ".join(out))
