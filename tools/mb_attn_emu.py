#!/usr/bin/env python3
"""emulated (bf16x3) attention next to the exact-f32 MFMA kernels: forward and backward time / TF at the step's shapes"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hoisdf_amd import ops
dev = "cuda"
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
for B, Lq, Lk, p in [(32, 2048, 2048, 0.1), (32, 2048, 2048, 0.0), (32, 1536, 2048, 0.1), (32, 512, 2048, 0.1), (4, 8192, 8192, 0.0)]:
    E, H = 256, 4
    q = torch.randn(B, Lq, E, device=dev); kv = torch.randn(B, Lk, 2 * E, device=dev); do = torch.randn(B, Lq, E, device=dev)
    k, v = kv[:, :, :E], kv[:, :, E:]
    dq = torch.empty_like(q); dkv = torch.empty_like(kv)
    fl = 4.0 * B * H * Lq * Lk * 64
    o, lse = ops._attn_fwd(q, k, v, H, Lk, p, 1234)
    t1 = timeit(lambda: ops._attn_fwd(q, k, v, H, Lk, p, 1234))
    t2 = timeit(lambda: ops._attn_bwd(q, k, v, o, lse, do, dq, dkv[:, :, :E], dkv[:, :, E:], H, Lk, p, 1234))
    oe, lsee = ops._attn_fwd_emu(q, k, v, H, Lk, p, 1234, keep=False)
    t3 = timeit(lambda: ops._attn_fwd_emu(q, k, v, H, Lk, p, 1234, keep=False))
    t3k = timeit(lambda: ops._attn_fwd_emu(q, k, v, H, Lk, p, 1234, keep=True))
    def bwd_kept():
        ops._attn_fwd_emu(q, k, v, H, Lk, p, 1234, keep=True)
        ops._attn_bwd_emu(q, k, v, oe, lsee, do, dq, dkv[:, :, :E], dkv[:, :, E:], H, Lk, p, 1234)
    t4 = timeit(lambda: ops._attn_bwd_emu(q, k, v, oe, lsee, do, dq, dkv[:, :, :E], dkv[:, :, E:], H, Lk, p, 1234))
    t5 = timeit(bwd_kept) - t3k
    print(f"B={B} Lq={Lq} Lk={Lk} p={p}: f32 fwd {t1*1e3:.2f} ms ({fl/t1/1e12:.0f} TF) bwd {t2*1e3:.2f} ms ({2.5*fl/t2/1e12:.0f} TF) | "
          f"emu fwd {t3*1e3:.2f} ms ({fl/t3/1e12:.0f}) [keep {t3k*1e3:.2f}] bwd {t4*1e3:.2f} ms ({2.5*fl/t4/1e12:.0f}) [kept planes {t5*1e3:.2f} ms]", flush=True)
