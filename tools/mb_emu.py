#!/usr/bin/env python3
"""per-shape timing of the emulated (bf16x3) linear kernels next to the exact-f32 MFMA kernels: fwd / grad-input / grad-weight"""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hoisdf_amd import ops as O
dev = "cuda"
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
shapes = [(294912, 60, 256), (294912, 20, 256), (65536, 1024, 256), (65536, 256, 1024), (65536, 768, 256), (65536, 256, 256), (294912, 256, 256), (49152, 1024, 992),
          (49152, 512, 512), (49152, 256, 256), (16384, 256, 256), (16384, 1024, 992), (49152, 512, 256)]
print(f"{'M':>7} {'N':>5} {'K':>5} | emulated fwd / dX / dW TF-eq (us) | exact-f32 fwd / dX / dW TF")
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / math.sqrt(K); dy = torch.randn(M, N, device=dev)
    y = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev); dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    fl = 2.0 * M * N * K
    row = []
    for emu in (True, False):
        O.set_gemm_emu(emu)
        t1 = timeit(lambda: O._gemm_fwd(x, K, W, None, y, N, M, N, K, 0, 0.0, 0, None))
        t2 = timeit(lambda: O._gemm_bwd_input(dy, N, None, 0.0, W, dx, K, M, N, K, 0))
        t3 = timeit(lambda: O._gemm_bwd_weight(dy, N, None, 0.0, x, K, dW, db, M, N, K))
        row.append((t1, t2, t3))
    e, f = row
    print(f"{M:7d} {N:5d} {K:5d} | {fl/e[0]/1e12:6.1f} {fl/e[1]/1e12:6.1f} {fl/e[2]/1e12:6.1f}  ({e[0]*1e6:5.0f} {e[1]*1e6:5.0f} {e[2]*1e6:5.0f}) | {fl/f[0]/1e12:6.1f} {fl/f[1]/1e12:6.1f} {fl/f[2]/1e12:6.1f}", flush=True)
