#!/usr/bin/env python3
"""small-M linear layers: the one-wave-per-tile emulated kernels (hoisdf_linear_*_emu_small) next to the exact-f32 tiled kernel
at the decoder stack's shapes: microseconds per call (fwd / grad-input / grad-weight), back-to-back launches on one stream."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hoisdf_amd import ops as O

dev = "cuda"


def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for M, N, K in [(544, 256, 256), (544, 1024, 256), (544, 256, 1024), (544, 512, 256), (32, 256, 256), (1536, 256, 256)]:
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / math.sqrt(K); b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev); y = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev)
    dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    bits = torch.empty(M, (N + 31) // 32, dtype=torch.int32, device=dev)
    row = []
    for small in (True, False):
        O._EMU_SMALL = small
        t1 = timeit(lambda: O._gemm_fwd(x, K, W, b, y, N, M, N, K, 1, 0.1, 1234, bits))
        t2 = timeit(lambda: O._gemm_bwd_input(dy, N, bits, 0.1, W, dx, K, M, N, K, 0))
        t3 = timeit(lambda: O._gemm_bwd_weight(dy, N, bits, 0.1, x, K, dW, db, M, N, K))
        row.append((t1, t2, t3))
    print(f"{M:5d} {N:5d} {K:5d} | small fwd/dx/dW {row[0][0]:6.1f} {row[0][1]:6.1f} {row[0][2]:6.1f} us | f32 tiled {row[1][0]:6.1f} {row[1][1]:6.1f} {row[1][2]:6.1f} us", flush=True)
