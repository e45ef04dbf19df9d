#!/usr/bin/env python3
"""Yardstick, not a product path: the library's plain bf16 GEMM (torch.mm on bf16 tensors -> hipBLASLt; inputs ALREADY bf16, f32
accumulate, bf16 and f32 outputs) at the step's linear-layer shapes.  Six such products make one emulated-fp32 product, so
library TF / 6 is what a conversion-free six-product scheme built from library calls would sustain on this chip: the practical
ceiling to hold emu_kc2 / emu_dw2 (which also convert their operands and apply the epilogue) against."""
import sys, os, math
import torch

dev = "cuda"


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


shapes = [(65536, 1024, 256), (65536, 256, 1024), (65536, 768, 256), (65536, 256, 256), (294912, 256, 256),
          (49152, 1024, 992), (49152, 512, 1024), (65536, 512, 992), (65536, 512, 512), (4096, 4096, 4096), (8192, 8192, 8192)]
print(f"{'M':>7} {'N':>5} {'K':>5} | library bf16 fwd/dX/dW TF (bf16 out) | fwd with f32 out | /6: fwd dX dW")
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    dy = torch.randn(M, N, device=dev).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16); dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    dW = torch.zeros(N, K, device=dev, dtype=torch.bfloat16)
    y32 = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K
    Wt, dyt = W.t(), dy.t()
    l1 = timeit(lambda: torch.mm(x, Wt, out=y))
    l2 = timeit(lambda: torch.mm(dy, W, out=dx))
    l3 = timeit(lambda: torch.mm(dyt, x, out=dW))
    try:
        l4 = timeit(lambda: torch.mm(x, Wt, out_dtype=torch.float32))
    except Exception as e:
        l4 = float("nan")
    f = lambda t: f"{fl / t / 1e12:7.1f}"
    g = lambda t: f"{fl / t / 6e12:6.1f}"
    print(f"{M:7d} {N:5d} {K:5d} | {f(l1)} {f(l2)} {f(l3)} | {f(l4)} | {g(l1)} {g(l2)} {g(l3)}", flush=True)
