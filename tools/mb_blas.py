#!/usr/bin/env python3
"""Yardstick, not a product path: exact-f32 library GEMM (torch.mm -> rocBLAS / hipBLASLt) at the step's linear-layer
shapes next to the hot path's own f32 MFMA kernel (no epilogue: bias / ReLU / bitmap off), forward / grad-input / grad-weight
orientation.  Says what this chip sustains on these shapes under its power cap, i.e. what is left for the hand-written kernel."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
import torch
from hoisdf_amd._lib import call, lib

dev = "cuda"
torch.backends.cuda.matmul.allow_tf32 = False


def timeit(fn, iters=10, warm=3):
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
          (49152, 1024, 992), (49152, 512, 1024), (65536, 512, 992), (65536, 512, 512), (4096, 4096, 4096)]
p = lambda t: C.c_void_p(t.data_ptr())
print(f"{'M':>7} {'N':>5} {'K':>5} | ours fwd/dX/dW TF | library fwd/dX/dW TF")
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) / math.sqrt(K)
    dy = torch.randn(M, N, device=dev)
    y = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev); dW = torch.zeros(N, K, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fl = 2.0 * M * N * K
    nws = lib().hoisdf_linear_bwd_weight_workspace(M, N, K)
    ws = torch.empty(max(nws, 1), device=dev)
    t1 = timeit(lambda: call("hoisdf_linear_fwd", p(x), K, p(W), K, None, p(y), N, M, N, K, 0, 0.0, 0, None, st))
    t2 = timeit(lambda: call("hoisdf_linear_bwd_input", p(dy), N, None, 0.0, p(W), K, p(dx), K, M, N, K, 0, st))
    t3 = timeit(lambda: call("hoisdf_linear_bwd_weight", p(dy), N, None, 0.0, p(x), K, p(dW), K, None, M, N, K, p(ws), nws, st))
    Wt = W.t()
    l1 = timeit(lambda: torch.mm(x, Wt, out=y))
    l2 = timeit(lambda: torch.mm(dy, W, out=dx))
    dyt = dy.t()
    l3 = timeit(lambda: torch.mm(dyt, x, out=dW))
    f = lambda t: f"{fl / t / 1e12:6.1f}"
    print(f"{M:7d} {N:5d} {K:5d} | {f(t1)} {f(t2)} {f(t3)} | {f(l1)} {f(l2)} {f(l3)}", flush=True)
