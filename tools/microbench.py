#!/usr/bin/env python3
"""Per-shape timing of the C-ABI kernels on one GPU (HIP events). Usage: python tools/microbench.py [gemm|attn|all]"""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hoisdf_amd import ops

dev = "cuda"


def timeit(fn, iters=20, warm=3):
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


def gemm():
    shapes = [(65536, 512, 992), (65536, 256, 512), (65536, 1024, 992), (65536, 512, 1024), (65536, 256, 512),
              (65536, 223, 256), (65536, 512, 289), (65536, 223, 512), (65536, 512, 512), (65536, 768, 256),
              (65536, 256, 256), (65536, 1024, 256), (65536, 256, 1024), (294912, 256, 256), (294912, 60, 256),
              (294912, 20, 256), (49152, 3, 256), (544, 256, 256), (544, 1024, 256), (544, 256, 1024), (544, 512, 256)]
    print(f"{'M':>7} {'N':>5} {'K':>5} | fwd TF   dX TF    dW TF")
    for M, N, K in shapes:
        x = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) / math.sqrt(K)
        b = torch.randn(N, device=dev)
        dy = torch.randn(M, N, device=dev); bits = torch.zeros(M, (N + 31) // 32, dtype=torch.int32, device=dev)
        fl = 2.0 * M * N * K
        y = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev); dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
        from hoisdf_amd._lib import call
        import ctypes as C
        p = lambda t: C.c_void_p(t.data_ptr())
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        t1 = timeit(lambda: call("hoisdf_linear_fwd", p(x), K, p(W), K, p(b), p(y), N, M, N, K, 1, 0.0, 0, p(bits), st))
        from hoisdf_amd import _lib
        nws = _lib.lib().hoisdf_linear_bwd_weight_workspace(M, N, K)
        ws = torch.empty(max(nws, 1), device=dev)
        t2 = timeit(lambda: call("hoisdf_linear_bwd_input", p(dy), N, p(bits), 0.0, p(W), K, p(dx), K, M, N, K, 0, st))
        t3 = timeit(lambda: call("hoisdf_linear_bwd_weight", p(dy), N, p(bits), 0.0, p(x), K, p(dW), K, p(db), M, N, K, p(ws), nws, st))
        print(f"{M:7d} {N:5d} {K:5d} | {fl/t1/1e12:6.1f}  {fl/t2/1e12:6.1f}  {fl/t3/1e12:6.1f}   ({t1*1e6:7.0f} {t2*1e6:7.0f} {t3*1e6:7.0f} us)")


def attn():
    for B, S, p in [(32, 2048, 0.0), (32, 2048, 0.1), (32, 800, 0.1), (8, 8192, 0.0)]:
        E, H = 256, 4
        qkv = torch.randn(B, S, 3 * E, device=dev)
        do = torch.randn(B, S, E, device=dev)
        fl = 4.0 * B * H * S * S * 64
        o, lse = ops._attn_fwd(qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:], H, S, p, 1234)
        d = torch.empty_like(qkv)
        t1 = timeit(lambda: ops._attn_fwd(qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:], H, S, p, 1234), iters=5)
        t2 = timeit(lambda: ops._attn_bwd(qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:], o, lse, do, d[:, :, :E], d[:, :, E:2 * E], d[:, :, 2 * E:], H, S, p, 1234), iters=5)
        if p == 0.0:
            with torch.no_grad():
                t3 = timeit(lambda: ops._attn_fwd_f16(qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:], H, S), iters=5)
            print(f"attn B={B} S={S} f16-MFMA fwd (incl. K/V conversion): {fl/t3/1e12:6.1f} TF ({t3*1e3:.2f} ms)")
        print(f"attn B={B} S={S} p={p}: fwd {fl/t1/1e12:6.1f} TF ({t1*1e3:.2f} ms)  bwd {2.5*fl/t2/1e12:6.1f} TF algorithmic ({t2*1e3:.2f} ms)")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("gemm", "all"):
        gemm()
    if which in ("attn", "all"):
        attn()
