#!/usr/bin/env python3
"""GEMM family at the training step's shapes x workgroups-per-CU (HOISDF_GEMM_OCC = 4 / 3 / 2): interleaved medians of
forward (bias + ReLU + sign bits), masked grad-input and masked grad-weight.  Feeds pick_occupancy() in csrc/gemm.hip."""
import sys, os, math, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, ctypes as C
from hoisdf_amd._lib import call

dev = "cuda"
p = lambda t: C.c_void_p(t.data_ptr())


def t_once(fn, iters=4):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


SHAPES = [(65536, 1024, 256), (65536, 256, 1024), (65536, 768, 256), (65536, 256, 256), (294912, 256, 256), (49152, 512, 512),
          (49152, 512, 992), (49152, 1024, 992), (49152, 512, 289), (16384, 512, 512), (49152, 256, 512), (49152, 512, 256),
          (16384, 512, 992), (49152, 256, 256), (49152, 223, 512), (294912, 60, 256), (294912, 20, 256), (98304, 256, 256),
          (16384, 256, 256), (16384, 1024, 992), (65536, 512, 256)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
print(f"{'M':>7} {'N':>5} {'K':>5} {'tiles':>6} | fwd TF @occ 4/3/2      | dX TF @occ 4/3/2       | dW TF @occ 4/3/2")
for M, N, K in SHAPES:
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / math.sqrt(K); b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev); bits = torch.randint(-2**31, 2**31 - 1, (M, (N + 31) // 32), dtype=torch.int32, device=dev)
    y = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev); dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    fl = 2.0 * M * N * K
    fns = [lambda: call("hoisdf_linear_fwd", p(x), K, p(W), K, p(b), p(y), N, M, N, K, 1, 0.0, 0, p(bits), st),
           lambda: call("hoisdf_linear_bwd_input", p(dy), N, p(bits), 0.0, p(W), K, p(dx), K, M, N, K, 0, st),
           lambda: call("hoisdf_linear_bwd_weight", p(dy), N, p(bits), 0.0, p(x), K, p(dW), K, p(db), M, N, K, None, 0, st)]
    res = {(f, o): [] for f in range(3) for o in (4, 3, 2)}
    for r in range(5):
        for o in (4, 3, 2):
            os.environ["HOISDF_GEMM_OCC"] = str(o)
            for f in range(3):
                res[(f, o)].append(t_once(fns[f]))
    row = []
    for f in range(3):
        row.append(" ".join(f"{fl / statistics.median(res[(f, o)]) / 1e12:6.1f}" for o in (4, 3, 2)))
    tiles = -(-M // 128) * -(-N // 128)
    print(f"{M:7d} {N:5d} {K:5d} {tiles:6d} | {row[0]}   | {row[1]}   | {row[2]}", flush=True)
