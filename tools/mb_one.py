#!/usr/bin/env python3
"""One shape of the emulated linear forward / grad-input / grad-weight, N iterations each (for rocprofv3 --kernel-trace --stats):
    python tools/mb_one.py M N K [iters] [what=fwd,dx,dw]"""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hoisdf_amd import ops as O

M, N, K = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
what = (sys.argv[5] if len(sys.argv) > 5 else "fwd,dx,dw").split(",")
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn(M, K, device=dev, generator=g); W = torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)
dy = torch.randn(M, N, device=dev, generator=g)
y = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev)
dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
for _ in range(iters):
    if "fwd" in what: O._gemm_fwd(x, K, W, None, y, N, M, N, K, 0, 0.0, 0, None)
    if "dx" in what: O._gemm_bwd_input(dy, N, None, 0.0, W, dx, K, M, N, K, 0)
    if "dw" in what: O._gemm_bwd_weight(dy, N, None, 0.0, x, K, dW, db, M, N, K)
torch.cuda.synchronize()
