#!/usr/bin/env python3
"""Accuracy of the fp32-emulating linear kernels (csrc/gemm_emu.hip) next to the exact-f32 MFMA kernels and torch's fp32 GEMM
(rocBLAS / hipBLASLt) on identical inputs: element-wise |err vs fp64| / sum_k |a_k||b_k| (max and RMS), forward and
grad-input orientation, several K and two input distributions."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hoisdf_amd import ops as O
DEV = "cuda"
torch.backends.cuda.matmul.allow_tf32 = False
print(f"{'shape':>18s} {'dist':>8s} {'op':>4s} | {'emulated max':>12s} {'rms':>9s} | {'exact-f32 max':>13s} {'rms':>9s} | {'library max':>11s} {'rms':>9s}")
for M, N, K in [(8192, 1024, 256), (4096, 256, 1024), (4096, 512, 992), (2048, 256, 4096)]:
    for dist in ("randn", "decades"):
        g = torch.Generator().manual_seed(K + len(dist))
        x = torch.randn(M, K, generator=g)
        dy = torch.randn(M, N, generator=g)
        if dist == "decades":
            x = x * torch.pow(10.0, -6.0 * torch.rand(M, 1, generator=g)) * torch.pow(10.0, 2.0 * torch.rand(1, K, generator=g) - 1)
            dy = dy * torch.pow(10.0, -6.0 * torch.rand(M, 1, generator=g))
        x, dy = x.to(DEV), dy.to(DEV)
        W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
        res = {}
        for mode in ("emu", "f32"):
            O.set_gemm_emu(mode == "emu")
            y = torch.empty(M, N, device=DEV); dx = torch.empty(M, K, device=DEV)
            O._gemm_fwd(x, K, W, None, y, N, M, N, K, 0, 0.0, 0, None)
            O._gemm_bwd_input(dy, N, None, 0.0, W, dx, K, M, N, K, 0)
            res[mode] = (y, dx)
        res["lib"] = (x @ W.t(), dy @ W)
        xd, Wd, dyd = x.double(), W.double(), dy.double()
        for i, (op, ref, den) in enumerate((("fwd", xd @ Wd.t(), xd.abs() @ Wd.abs().t()), ("dx", dyd @ Wd, dyd.abs() @ Wd.abs()))):
            den = den.clamp_min(1e-300)
            row = []
            for mode in ("emu", "f32", "lib"):
                e = (res[mode][i].double() - ref).abs() / den
                row += [float(e.max()), float((e ** 2).mean().sqrt())]
            print(f"{str((M, N, K)):>18s} {dist:>8s} {op:>4s} | {row[0]:12.2e} {row[1]:9.2e} | {row[2]:13.2e} {row[3]:9.2e} | {row[4]:11.2e} {row[5]:9.2e}")
