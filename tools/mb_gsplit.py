"""Split-precision GEMM (csrc/gemm_split.hip) vs the f32 MFMA GEMM: error against fp64 and time per call, per contraction.
usage: python tools/mb_gsplit.py [--check-only]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hoisdf_amd import ops
from hoisdf_amd._lib import lib

dev = torch.device("cuda:0")
SHAPES = [(65536, 512, 992), (65536, 256, 512), (65536, 768, 256), (65536, 1024, 256), (65536, 256, 1024), (49152, 512, 516),
          (16384, 512, 3968)]


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3     # us


def run(M, N, K, check=True):
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    dy = (torch.randn(M, N, generator=g) * 1e-4).to(dev)
    res = {}
    for split in (False, True):
        ops.set_gemm_split(split)
        y, bits = ops._lin_fwd(x, W, b, True, 0.0, 0, True)
        dx = torch.empty(M, K, device=dev)
        ops._lin_bwd_input(dy, bits, 0.0, W, dx, False)
        dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
        ops._lin_bwd_weight(dy, bits, 0.0, x, dW, db)
        res[split] = (y, dx, dW, db)
        t_f = timeit(lambda: ops._lin_fwd(x, W, b, True, 0.0, 0, True))
        t_i = timeit(lambda: ops._lin_bwd_input(dy, bits, 0.0, W, dx, False))
        def bw():
            dW.zero_(); db.zero_()
            ops._lin_bwd_weight(dy, bits, 0.0, x, dW, db)
        t_w = timeit(bw)
        gf = 2.0 * M * N * K / 1e3
        print(f"  {'split' if split else 'f32  '} fwd {t_f:8.1f} us ({gf / t_f / 1e3:6.1f} TF)  dX {t_i:8.1f} us ({gf / t_i / 1e3:6.1f} TF)"
              f"  dW {t_w:8.1f} us ({gf / t_w / 1e3:6.1f} TF)")
    ops.set_gemm_split(False)
    if check:
        rows = slice(0, min(M, 4096))
        y64 = torch.relu(x[rows].double() @ W.double().t() + b.double())
        mask = (y64 > 0).double()
        # bits of the two runs may differ where y ~ 0; compare dx on the f32 run's own mask is not possible across runs, so
        # use fp64's mask and ignore the (rare) rows whose masks differ
        dyd = dy.double()
        dx64 = (dyd[rows] * mask) @ W.double()
        for name, i, ref in (("y", 0, y64), ("dx", 1, dx64)):
            for split in (False, True):
                e = (res[split][i][rows].double() - ref).abs().max().item() / ref.abs().max().item()
                print(f"    {name:3s} {'split' if split else 'f32  '} max err / max |ref| = {e:.2e}")
        yfull = torch.relu(x.double() @ W.double().t() + b.double())
        dye = dyd * (yfull > 0)
        dW64 = dye.t() @ x.double(); db64 = dye.sum(0)
        for name, i, ref in (("dW", 2, dW64), ("db", 3, db64)):
            for split in (False, True):
                e = (res[split][i].double() - ref).abs().max().item() / ref.abs().max().item()
                print(f"    {name:3s} {'split' if split else 'f32  '} max err / max |ref| = {e:.2e}")


if __name__ == "__main__":
    print(lib().hoisdf_version().decode())
    if "--one" in sys.argv:
        SHAPES = SHAPES[:1]
    if "--shape" in sys.argv:
        i = int(sys.argv[sys.argv.index("--shape") + 1])
        SHAPES = SHAPES[i:i + 1]
    for (M, N, K) in SHAPES:
        print(f"M={M} N={N} K={K}")
        run(M, N, K, check="--no-check" not in sys.argv)
