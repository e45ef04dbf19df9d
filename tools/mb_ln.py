#!/usr/bin/env python3
"""add + LayerNorm forward + backward at the encoder's shape (65536 x 256, dropout 0.1) for a few block caps of the backward
(HOISDF_LN_BWD_BLOCKS; children):  python tools/mb_ln.py"""
import os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def child():
    import time, torch
    from hoisdf_amd import ops as O
    M, D = 65536, 256
    x = torch.randn(M, D, device="cuda", requires_grad=True)
    r = torch.randn(M, D, device="cuda", requires_grad=True)
    g = torch.ones(D, device="cuda", requires_grad=True)
    b = torch.zeros(D, device="cuda", requires_grad=True)
    gy = torch.randn(M, D, device="cuda")

    def run():
        O.add_layernorm(x, r, g, b, 1e-5, 0.1).backward(gy)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        run()
    torch.cuda.synchronize()
    print("RESULT %.1f us per fwd+bwd" % ((time.perf_counter() - t0) / 50 * 1e6))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(); sys.exit(0)
    for cap in (64, 128, 256, 512, 1024):
        env = dict(os.environ, HOISDF_LN_BWD_BLOCKS=str(cap))
        p = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print(cap, [l for l in p.stdout.splitlines() if l.startswith("RESULT")] or p.stderr[-500:])
