import sys, os, math, torch
sys.path.insert(0, "/root/repo")
from hoisdf_amd import ops as O
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(1)
for (M, N, K, p) in [(4096, 512, 992, 0.0), (4096, 512, 512, 0.2), (4096, 64, 256, 0.0), (300, 128, 128, 0.1)]:
    x = torch.randn(M, K, device=dev, generator=g); W = torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)
    dy = torch.randn(M, N, device=dev, generator=g)
    bits = torch.randint(-2**31, 2**31 - 1, (M, (N + 31) // 32), device=dev, dtype=torch.int32)
    dx = torch.empty(M, K, device=dev)
    O._gemm_bwd_input(dy, N, bits, p, W, dx, K, M, N, K, 0)
    b = bits.view(torch.int32)
    idx = torch.arange(N, device=dev)
    keep = ((b[:, idx // 32] >> (idx % 32)) & 1).double()
    ref = (dy.double() * keep / (1 - p)) @ W.double()
    err = ((dx.double() - ref).abs().max() / ref.abs().max()).item()
    bad = ((dx.double() - ref).abs() > 1e-4 * ref.abs().max())
    print(M, N, K, p, "rel err", err, "bad rows", bad.any(1).sum().item(), "bad cols", bad.any(0).sum().item(), flush=True)
    if bad.any():
        r = bad.any(1).nonzero()[:8].flatten().tolist(); c = bad.any(0).nonzero()[:8].flatten().tolist()
        print("  first bad rows", r, "cols", c)
