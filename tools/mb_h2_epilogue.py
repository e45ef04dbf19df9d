"""What does each piece of the emulated forward's epilogue cost?  65536 x 1024 x 256 and 49152 x 512 x 512 with the pieces switched on
one by one (bias is always on)."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hoisdf_amd import ops
dev = "cuda"
def timeit(fn, iters=40, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
for M, N, K in [(65536, 1024, 256), (49152, 512, 512), (65536, 256, 256)]:
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / math.sqrt(K); b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev); mag = ops._mag_measure(x, K, M, K); ym = torch.zeros(M, device=dev, dtype=torch.int32)
    bits = torch.empty(M, (N + 31) // 32, device=dev, dtype=torch.int32)
    row = []
    for name, act, p, bt, om in [("plain", False, 0.0, None, None), ("+mag", False, 0.0, None, ym), ("+relu", True, 0.0, None, None),
                                 ("+relu+bits", True, 0.0, bits, None), ("+relu+drop", True, 0.1, None, None),
                                 ("+relu+drop+bits", True, 0.1, bits, None), ("+relu+drop+bits+mag", True, 0.1, bits, ym)]:
        t = timeit(lambda: ops._gemm_fwd(x, K, W, b, y, N, M, N, K, act, p, 1234, bt, x_mag=mag, y_mag=om))
        row.append(f"{name} {t*1e6:6.1f}")
    print(f"{M}x{N}x{K}: " + " | ".join(row))
