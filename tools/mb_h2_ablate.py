"""What would keeping an MLP chain's intermediate on chip buy?  The emulated (f16x2) forward at the FFN / SDF-decoder shapes, timed in
the library HOISDF_LIB points at (tools/ablate_h2.sh builds: the product, one whose output tile never leaves the CU, one whose row
operand is served by the caches, one with both) - the difference to the product build is ALL a fused chain could remove from that
launch (it would still have to do everything else: the MFMAs, the splits, the LDS traffic, the weight stream)."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hoisdf_amd import ops
dev = "cuda"
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
out = []
for M, N, K, act, p in [(65536, 1024, 256, True, 0.1), (65536, 256, 1024, False, 0.0), (65536, 768, 256, False, 0.0), (65536, 256, 256, False, 0.0),
                        (320000, 512, 512, True, 0.0), (320000, 512, 992, True, 0.0), (320000, 256, 512, True, 0.0)]:
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / math.sqrt(K); b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev); mag = ops._mag_measure(x, K, M, K); ym = torch.zeros(M, device=dev, dtype=torch.int32)
    bits = torch.empty(M, (N + 31) // 32, device=dev, dtype=torch.int32) if act else None
    t = timeit(lambda: ops._gemm_fwd(x, K, W, b, y, N, M, N, K, act, p, 1234, bits, x_mag=mag, y_mag=ym))
    out.append(f"{M}x{N}x{K}{' +relu' if act else ''}{' +dropout' if p else ''}: {t*1e6:7.1f} us {2.0*M*N*K/t/1e12:6.1f} TF")
print(os.environ.get("HOISDF_LIB", "product build"), "|", " | ".join(out))
