"""Does the row stride of the SDF decoder's concatenated row matter to the emulated (f16x2) forward?  M x 512 x K for the decoder's
ragged contractions (292 of a 516-wide row, the 516-wide row itself) at several leading dimensions, next to 512 x 512."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hoisdf_amd import ops
dev = "cuda"
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
N = 512
for M in (49152, 320000):
    for K, ld, off in [(512, 512, 0), (516, 516, 0), (516, 528, 0), (516, 544, 0), (516, 576, 0), (528, 528, 0), (544, 544, 0),
                       (292, 516, 224), (292, 528, 224), (292, 576, 256), (304, 576, 256), (320, 576, 256)]:
        xb = torch.randn(M, ld, device=dev); x = xb[:, off:off + K]
        W = torch.randn(N, K, device=dev) / math.sqrt(K); b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev)
        mag = ops._mag_measure(x, ld, M, K)
        fl = 2.0 * M * N * K
        t1 = timeit(lambda: ops._gemm_fwd(x, ld, W, b, y, N, M, N, K, True, 0.0, 0, None, x_mag=mag))
        print(f"M={M:7d} K={K:4d} ld={ld:4d} col0={off:3d} | fwd {fl/t1/1e12:6.1f} TF ({t1*1e6:6.0f} us)")
