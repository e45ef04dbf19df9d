import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hoisdf_amd import ops as O
dev = "cuda"
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
shapes = [(65536, 1024, 256), (65536, 256, 1024), (65536, 768, 256), (65536, 256, 256), (294912, 256, 256), (49152, 1024, 992), (49152, 512, 512), (49152, 256, 256)]
out = []
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / math.sqrt(K); dy = torch.randn(M, N, device=dev)
    y = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev)
    bits = torch.randint(-2**31, 2**31 - 1, (M, (N + 31) // 32), device=dev, dtype=torch.int32)
    t1 = timeit(lambda: O._gemm_fwd(x, K, W, None, y, N, M, N, K, 0, 0.0, 0, None))
    t2 = timeit(lambda: O._gemm_bwd_input(dy, N, None, 0.0, W, dx, K, M, N, K, 0))
    t3 = timeit(lambda: O._gemm_bwd_input(dy, N, bits, 0.1, W, dx, K, M, N, K, 0))
    out.append(f"{t1*1e6:4.0f}/{t2*1e6:4.0f}/{t3*1e6:4.0f}")
print(os.environ.get("TAG", "?").ljust(6), " ".join(out), flush=True)
