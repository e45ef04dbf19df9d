import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, ctypes as C
from hoisdf_amd._lib import call
dev = "cuda"
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
M = 65536
for N, K in [(512, 992), (256, 256), (1024, 256)]:
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / math.sqrt(K); b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev); bits = torch.zeros(M, (N + 31) // 32, dtype=torch.int32, device=dev)
    fl = 2.0 * M * N * K
    for name, act, bt, dp in [("act=0 nobits", 0, None, 0.0), ("act=1 nobits", 1, None, 0.0), ("act=1 bits", 1, bits, 0.0), ("act=1 bits drop.2", 1, bits, 0.2), ("act=0 nobits", 0, None, 0.0)]:
        t = timeit(lambda: call("hoisdf_linear_fwd", p(x), K, p(W), K, p(b), p(y), N, M, N, K, act, dp, 1234, p(bt), st))
        print(f"N={N} K={K} {name:20s} {fl/t/1e12:6.1f} TF ({t*1e6:6.0f} us)")
