"""Does the row stride of x matter?  linear fwd/dX/dW at M=65536, N=512 for several K and leading dims."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, ctypes as C
from hoisdf_amd._lib import call
dev = "cuda"
p = lambda t: C.c_void_p(t.data_ptr())
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
for N in (512, 1024):
    for K, ld in [(960, 960), (992, 992), (992, 1024), (992, 1056), (1008, 1008), (1024, 1024), (1024, 1056), (1056, 1056), (1088, 1088)]:
        xb = torch.randn(M, ld, device=dev); x = xb[:, :K]
        W = torch.randn(N, K, device=dev) / math.sqrt(K); b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev); dy = torch.randn(M, N, device=dev); dxb = torch.empty(M, ld, device=dev)
        dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
        fl = 2.0 * M * N * K
        t1 = timeit(lambda: call("hoisdf_linear_fwd", p(x), ld, p(W), K, p(b), p(y), N, M, N, K, 1, 0.0, 0, None, st))
        t2 = timeit(lambda: call("hoisdf_linear_bwd_input", p(dy), N, None, 0.0, p(W), K, p(dxb), ld, M, N, K, 0, st))
        t3 = timeit(lambda: call("hoisdf_linear_bwd_weight", p(dy), N, None, 0.0, p(x), ld, p(dW), K, p(db), M, N, K, None, 0, st))
        print(f"N={N:5d} K={K:5d} ld={ld:5d} | fwd {fl/t1/1e12:6.1f} TF ({t1*1e6:6.0f} us)  dX {fl/t2/1e12:6.1f} ({t2*1e6:6.0f})  dW {fl/t3/1e12:6.1f} ({t3*1e6:6.0f})")
