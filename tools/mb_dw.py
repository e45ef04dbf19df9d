import os, sys
sys.path.insert(0, "/root/repo")
import torch
from hoisdf_amd import ops
from hoisdf_amd._lib import lib
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (M, N, K) in [(65536, 256, 256), (49152, 256, 256), (49152, 512, 256), (294912, 256, 256), (65536, 768, 256), (65536, 1024, 256), (65536, 256, 1024)]:
    x = torch.randn(M, K, device=dev); dy = torch.randn(M, N, device=dev) * 1e-3
    dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    def atomic():
        dW.zero_(); db.zero_()
        ops._lin_bwd_weight(dy, None, 0.0, x, dW, db)
    nws = lib().hoisdf_linear_bwd_weight_workspace(M, N, K)
    ws = torch.empty(max(nws, 1), device=dev)
    def wsp():
        ops.call("hoisdf_linear_bwd_weight", ops._p(dy), N, None, 0.0, ops._p(x), K, ops._p(dW), K, ops._p(db), M, N, K, ops._p(ws), nws, ops._st())
    ta, tw = timeit(atomic), timeit(wsp)
    gf = 2.0 * M * N * K / 1e3
    print(f"{M}x{N}x{K}: atomics (+2 fills) {ta:7.1f} us {gf/ta/1e3:6.1f} TF | workspace+reduce {tw:7.1f} us {gf/tw/1e3:6.1f} TF  (ws {nws*4/1e6:.0f} MB)")
