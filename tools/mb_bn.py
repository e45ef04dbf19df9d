"""fused BatchNorm+ReLU (csrc/bnorm.hip) vs torch (MIOpen BN + ReLU [+ add]) at the ResNet-50 activation shapes, fwd+bwd ms"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, torch.nn.functional as F
from hoisdf_amd import ops as O
dev = "cuda"
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for shape, res in [((32, 64, 128, 128), False), ((32, 64, 64, 64), False), ((32, 256, 64, 64), True), ((32, 128, 32, 32), False),
                   ((32, 512, 32, 32), True), ((32, 1024, 16, 16), True), ((32, 2048, 8, 8), True), ((32, 32, 128, 128), False)]:
    N, C, H, W = shape
    x = torch.randn(*shape, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r = torch.randn(*shape, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True) if res else None
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    go = torch.randn(*shape, device=dev).contiguous(memory_format=torch.channels_last)
    def fused():
        y = O.batchnorm_relu(x, g, b, rm, rv, 0.1, 1e-5, True, r); y.backward(go)
    def plain():
        y = F.batch_norm(x, rm, rv, g, b, True, 0.1, 1e-5)
        if r is not None: y = y + r
        y = F.relu(y); y.backward(go)
    mb = x.numel() * 4 / 1e6
    print(f"{str(shape):22s} res={res!s:5s} {mb:7.1f} MB  fused {t(fused):.3f} ms   torch {t(plain):.3f} ms")
