"""The encoder's three auxiliary 1x1 heads (32 -> 32 -> 1 at 128 x 128, B = 32): nn.Conv2d on MIOpen / rocBLAS vs a
matmul (gemv) formulation of the final 32 -> 1 convolution.  fwd+bwd time per variant."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, torch.nn as nn
from hoisdf_amd import miopen_tuning
miopen_tuning.enable()
dev = "cuda"
x = torch.randn(32, 32, 128, 128, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
conv = nn.Conv2d(32, 1, 1).to(dev).to(memory_format=torch.channels_last)
conv32 = nn.Conv2d(32, 32, 1).to(dev).to(memory_format=torch.channels_last)


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


def f_conv():
    y = conv(x); y.sum().backward()
def f_mm():
    xl = x.permute(0, 2, 3, 1)                      # NHWC view of the channels_last tensor (no copy)
    y = torch.matmul(xl, conv.weight.view(32)) + conv.bias
    y.sum().backward()
def f_mul():
    xl = x.permute(0, 2, 3, 1)
    y = (xl * conv.weight.view(32)).sum(-1) + conv.bias
    y.sum().backward()
def f_conv32():
    y = conv32(x); y.sum().backward()
print(f"conv 32->1 fwd+bwd: {t(f_conv):.3f} ms | matmul: {t(f_mm):.3f} ms | mul+sum: {t(f_mul):.3f} ms | conv 32->32: {t(f_conv32):.3f} ms")
