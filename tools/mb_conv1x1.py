"""Would the encoder's 1x1 stride-1 convolutions be faster on the hot path's f32 GEMM than on MIOpen?
ResNet-50 @ 256x256, B = 32, channels_last, fwd + bwd (dX + dW) per shape, times summed over the network."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, torch.nn.functional as F
from hoisdf_amd import ops, miopen_tuning
miopen_tuning.enable()
dev = "cuda"
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters * 1e3
B = 32
# (H, Cin, Cout, count in ResNet-50) for stride-1 1x1 convs
shapes = [(64, 64, 64, 1), (64, 64, 256, 3 + 1), (64, 256, 64, 2), (64, 256, 128, 1),
          (32, 128, 512, 4), (32, 512, 128, 3), (32, 512, 256, 1),
          (16, 256, 1024, 6), (16, 1024, 256, 5), (16, 1024, 512, 1),
          (8, 512, 2048, 3), (8, 2048, 512, 2)]
tot_c = tot_l = 0.0
for H, ci, co, cnt in shapes:
    x = torch.randn(B, ci, H, H, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(co, ci, 1, 1, device=dev) / ci ** 0.5).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gy = torch.randn(B, co, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    def conv():
        x.grad = w.grad = None
        F.conv2d(x, w).backward(gy)
    x2 = x.detach().permute(0, 2, 3, 1).reshape(-1, ci).requires_grad_(True)
    w2 = w.detach().view(co, ci).requires_grad_(True)
    g2 = gy.permute(0, 2, 3, 1).reshape(-1, co)
    def lin():
        x2.grad = w2.grad = None
        ops.linear(x2, w2, None).backward(g2)
    tc, tl = timeit(conv), timeit(lin)
    tot_c += tc * cnt; tot_l += tl * cnt
    print(f"H={H:3d} {ci:5d}->{co:5d} x{cnt}: MIOpen {tc:7.1f} us  hoisdf linear {tl:7.1f} us")
print(f"sum over ResNet-50 stride-1 1x1 convs (fwd+bwd): MIOpen {tot_c/1e3:.2f} ms, hoisdf GEMM {tot_l/1e3:.2f} ms")
