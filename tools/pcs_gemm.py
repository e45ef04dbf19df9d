"""One GEMM family in a loop, for rocprofv3 PC sampling (stall attribution).  usage: pcs_gemm.py fwd|dx|dw|attn"""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, ctypes as C
from hoisdf_amd._lib import call
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
dev = "cuda"
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K = 65536, 512, 992
x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / math.sqrt(K); b = torch.randn(N, device=dev)
dy = torch.randn(M, N, device=dev); bits = torch.zeros(M, (N + 31) // 32, dtype=torch.int32, device=dev)
out = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev); dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
for _ in range(40):
    if which == "fwd":
        call("hoisdf_linear_fwd", p(x), K, p(W), K, p(b), p(out), N, M, N, K, 1, 0.0, 0, p(bits), st)
    elif which == "dx":
        call("hoisdf_linear_bwd_input", p(dy), N, p(bits), 0.0, p(W), K, p(dx), K, M, N, K, 0, st)
    elif which == "dw":
        call("hoisdf_linear_bwd_weight", p(dy), N, None, 0.0, p(x), K, p(dW), K, p(db), M, N, K, None, 0, st)
torch.cuda.synchronize()
