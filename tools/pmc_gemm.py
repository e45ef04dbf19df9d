"""Small driver for PMC collection: one launch family per shape (run under rocprofv3 --pmc ...)."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, ctypes as C
from hoisdf_amd._lib import call
dev = "cuda"
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K = 49152, 1024, 992      # the largest linear of the training step (linear_transformerin layer 0 on the hand points)
x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / math.sqrt(K); b = torch.randn(N, device=dev)
dy = torch.randn(M, N, device=dev); bits = torch.zeros(M, (N + 31) // 32, dtype=torch.int32, device=dev); y = torch.relu(torch.randn(M, N, device=dev))
out = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev); dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
for _ in range(3):
    call("hoisdf_linear_fwd", p(x), K, p(W), K, p(b), p(out), N, M, N, K, 1, 0.0, 0, p(bits), st)
    call("hoisdf_linear_bwd_input", p(dy), N, None, 0.0, p(W), K, p(dx), K, M, N, K, 0, st)
    call("hoisdf_linear_bwd_input", p(dy), N, p(bits), 0.0, p(W), K, p(dx), K, M, N, K, 0, st)
    call("hoisdf_linear_bwd_weight", p(dy), N, None, 0.0, p(x), K, p(dW), K, p(db), M, N, K, None, 0, st)
torch.cuda.synchronize()
