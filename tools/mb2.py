import sys, os, math
sys.path.insert(0, "/root/repo")
import torch, ctypes as C
from hoisdf_amd import _lib
from hoisdf_amd._lib import call
sys.path.insert(0, "/root/repo/tools")
from microbench import timeit
dev="cuda"
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for M,N,K in [(65536,512,992),(65536,256,256)]:
    x=torch.randn(M,K,device=dev); W=torch.randn(N,K,device=dev)/math.sqrt(K); dy=torch.randn(M,N,device=dev); bits=torch.randint(-2**31,2**31-1,(M,(N+31)//32),dtype=torch.int32,device=dev)
    y=torch.relu(torch.randn(M,N,device=dev)); dx=torch.empty(M,K,device=dev); dW=torch.zeros(N,K,device=dev); db=torch.zeros(N,device=dev)
    nws=_lib.lib().hoisdf_linear_bwd_weight_workspace(M,N,K); ws=torch.empty(max(nws,1),device=dev)
    fl=2.0*M*N*K
    r={}
    r['dX nomask']=timeit(lambda: call("hoisdf_linear_bwd_input", p(dy), N, None, 0.0, p(W), K, p(dx), K, M, N, K, 0, st))
    r['dX mask']=timeit(lambda: call("hoisdf_linear_bwd_input", p(dy), N, p(bits), 0.0, p(W), K, p(dx), K, M, N, K, 0, st))
    r['dW nomask ws']=timeit(lambda: call("hoisdf_linear_bwd_weight", p(dy), N, None, 0.0, p(x), K, p(dW), K, p(db), M, N, K, p(ws), nws, st))
    r['dW nomask ws nodb']=timeit(lambda: call("hoisdf_linear_bwd_weight", p(dy), N, None, 0.0, p(x), K, p(dW), K, None, M, N, K, p(ws), nws, st))
    r['dW mask ws']=timeit(lambda: call("hoisdf_linear_bwd_weight", p(dy), N, p(bits), 0.0, p(x), K, p(dW), K, p(db), M, N, K, p(ws), nws, st))
    r['dW nomask atomic']=timeit(lambda: call("hoisdf_linear_bwd_weight", p(dy), N, None, 0.0, p(x), K, p(dW), K, p(db), M, N, K, None, 0, st))
    print(M,N,K,{k: f"{v*1e6:.0f}us {fl/v/1e12:.1f}TF" for k,v in r.items()})
