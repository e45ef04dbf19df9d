import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hoisdf_amd import ops as O
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: torch.randn(*s, generator=g, device=dev)
def rel(a, b): return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
E, H = 256, 4
for (B, Lq, Lk, kv, scale) in [(2, 2048, 2048, 2048, 1.0), (2, 512, 2048, 2048, 1.0), (2, 1536, 2048, 2048, 1.0), (2, 2048, 2048, 2048, 4.0), (2, 512, 2048, 2048, 4.0)]:
    q = rnd(B, Lq, E) * scale; kvt = rnd(B, Lk, 2 * E) * scale; go = rnd(B, Lq, E)
    res = []
    for split in (False, True):
        O.set_attention_split(split)
        qq = q.clone().requires_grad_(True); kk = kvt.clone().requires_grad_(True)
        o = O.attention_cross(qq, kk, H, kv)
        o.backward(go)
        res.append((o.detach(), qq.grad, kk.grad))
    O.set_attention_split(False)
    print(B, Lq, Lk, scale, "out", f"{rel(res[1][0], res[0][0]):.2e}", "dq", f"{rel(res[1][1], res[0][1]):.2e}", "dkv", f"{rel(res[1][2], res[0][2]):.2e}",
          "nan:", bool(torch.isnan(res[1][1]).any()), bool(torch.isnan(res[1][2]).any()))
