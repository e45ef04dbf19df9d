"""Which loss terms produce run-to-run different gradients under HOISDF_DETERMINISTIC? (debug helper)"""
import sys, os, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hoisdf_amd import ops, testing as T
from hoisdf_amd.config import Config
from hoisdf_amd.model import get_model
ops.set_deterministic(True)
c = Config(); c.resnet_type = 18; c.apply_setting("dexycb"); c.num_samp_hand, c.num_samp_obj = 384, 128
torch.manual_seed(0)
model = get_model("train", cfg=c).to("cuda").train()
batch = tuple(T.to_device(x, "cuda") for x in T.synthetic_batch(4, 384, 128, seed=5))


def step(keys):
    model.zero_grad(set_to_none=True)
    model._py_random = random.Random(0)
    torch.manual_seed(3); ops.manual_seed(77)
    out = model(*batch, "train", 0, 0.1)
    total = sum(v.mean() for k, v in out.items() if "_out" not in k and (keys is None or k in keys))
    total.backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


out = model(*batch, "train", 0, 0.1)
names = [k for k in out if "_out" not in k]
for k in names + [None]:
    g1, g2 = step([k] if k else None), step([k] if k else None)
    hot = [n for n in g1 if not n.startswith(("backbone_net", "decoder_net"))]
    bad = [n for n in hot if not torch.equal(g1[n], g2[n])]
    print(f"{str(k):22s} hot-path grads {len(hot):4d}  differing {len(bad):4d}  e.g. {bad[:3]}")
