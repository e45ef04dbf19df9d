"""One training step with and without the two-stream overlap: parameter gradients must agree to float-atomic noise
(calibrated by running the single-stream step twice)."""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hoisdf_amd import ops, testing as T
from hoisdf_amd.config import Config
from hoisdf_amd.model import get_model
dev = torch.device("cuda", 0)
c = Config(); c.resnet_type = 50; c.apply_setting("dexycb"); c.num_samp_hand, c.num_samp_obj = 1536, 512
torch.manual_seed(0)
model = get_model("train", cfg=c).to(dev).eval()      # dropout off: the two modes issue ops (and draw seeds) in a different order
for m in model.modules():
    if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
        m.train()
batch = tuple(T.to_device(x, dev) for x in T.synthetic_batch(8, 1536, 512, seed=5))
def grads(two):
    c.overlap_streams = two
    model.zero_grad(set_to_none=True)
    model._py_random = random.Random(0); ops.manual_seed(77); torch.manual_seed(3)
    out = model(*batch, "train", 0, 0.1)
    total = sum(v.mean() for k, v in out.items() if "_out" not in k)
    total.backward(); torch.cuda.synchronize()
    return float(total), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
def cmp(a, b, hot_only=True):
    worst = (0.0, "")
    num = den = 0.0
    for n in a[1]:
        if hot_only and n.startswith(("backbone_net", "decoder_net")):
            continue
        num += float((a[1][n] - b[1][n]).double().pow(2).sum()); den += float(a[1][n].double().pow(2).sum())
        d = (a[1][n] - b[1][n]).abs().max().item() / (a[1][n].abs().max().item() + 1e-30)
        if a[1][n].abs().max().item() > 1e-6 and d > worst[0]: worst = (d, n)
    return f"|dloss| {abs(a[0] - b[0]):.3e} of {a[0]:.3f}; global grad rel L2 diff {(num / den) ** 0.5:.3e}; worst param {worst}"
s1, s2 = grads(False), grads(False)
t1, t2 = grads(True), grads(True)
print("single vs single :", cmp(s1, s2))
print("two    vs two    :", cmp(t1, t2))
print("single vs two    :", cmp(s1, t1))
print("encoder included, single vs single:", cmp(s1, s2, False))
for name in ("linear_sdfin.layers.0.weight", "linear_sdfin.layers.1.weight", "linear_transformerin.layers.0.weight"):
    a, b = s1[1][name], s2[1][name]
    d = (a - b).abs(); mx = a.abs().max().item()
    bad = (d > 1e-4 * mx).nonzero()
    print(name, tuple(a.shape), "max|g|", mx, "n_bad", bad.shape[0], "max diff", d.max().item())
    if bad.shape[0]:
        rows = bad[:, 0].unique(); cols = bad[:, 1].unique()
        print("   rows", rows[:12].tolist(), "... cols", cols[:12].tolist(), "n_rows", rows.numel(), "n_cols", cols.numel())
        i, j = bad[0].tolist(); print("   sample", a[i, j].item(), b[i, j].item())

# the bucketed reducer on top of the two-stream step must hand the optimizer the same gradients
from hoisdf_amd.ddp import GradReducer, reducible_parameters
red = GradReducer(reducible_parameters(model))
def grads_reducer():
    c.overlap_streams = True
    red.zero_grad()
    model._py_random = random.Random(0); ops.manual_seed(77); torch.manual_seed(3)
    out = model(*batch, "train", 0, 0.1)
    total = sum(v.mean() for k, v in out.items() if "_out" not in k)
    total.backward(); red.finish(); torch.cuda.synchronize()
    return float(total), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
for i in range(3):
    print("single (plain autograd) vs two-stream + GradReducer:", cmp(s1, grads_reducer()))
