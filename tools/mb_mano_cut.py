import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
sys.argv = [sys.argv[0]]
import bench
from hoisdf_amd import _lib, ops, testing as T
from hoisdf_amd.config import Config
from hoisdf_amd.ddp import GradReducer, reducible_parameters
from hoisdf_amd.model import get_model
from hoisdf_amd.optim import FusedAdamW
dev = torch.device("cuda", 0)
cfg = Config(); cfg.resnet_type = 50; cfg.apply_setting("dexycb"); cfg.num_samp_hand, cfg.num_samp_obj, cfg.bins_n = 1536, 512, 64
torch.manual_seed(0)
model = get_model("train", cfg=cfg).to(dev).train()
model.backbone_net.to(memory_format=torch.channels_last); model.decoder_net.to(memory_format=torch.channels_last)
reducer = GradReducer(reducible_parameters(model), bucket_mb=64.0, average=False)
opt = FusedAdamW(list(model.parameters()), lr=cfg.lr)
inputs, targets, meta = (T.to_device(x, dev) for x in T.synthetic_batch(32, 1536, 512, seed=1234))
inputs["img"] = inputs["img"].contiguous(memory_format=torch.channels_last)
import random
model._py_random = random.Random(1)
def step():
    reducer.zero_grad()
    out = model(inputs, targets, meta, "train", 0, 0.1)
    loss = {k: v.mean() for k, v in out.items() if "_out" not in k}
    total = sum(v * bench.LOSS_WEIGHTS.get(k, 1.0) for k, v in loss.items())
    total.backward(); reducer.finish(); opt.step()
def timeit(n=15):
    for _ in range(4): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
base = timeit()
# cut the MANO head + its losses (keep the linear heads): upper bound of what a fused MANO kernel could save
orig_fwd = model.mano_head.forward_batch_first
cache = {}
def fake_fwd(pose6d, shape, mp):
    if "v" not in cache:
        with torch.no_grad():
            cache["v"] = orig_fwd(pose6d.detach(), shape.detach(), mp)
    return cache["v"]
def fake_loss(pm, gm):
    z = torch.zeros((), device=dev)
    return z, z, z, z, None, None
model.mano_head.forward_batch_first = fake_fwd
model.mano_loss.forward = fake_loss
cut = timeit()
print(f"with MANO head + losses {base:.2f} ms/step, without {cut:.2f} ms/step")
