"""How long does the host take to *issue* one training step vs the GPU to execute it?"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
sys.argv = [sys.argv[0]]
import bench
from hoisdf_amd import _lib, ops, testing as T
from hoisdf_amd.config import Config
from hoisdf_amd.ddp import GradReducer, reducible_parameters
from hoisdf_amd.model import get_model
dev = torch.device("cuda", 0)
cfg = Config(); cfg.resnet_type = 50; cfg.apply_setting("dexycb"); cfg.num_samp_hand, cfg.num_samp_obj = 1536, 512
torch.manual_seed(0)
model = get_model("train", cfg=cfg).to(dev).train()
model.backbone_net.to(memory_format=torch.channels_last); model.decoder_net.to(memory_format=torch.channels_last)
reducer = GradReducer(reducible_parameters(model))
from hoisdf_amd.optim import FusedAdamW
opt = FusedAdamW(list(model.parameters()), lr=1e-4)          # (as bench.py)
inputs, targets, meta = (T.to_device(x, dev) for x in T.synthetic_batch(32, 1536, 512, seed=1234))
inputs["img"] = inputs["img"].contiguous(memory_format=torch.channels_last)
def step():
    reducer.zero_grad()
    out = model(inputs, targets, meta, "train", 0, 0.1)
    loss = {k: v.mean() for k, v in out.items() if "_out" not in k}
    total = sum(v * bench.LOSS_WEIGHTS.get(k, 1.0) for k, v in loss.items())
    t1 = time.perf_counter()
    total.backward()
    t2 = time.perf_counter()
    reducer.finish(); opt.step()
    return t1, t2
for _ in range(3): step()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); t1, t2 = step(); t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"host: fwd issue {1e3*(t1-t0):.1f} ms, bwd issue {1e3*(t2-t1):.1f} ms, opt {1e3*(t3-t2):.1f} ms, total issue {1e3*(t3-t0):.1f} ms; GPU done after {1e3*(t4-t0):.1f} ms")
# steady state: 8 steps issued back to back without a sync - does the host stay ahead of the device?
torch.cuda.synchronize()
t0 = time.perf_counter()
marks = []
for _ in range(8):
    t1, t2 = step(); marks.append(time.perf_counter())
t_issue = time.perf_counter(); torch.cuda.synchronize(); t_done = time.perf_counter()
print("issue time of consecutive steps (ms):", [round(1e3 * (b - a), 1) for a, b in zip([t0] + marks[:-1], marks)])
print(f"8 steps: issued in {1e3*(t_issue-t0):.1f} ms, device done after {1e3*(t_done-t0):.1f} ms ({1e3*(t_done-t0)/8:.1f} ms/step)")
import cProfile, pstats
pr = cProfile.Profile()
reducer.zero_grad()
pr.enable()
out = model(inputs, targets, meta, "train", 0, 0.1)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(25)
pr = cProfile.Profile(); pr.enable(); reducer.finish() if False else None; opt.step(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
