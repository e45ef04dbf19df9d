#!/usr/bin/env python3
"""diagnostic: the stride-32 pyramid gradient of the N=2048 training fixture in the three GEMM modes (exact f32 MFMA, bf16x3
emulation) against the reference golden and against each other"""
import sys, os, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from hoisdf_amd import ops, testing as T
from hoisdf_amd.config import Config
from hoisdf_amd.model import get_model
from hoisdf_amd.nets import mano as MANO
DEV = "cuda"
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g8_train_dexycb_n2048.npz")))
NH, NO = 1536, 512
def run(mode, rep=1):
    c = Config(); c.resnet_type = 18; c.apply_setting("dexycb")
    c.num_samp_hand, c.num_samp_obj, c.bins_n, c.dropout = NH, NO, 16, 0.0
    model = get_model("test", cfg=c, mano_layer=MANO.ManoLayer(MANO.synthetic_assets(0)), with_encoder=False)
    sd = model.state_dict()
    for k in sd:
        if not k.startswith("mano_head"): sd[k] = T.det_param(k, sd[k].shape)
    model.load_state_dict(sd, strict=True); model = model.to(DEV).train()
    for m in model.modules():
        if hasattr(m, "p"): m.p = 0.0
        if hasattr(m, "dropout_prob"): m.dropout_prob = 0.0
    levels = [v.to(DEV).permute(0, 2, 3, 1).contiguous().requires_grad_(True) for v in T.synthetic_pyramid(2, big=False, seed=3).values()]
    pyr = ops.PyramidNHWC(levels)
    inputs, targets, meta = T.synthetic_batch(2, NH, NO, seed=31)
    torch.manual_seed(1234)
    jit = [torch.empty_like(inputs["hand_pre_points"]).uniform_(-0.05, 0.05), torch.empty_like(inputs["obj_pre_points"]).uniform_(-0.05, 0.05)]
    model._jitter = lambda like, d: jit.pop(0).to(DEV)
    model._py_random = random.Random(0)
    inputs, targets, meta = ({k: v.to(DEV) for k, v in d.items()} for d in (inputs, targets, meta))
    ops.set_gemm_emu(mode == "emu")
    loss, out = model.hot_path(pyr, inputs, targets, meta, "train", 0, 0.5)
    total = sum(v.mean() for v in loss.values()); total.backward()
    return levels[4].grad.permute(0, 3, 1, 2)[:, ::16].detach().cpu().double(), {k: float(v.mean()) for k, v in loss.items()}
ref = torch.from_numpy(g["grad.pyr.stride32"]).double()
res = {m: run(m) for m in ("f32", "emu", "f32", "emu")}
res2 = {}
for i, m in enumerate(("f32", "emu")):
    res2[m] = res[m]
mx = float(ref.abs().max())
print("max |ref| =", mx)
for m, (gr, ls) in res2.items():
    d = (gr - ref).abs()
    idx = np.unravel_index(int(d.argmax()), d.shape)
    print(f"{m:6s} vs golden: max {float(d.max()) / mx:.3e} of max at {idx}: got {float(gr[idx]):+.6e} ref {float(ref[idx]):+.6e}   rms {float((d**2).mean().sqrt()) / mx:.3e}")
for a, b in (("f32", "emu"),):
    d = (res2[a][0] - res2[b][0]).abs()
    print(f"{a} vs {b}: max {float(d.max()) / mx:.3e} of max, rms {float((d**2).mean().sqrt()) / mx:.3e}")
