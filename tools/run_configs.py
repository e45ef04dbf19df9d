#!/usr/bin/env python3
"""BASELINE.json configs[3] and configs[4] (inference through sdf_infer) on one GPU: functional check + timing."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hoisdf_amd import testing as T
from hoisdf_amd.config import Config
from hoisdf_amd.model import get_model

dev = "cuda"


def run(name, setting, B, nh, no, steps=3, f16=False, ref=None):
    c = Config(); c.resnet_type = 50; c.apply_setting(setting); c.attention_f16_eval = f16
    c.num_samp_hand, c.num_samp_obj, c.bins_n = nh, no, 64
    torch.manual_seed(0)
    model = get_model("test", cfg=c).to(dev).eval()
    inputs, targets, meta = (T.to_device(x, dev) for x in T.synthetic_batch(B, nh, no, seed=7))
    with torch.no_grad():
        out = model(inputs, targets, meta, "eval")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = model(inputs, targets, meta, "eval")
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    ok = all(torch.isfinite(v).all() for k, v in out.items() if k.endswith("_out") and v.dtype.is_floating_point)
    print(f"{name}: B={B} N={nh}+{no} {setting}: {dt*1e3:.1f} ms/iter = {B/dt:.1f} samples/s, finite={ok}, "
          f"hand_joints {tuple(out['hand_joints_out'].shape)}, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
    if ref is not None:
        for k in ("hand_joints_out", "mano_joints_out", "mano_mesh_out"):
            if k in out and k in ref:
                print(f"    max |{k} (f16 attention) - (f32)| = {float((out[k] - ref[k]).abs().max()):.3e} m")
    return out


if __name__ == "__main__":
    run("config4 (HO3Dv2-shape, IK variant, inference)", "ho3d_render", 16, 3072, 1024)
    r32 = run("config5 (dense eval, per-GPU half of batch 8), f32 attention", "dexycb", 4, 6144, 2048)
    run("config5 (dense eval, per-GPU half of batch 8), f16 MFMA attention", "dexycb", 4, 6144, 2048, f16=True, ref=r32)
    run("config1-shape on GPU", "dexycb", 1, 384, 128)
