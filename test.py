#!/usr/bin/env python3
"""Inference driver with the reference's CLI (main/test.py:51-74): --gpu_ids --ckpt_path.
Loads a reference-format checkpoint (strict), runs the eval forward (dense-grid sdf_infer branch) and writes
results.txt with MPJPE / PA-MPJPE in cm (main/test.py:229-261).  Without a real dataset it evaluates on synthetic
DexYCB/HO3D-shaped samples (the metrics are then only a plumbing check)."""
import argparse
import os

import torch

from hoisdf_amd.config import cfg
from hoisdf_amd.engine import SyntheticDataset, Tester, mpjpe, pa_mpjpe


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu_ids", type=str, default="0")
    ap.add_argument("--ckpt_path", type=str, default=None)
    ap.add_argument("--setting", type=str, default="dexycb")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--n_batches", type=int, default=2)
    ap.add_argument("--out_dir", type=str, default="outputs/result")
    a = ap.parse_args()
    cfg.apply_setting(a.setting)
    dev = torch.device("cuda", 0)
    tester = Tester(cfg, dev, a.ckpt_path)
    loader = torch.utils.data.DataLoader(SyntheticDataset(cfg, a.batch * a.n_batches, seed=1), batch_size=a.batch)
    preds, gts = [], []
    for inputs, targets, meta in loader:
        out = tester.predict(inputs, targets, meta)
        key = "mano_joints_out" if "mano_joints_out" in out else None
        if key and "mano_joints_gt_out" in out:
            preds.append(out[key].cpu())
            gts.append(out["mano_joints_gt_out"].cpu())
        else:                                               # ho3d: 20 voted joints + zero root (main/test.py:139-142)
            j = torch.cat([torch.zeros_like(out["hand_joints_out"][:, :1]), out["hand_joints_out"]], 1).cpu()
            preds.append(j)
            gts.append(targets["joint_cam_no_trans"] / 1000)
    P, G = torch.cat(preds), torch.cat(gts)
    os.makedirs(a.out_dir, exist_ok=True)
    with open(os.path.join(a.out_dir, "results.txt"), "w") as f:
        f.write(f"MPJPE (cm): {100 * mpjpe(P, G):.4f}\nPA-MPJPE (cm): {100 * pa_mpjpe(P, G):.4f}\n")
    print(open(os.path.join(a.out_dir, "results.txt")).read())


if __name__ == "__main__":
    main()
