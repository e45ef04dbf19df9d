#!/usr/bin/env python3
"""Inference driver with the reference's CLI (main/test.py:51-74): --gpu_ids --ckpt_path.
Loads a reference-format checkpoint (strict), runs the eval forward (dense-grid sdf_infer branch) and writes, next to
the checkpoint (main/test.py:88-90) or under --out_dir,
  * results.txt in the reference's layout (main/test.py:229-261): ``key :  value`` lines - ADDS_error, and for dexycb
    mano_mje / mano_pamje / OCE_error / MCE_error (cm) + the 3D-mesh AUC block and the F-scores, for ho3d MME_error;
  * pred_mano.json for ho3d (main/test.py:263-265, the HO3D submission format), IK post-process included for the IK
    variant (main/test.py:139-160).
Datasets and the YCB object models are licence-gated and absent offline: without them the driver evaluates DexYCB /
HO3D-shaped synthetic samples against synthetic object templates (the numbers are then a plumbing check only)."""
import argparse
import os

import numpy as np
import torch

from hoisdf_amd import metrics as M
from hoisdf_amd.config import cfg
from hoisdf_amd.engine import SyntheticDataset, Tester

# data/ho3d.py:47-70: jointsMapSimpleToMano = argsort(jointsMapManoToSimple) - the order of the HO3D submission file
JOINTS_SIMPLE_TO_MANO = [0, 5, 6, 7, 9, 10, 11, 17, 18, 19, 13, 14, 15, 1, 2, 3, 4, 8, 12, 16, 20]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu_ids", type=str, default="0")
    ap.add_argument("--ckpt_path", type=str, default=None, help="Full path to the checkpoint file")
    ap.add_argument("--setting", type=str, default="dexycb", help="the reference edits Config.setting in config.py")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--n_batches", type=int, default=2)
    ap.add_argument("--out_dir", type=str, default=None, help="default: the checkpoint's directory, else outputs/result")
    a = ap.parse_args()
    assert a.gpu_ids, "Please set propoer gpu ids"
    if "-" in a.gpu_ids:                                   # "0-3" -> "0,1,2,3" (main/test.py:66-70)
        lo, hi = a.gpu_ids.split("-")
        a.gpu_ids = ",".join(str(i) for i in range(int(lo), int(hi) + 1))
    return a


def main():
    a = parse_args()
    cfg.apply_setting(a.setting)
    # one process drives one GPU: the first id of --gpu_ids (the reference wraps the model in DataParallel over all of them)
    dev = torch.device("cuda", int(a.gpu_ids.split(",")[0]))
    torch.cuda.set_device(dev)
    tester = Tester(cfg, dev, a.ckpt_path)
    mano_layer = getattr(getattr(tester.model, "mano_head", None), "mano_layer", None)
    if mano_layer is None:
        from hoisdf_amd.nets.mano import ManoLayer
        mano_layer = ManoLayer().to(dev)
    loader = torch.utils.data.DataLoader(SyntheticDataset(cfg, a.batch * a.n_batches, seed=1), batch_size=a.batch)
    g = torch.Generator().manual_seed(0)
    templates = (0.05 * torch.randn(4, 500, 3, generator=g)).to(dev)         # stand-ins for the YCB models
    ho3d = cfg.dataset == "ho3d"
    results = {"ADDS_error": 0.0}
    if ho3d:
        results["MME_error"] = 0.0
        coord_change = torch.tensor([[1.0, 0, 0], [0, -1.0, 0], [0, 0, -1.0]], device=dev)
        joint_list, mesh_list = [], []
    else:
        results.update(mano_mje=0.0, mano_pamje=0.0, OCE_error=0.0, MCE_error=0.0)
        mesh_err, mesh_err_al = M.MeshEval(), M.MeshEval()
        f_score, f_score_al, f_threshs = [], [], [0.005, 0.015]
    total = 0
    for it, (inputs, targets, meta) in enumerate(loader):
        out = tester.predict(inputs, targets, meta, mano_layer=mano_layer)
        B = meta["mano_root"].shape[0]
        tg = {k: v.to(dev) for k, v in targets.items()}
        root = meta["mano_root"].to(dev)
        obj_cls = (torch.arange(B) + it) % templates.shape[0]
        om = M.obj_metrics(out["obj_rot_out"], out["obj_trans_out"], tg["obj_rot"], tg["rel_obj_trans"], templates[obj_cls], ho3d)
        total += B
        results["ADDS_error"] += om["ADDS"] * B * 100
        if ho3d:                                                                  # main/test.py:133-176
            if cfg.use_inverse_kinematics:
                joints, mesh = out["ik_joints_out"], out["ik_verts_out"]
            else:
                joints, mesh = out["mano_joints_out"], out["mano_mesh_out"]
            joints = (joints + root[:, None]) @ coord_change
            mesh = (mesh + root[:, None]) @ coord_change
            results["MME_error"] += om["MME"] * B * 100
            joint_list += [j[JOINTS_SIMPLE_TO_MANO] for j in joints.cpu().numpy()]
            mesh_list += list(mesh.cpu().numpy())
        else:                                                                     # main/test.py:178-225
            if cfg.use_inverse_kinematics:
                mje, pamje = M.eval_hand_joint(out["ik_joints_out"] - out["ik_joints_out"][:, :1], tg["joint_cam_no_trans"] / 1000)
            else:
                mje, pamje = M.eval_hand_joint(out["mano_joints_out"], out["mano_joints_gt_out"])
            results["mano_mje"] += mje * B * 100
            results["mano_pamje"] += pamje * B * 100
            results["OCE_error"] += om["OCE"] * B * 100
            results["MCE_error"] += om["MCE"] * B * 100
            if cfg.eval_mesh and "mano_mesh_out" in out:
                pv, gv = out["mano_mesh_out"], out["mano_mesh_gt_out"]
                al = M.rigid_align(pv, gv)
                mesh_err.feed(gv, pv)
                mesh_err_al.feed(gv, al)
                f_score.append(torch.stack([M.fscore(gv, pv, t) for t in f_threshs], 1).cpu().numpy())
                f_score_al.append(torch.stack([M.fscore(gv, al, t) for t in f_threshs], 1).cpu().numpy())
    out_dir = a.out_dir or (os.path.dirname(a.ckpt_path) if a.ckpt_path else "outputs/result")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "results.txt")
    if not ho3d and cfg.eval_mesh and f_score:
        M.write_results(path, results, total, mesh=(mesh_err, mesh_err_al),
                        fscores=(np.concatenate(f_score).T, np.concatenate(f_score_al).T, f_threshs))
    else:
        M.write_results(path, results, total)
    if ho3d:
        M.dump_pred_mano(os.path.join(out_dir, "pred_mano.json"), joint_list, mesh_list)
        print(f"Dumped {len(joint_list)} joints and {len(mesh_list)} verts predictions to {out_dir}/pred_mano.json")
    print(open(path).read())


if __name__ == "__main__":
    main()
