"""Evaluation metrics of the reference's test driver, batched on the device.

  hand:    MJE / PA-MJE                         common/metrics.py:188-232 (rigid_transform_3D, rigid_align, eval_hand_joint)
  object:  ADD-S, MCE (bbox-corner error), OCE (centre error), MME (mean mesh error)
                                                 common/metrics.py:62-185 (compute_obj_metrics_*, eval_batched_obj_direct)
  mesh:    per-vertex EPE mean / AUC over thresholds (EvalUtil.get_measures) and F-scores at 5 / 15 mm
                                                 common/eval_util.py:11-136, main/test.py:204-261
  results.txt / pred_mano.json writers           main/test.py:229-265, data/ho3d_util.py:123-134

The reference loops over samples on the host (numpy SVD per sample, open3d nearest neighbours per mesh); here every
metric is one batched torch expression on the GPU (the N x N distance matrices of ADD-S / F-score are 778^2 ... 2000^2
per sample).  Object templates are dataset assets (YCB models) - callers pass ``templates[obj_id] = (V, 3)`` tensors.
"""
from __future__ import annotations

import json
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

_trapz = getattr(np, "trapezoid", None) or np.trapz


# ---- rotations -------------------------------------------------------------------------------------------------
def batch_rodrigues(aa: torch.Tensor) -> torch.Tensor:
    """axis-angle (B,3) -> (B,3,3), the quaternion form of manopth/rodrigues_layer.py:38-89 (|aa| + 1e-8)."""
    ang = (aa + 1e-8).norm(dim=1, keepdim=True)
    n = aa / ang
    h = 0.5 * ang
    w, xyz = torch.cos(h), torch.sin(h) * n
    q = torch.cat([w, xyz], 1)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([w * w + x * x - y * y - z * z, 2 * x * y - 2 * w * z, 2 * w * y + 2 * x * z,
                     2 * w * z + 2 * x * y, w * w - x * x + y * y - z * z, 2 * y * z - 2 * w * x,
                     2 * x * z - 2 * w * y, 2 * w * x + 2 * y * z, w * w - x * x - y * y + z * z], 1)
    return R.view(-1, 3, 3)


# ---- hand ------------------------------------------------------------------------------------------------------
def rigid_align(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """similarity (scale + rotation + translation) alignment of A (N,P,3) onto B (N,P,3), batched
    (common/metrics.py:188-211: H = (A-ca)^T (B-cb) / n, SVD, reflection fix on the last singular vector)."""
    A64, B64 = A.double(), B.double()
    ca, cb = A64.mean(1, keepdim=True), B64.mean(1, keepdim=True)
    H = (A64 - ca).transpose(1, 2) @ (B64 - cb) / A.shape[1]
    U, s, Vh = torch.linalg.svd(H)
    R = Vh.transpose(1, 2) @ U.transpose(1, 2)
    neg = torch.linalg.det(R) < 0
    s = s.clone()
    s[neg, -1] = -s[neg, -1]
    Vh = Vh.clone()
    Vh[neg, 2] = -Vh[neg, 2]
    R = Vh.transpose(1, 2) @ U.transpose(1, 2)
    varP = A64.var(dim=1, unbiased=False).sum(1)
    c = s.sum(1) / varP
    t = -(c[:, None, None] * R @ ca.transpose(1, 2)).transpose(1, 2) + cb
    return ((c[:, None, None] * R @ A64.transpose(1, 2)).transpose(1, 2) + t).to(A.dtype)


def eval_hand_joint(pred: torch.Tensor, gt: torch.Tensor) -> Tuple[float, float]:
    """(MJE, PA-MJE): mean over samples of the mean per-joint error (common/metrics.py:214-232)."""
    mje = (pred - gt).norm(dim=-1).mean(1)
    pa = (rigid_align(pred, gt) - gt).norm(dim=-1).mean(1)
    return float(mje.mean()), float(pa.mean())


# ---- object ----------------------------------------------------------------------------------------------------
_CORNERS = torch.tensor([[0, 1, 0, 0, 1, 0, 1, 1], [0, 0, 1, 0, 1, 1, 0, 1], [0, 0, 0, 1, 0, 1, 1, 1]])


def _adds(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """ADD-S per sample: mean over PREDICTED vertices of the distance to the closest target vertex."""
    return torch.cdist(pred, target).min(dim=2)[0].mean(1)


def _bbox_corners(m: torch.Tensor) -> torch.Tensor:
    mm = torch.stack([m.min(1)[0], m.max(1)[0]], 2)                       # (B,3,2)
    idx = _CORNERS.to(m.device)
    return torch.stack([mm[:, 0, idx[0]], mm[:, 1, idx[1]], mm[:, 2, idx[2]]], 2)   # (B,8,3)


def obj_metrics(obj_rot, obj_trans, obj_rot_gt, obj_trans_gt, template_verts, ho3d: bool):
    """common/metrics.py:118-185.  obj_rot / obj_trans: per-point predictions (B,P,3) -> averaged over points;
    template_verts (B,V,3).  Returns dict of per-batch means: ADDS + (MCE, OCE | MME)."""
    rot, trans = obj_rot.detach().mean(1), obj_trans.detach().mean(1)
    tgt = template_verts @ batch_rodrigues(obj_rot_gt).transpose(1, 2) + obj_trans_gt[:, None]
    prd = template_verts @ batch_rodrigues(rot).transpose(1, 2) + trans[:, None]
    out = {"ADDS": float(_adds(prd, tgt).mean())}
    if ho3d:
        out["MME"] = float((tgt - prd).norm(dim=-1).mean(-1).mean())
    else:
        out["MCE"] = float((_bbox_corners(prd) - _bbox_corners(tgt)).norm(dim=-1).mean(-1).mean())
        out["OCE"] = float((trans - obj_trans_gt).norm(dim=-1).mean())
    return out


# ---- mesh ------------------------------------------------------------------------------------------------------
def fscore(gt: torch.Tensor, pr: torch.Tensor, th: float) -> torch.Tensor:
    """per-sample F-score at threshold th (common/eval_util.py:117-136; nearest-neighbour distances both ways)."""
    d = torch.cdist(gt, pr)
    d1, d2 = d.min(2)[0], d.min(1)[0]                 # gt -> closest pred, pred -> closest gt
    recall = (d2 < th).float().mean(1)
    precision = (d1 < th).float().mean(1)
    s = recall + precision
    return torch.where(s > 0, 2 * recall * precision / s.clamp_min(1e-30), torch.zeros_like(s))


class MeshEval:
    """EvalUtil(num_kp=778).feed / get_measures (common/eval_util.py:11-103) for fully visible meshes."""

    def __init__(self):
        self.dist: List[torch.Tensor] = []

    def feed(self, gt: torch.Tensor, pred: torch.Tensor):
        self.dist.append((gt - pred).norm(dim=-1).double().cpu())          # (B,V)

    def get_measures(self, val_min: float, val_max: float, steps: int):
        d = torch.cat(self.dist, 0).numpy()                                   # (N,V)
        th = np.linspace(val_min, val_max, steps)
        norm = _trapz(np.ones_like(th), th)
        epe_mean = d.mean(0).mean()
        pck = (d[None] <= th[:, None, None]).mean(1)                          # (steps,V)
        auc = (_trapz(pck, th, axis=0) / norm).mean()
        return float(epe_mean), float(np.median(d, 0).mean()), float(auc), pck.mean(1), th


# ---- writers ---------------------------------------------------------------------------------------------------
def write_results(path: str, results: Dict[str, float], total_samples: int, mesh: Optional[Tuple[MeshEval, MeshEval]] = None,
                  fscores: Optional[Tuple[np.ndarray, np.ndarray, Sequence[float]]] = None) -> None:
    """results.txt in the reference's layout (main/test.py:229-261): ``key :  value`` lines, then the mesh block."""
    with open(path, "w+") as f:
        for k, v in results.items():
            print(k, ": ", v / total_samples, file=f)
        if mesh is not None:
            m3d, _, auc, _, _ = mesh[0].get_measures(0.0, 0.05, 100)
            print("Evaluation 3D MESH results:", file=f)
            print("auc=%.3f, mean_vert3d_avg=%.2f cm" % (auc, m3d * 100.0), file=f)
            m3d, _, auc, _, _ = mesh[1].get_measures(0.0, 0.05, 100)
            print("Evaluation 3D MESH ALIGNED results:", file=f)
            print("auc=%.3f, mean_vert3d_avg=%.2f cm\n" % (auc, m3d * 100.0), file=f)
        if fscores is not None:
            print("F-scores", file=f)
            fs, fa, ths = fscores
            for a, b, t in zip(fs, fa, ths):
                print("F@%.1fmm = %.3f" % (t * 1000, a.mean()), "\tF_aligned@%.1fmm = %.3f" % (t * 1000, b.mean()), file=f)


def dump_pred_mano(path: str, xyz_pred_list, verts_pred_list) -> None:
    """pred_mano.json of the HO3D submission format (data/ho3d_util.py:123-134): [[joints...], [verts...]]."""
    with open(path, "w") as fo:
        json.dump([[np.asarray(x).tolist() for x in xyz_pred_list], [np.asarray(v).tolist() for v in verts_pred_list]], fo)
