"""(f3) closed-form inverse kinematics post-process of the IK variant (``cfg.use_inverse_kinematics``, setting
"ho3d_render"): MANO pose from the 21 predicted joints + the predicted shape, then MANO vertices.

Restates common/utils/inverse_kinematics.py:15-150 (global rotation = Kabsch/SVD fit of the five palm bones,
then per finger and per joint the axis-angle that swings the template bone onto the target bone, expressed in the
accumulated parent frame), batched, on whatever device the inputs live on (no host round trip, no per-call
``ManoLayer`` construction).  **Pinned** (round 6): tests/golden/g12_ik.npz holds the outputs of the reference's own
``ik_solver_mano`` on seeded joints (proper fits and reflected palms).  The ONE function the reference takes from the un-vendored
kornia (``rotation_matrix_to_axis_angle``, :9 / :70) is supplied to it as a restatement of kornia's published algorithm
(four-branch matrix -> quaternion with the 1e-8 guard, then 2 atan2 / sin: tests/golden/make_golden.py ik_golden) instead of
round 5's scipy log map, so the fixture carries the reference's arithmetic; poses (as vectors and as rotations), joints and
vertices agree to 2e-5 (tests/test_ik.py, CPU and on the device), next to the property the algorithm guarantees (MANO(pose from
IK) reproduces the target joints)."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .nets.heads import matrix_to_axis_angle
from .nets.mano import ManoLayer, axis_angle_to_matrix

# joint chains in the 21-joint output order; group g drives MANO pose joints 3g+1 .. 3g+3   (reference :73-79)
FINGERS = [[0, 5, 6, 7, 8], [0, 9, 10, 11, 12], [0, 17, 18, 19, 20], [0, 13, 14, 15, 16], [0, 1, 2, 3, 4]]
PALM = [1, 5, 9, 13, 17]


@torch.no_grad()
def ik_solver_mano(mano_layer: ManoLayer, mano_shape: Optional[torch.Tensor], pred_joints: torch.Tensor) -> Dict[str, torch.Tensor]:
    """pred_joints (B, >=21, 3) metres; mano_shape (B, 10) or None -> verts (B,778,3) m, joints (B,21,3) m, shape, pose
    (B,48) axis-angle, vis (B,1) = 1 where the palm fit is a proper rotation."""
    B, dev = pred_joints.shape[0], pred_joints.device
    root = pred_joints[:, :1]
    tgt = (pred_joints[:, :21] - root).float()
    shape = torch.zeros(B, 10, device=dev) if mano_shape is None else mano_shape.detach().float()
    pose_R = torch.eye(3, device=dev).repeat(B, 16, 1, 1)
    pose_aa = torch.zeros(B, 16, 3, device=dev)
    _, tpl = mano_layer(pose_aa.reshape(B, -1), shape)
    tpl = tpl / 1000.0
    P0 = (tgt[:, PALM] - tgt[:, :1]).transpose(1, 2)                      # (B,3,5)
    T0 = (tpl[:, PALM] - tpl[:, :1]).transpose(1, 2)
    U, _, Vt = torch.linalg.svd(T0 @ P0.transpose(1, 2))
    R = Vt.transpose(1, 2) @ U.transpose(1, 2)
    valid = (torch.linalg.det(R) + 1).abs() > 1e-6                          # reflections are left at identity (:66-71)
    vm = valid[:, None]
    pose_aa[:, 0] = torch.where(vm, matrix_to_axis_angle(R), pose_aa[:, 0])
    pose_R[:, 0] = torch.where(vm[:, :, None], R, pose_R[:, 0])
    for g, chain in enumerate(FINGERS):
        recon = torch.zeros(B, 5, 3, device=dev)
        for ji in range(2, 5):
            vec_tpl = tpl[:, chain[ji]] - tpl[:, chain[ji - 1]]
            R_pa = R.clone()
            for i in range(ji - 2):
                R_pa = R_pa @ pose_R[:, g * 3 + i + 1]
            recon[:, ji - 1] = (R_pa @ (tpl[:, chain[ji - 1]] - tpl[:, chain[ji - 2]]).unsqueeze(-1)).squeeze(-1) \
                + recon[:, ji - 2]
            vec_tgt = (R_pa.transpose(1, 2) @ (tgt[:, chain[ji]] - recon[:, ji - 1]).unsqueeze(-1)).squeeze(-1)
            axis = torch.cross(vec_tpl, vec_tgt, dim=-1)
            axis = axis / (axis.norm(dim=-1, keepdim=True) + 1e-7)
            cosang = (vec_tpl * vec_tgt).sum(-1, keepdim=True) / (vec_tpl.norm(dim=-1, keepdim=True) + 1e-7) \
                / (vec_tgt.norm(dim=-1, keepdim=True) + 1e-7)
            aa = torch.acos(cosang.clamp(-1 + 1e-7, 1 - 1e-7)) * axis
            j = g * 3 + ji - 1
            pose_aa[:, j] = torch.where(vm, aa, pose_aa[:, j])
            pose_R[:, j] = torch.where(vm[:, :, None], axis_angle_to_matrix(aa), pose_R[:, j])
    verts, joints = mano_layer(pose_aa.reshape(B, -1), shape)
    return {"verts": verts / 1000.0 + root, "joints": joints / 1000.0 + root, "shape": shape,
            "pose": pose_aa.reshape(B, -1), "vis": valid[:, None].long()}
