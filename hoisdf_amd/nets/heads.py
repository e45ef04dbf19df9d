"""Vote aggregation + losses (reference common/nets/loss.py:23-171) and the MANO head
(common/nets/mano_head.py:185-278).

``JointvoteLoss`` produces ``hand_joints_out`` through the HIP vote kernel (K12, its three reductions fused).
The MANO head (a16 / f3) is one HIP kernel forward and one backward for all L*B hands with the four ManoLoss
squared-error sums fused (csrc/mano.hip); the PyTorch chain in ``ManoHead`` remains for MANO layers that are not
``nets.mano.ManoLayer`` or carry a hand mean.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class JointvoteLoss(nn.Module):
    def __init__(self, hand_cls_dist: float = 0.04):
        super().__init__()
        self.hand_cls_dist = hand_cls_dist

    def forward(self, hand_points, hand_off, hand_cls, joint_gt, batch_first: bool = False):
        """hand_points (B,P,3) metres; hand_off (L,P,B,60) / hand_cls (L,P,B,20) as in the reference,
        or (L,B,P,*) with batch_first=True; joint_gt (B,20,3) millimetres.
        Returns loss_joint_3d, loss_joint_cls, loss_all_joint_3d, joints (L,B,20,3)."""
        if not batch_first:
            hand_off = hand_off.permute(0, 2, 1, 3)
            hand_cls = hand_cls.permute(0, 2, 1, 3)
        L, B, P, J = hand_cls.shape
        # one HIP pass: joints + the masked SmoothL1 / BCE / near-count reductions (K12); the three scalars
        # below are means of those (L,B)-sized sums
        joints, l3d_sum, bce_sum, near_sum = ops.vote_loss(hand_off, hand_cls, hand_points, joint_gt,
                                                           self.hand_cls_dist)
        return self._finish(joints, l3d_sum, bce_sum, near_sum, joint_gt, P)

    def forward_fused(self, hand_points, hand_enc, vote_mlp, cls_mlp, joint_gt):
        """the two head MLPs on all depths + the vote aggregation + the reductions as ONE C call (hoisdf_heads_vote_fwd, K11 + K12);
        hand_enc (L,B,P,E) batch-first."""
        joints, l3d_sum, bce_sum, near_sum = ops.heads_vote(
            hand_enc, hand_points, joint_gt, self.hand_cls_dist, [l.weight for l in vote_mlp.layers], [l.bias for l in vote_mlp.layers],
            [l.weight for l in cls_mlp.layers], [l.bias for l in cls_mlp.layers])
        return self._finish(joints, l3d_sum, bce_sum, near_sum, joint_gt, hand_enc.shape[2])

    def _finish(self, joints, l3d_sum, bce_sum, near_sum, joint_gt, P):
        L, B, J = joints.shape[0], joints.shape[1], joints.shape[2]
        # the reference sums over (b, p, j), divides by near.sum() and then averages over (l, xyz)
        l3d = l3d_sum.sum(1) / near_sum.sum() / 3.0
        lcls = bce_sum.sum() / float(L * B * P * J)
        # a15: HIP reduction (hoisdf_point_loss_fwd), the (B, J, 3) target broadcast over depth (no eager form: a CPU tensor raises in ops)
        lall = ops.smooth_l1_loss_broadcast(joints, joint_gt.reshape(B, J * 3), 1, 1000.0)
        return l3d.mean(), lcls, lall, joints


class SepSDFLoss(nn.Module):
    """common/nets/loss.py:64-78.  ``clamp``: the ground truth is clamped to +-clamp inside the HIP reduction (what
    main/model.py:393-400 does before calling the loss); None = the targets are taken as they are."""

    def forward(self, hand_sdf, obj_sdf, hand_sdf_gt, obj_sdf_gt, clamp=None):
        c = 0.0 if clamp is None else float(clamp)              # a15: hoisdf_point_loss_fwd / _bwd (no eager form)
        return (ops.l1_loss_clamped_target(hand_sdf, hand_sdf_gt, c), ops.l1_loss_clamped_target(obj_sdf, obj_sdf_gt, c))


class ManoLoss(nn.Module):
    def __init__(self, lambda_verts3d, lambda_joints3d, lambda_manopose, lambda_manoshape):
        super().__init__()
        self.lv, self.lj, self.lp, self.ls = lambda_verts3d, lambda_joints3d, lambda_manopose, lambda_manoshape

    def forward(self, preds, gts):
        if "loss_sums" in preds:                          # the MANO-head kernel already reduced the four squared errors per hand
            s = preds["loss_sums"].sum(0)
            n = preds["loss_sums"].shape[0]
            return (s[0] * (self.lv / (n * 2334)), s[1] * (self.lj / (n * 63)), s[2] * (self.lp / (n * 144)),
                    s[3] * (self.ls / (n * 10)), None, None)

        def mse(a, b):
            return F.mse_loss(a, b.unsqueeze(0).expand(a.shape))
        return (self.lv * mse(preds["verts3d"], gts["verts3d"]), self.lj * mse(preds["joints3d"], gts["joints3d"]),
                self.lp * mse(preds["mano_pose"], gts["mano_pose"]),
                self.ls * mse(preds["mano_shape"], gts["mano_shape"]), None, None)


class ManoShapeLoss(nn.Module):
    def __init__(self, lambda_manoshape, lambda_regulshape):
        super().__init__()
        self.ls, self.lr = lambda_manoshape, lambda_regulshape

    def forward(self, pred_shape, gt_shape):
        return (self.ls * F.mse_loss(pred_shape, gt_shape.unsqueeze(0).expand(pred_shape.shape)),
                self.lr * F.mse_loss(pred_shape, torch.zeros_like(pred_shape)))


# ---- rotation conversions ---------------------------------------------------------------------
def rot6d_to_matrix(x):
    a1, a2 = x[:, :3], x[:, 3:6]
    b1 = F.normalize(a1)
    b2 = F.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=1)), dim=-1)


def matrix_to_quaternion(R, eps=1e-6):
    """(N,3,3) -> (N,4) [w,x,y,z]; four-branch form on R^T selected by the diagonal, unnormalised
    sign convention of the reference (mano_head.py:90-182)."""
    T = R.transpose(1, 2)
    d0, d1, d2 = T[:, 0, 0], T[:, 1, 1], T[:, 2, 2]
    neg_z = d2 < eps
    x_big = d0 > d1
    x_small = d0 < -d1
    cands = [
        (1 + d0 - d1 - d2, [T[:, 1, 2] - T[:, 2, 1], None, T[:, 0, 1] + T[:, 1, 0], T[:, 2, 0] + T[:, 0, 2]], 1),
        (1 - d0 + d1 - d2, [T[:, 2, 0] - T[:, 0, 2], T[:, 0, 1] + T[:, 1, 0], None, T[:, 1, 2] + T[:, 2, 1]], 2),
        (1 - d0 - d1 + d2, [T[:, 0, 1] - T[:, 1, 0], T[:, 2, 0] + T[:, 0, 2], T[:, 1, 2] + T[:, 2, 1], None], 3),
        (1 + d0 + d1 + d2, [None, T[:, 1, 2] - T[:, 2, 1], T[:, 2, 0] - T[:, 0, 2], T[:, 0, 1] - T[:, 1, 0]], 0),
    ]
    sel = [neg_z & x_big, neg_z & ~x_big, ~neg_z & x_small, ~neg_z & ~x_small]
    q = torch.zeros(R.shape[0], 4, dtype=R.dtype, device=R.device)
    den = torch.zeros(R.shape[0], dtype=R.dtype, device=R.device)
    for (t, comps, slot), m in zip(cands, sel):
        comps = [t if c is None else c for c in comps]
        mf = m.to(R.dtype)
        q = q + torch.stack(comps, -1) * mf[:, None]
        den = den + t * mf
    return 0.5 * q / torch.sqrt(den)[:, None]


def quaternion_to_axis_angle(q):
    v = q[..., 1:]
    s2 = (v * v).sum(-1)
    s = torch.sqrt(s2)
    c = q[..., 0]
    two_theta = 2.0 * torch.where(c < 0.0, torch.atan2(-s, -c), torch.atan2(s, c))
    k = torch.where(s2 > 0.0, two_theta / s, torch.full_like(s, 2.0))
    return v * k[..., None]


def matrix_to_axis_angle(R):
    aa = quaternion_to_axis_angle(matrix_to_quaternion(R))
    return torch.where(torch.isnan(aa), torch.zeros_like(aa), aa)


class ManoHead(nn.Module):
    def __init__(self, mano_layer, coord_change_mat=None):
        super().__init__()
        self.mano_layer = mano_layer
        self.mano_pose_size = 48
        if coord_change_mat is not None:
            self.register_buffer("coord_change_mat", coord_change_mat)
        else:
            self.coord_change_mat = None

    def forward(self, pose6d, shape, mano_params=None):
        """pose6d (L,16,B,6) [or (L,B,16,6) via forward_batch_first], shape (L,B,10)."""
        return self.forward_batch_first(pose6d.permute(0, 2, 1, 3), shape, mano_params)

    def forward_batch_first(self, pose6d, shape, mano_params=None):
        from .mano import axis_angle_to_matrix, ManoLayer
        L, B, N, C = pose6d.shape
        assets = self.mano_layer.kernel_assets() if (pose6d.is_cuda and isinstance(self.mano_layer, ManoLayer)) else None
        if assets is not None:
            # one HIP launch for the predictions (+ one for the ground truth): csrc/mano.hip.  The torch chain below stays for
            # MANO layers that are not this package's (get_model takes any module with manopth's interface) or carry a hand mean.
            gt = pack = None
            if mano_params is not None:
                mp = mano_params.contiguous().float()
                gv, gj, gr = ops.mano_gt(mp, assets)
                gt = {"verts3d": gv, "joints3d": gj, "mano_shape": mp[:, self.mano_pose_size:], "mano_pose": gr}
                pack = (gv, gj, gr, mp[:, self.mano_pose_size:])
            verts, joints, R, sums = ops.mano_head(pose6d.reshape(L * B, N, C), shape.reshape(L * B, 10), assets, pack)
            pred = {"verts3d": verts.view(L, B, -1, 3), "joints3d": joints.view(L, B, -1, 3), "mano_pose": R.view(L, B, N, 3, 3),
                    "mano_shape": shape.reshape(L, B, 10)}
            if sums is not None:
                pred["loss_sums"] = sums                  # (L*B, 4) squared-error sums of the four ManoLoss terms
            return pred, gt
        R = rot6d_to_matrix(pose6d.reshape(L * B * N, C))
        pose = matrix_to_axis_angle(R).reshape(-1, self.mano_pose_size)
        betas = shape.reshape(-1, 10)
        verts, joints = self.mano_layer(th_pose_coeffs=pose, th_betas=betas)
        pred = {"verts3d": verts.view(L, B, -1, 3) / 1000, "joints3d": joints.view(L, B, -1, 3) / 1000,
                "mano_pose": R.view(L, B, N, 3, 3), "mano_shape": betas.view(L, B, 10)}
        gt = None
        if mano_params is not None:
            gt_shape = mano_params[:, self.mano_pose_size:]
            gt_pose = mano_params[:, :self.mano_pose_size].clone()
            gt_pose[:, 3:] = gt_pose[:, 3:] - self.mano_layer.th_hands_mean
            gv, gj = self.mano_layer(th_pose_coeffs=gt_pose, th_betas=gt_shape)
            gt = {"verts3d": gv / 1000, "joints3d": gj / 1000, "mano_shape": gt_shape,
                  "mano_pose": axis_angle_to_matrix(gt_pose.reshape(-1, 3)).view(-1, 16, 3, 3)}
        return pred, gt
