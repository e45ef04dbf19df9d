"""MANO hand model layer (linear blend skinning): the module with manopth's interface and buffers.

Scope: SURVEY.md section 8 row a16 / (f3).  On the GPU the head (nets/heads.py ManoHead) runs this layer's
maths inside one HIP kernel per direction (csrc/mano.hip, reading this module's buffers through
``kernel_assets``); ``forward`` below is the same layer in plain PyTorch - the module's public manopth-style
call, the fp64 reference of tests/test_gpu_mano.py, and the path a layer with a non-zero hand mean takes.
It sits between the HIP hot path (which produces the 6D pose and shape parameters) and
the reported vertex / joint coordinates.  The licensed MANO_RIGHT.pkl asset is not available
offline, so ``synthetic_assets`` builds a MANO-*shaped* random asset with the same buffer names
and shapes the reference registers (manopth/manopth/manolayer.py:72-101) - checkpoints that
carry real ``mano_head.mano_layer.th_*`` buffers load over it with ``strict=True``.

Semantics follow manopth/manopth/manolayer.py:111-276 for the configuration the reference uses
(main/model.py:735-742: use_pca=False, flat_hand_mean=True, center_idx=0, side="right",
axis-angle root and joints).  Units: inputs in MANO metres, outputs in millimetres.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

# kinematic tree of the 16 MANO joints (wrist + 5 fingers x 3)
_PARENTS = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]
_TIP_VERTS = [745, 317, 444, 556, 673]
_JOINT_ORDER = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]


def synthetic_assets(seed: int = 0) -> Dict[str, torch.Tensor]:
    r = np.random.default_rng(seed)
    f32 = lambda a: torch.from_numpy(np.asarray(a, np.float32))
    w = r.random((778, 16)) ** 8
    w = w / w.sum(1, keepdims=True)
    jr = r.random((16, 778)) ** 6
    jr = jr / jr.sum(1, keepdims=True)
    return dict(
        th_betas=torch.zeros(1, 10),
        th_shapedirs=f32(0.004 * r.standard_normal((778, 3, 10))),
        th_posedirs=f32(0.002 * r.standard_normal((778, 3, 135))),
        th_v_template=f32(0.05 * r.standard_normal((1, 778, 3))),
        th_J_regressor=f32(jr),
        th_weights=f32(w),
        th_faces=torch.from_numpy(r.integers(0, 778, (1538, 3))).long(),
        th_hands_mean=torch.zeros(1, 45),
        th_selected_comps=f32(r.standard_normal((45, 45)) / 6.0),
    )


def axis_angle_to_matrix(aa: torch.Tensor) -> torch.Tensor:
    """(N,3) -> (N,3,3) through a unit quaternion with the reference's +1e-8 norm guard
    (manopth/manopth/rodrigues_layer.py:43-54)."""
    ang = (aa + 1e-8).norm(dim=1, keepdim=True)
    axis = aa / ang
    q = torch.cat([torch.cos(0.5 * ang), torch.sin(0.5 * ang) * axis], dim=1)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    rows = [w * w + x * x - y * y - z * z, 2 * x * y - 2 * w * z, 2 * w * y + 2 * x * z,
            2 * w * z + 2 * x * y, w * w - x * x + y * y - z * z, 2 * y * z - 2 * w * x,
            2 * x * z - 2 * w * y, 2 * w * x + 2 * y * z, w * w - x * x - y * y + z * z]
    return torch.stack(rows, dim=1).view(-1, 3, 3)


class ManoLayer(nn.Module):
    def __init__(self, assets: Optional[Dict[str, torch.Tensor]] = None, center_idx: int = 0):
        super().__init__()
        assets = synthetic_assets() if assets is None else assets
        for k, v in assets.items():
            self.register_buffer(k, v.clone())
        # constants live on the device with the module (no per-call host->device copies = no host syncs)
        self.register_buffer("_eye3", torch.eye(3), persistent=False)
        self.register_buffer("_bottom", torch.tensor([[0.0, 0, 0, 1]]), persistent=False)
        self.register_buffer("_tip_idx", torch.tensor(_TIP_VERTS, dtype=torch.long), persistent=False)
        self.register_buffer("_joint_idx", torch.tensor(_JOINT_ORDER, dtype=torch.long), persistent=False)
        self.center_idx = center_idx
        self._kernel_assets = None                      # ((versions, pointers), assets tuple) of the HIP MANO-head kernels

    def kernel_assets(self):
        """The asset tuple ops.mano_head / ops.mano_gt take - (transposed blend-shape image, v_template, J_regressor,
        skinning weights, hands_mean) - or None when the one-kernel head does not apply: tensors not on a GPU, a centre
        joint other than the wrist, or a non-zero hand mean (the reference builds the layer with flat_hand_mean=True,
        main/model.py:735-742; the kernel's backward relies on it).  Rebuilt when a buffer is reloaded or moved."""
        bufs = (self.th_shapedirs, self.th_posedirs, self.th_v_template, self.th_J_regressor, self.th_weights, self.th_hands_mean)
        key = tuple((b._version, b.data_ptr()) for b in bufs)
        if self._kernel_assets is None or self._kernel_assets[0] != key:
            assets = None
            if self.th_shapedirs.is_cuda and self.center_idx == 0 and not bool(self.th_hands_mean.ne(0).any()):
                from .. import ops
                f = lambda t: t.contiguous().float()
                assets = (ops.mano_dirs_image(self.th_shapedirs, self.th_posedirs, self.th_weights), f(self.th_v_template).view(-1),
                          f(self.th_J_regressor), f(self.th_weights), f(self.th_hands_mean).view(-1))
            self._kernel_assets = (key, assets)
        return self._kernel_assets[1]

    def forward(self, th_pose_coeffs: torch.Tensor, th_betas: torch.Tensor):
        B = th_pose_coeffs.shape[0]
        full = torch.cat([th_pose_coeffs[:, :3], self.th_hands_mean + th_pose_coeffs[:, 3:48]], 1)
        R = axis_angle_to_matrix(full.reshape(-1, 3)).view(B, 16, 3, 3)
        eye = self._eye3
        pose_map = (R[:, 1:] - eye).reshape(B, 135)

        v_shaped = torch.einsum("vck,bk->bvc", self.th_shapedirs, th_betas) + self.th_v_template
        J = torch.einsum("jv,bvc->bjc", self.th_J_regressor, v_shaped)
        v_posed = v_shaped + torch.einsum("vck,bk->bvc", self.th_posedirs, pose_map)

        # forward kinematics: world transform of every joint
        bottom = self._bottom.expand(B, 1, 4)
        G = []
        for j, par in enumerate(_PARENTS):
            t = J[:, j] if par < 0 else J[:, j] - J[:, par]
            local = torch.cat([torch.cat([R[:, j], t.unsqueeze(2)], 2), bottom], 1)
            # 4x4 chain products as broadcast multiply-sums: a batched matmul over B tiny matrices goes to a rocBLAS kernel
            # that takes ~0.8 ms per call at B = 160 (profiles/r02_bench_kernel_stats_final.csv, Cijk_* MT16x16 rows)
            G.append(local if par < 0 else (G[par].unsqueeze(3) * local.unsqueeze(1)).sum(2))
        G = torch.stack(G, 1)                                        # (B,16,4,4)
        # skinning transforms: remove the rest-pose joint location
        rest = (G[:, :, :3, :3] * J.unsqueeze(2)).sum(3, keepdim=True)   # (B,16,3,1) = G[:3,:3] . J
        A = G[:, :, :3, :].clone()
        A[:, :, :, 3:] = A[:, :, :, 3:] - rest
        T = torch.einsum("vj,bjrc->bvrc", self.th_weights, A)       # (B,778,3,4)
        verts = (T[..., :3] * v_posed.unsqueeze(2)).sum(3) + T[..., 3]   # per-vertex 3x3 . 3 (B*778 of them)

        jtr = torch.cat([G[:, :, :3, 3], verts.index_select(1, self._tip_idx)], 1).index_select(1, self._joint_idx)
        if self.center_idx is not None:
            c = jtr[:, self.center_idx].unsqueeze(1)
            jtr = jtr - c
            verts = verts - c
        return verts * 1000, jtr * 1000
