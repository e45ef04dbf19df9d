"""(f2) dataset-side SDF sample selection on the device.

The reference's dataset object loads one ``sdf_processed/<frame>.npy`` per sample ((N_h + N_o, 6) float32 rows
``[x y z sdf_hand sdf_obj label]``, written by tool/pre_process_sdf.py:140-148 together with ``sdf_index.npy`` =
``[[N_h, N_o], ...]``) and draws the query points with ``np.random.choice(..., replace=False)``
(data/dexycb.py:514-546).  At > 200 samples/s per GPU that CPU path is the bottleneck, so here the rows of all frames
of a shard live in HBM (288 GB per MI355X: ~10^10 rows) and a batch is drawn by three kernels:
uniform keys (``hoisdf_sdf_sample_keys``) -> k smallest keys per segment (``hoisdf_select_smallest_abs``) -> row fetch.
Same distribution as the reference (uniform without replacement, optional |sdf| < dist pre-filter), different random
stream.  Output layout = what the reference's ``__getitem__`` produces before augmentation: rows ordered
[hand_sdf | obj_sdf | hand_pre | obj_pre], ``sdf_points`` = columns 0-4, ``sdf_raw_label`` = column 5."""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import ops
from ._lib import call

_p, _st = ops._p, ops._st


def load_frame(path: str) -> np.ndarray:
    """one ``sdf_processed/<frame>.npy`` file"""
    a = np.load(path)
    if a.ndim != 2 or a.shape[1] != 6 or a.dtype != np.float32:
        raise ValueError(f"{path}: expected float32 (N, 6) rows [x y z sdf_hand sdf_obj label], got {a.dtype} {a.shape}")
    return a


class SdfStore:
    """HBM-resident rows of many frames + per-frame [first_row, n_hand, n_obj]."""

    def __init__(self, frames: Sequence[np.ndarray], index: np.ndarray, device="cuda"):
        index = np.asarray(index, dtype=np.int64).reshape(-1, 2)
        if len(frames) != len(index):
            raise ValueError("one sdf_index row per frame")
        for f, (nh, no) in zip(frames, index):
            if f.shape[0] != nh + no:             # data/dexycb.py:517
                raise ValueError(f"frame has {f.shape[0]} rows but sdf_index says {nh} + {no}")
        self.n_frames = len(frames)
        self.index = torch.from_numpy(index)                                         # host copy
        self.row0 = torch.from_numpy(np.concatenate([[0], np.cumsum(index.sum(1))]).astype(np.int64))
        rows = np.concatenate(frames, 0).astype(np.float32) if frames else np.zeros((0, 6), np.float32)
        self.rows = torch.from_numpy(rows).to(device)
        self.device = self.rows.device

    @classmethod
    def from_directory(cls, sdf_dir: str, frame_names: Optional[Sequence[str]] = None, device="cuda"):
        """``sdf_dir`` holds ``sdf_index.npy`` and ``sdf_processed/*.npy`` (tool/pre_process_sdf.py layout); the index
        rows follow the sorted file list unless ``frame_names`` gives the order."""
        proc = os.path.join(sdf_dir, "sdf_processed")
        names = list(frame_names) if frame_names is not None else sorted(
            f[:-4] for f in os.listdir(proc) if f.endswith(".npy"))
        index = np.load(os.path.join(sdf_dir, "sdf_index.npy"))
        return cls([load_frame(os.path.join(proc, n + ".npy")) for n in names], index[:len(names)], device)

    def sample(self, frame_ids, num_hand: int, num_obj: int, dist: float, train: bool, seed: int,
               validate: bool = True) -> Dict[str, torch.Tensor]:
        """frame_ids (B,) ints -> ``sdf_points`` (B, n, 5), ``sdf_raw_label`` (B, n), ``rows`` (B, n) store row ids;
        n = num_hand + num_obj (+ the same again in training: the |sdf| < dist "pre" points)."""
        fid = torch.as_tensor(frame_ids, dtype=torch.int64).reshape(-1)
        B = fid.numel()
        nh, no = self.index[fid, 0], self.index[fid, 1]
        base = self.row0[fid]
        # segments per sample: hand, obj, [hand_pre, obj_pre]; all segments of a kind share k
        kinds = [(base, nh, -1, num_hand), (base + nh, no, -1, num_obj)]
        if train:
            kinds += [(base, nh, 3, num_hand), (base + nh, no, 4, num_obj)]
        out_rows = []
        for gi, (r0, ln, col, k) in enumerate(kinds):
            off = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(ln, 0)])
            n_keys = int(off[-1])
            if n_keys >= 2 ** 31:
                raise ValueError("batch too large for int32 key offsets")
            d_r0 = r0.to(self.device)
            d_len = ln.to(self.device, torch.int32)
            d_off = off[:-1].to(self.device, torch.int32)
            d_col = torch.full((B,), col, dtype=torch.int32, device=self.device)
            keys = torch.empty(max(n_keys, 1), device=self.device, dtype=torch.float32)
            elig = torch.empty(B, device=self.device, dtype=torch.int32)
            call("hoisdf_sdf_sample_keys", _p(self.rows), 6, _p(d_r0), _p(d_len), _p(d_off), _p(d_col), B,
                 int(ln.max()) if B else 0, float(dist), int(seed) * 4 + gi, _p(keys), _p(elig), _st())
            if validate and B and int(elig.min()) < k:                     # np.random.choice would raise ValueError
                raise ValueError(f"a frame has only {int(elig.min())} eligible rows for {k} draws (segment kind {gi})")
            sel = ops.select_smallest_abs(keys, d_off, d_len, k)          # (B, k) indices into keys
            out_rows.append(sel.long() - d_off.long()[:, None] + d_r0[:, None])
        rows = torch.cat(out_rows, 1)                                     # (B, n) store rows, reference order
        data = self.rows[rows.reshape(-1)].view(B, rows.shape[1], 6)
        return {"sdf_points": data[..., :5], "sdf_raw_label": data[..., 5], "rows": rows}


    def make_inputs(self, frame_ids, hand_root: torch.Tensor, obj_center: torch.Tensor, num_hand: int, num_obj: int,
                    dist: float, hand_scale: float, obj_scale: float, train: bool, seed: int,
                    do_flip: Optional[torch.Tensor] = None, rot_mat: Optional[torch.Tensor] = None,
                    validate: bool = True, rows: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """The point entries of one batch exactly as the reference's ``__getitem__`` hands them to the model
        (data/dexycb.py:514-549 draw, :548-549 flip ``x *= -1``, data_aug :288 in-plane rotation ``p . rot_mat^T``,
        :593-620 centre + scale), all on the device:
          inputs : hand_sdf_points (B,N_h,3), obj_sdf_points (B,N_o,3) [, hand_pre_points, obj_pre_points]
          targets: hand_sdf (B,N_h) = column 3 * hand_scale, obj_sdf (B,N_o) = column 4 * obj_scale
        ``hand_root`` / ``obj_center`` (B,3) are the centres AFTER the same flip / rotation (the dataset object computes
        them from the augmented joints / bbox); ``do_flip`` (B,) bool, ``rot_mat`` (B,3,3).
        ``rows`` (B, n) store row ids in the reference's order [hand_sdf | obj_sdf | hand_pre | obj_pre]: hand off THESE rows
        instead of drawing (tests/golden/g14_sampler.npz feeds numpy's own draws through the device hand-off)."""
        if rows is None:
            smp = self.sample(frame_ids, num_hand, num_obj, dist, train, seed, validate)
        else:
            rows = torch.as_tensor(rows, dtype=torch.int64, device=self.device)
            smp = {"sdf_points": self.rows[rows.reshape(-1)].view(rows.shape[0], rows.shape[1], 6)[..., :5], "rows": rows}
        pts = smp["sdf_points"].clone()                                   # (B, n, 5): xyz, sdf_hand, sdf_obj
        B = pts.shape[0]
        if do_flip is not None:
            sgn = torch.where(do_flip.to(self.device).bool(), -1.0, 1.0).view(B, 1)
            pts[..., 0] = pts[..., 0] * sgn
        if rot_mat is not None:
            pts[..., :3] = pts[..., :3] @ rot_mat.to(self.device).transpose(1, 2)
        hr, oc = hand_root.to(self.device)[:, None], obj_center.to(self.device)[:, None]
        a, b = num_hand, num_hand + num_obj
        hand, obj = pts[:, :a].clone(), pts[:, a:b].clone()
        hand[..., :3] -= hr
        obj[..., :3] -= oc
        hand, obj = hand * hand_scale, obj * obj_scale
        out = {"hand_sdf_points": hand[..., :3].contiguous(), "obj_sdf_points": obj[..., :3].contiguous(),
               "hand_sdf": hand[..., 3].contiguous(), "obj_sdf": obj[..., 4].contiguous(), "rows": smp["rows"]}
        if train:
            c = b + num_hand
            out["hand_pre_points"] = ((pts[:, b:c, :3] - hr) * hand_scale).contiguous()
            out["obj_pre_points"] = ((pts[:, c:, :3] - oc) * obj_scale).contiguous()
        return out


def synthetic_store(n_frames: int, rows_hand: int = 6000, rows_obj: int = 4000, seed: int = 0, device="cuda") -> "SdfStore":
    """A store shaped like tool/pre_process_sdf.py's output (points in a ~0.3 m box around the hand at z ~ 0.7 m,
    |sdf| small for ~half of the rows) for smoke runs without the licence-gated datasets."""
    r = np.random.default_rng(seed)
    frames, index = [], []
    for _ in range(n_frames):
        nh, no = rows_hand + int(r.integers(0, 200)), rows_obj + int(r.integers(0, 200))
        a = np.zeros((nh + no, 6), np.float32)
        a[:, :3] = r.uniform(-0.1, 0.1, (nh + no, 3)) + np.array([0.0, 0.0, 0.7], np.float32)
        a[:, 3] = r.uniform(-0.02, 0.1, nh + no)
        a[:, 4] = r.uniform(-0.02, 0.1, nh + no)
        a[:, 5] = r.integers(0, 6, nh + no)
        frames.append(a)
        index.append([nh, no])
    return SdfStore(frames, np.asarray(index), device)
