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
