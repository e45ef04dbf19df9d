"""Runtime harness with the reference's driver surface:

  * ``Trainer`` / ``Tester``: same roles as common/base.py:59-193 - AdamW(lr=cfg.lr) +
    StepLR(cfg.lr_drop, cfg.lr_decay_gamma), the 1e-5 learning-rate floor (base.py:30-32), the loss
    weighting of main/train.py:113-127;
  * checkpoints in the reference's on-disk format: ``snapshot_{epoch}_{iter}.pth.tar`` holding
    {"epoch", "network", "optimizer", "lr_scheduler"} with every network key prefixed ``module.``
    (the DataParallel prefix, common/base.py:113-150), resumed from the highest (epoch, iter);
  * released reference checkpoints load with ``strict=True`` (state-dict schema checked against the
    reference in tests/test_checkpoint_schema.py).
Redesigned for MI355X: one process per GPU (torchrun), gradients reduced by
``hoisdf_amd.ddp.GradReducer`` over RCCL instead of single-process ``nn.DataParallel``.
Datasets are licence-gated and absent offline: ``SyntheticDataset`` yields DexYCB / HO3D shaped
samples with the schema of data/dexycb.py:627-655; a real dataset object with the same
``__getitem__`` contract can be passed instead.
"""
from __future__ import annotations

import glob
import os
import os.path as osp
import re
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist

from . import testing as T
from .config import Config
from .ddp import GradReducer, reducible_parameters
from .model import get_model

LOSS_WEIGHTS = {  # main/train.py:115-127 <- cfg attribute
    "sdfhand_loss": "sdf_hand_weight", "sdfobj_loss": "sdf_obj_weight", "joint_heatmap": "hm_weight",
    "obj_seg": "obj_hm_weight", "hand_seg": "obj_hm_weight", "obj_rot": "obj_rot_weight",
    "obj_trans": "obj_trans_weight", "loss_joint_3d": "joint_weight", "loss_joint_cls": "cls_weight",
    "loss_all_joint_3d": "joint_weight"}


def _mix_seed(epoch_seed: int, draw: int) -> int:
    """splitmix64 of (per-epoch per-rank seed, draw index within the epoch): streams of different ranks / epochs / resumed
    runs are unrelated instead of shifted copies of each other"""
    x = (epoch_seed * 0x9E3779B97F4A7C15 + draw * 0xBF58476D1CE4E5B9 + 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 31
    return x >> 3            # 61 bits: SdfStore.sample derives 4 sub-streams as seed * 4 + i in 64 bits


def weighted_total(model_out: Dict[str, torch.Tensor], cfg: Config):
    """main/train.py:111-138: split on the ``_out`` suffix, mean every loss, apply cfg weights, sum."""
    out = {k[:-4]: v for k, v in model_out.items() if "_out" in k}
    loss = {k: v.mean() for k, v in model_out.items() if "_out" not in k}
    for k, attr in LOSS_WEIGHTS.items():
        if k in loss:
            loss[k] = loss[k] * getattr(cfg, attr)
    return sum(loss.values()), loss, out


def adjust_learning_rate(lr_scheduler, optimizer, floor: float = 1e-5):
    """common/base.py:30-32."""
    if lr_scheduler.get_last_lr()[-1] < floor:
        for g in optimizer.param_groups:
            g["lr"] = floor


class SyntheticSdfDataset(torch.utils.data.Dataset):
    """(f2) the dataset contract when the SDF rows live in HBM (``hoisdf_amd.sdf_data.SdfStore``): ``__getitem__`` yields
    everything the reference's does EXCEPT the four point sets and the two SDF targets; in their place meta_info carries
    ``sdf_frame`` (row of the store), ``do_flip`` and ``aug_rot`` (the 3x3 in-plane augmentation rotation) and
    ``Trainer(sdf_store=...)`` draws the points on the device (data/dexycb.py:514-549 + :288, :593-620)."""

    def __init__(self, cfg: Config, n_frames: int, length: int = 64, seed: int = 0):
        self.cfg, self.n_frames, self.length, self.seed = cfg, n_frames, length, seed

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        import math
        ins, tg, mt = T.synthetic_batch(1, 1, 1, seed=self.seed * 100003 + i)
        sq = lambda d: {k: v[0] for k, v in d.items()}
        ins, tg, mt = sq(ins), sq(tg), sq(mt)
        for k in ("hand_sdf_points", "obj_sdf_points", "hand_pre_points", "obj_pre_points"):
            ins.pop(k)
        tg.pop("hand_sdf"), tg.pop("obj_sdf")
        g = torch.Generator().manual_seed(self.seed * 7919 + i)
        ang = float((torch.rand(1, generator=g) - 0.5) * math.pi / 3)             # +-30 degrees, like cfg.max_rot
        c, s_ = math.cos(ang), math.sin(ang)
        mt["sdf_frame"] = torch.tensor(i % self.n_frames)
        mt["do_flip"] = torch.rand(1, generator=g)[0] < 0.5
        mt["aug_rot"] = torch.tensor([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]])
        return ins, tg, mt


class SyntheticDataset(torch.utils.data.Dataset):
    def __init__(self, cfg: Config, length: int = 64, seed: int = 0):
        self.cfg, self.length, self.seed = cfg, length, seed

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        ins, tg, mt = T.synthetic_batch(1, self.cfg.num_samp_hand, self.cfg.num_samp_obj, seed=self.seed * 100003 + i)
        sq = lambda d: {k: v[0] for k, v in d.items()}
        return sq(ins), sq(tg), sq(mt)


def reserve_hbm_pool(model: torch.nn.Module, fraction: float = 0.5, device=None) -> int:
    """After the first steps: hand PyTorch's caching allocator `fraction` of the peak it has seen so far as FREE cached blocks, one per
    stream the model launches on (blocks are owned by the stream they were freed on).  The step's allocation pattern is not exactly
    repeatable - two streams race, a training step after `cfg.point_sampling_epoch` draws its point branch - and a request the cache
    cannot serve costs a hipMalloc: synchronous, 50-250 ms for the multi-GB workspaces of `sdf_infer` (measured: the first branch-B
    step of a run took 383 ms instead of 90).  288 GB of HBM per GPU is there to be laid out once.  Returns the bytes reserved."""
    if not torch.cuda.is_available():
        return 0
    dev = torch.device(device) if device is not None else next(model.parameters()).device
    if dev.type != "cuda":
        return 0
    peak = torch.cuda.max_memory_allocated(dev)
    free, _ = torch.cuda.mem_get_info(dev)
    extra = min(int(fraction * peak), int(0.25 * free))
    streams = [torch.cuda.current_stream(dev)]
    side = getattr(model, "_side_stream", None) or getattr(getattr(model, "module", None), "_side_stream", None)
    if side is not None:
        streams.append(side)
    per = extra // len(streams) // (1 << 20) * (1 << 20)
    if per <= 0:
        return 0
    for st in streams:
        with torch.cuda.stream(st):
            t = torch.empty(per, dtype=torch.uint8, device=dev)
            del t
    return per * len(streams)


def snapshot_path(model_dir: str, epoch: int, itr: int) -> str:
    return osp.join(model_dir, f"snapshot_{epoch}_{itr}.pth.tar")


def latest_snapshot(model_dir: str) -> Optional[Tuple[str, int, int]]:
    """highest epoch, then highest iteration (common/base.py:121-142)."""
    best = None
    for f in glob.glob(osp.join(model_dir, "snapshot_*.pth.tar")):
        m = re.search(r"snapshot_(\d+)_(\d+)\.pth\.tar$", f)
        if m:
            key = (int(m.group(1)), int(m.group(2)))
            if best is None or key > best[1:]:
                best = (f, *key)
    return best


def to_reference_state_dict(model: torch.nn.Module) -> Dict[str, torch.Tensor]:
    return {"module." + k: v for k, v in model.state_dict().items()}


def load_reference_state_dict(model: torch.nn.Module, network: Dict[str, torch.Tensor], strict: bool = True):
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in network.items()}
    return model.load_state_dict(sd, strict=strict)


class Trainer:
    def __init__(self, cfg: Config, device: torch.device, dataset=None, batch_size: Optional[int] = None,
                 channels_last: bool = True, tune_encoder: bool = True, sdf_store=None):
        self.cfg, self.device = cfg, device
        self.sdf_store = sdf_store          # (f2) HBM-resident sdf_processed rows: the query points are drawn on the device
        self._sdf_draws = 0
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.base_seed = int(getattr(cfg, "seed", 0))
        torch.manual_seed(self.base_seed)                        # identical initial weights on every rank
        self.model = get_model("train", cfg=cfg).to(device).train()
        if channels_last:
            if device.type == "cuda" and tune_encoder:
                from . import miopen_tuning
                miopen_tuning.enable()       # tuned fp32 NHWC conv solvers for the encoder (shipped find-db)
            self.model.backbone_net.to(memory_format=torch.channels_last)
            self.model.decoder_net.to(memory_format=torch.channels_last)
        self.channels_last = channels_last
        on_gpu = device.type == "cuda"
        self.reducer = GradReducer(reducible_parameters(self.model), average=not on_gpu)
        # reference: AdamW over ALL named parameters in one group, frozen BN affine included (common/base.py:64-73):
        # the optimizer state_dict's ``params`` index list must have that length/order for snapshots to be
        # interchangeable.  Parameters without a gradient are skipped by the update either way.
        params = list(self.model.parameters())
        if on_gpu:
            from .optim import FusedAdamW                       # same rule / state layout, one launch, 1/world folded in
            self.optimizer = FusedAdamW(params, lr=cfg.lr, grad_scale=1.0 / self.world)
        else:
            self.optimizer = torch.optim.AdamW(params, lr=cfg.lr)
        self.lr_scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=cfg.lr_drop,
                                                            gamma=cfg.lr_decay_gamma)
        self.start_epoch = 0
        if dataset is None and sdf_store is not None:
            dataset = SyntheticSdfDataset(cfg, sdf_store.n_frames, seed=self.rank)
        ds = dataset if dataset is not None else SyntheticDataset(cfg, seed=self.rank)
        bs = batch_size or cfg.train_batch_size
        sampler = torch.utils.data.distributed.DistributedSampler(ds, self.world, self.rank, shuffle=True) \
            if self.world > 1 else None
        self.batch_generator = torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=sampler is None, sampler=sampler,
                                                           num_workers=0, drop_last=True, pin_memory=True)
        self.sampler = sampler
        self.itr_per_epoch = len(self.batch_generator)
        self.begin_epoch(self.start_epoch)

    def begin_epoch(self, epoch: int):
        """Per-epoch, per-rank random streams: a new DistributedSampler shuffle every epoch, and distinct dropout masks /
        point jitter on every rank (the weights were initialised from the shared seed above).  Also what --continue
        resumes with, so a resumed epoch does not replay the streams of epoch 0."""
        if self.sampler is not None:
            self.sampler.set_epoch(epoch)
        from . import ops
        s = self.base_seed + 1000003 * (epoch + 1) + 7919 * self.rank
        torch.manual_seed(s)
        ops.manual_seed(s)
        self.model._py_random.seed(s)                            # the branch A / B draw (main/model.py:426)
        self._epoch_seed, self._sdf_draws = s, 0                 # the device-side SDF point draw restarts per epoch too

    def train_step(self, inputs, targets, meta, epoch: int, batch_ratio: float):
        dev = self.device
        inputs, targets, meta = (T.to_device(x, dev) for x in (inputs, targets, meta))
        if self.sdf_store is not None and "sdf_frame" in meta:
            c = self.cfg
            self._sdf_draws += 1
            pts = self.sdf_store.make_inputs(
                meta["sdf_frame"].cpu(), meta["mano_root"], meta["obj_center_cam"], c.num_samp_hand, c.num_samp_obj,
                c.points_filter_dist, c.hand_sdf_scale, c.obj_sdf_scale, train=True,
                seed=_mix_seed(self._epoch_seed, self._sdf_draws),
                do_flip=meta.get("do_flip"), rot_mat=meta.get("aug_rot"))
            for k in ("hand_sdf_points", "obj_sdf_points", "hand_pre_points", "obj_pre_points"):
                inputs[k] = pts[k]
            targets["hand_sdf"], targets["obj_sdf"] = pts["hand_sdf"], pts["obj_sdf"]
        if self.channels_last:
            inputs["img"] = inputs["img"].contiguous(memory_format=torch.channels_last)
        self.reducer.zero_grad()
        out = self.model(inputs, targets, meta, "train", epoch, batch_ratio)
        total, loss, _ = weighted_total(out, self.cfg)
        total.backward()
        self.reducer.finish()
        self.optimizer.step()
        self._steps_done = getattr(self, "_steps_done", 0) + 1
        if self._steps_done == 3 and dev.type == "cuda":
            # the allocator has seen the step's pattern: leave it free cached blocks on both streams, so that a later step that
            # allocates differently (the other point branch, the two streams racing) does not stall in a hipMalloc
            self.hbm_pool_bytes = reserve_hbm_pool(self.model, float(getattr(self.cfg, "hbm_pool_fraction", 0.5)), dev)
        return total.detach(), {k: v.detach() for k, v in loss.items()}

    def save_model(self, epoch: int, itr: int) -> Optional[str]:
        if self.rank != 0:
            return None
        os.makedirs(self.cfg.model_dir, exist_ok=True)
        path = snapshot_path(self.cfg.model_dir, epoch, itr)
        torch.save({"epoch": epoch, "network": to_reference_state_dict(self.model),
                    "optimizer": self.optimizer.state_dict(), "lr_scheduler": self.lr_scheduler.state_dict()}, path)
        return path

    def load_model(self) -> int:
        found = latest_snapshot(self.cfg.model_dir)
        if found is None:
            return 0
        ckpt = torch.load(found[0], map_location=self.device)
        load_reference_state_dict(self.model, ckpt["network"], strict=True)
        self.optimizer.load_state_dict(ckpt["optimizer"])
        self.lr_scheduler.load_state_dict(ckpt["lr_scheduler"])
        self.start_epoch = ckpt["epoch"] + 1
        return self.start_epoch


class Tester:
    def __init__(self, cfg: Config, device: torch.device, ckpt_path: Optional[str] = None):
        self.cfg, self.device = cfg, device
        self.model = get_model("test", cfg=cfg).to(device).eval()
        if ckpt_path:
            ckpt = torch.load(ckpt_path, map_location=device)
            load_reference_state_dict(self.model, ckpt["network"], strict=True)

    @torch.no_grad()
    def predict(self, inputs, targets, meta, mano_layer=None):
        """eval forward; for the IK variant (``cfg.use_inverse_kinematics``) the closed-form IK post-process of
        main/test.py:139-160 runs on the device as well when a MANO layer is given: adds ``ik_joints_out`` /
        ``ik_verts_out`` (root-relative metres, before the caller's mano_root shift) and ``ik_pose_out``."""
        inputs, targets, meta = (T.to_device(x, self.device) for x in (inputs, targets, meta))
        out = self.model(inputs, targets, meta, "eval")
        if self.cfg.use_inverse_kinematics and mano_layer is not None:
            from .ik import ik_solver_mano
            hj = torch.cat([torch.zeros_like(out["hand_joints_out"][:, :1]), out["hand_joints_out"]], 1)
            r = ik_solver_mano(mano_layer.to(self.device), out.get("mano_shape_out"), hj)
            out["ik_joints_out"], out["ik_verts_out"], out["ik_pose_out"] = r["joints"], r["verts"], r["pose"]
        return out


def mpjpe(pred: torch.Tensor, gt: torch.Tensor) -> float:
    """mean per-joint position error (common/metrics.py:213-232), inputs (N,J,3)."""
    return float((pred - gt).norm(dim=-1).mean())


def pa_mpjpe(pred: torch.Tensor, gt: torch.Tensor) -> float:
    """MPJPE after a per-sample similarity (Procrustes) alignment (common/metrics.py:188-232)."""
    errs = []
    for p, g in zip(pred.double().cpu(), gt.double().cpu()):
        mp, mg = p.mean(0), g.mean(0)
        p0, g0 = p - mp, g - mg
        U, S, Vt = torch.linalg.svd(p0.T @ g0)
        d = torch.sign(torch.linalg.det(U @ Vt))
        D = torch.diag(torch.tensor([1.0, 1.0, float(d)], dtype=torch.float64))
        R = U @ D @ Vt
        s = (S * torch.diag(D)).sum() / (p0 ** 2).sum()
        errs.append(float(((s * p0 @ R + mg) - g).norm(dim=-1).mean()))
    return sum(errs) / max(len(errs), 1)
