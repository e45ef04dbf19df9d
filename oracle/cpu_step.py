"""CPU baseline step = the reference's training step restated on host cores: PyTorch-CPU CNN
encoder/decoder + the oracle hot path (oracle/hoisdf_oracle.py) + aux image losses + loss
weighting (main/train.py:113-127) + backward + AdamW.  TEST / BENCH INFRASTRUCTURE ONLY
(``bench.py``'s ``cpu_baseline`` leg times it; the product never imports it)."""
from __future__ import annotations

import random
import time
from typing import Dict

import torch
import torch.nn.functional as F

from hoisdf_amd import testing as T
from hoisdf_amd.nets import encoder as ENC
from hoisdf_amd.nets import mano as MANO
from oracle import hoisdf_oracle as R

LOSS_WEIGHTS = dict(sdfhand_loss=50, sdfobj_loss=25, joint_heatmap=100 / 100000, obj_seg=1, hand_seg=1,
                    obj_rot=0.7, obj_trans=100.0, loss_joint_3d=0.1, loss_joint_cls=1.0, loss_all_joint_3d=0.1)


class CpuTrainer:
    def __init__(self, n_hand: int, n_obj: int, resnet_type: int = 50, seed: int = 0, train: bool = True,
                 setting: str = "dexycb"):
        torch.manual_seed(seed)
        ik = setting == "ho3d_render"
        self.cfg = R.OracleCfg(num_samp_hand=n_hand, num_samp_obj=n_obj, bins_n=64, use_inverse_kinematics=ik,
                               dataset="ho3d" if "ho3d" in setting else "dexycb")
        self.backbone = ENC.BackboneNet(resnet_type)
        self.decoder = ENC.DecoderNet(resnet_type, big=False)
        for n, p in self.backbone.named_parameters():
            if "bn" in n:
                p.requires_grad = False
        self.P = T.det_params(T.hot_path_param_shapes(992, ik=ik))
        for v in self.P.values():
            v.requires_grad_(True)
        self.mano = MANO.ManoLayer(MANO.synthetic_assets(0))
        self.training = train
        self.backbone.train(train)
        self.decoder.train(train)
        params = [p for p in list(self.backbone.parameters()) + list(self.decoder.parameters()) if p.requires_grad]
        params += list(self.P.values())
        self.opt = torch.optim.AdamW(params, lr=1e-4)
        self.rng = random.Random(seed)

    def forward_loss(self, inputs, targets, meta):
        if not self.training:
            with torch.no_grad():
                img_feat, skips = self.backbone(inputs["img"])
                pyr, dec_out = self.decoder(img_feat, skips)
                R.hot_path_forward(self.P, self.cfg, pyr, inputs, targets, meta, "eval", mano_layer=self.mano,
                                   hands_mean=self.mano.th_hands_mean, rng=self.rng)
            return None
        img_feat, skips = self.backbone(inputs["img"])
        pyr, dec_out = self.decoder(img_feat, skips)
        out = R.hot_path_forward(self.P, self.cfg, pyr, inputs, targets, meta, "train" if self.training else "eval",
                                 0, 0.1, mano_layer=self.mano, hands_mean=self.mano.th_hands_mean, rng=self.rng)
        if not self.training:
            return None
        loss = {k: v.mean() for k, v in out.items() if "_out" not in k}
        # aux image losses (main/model.py:404-422)
        c = self.cfg
        x = torch.arange(128).float()
        yy, xx = torch.meshgrid(x, x, indexing="ij")
        jc = targets["joint_coord"]
        hm = torch.exp(-(((xx[None, None] - jc[:, :, 0, None, None]) / 1.25) ** 2) / 2
                       - (((yy[None, None] - jc[:, :, 1, None, None]) / 1.25) ** 2) / 2).sum(1) * 255
        loss["joint_heatmap"] = ((dec_out[:, 0] - hm) ** 2).mean()
        loss["obj_seg"] = F.binary_cross_entropy(dec_out[:, 2], targets["obj_seg"])
        loss["hand_seg"] = F.binary_cross_entropy(dec_out[:, 1], targets["hand_seg"])
        return sum(v * LOSS_WEIGHTS.get(k, 1.0) for k, v in loss.items())

    def step(self, inputs, targets, meta) -> float:
        self.opt.zero_grad()
        total = self.forward_loss(inputs, targets, meta)
        if total is None:
            return 0.0
        total.backward()
        self.opt.step()
        return float(total.detach())


def time_cpu_baseline(n_hand: int, n_obj: int, batch: int, iters: int = 3, warmup: int = 1,
                      resnet_type: int = 50, train: bool = True, threads: int = 0, setting: str = "dexycb") -> Dict:
    import os
    torch.set_num_threads(threads if threads > 0 else (os.cpu_count() or 1))
    tr = CpuTrainer(n_hand, n_obj, resnet_type, train=train, setting=setting)
    inputs, targets, meta = T.synthetic_batch(batch, n_hand, n_obj, seed=1234)
    for _ in range(warmup):
        tr.step(inputs, targets, meta)
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        tr.step(inputs, targets, meta)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    return dict(value=batch / med, unit="samples/s", cores=torch.get_num_threads(), kind="port",
                sample=(f"{warmup} warm-up + {iters} timed fwd+bwd+AdamW steps of batch {batch} "
                        f"(ResNet-{resnet_type}, {n_hand}+{n_obj} query points, branch A, dropout on), median "
                        f"{med:.2f} s/step") if train else
                       (f"{warmup} warm-up + {iters} timed eval forwards (64^3-lattice sdf_infer) of batch {batch} "
                        f"(ResNet-{resnet_type}, {n_hand}+{n_obj} query points, {setting}), median {med:.2f} s/iter"),
                threads=torch.get_num_threads(), seconds_per_step=med)
