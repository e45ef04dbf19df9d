#!/usr/bin/env python3
"""Training driver with the reference's CLI (main/train.py:24-49): --gpu --continue --run_dir_name --end_epoch
--point_sampling_epoch --lr_drop.  One process per GPU:
    python train.py --run_dir_name demo --gpu 0
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py --run_dir_name demo --gpu 0-7
Datasets are licence-gated and absent here: the loop runs on DexYCB/HO3D-shaped synthetic samples (a dataset object with
the schema of data/dexycb.py:627-655 can be handed to hoisdf_amd.engine.Trainer(dataset=...) from Python)."""
import argparse
import os
import time

# the host driver only supports dmabuf IPC: without this RCCL / cross-process CUDA-tensor sharing fails with
# `hipIpcGetMemHandle: invalid argument` (already exported on the GPU box; harmless to repeat)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

from hoisdf_amd.config import cfg
from hoisdf_amd.engine import Trainer, adjust_learning_rate


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", type=str, dest="gpu_ids", default="0")
    ap.add_argument("--continue", dest="continue_train", action="store_true")
    ap.add_argument("--run_dir_name", type=str, required=True)
    ap.add_argument("--end_epoch", type=int, default=None)
    ap.add_argument("--point_sampling_epoch", type=int, default=None)
    ap.add_argument("--lr_drop", type=int, default=None)
    ap.add_argument("--setting", type=str, default="dexycb", help="the reference edits Config.setting in config.py")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--max_iters", type=int, default=None, help="stop after this many iterations (smoke runs)")
    a = ap.parse_args()
    if "-" in a.gpu_ids:                                   # "0-3" -> "0,1,2,3" (main/train.py:43-47)
        lo, hi = a.gpu_ids.split("-")
        a.gpu_ids = ",".join(str(i) for i in range(int(lo), int(hi) + 1))
    return a


def main():
    a = parse_args()
    cfg.apply_setting(a.setting)
    cfg.set_args(a.gpu_ids, a.run_dir_name, a.continue_train)
    for k in ("end_epoch", "point_sampling_epoch", "lr_drop"):
        if getattr(a, k) is not None:
            setattr(cfg, k, getattr(a, k))
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", 1), ("RANK", 0), ("LOCAL_RANK", 0)))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    trainer = Trainer(cfg, dev, batch_size=a.batch)
    if a.continue_train:
        trainer.load_model()
    it_total = 0
    for epoch in range(trainer.start_epoch, cfg.end_epoch):
        adjust_learning_rate(trainer.lr_scheduler, trainer.optimizer)
        trainer.begin_epoch(epoch)
        t0 = time.time()
        for itr, (inputs, targets, meta) in enumerate(trainer.batch_generator):
            total, loss = trainer.train_step(inputs, targets, meta, epoch, itr / max(trainer.itr_per_epoch, 1))
            it_total += 1
            if rank == 0 and itr % 10 == 0:
                print(f"Epoch {epoch}/{cfg.end_epoch} itr {itr}/{trainer.itr_per_epoch}: lr {trainer.lr_scheduler.get_last_lr()[-1]:g} "
                      f"loss {float(total):.4f} " + " ".join(f"loss_{k}: {float(v):.4f}" for k, v in loss.items()), flush=True)
            if a.max_iters and it_total >= a.max_iters:
                break
        trainer.lr_scheduler.step()
        trainer.save_model(epoch, itr)
        if rank == 0:
            print(f"epoch {epoch} done in {time.time() - t0:.1f}s")
        if a.max_iters and it_total >= a.max_iters:
            break
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
