#!/usr/bin/env python3
"""HOISDF hot-path benchmark (driver contract: one JSON line on rank 0).

Workload = BASELINE.json configs[1]: DexYCB-shaped synthetic batch, B = 32 per GPU, 2048 SDF
query points (1536 hand + 512 object, the reference's 3:1 split), ResNet-50 encoder (the
reference has no HRNet; the encoder is outside the HIP scope and stays PyTorch/MIOpen), one
full training step = encoder fwd + HIP hot path fwd + losses + backward + gradient all-reduce
(RCCL) + AdamW, train mode (dropout on), point branch A (reference main/model.py:427-460,
what the first 40 epochs run).  `value` = samples/s of the whole job, inputs resident in HBM.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --config 3          # BASELINE configs[3]: HO3Dv2-shape B=16, 3072+1024 points, IK variant, inference
    python bench.py --config 4          # BASELINE configs[4]: dense eval, 6144+2048 points, f16-MFMA attention, B=8 over 2 GPUs
    python bench.py --branch-mix        # configs[1] with the epoch >= 40 point-branch mix (40 % pre-points, 60 % sdf_infer)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 10 --warmup 3
"""
import argparse
import json
import os
import sys
import time

# the host driver only supports dmabuf IPC: without this RCCL / cross-process CUDA-tensor sharing fails with
# `hipIpcGetMemHandle: invalid argument` (already exported on the GPU box; harmless to repeat)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_F32_TFLOPS = 157.3          # MI355X_MICROARCH.md: f32 vector = f32 MFMA peak
PEAK_F16_TFLOPS = 2500.0         # MI355X_MICROARCH.md:42: dense f16/bf16 MFMA (no sparsity)
# the arithmetic type of the headline line: every contraction is fp32 - either the exact-f32 MFMA or (default for the large
# linear layers and the attention forward) fp32 emulated with EXACT 3-way bf16 operand splits, 6 products, f32 accumulation:
# error vs fp64 at or below the exact-f32 kernels' and the vendor fp32 GEMM's (tests/test_gpu_emu.py, tools/emu_accuracy.py)
DTYPE_F32 = "f32"                  # every contraction on the exact-f32 MFMA (--gemm f32 --attention f32)
DTYPE_EMU = "f32 (bf16x3-emulated contractions, f32 accumulate)"   # HOISDF_EMU_FORM=b3: 3-way exact bf16 split, 6 products, f32 accumulation
# the default since round 5: linear layers AND the encoder attention in the f16x2 form (two scaled f16 pieces per operand, 3 products:
# include/hoisdf.h); HOISDF_EMU_FORM=b3 brings the bf16x3 arithmetic back for both
DTYPE_EMU_H2 = ("f32 (emulated contractions, f32 accumulate, f16x2 form: operands as scaled hi + lo f16 pieces x 3 products - 22 operand bits, one power-of-two "
                "scale per ROW of a linear layer's row operand, per (sample, head) of the attention operands, per weight matrix; the attention backward's dS as three pieces x 5)")
PMC_FILE = "r06_pmc.json"
# what a BARE v_mfma_f32_32x32x16_bf16 stream (registers only, one wave per SIMD) sustains on this power-capped board (1400 W) when the
# operands are the bf16x3 pieces of N(0,1) values / uniform random values: 1542-1568 TF of the 2500 TF datasheet peak (2044-2100 TF on
# zero / constant operands) - tools/ubench/mfma_data.hip, profiles/r05_mfma_rate_vs_operand_data.txt
MEASURED_MFMA_CEILING_TFLOPS = 1555.0
# the same probe with v_mfma_f32_32x32x16_f16 on the hi / lo f16 pieces of N(0,1) values: 1345-1380 TF (tools/ubench/mfma_data_f16.hip,
# profiles/r05_mfma_rate_vs_operand_data_f16.txt)
MEASURED_MFMA_CEILING_F16_TFLOPS = 1365.0
STEP_TRACE = "r06_bench_kernel_stats.csv"      # rocprofv3 kernel trace of this bench restricted to the timed steps (tools/trace_stats.py)
LOSS_WEIGHTS = dict(sdfhand_loss=50, sdfobj_loss=25, joint_heatmap=100 / 100000, obj_seg=1, hand_seg=1,
                    obj_rot=0.7, obj_trans=100.0, loss_joint_3d=0.1, loss_joint_cls=1.0, loss_all_joint_3d=0.1)


class KernelTimer:
    """HIP events (torch.cuda.Event on the current stream = the stream the kernels are launched on)
    around selected C-ABI calls; accumulates algorithmic FLOPs per call family."""

    FLOPS = {
        # (x, ldx, W, ldw, bias, y, ldy, M, N, K, ...)
        "hoisdf_linear_fwd": lambda a: 2.0 * a[7] * a[8] * a[9],
        # (dy, lddy, bits, p, W, ldw, dx, lddx, M, N, K)
        "hoisdf_linear_bwd_input": lambda a: 2.0 * a[8] * a[9] * a[10],
        # (dy, lddy, bits, p, x, ldx, dW, lddw, db, M, N, K, ws, nws)
        "hoisdf_linear_bwd_weight": lambda a: 2.0 * a[9] * a[10] * a[11],
        # (q,ldq,k,ldk,v,ldv,o,ldo,lse,B,H,Lq,Lk,kv_len,...): QK^T + PV
        "hoisdf_attention_fwd": lambda a: 4.0 * a[9] * a[10] * a[11] * a[13] * 64,
        # (q,ldq,k,ldk,v,ldv,o,ldo,do,lddo,lse,delta,dq,dk,dv,B,H,Lq,Lk,kv_len,...): 5 GEMM-equivalents
        "hoisdf_attention_bwd": lambda a: 10.0 * a[15] * a[16] * a[17] * a[19] * 64,
        # (q,ldq,k,ldk,v,ldv,o,ldo,B,H,Lq,Lk,kv_len,ws,nws): algorithmic QK^T + PV (the 3x split products are not counted)
        "hoisdf_attention_fwd_f16": lambda a: 4.0 * a[8] * a[9] * a[10] * a[12] * 64,
        "hoisdf_attention_fwd_bf16x2": lambda a: 4.0 * a[8] * a[9] * a[10] * a[12] * 64,      # same argument list
        # emulated fp32 attention: (q,ldq,k,ldk,v,ldv,o,ldo,lse,B,H,Lq,Lk,kv_len,...) / (q,..,o,ldo,do,lddo,lse,delta,dq,dk,dv,B,H,Lq,Lk,kv_len,...)
        "hoisdf_attention_fwd_emu": lambda a: 4.0 * a[9] * a[10] * a[11] * a[13] * 64,
        "hoisdf_attention_bwd_emu": lambda a: 10.0 * a[15] * a[16] * a[17] * a[19] * 64,
        # the f16x2 form: the same argument lists with the magnitude words appended
        "hoisdf_attention_fwd_emu_mag": lambda a: 4.0 * a[9] * a[10] * a[11] * a[13] * 64,
        "hoisdf_attention_bwd_emu_mag": lambda a: 10.0 * a[15] * a[16] * a[17] * a[19] * 64,
        # (q,ldq,k,ldk,v,ldv,o,ldo,lse,B,H,Lq,Lk,kv_len,...) / (q,..,o,ldo,do,lddo,lse,delta,dq,dk,dv,B,H,Lq,Lk,kv_len,...)
        # fp32 emulated on the bf16 pipe: (x, ldx, image, bias, y, ldy, M, N, K, ...) / (dy, lddy, bits, p, image, dx, lddx, M, N, K, ...);
        # algorithmic FLOPs (the six bf16 products per product are not counted)
        "hoisdf_linear_fwd_emu": lambda a: 2.0 * a[6] * a[7] * a[8],
        "hoisdf_linear_bwd_input_emu": lambda a: 2.0 * a[7] * a[8] * a[9],
        "hoisdf_linear_bwd_weight_emu": lambda a: 2.0 * a[9] * a[10] * a[11],
        # the same entries with magnitude words (f16x2 form): the words are appended to the argument lists
        "hoisdf_linear_fwd_emu_mag": lambda a: 2.0 * a[6] * a[7] * a[8],
        "hoisdf_linear_bwd_input_emu_mag": lambda a: 2.0 * a[7] * a[8] * a[9],
        "hoisdf_linear_bwd_weight_emu_mag": lambda a: 2.0 * a[9] * a[10] * a[11],
        # the one-wave-per-tile emulated form for < 2048 rows: the argument lists of the exact-f32 entries (no workspace)
        "hoisdf_linear_fwd_emu_small": lambda a: 2.0 * a[7] * a[8] * a[9],
        "hoisdf_linear_bwd_input_emu_small": lambda a: 2.0 * a[8] * a[9] * a[10],
        "hoisdf_linear_bwd_weight_emu_small": lambda a: 2.0 * a[9] * a[10] * a[11],
        # the gradient-free SDF query (K1-K4 behind one C-ABI call): its six GEMMs, 2 (C*512 + 512*256) +
        # 2 (289*512 + 512*223 + 512*512 + 512*512 + 512) FLOP per point; the gather / posenc time inside the call is
        # charged to the GEMM family as well
        "hoisdf_sdf_query_fwd": lambda a: a[3] * (2.0 * (a[12]._obj.C * 512 + 512 * 256) + 2.0 * (289 * 512 + 512 * 223 + 2 * 512 * 512 + 512)),
    }

    def __init__(self):
        self.names = set(self.FLOPS)
        self.records = {n: [] for n in self.names}
        self._open = None

    SHAPE = {"hoisdf_linear_fwd": (7, 8, 9), "hoisdf_linear_bwd_input": (8, 9, 10), "hoisdf_linear_bwd_weight": (9, 10, 11),
             "hoisdf_attention_fwd": (9, 11, 13), "hoisdf_attention_bwd": (15, 17, 19),
             "hoisdf_attention_fwd_f16": (8, 10, 12), "hoisdf_attention_fwd_bf16x2": (8, 10, 12), "hoisdf_sdf_query_fwd": (3, 3, 3),
             "hoisdf_attention_fwd_emu": (9, 11, 13), "hoisdf_attention_bwd_emu": (15, 17, 19),
             "hoisdf_attention_fwd_emu_mag": (9, 11, 13), "hoisdf_attention_bwd_emu_mag": (15, 17, 19),
             "hoisdf_linear_fwd_emu": (6, 7, 8), "hoisdf_linear_bwd_input_emu": (7, 8, 9), "hoisdf_linear_bwd_weight_emu": (9, 10, 11),
             "hoisdf_linear_fwd_emu_mag": (6, 7, 8), "hoisdf_linear_bwd_input_emu_mag": (7, 8, 9), "hoisdf_linear_bwd_weight_emu_mag": (9, 10, 11),
             "hoisdf_linear_fwd_emu_small": (7, 8, 9), "hoisdf_linear_bwd_input_emu_small": (8, 9, 10),
             "hoisdf_linear_bwd_weight_emu_small": (9, 10, 11),
             }

    def begin(self, name, args):
        s = torch.cuda.Event(enable_timing=True)
        s.record()
        self._open = (s, self.FLOPS[name](args), tuple(int(args[i]) for i in self.SHAPE[name]))

    def end(self, name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        s, fl, shape = self._open
        self.records[name].append((s, e, fl, shape))

    def summary(self):
        out = {}
        for n, recs in self.records.items():
            if not recs:
                continue
            ms = sum(r[0].elapsed_time(r[1]) for r in recs)
            fl = sum(r[2] for r in recs)
            out[n] = dict(launches=len(recs), total_ms=ms, avg_us=1e3 * ms / len(recs), gflop=fl / 1e9,
                          tflops=fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0)
        return out

    def largest_launch_us(self, name):
        """average duration of the launches of `name` with the most algorithmic FLOPs (the shape the PMC traffic was
        collected at)"""
        recs = self.records[name]
        top = max(r[2] for r in recs)
        sel = [r for r in recs if r[2] == top]
        return 1e3 * sum(r[0].elapsed_time(r[1]) for r in sel) / len(sel)

    def by_shape(self, steps):
        """per (family, shape) time / TF table (stderr, --shape-report)"""
        agg = {}
        for n, recs in self.records.items():
            for s, e, fl, shape in recs:
                a = agg.setdefault((n, shape), [0, 0.0, 0.0])
                a[0] += 1
                a[1] += s.elapsed_time(e)
                a[2] += fl
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
        return [f"{n:26s} {str(shape):24s} x{c / steps:5.1f}/step {ms / steps:7.3f} ms/step {fl / (ms * 1e-3) / 1e12:6.1f} TF"
                for (n, shape), (c, ms, fl) in rows]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=1, choices=(1, 3, 4),
                    help="BASELINE.json configs index: 1 = training step (the headline metric), 3 = HO3Dv2-shape inference "
                         "(IK variant, 4096 points), 4 = dense eval (8192 points, f16-MFMA attention, batch 8 over 2 GPUs)")
    ap.add_argument("--branch-mix", action="store_true",
                    help="configs[1] after cfg.point_sampling_epoch: per step draw p ~ U(0,1), p < 0.4 -> pre-points "
                         "(branch A), else the dense-lattice sdf_infer (branch B) - main/model.py:426-481")
    ap.add_argument("--aten-report", action="store_true",
                    help="after the warm-up, run one extra step under torch.profiler and print the ATen ops by name and input "
                         "shape (stderr): where the glue launches come from")
    ap.add_argument("--gemm", choices=("emu", "f32"), default="emu",
                    help="linear layers: emu = fp32 emulated on the bf16 MFMA pipe (exact 3-way bf16 operand splits, 6 products, f32 "
                         "accumulation: fp32-equivalent, the default); f32 = the exact-f32 MFMA GEMM")
    ap.add_argument("--attention", choices=("emu", "f32"), default="emu",
                    help="attention: emu = forward and backward emulated like --gemm emu (HOISDF_ATTN_BWD=f32 keeps the exact-f32 fused "
                         "backward); f32 = exact-f32 MFMA kernels")
    ap.add_argument("--f16-attention", type=int, default=-1, choices=(-1, 0, 1),
                    help="configs[4] words its attention as reduced precision: 1 (its default) = the f16-MFMA eval kernel (f16 hi+lo operands, "
                         "3 products), 0 = the emulated fp32 attention of the other configs (6 bf16 products, fp32-equivalent)")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (configs[1]: 32, [3]: 16, [4]: 8 / max(2, gpus))")
    ap.add_argument("--n-hand", type=int, default=None)
    ap.add_argument("--n-obj", type=int, default=None)
    ap.add_argument("--resnet", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bf16x3-leg", type=int, default=1, help="also time 10 steps with HOISDF_EMU_FORM=b3 in a child process (outside the timed region) into `bf16x3`")
    ap.add_argument("--exact-f32", type=int, default=1, help="also time 10 steps of the exact-f32 path (outside the timed region) into `exact_f32`")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--time-every", type=int, default=20, help="record per-kernel HIP events on every n-th timed step (such a step runs single-stream with ~1400 event records: ~10 ms slower than a plain one)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the gradient all-reduce even with one rank (single-GPU check of the N>1 path)")
    ap.add_argument("--shape-report", action="store_true", help="per-shape kernel table on stderr")
    ap.add_argument("--channels-last", type=int, default=1,
                    help="run the CNN encoder in channels_last (the pyramid is then consumed zero-copy)")
    ap.add_argument("--miopen-find", type=int, default=1,
                    help="encoder convs: MIOpen find with the shipped find-db (hoisdf_amd/miopen_db); 0 = MIOpen defaults")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--cpu-threads", type=int, default=32,
                    help="host threads for the CPU baseline (32 was the best of 16/32/64/256 probed on the "
                         "2x EPYC 9575F GPU-box host; 256 threads is >10x slower)")
    args = ap.parse_args()
    dflt = {1: (32, 1536, 512), 3: (16, 3072, 1024), 4: (8 // max(2, args.gpus), 6144, 2048)}[args.config]
    args.batch = args.batch or dflt[0]
    args.n_hand = args.n_hand or dflt[1]
    args.n_obj = args.n_obj or dflt[2]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP hot path has no CPU fallback)"
    # functional test hook (tests/test_gpu_model.py): on a one-GPU box every rank runs on cuda:0 and the collectives go over
    # gloo (RCCL refuses two ranks on one device) - exercises the N > 1 control flow of this file, says nothing about speed
    one_gpu = os.environ.get("HOISDF_BENCH_ONE_GPU_GLOO", "") == "1"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.miopen_find:
        from hoisdf_amd import miopen_tuning
        miopen_tuning.enable()
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from hoisdf_amd import _lib, ops, testing as T
    from hoisdf_amd.config import Config
    from hoisdf_amd.ddp import GradReducer, reducible_parameters
    from hoisdf_amd.model import get_model

    _lib.lib()                     # fail loudly if the extension is missing
    train = args.config == 1
    cfg = Config()
    cfg.resnet_type = args.resnet
    cfg.apply_setting({1: "dexycb", 3: "ho3d_render", 4: "dexycb"}[args.config])
    cfg.num_samp_hand, cfg.num_samp_obj, cfg.bins_n = args.n_hand, args.n_obj, 64
    cfg.attention_f16_eval = (args.config == 4) if args.f16_attention < 0 else bool(args.f16_attention)
    cfg.gemm_emu = args.gemm == "emu"
    cfg.attention_emu = args.attention == "emu"
    ops.set_gemm_emu(cfg.gemm_emu)
    ops.set_attention_emu(cfg.attention_emu)
    torch.manual_seed(0)           # identical initial weights on every rank
    model = get_model("train" if train else "test", cfg=cfg).to(dev).train(train)
    if args.channels_last:
        model.backbone_net.to(memory_format=torch.channels_last)
        model.decoder_net.to(memory_format=torch.channels_last)
    from hoisdf_amd.optim import FusedAdamW
    reducer = opt = None
    if train:
        reducer = GradReducer(reducible_parameters(model), bucket_mb=64.0, always_reduce=args.force_dist, average=False)
        # the reference's optimizer (torch.optim.AdamW(lr=cfg.lr) over ALL parameters, common/base.py:64-73) as one HIP
        # launch; 1/world of the gradient mean is folded into it
        opt = FusedAdamW(list(model.parameters()), lr=cfg.lr, grad_scale=1.0 / world)
    ops.manual_seed(1000 + rank)   # dropout stream: distinct per rank
    model._py_random = __import__("random").Random(4321 + rank)      # the branch A / B draw (main/model.py:426)
    if use_dist and world > 1:
        # data-parallel sanity: every rank starts from the same weights and draws different dropout masks
        chk = torch.stack([p.detach().double().abs().sum() for p in model.parameters()]).sum()
        lo, hi = chk.clone(), chk.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        assert float(lo) == float(hi), "ranks start from different weights"
        seeds = [None] * world
        torch.distributed.all_gather_object(seeds, ops._SEED[0])
        assert len(set(seeds)) == world, f"dropout seeds collide across ranks: {seeds}"
    inputs, targets, meta = (T.to_device(x, dev) for x in T.synthetic_batch(args.batch, args.n_hand, args.n_obj,
                                                                           seed=1234 + rank))

    if args.channels_last:
        inputs["img"] = inputs["img"].contiguous(memory_format=torch.channels_last)
    epoch_cnt = 1e8 if args.branch_mix else 0          # epoch >= cfg.point_sampling_epoch: p < 0.4 -> A, else B
    branches = {"A": 0, "B": 0}

    def train_step():
        reducer.zero_grad()
        out = model(inputs, targets, meta, "train", epoch_cnt, 0.1)
        loss = {k: v.mean() for k, v in out.items() if "_out" not in k}
        total = sum(v * LOSS_WEIGHTS.get(k, 1.0) for k, v in loss.items())     # main/train.py:113-127,138
        total.backward()
        reducer.finish()
        opt.step()
        return total

    @torch.no_grad()
    def eval_step():
        out = model(inputs, targets, meta, "eval")                           # main/test.py:126 (dense-lattice sdf_infer)
        return out["hand_joints_out"].sum()

    step = train_step if train else eval_step

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    timer = None if args.no_kernel_timing else KernelTimer()
    draw = getattr(model, "draw_branch", None)
    for w in range(args.warmup):
        if args.branch_mix and train and draw is not None:
            # the warm-up visits BOTH point branches (alternating, B first): whichever a run's random draws leave out would pay its
            # first-time costs - multi-GB hipMallocs of sdf_infer's workspaces - inside the timed region
            # (the class's draw is still made and dropped, so that the timed steps see the draws they always saw: 8 A / 12 B of 20)
            model.draw_branch = (lambda mode, epoch_cnt=1e8, _a=bool(w & 1): (draw(mode, epoch_cnt), _a and mode == "train")[1])
        # the event-bracketed timed steps run single-stream (see below): warm that allocation pattern up as well
        sampled_like = timer is not None and w == args.warmup - 2
        cfg.overlap_streams = not sampled_like
        # ... with a throw-away timer: a bracketed step issues the encoder layers call by call (ops._coarse_layer_ok), which
        # is another allocation pattern the caching allocator should have seen before the timed region
        _lib.set_timer(KernelTimer() if sampled_like else None)
        step()
        _lib.set_timer(None)
    cfg.overlap_streams = True
    if args.branch_mix and train and draw is not None:
        del model.draw_branch                   # (back to the class's random draw)
    from hoisdf_amd.engine import reserve_hbm_pool
    hbm_pool = reserve_hbm_pool(model)          # free cached blocks on both streams: no hipMalloc inside the timed region
    barrier()
    if args.aten_report and rank == 0:
        from torch.profiler import profile, ProfilerActivity
        from collections import Counter
        # record_shapes cannot marshal 64-bit unsigned seeds of the custom autograd functions: 63-bit ones for this step
        _ns = ops.next_seed
        ops.next_seed = lambda: _ns() & 0x7FFFFFFFFFFFFFFF
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            step()
            torch.cuda.synchronize()
        ops.next_seed = _ns
        cnt, dev_us = Counter(), Counter()
        for e in prof.events():
            if e.name.startswith("aten::") and e.device_time_total > 0 and e.cpu_parent is not None and \
                    not e.cpu_parent.name.startswith("aten::"):
                key = (e.name, e.cpu_parent.name[:40], str(e.input_shapes)[:90])
                cnt[key] += 1
                dev_us[key] += e.device_time_total
        for key, us in dev_us.most_common(60):
            print(f"{us / 1e3:7.3f} ms {cnt[key]:4d}x  {key[0]:24s} <- {key[1]:40s} {key[2]}", file=sys.stderr)
        barrier()
    # per-kernel HIP events live inside the timed region, on every `--time-every`-th step only: ~1400 event records per
    # step cost 1.7 % of the step (they serialise consecutive kernels), which `value` should not pay on every step
    timed_steps = 0
    model.branch_log = []
    t0 = time.perf_counter()
    marks = []                                  # host clock after each step's launches were queued (no synchronisation: diagnostics only)
    for i in range(args.steps):
        sample = timer is not None and i % args.time_every == 0
        _lib.set_timer(timer if sample else None)
        cfg.overlap_streams = not sample        # event-bracketed steps run single-stream: with the object stack on a second
        timed_steps += int(sample)              # stream a kernel's events would also span the other stream's kernels
        last = step()
        marks.append(time.perf_counter())
    barrier()
    dt = time.perf_counter() - t0
    if os.environ.get("HOISDF_BENCH_STEP_LOG") and rank == 0:
        print("host clock per step (ms, launches queued, no sync):", [round(1e3 * (b - a), 1) for a, b in zip([t0] + marks[:-1], marks)],
              "drain %.1f" % (1e3 * (t0 + dt - marks[-1])), file=sys.stderr)
    _lib.set_timer(None)
    for bname in getattr(model, "branch_log", []):
        branches[bname] += 1
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    assert torch.isfinite(last), "non-finite loss"

    # ---- communication diagnostics (every rank takes part, rank 0 reports; outside the timed region) -----------------------
    comm = None
    if use_dist and train:
        import torch.distributed as dist
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        assert int(ones.item()) == world, f"{int(ones.item())} ranks joined the job, expected {world}"
        # (a) exposed communication: how long the compute stream stalls in reducer.finish() (outstanding bucket all-reduces +
        #     the unused-parameter bookkeeping) over three extra steps
        exposed = []
        for _ in range(3):
            reducer.zero_grad()
            out = model(inputs, targets, meta, "train", epoch_cnt, 0.1)
            total = sum(v.mean() * LOSS_WEIGHTS.get(k, 1.0) for k, v in out.items() if "_out" not in k)
            total.backward()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            reducer.finish()
            e1.record()
            opt.step()
            torch.cuda.synchronize()
            exposed.append(e0.elapsed_time(e1))
        # (b) every bucket's all-reduce on its own (nothing else on the device): latency / bus bandwidth of the collective itself
        per_bucket = []
        for flat in reducer.buckets:
            torch.cuda.synchronize(); dist.barrier()
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            nbytes = flat.numel() * 4
            ms = sorted(ts)[1]
            per_bucket.append({"mbytes": round(nbytes / 2 ** 20, 1), "ms": round(ms, 3),
                               "busbw_gbps": round(2.0 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9, 1) if world > 1 else None})
        # (c) do the model's two streams still run side by side next to the process group's streams?  One single-workgroup spin
        #     kernel (torch.cuda._sleep) on the compute stream alone, then one on each stream at once: ~1.0 = overlapped, ~2.0 = one
        #     hardware queue (what a normal-priority second stream got before round 3: profiles/r03_second_stream_hw_queue.txt)
        side = getattr(model, "_side_stream", None)
        overlap_ratio = None
        if side is not None and hasattr(torch.cuda, "_sleep"):
            cur = torch.cuda.current_stream(dev)

            def probe(both):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if both:
                    side.wait_stream(cur)
                    with torch.cuda.stream(side):
                        torch.cuda._sleep(2000000)
                torch.cuda._sleep(2000000)
                if both:
                    cur.wait_stream(side)
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1)
            probe(True)
            overlap_ratio = round(min(probe(True) for _ in range(3)) / min(probe(False) for _ in range(3)), 2)
        t = torch.tensor([sorted(exposed)[1]], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            ver = "unknown"
        comm = {"backend": dist.get_backend(), "rccl_version": ver, "world": world, "ranks_joined": world,
                "gradient_mbytes": round(reducer.total_bytes() / 2 ** 20, 1), "buckets": per_bucket,
                "exposed_ms_per_step_max_over_ranks": round(float(t), 3),
                "all_reduce_ms_sum": round(sum(b["ms"] for b in per_bucket), 3),
                "second_stream_time_ratio_both_vs_one": overlap_ratio,
                "env": {k: v for k, v in os.environ.items()
                        if k.startswith(("NCCL_", "RCCL_", "HSA_ENABLE_IPC", "GPU_MAX_HW_QUEUES", "HOISDF_"))}}
        if rank == 0:
            print(f"[comm] {json.dumps(comm)}", file=sys.stderr)

    # ---- the same step with every contraction on the exact-f32 MFMA kernels (round 2's headline path), OUTSIDE the timed region:
    # the driver-run record then carries both numbers (2 warm-up + 10 steps; every rank takes part, max over ranks)
    exact = None
    if train and args.exact_f32 and (args.gemm == "emu" or args.attention == "emu") and not args.branch_mix:
        cfg.gemm_emu = cfg.attention_emu = False
        ops.set_gemm_emu(False)
        ops.set_attention_emu(False)
        for _ in range(2):
            step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(10):
            step()
        barrier()
        dt1 = time.perf_counter() - t1
        if use_dist:
            t = torch.tensor([dt1], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt1 = float(t)
        exact = {"what": "the same train step with --gemm f32 --attention f32 (exact-f32 MFMA linear layers and attention forward; backward "
                         "as HOISDF_ATTN_BWD says), 10 steps outside the timed region", "dtype": DTYPE_F32,
                 "value": round(world * args.batch * 10 / dt1, 3), "unit": "samples/s", "ms_per_step": round(1e3 * dt1 / 10, 3)}
        cfg.gemm_emu, cfg.attention_emu = args.gemm == "emu", args.attention == "emu"
        ops.set_gemm_emu(cfg.gemm_emu)
        ops.set_attention_emu(cfg.attention_emu)

    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    ms_per_step = 1e3 * dt / args.steps
    value = world * args.batch * args.steps / dt
    enc = f"ResNet-{args.resnet} encoder (PyTorch/MIOpen; the reference has no HRNet)"
    if args.config == 1:
        mix = (f"epoch >= {cfg.point_sampling_epoch} point-branch mix: {branches['A']} steps pre-points (A) / "
               f"{branches['B']} steps dense-lattice sdf_infer (B)") if args.branch_mix else "point branch A (pre-points)"
        workload = (f"BASELINE configs[1]: DexYCB-shape synthetic batch {args.batch}/GPU, {args.n_hand}+{args.n_obj} SDF "
                    f"query points, {enc}, train step = fwd+bwd+grad all-reduce+AdamW, dropout on, {mix}")
    elif args.config == 3:
        workload = (f"BASELINE configs[3]: HO3Dv2-shape synthetic batch {args.batch}/GPU, {args.n_hand}+{args.n_obj} SDF "
                    f"query points selected by sdf_infer on the 64^3 lattice, IK variant (rendered-aug setting "
                    f"ho3d_render), {enc}, inference only (test.py forward)")
    else:
        workload = (f"BASELINE configs[4]: dense eval, batch 8 over 2 GPUs = {args.batch}/GPU, {args.n_hand}+{args.n_obj} "
                    f"query points through sdf_infer, f16-MFMA attention (hi+lo split operands, f32 softmax/accumulate), "
                    f"{enc}, inference only")
    h2_form = _lib.lib().hoisdf_linear_emu_pieces() == 2          # the linear layers' emulation form of this process (HOISDF_EMU_FORM)
    res = {
        "metric": "samples/sec (img + 2048 SDF queries) fwd+bwd at 1/2/4/8 MI355X" if args.config == 1 else
                  f"samples/sec, inference (BASELINE configs[{args.config}])",
        "value": round(value, 3), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ((DTYPE_EMU_H2 if (h2_form and args.gemm == "emu") else DTYPE_EMU) if (args.gemm == "emu" or args.attention == "emu") else DTYPE_F32)
                 if args.config != 4 else
                 "f32 (attention contractions: bf16 hi+lo split operands x3 products, f32 accumulate / softmax; linear layers " +
                 (("f16x2-emulated fp32)" if h2_form else "bf16x3-emulated fp32)") if args.gemm == "emu" else "exact f32)"),
        "data": "synthetic",
        "config": {"workload": workload, "baseline_config": args.config,
                   "global_batch": world * args.batch, "points": args.n_hand + args.n_obj,
                   "parallelism": f"dp{world}", ("final_loss" if train else "checksum"): float(last.detach()),
                   # free cached allocator blocks handed out behind the warm-up (engine.reserve_hbm_pool): no hipMalloc in the timed region
                   "hbm_pool_reserved_mb": round(hbm_pool / 2 ** 20, 1)},
    }
    res["config"]["arithmetic"] = {
        "linear_layers": {"emu": ("fp32 emulated on the f16 MFMA pipe (f16x2 form, HOISDF_EMU_FORM=h2): both f32 operands scaled by a power of two "
                                  "(largest magnitude of each ROW of the activation / gradient operand, of the weight matrix -> [2^13, 2^14)) and split into hi + lo f16 pieces, "
                                  "3 products, f32 accumulate; the row magnitudes travel from the producing kernel's epilogue to the consuming contraction "
                                  "(round 6: a sample's result no longer depends on its batch companions)" if h2_form else
                                  "fp32 emulated on the bf16 MFMA pipe: exact 3-way bf16 split of both f32 operands, 6 products, f32 accumulate"),
                          "f32": "exact-f32 MFMA"}[args.gemm],
        "attention": {"emu": ("forward and backward emulated fp32 in the f16x2 form (hi + lo f16 planes of Q, K, V, dO scaled per (sample, head), P: 3 products; dS three pieces: 5 products; "
                              "HOISDF_ATTN_FORM=b3: the bf16x3 kernels; the 17-query decoder attention exact-f32)" if (h2_form and os.environ.get("HOISDF_ATTN_FORM", "h2")[:1].lower() != "b") else
                              "forward and backward emulated fp32 in the bf16x3 form (exact 3-way bf16 split, 6 products; the 17-query decoder attention exact-f32)")
                             if os.environ.get("HOISDF_ATTN_BWD", "emu") != "f32" else
                             "forward emulated fp32 (bf16x3), backward exact-f32 MFMA fused kernel (HOISDF_ATTN_BWD=f32)",
                      "f32": "exact-f32 MFMA", "f16": "f16-MFMA eval kernel (BASELINE configs[4]): f16 hi+lo operands, 3 products"}[
                          "f16" if cfg.attention_f16_eval else ("f32" if args.attention == "f32" else "emu")],
        "accuracy_evidence": "tests/test_gpu_emu.py, tools/emu_accuracy.py: error vs fp64 <= the exact-f32 kernels' and hipBLASLt fp32's"}
    if timer is not None:
        ks = timer.summary()
        if args.shape_report:
            print("\n".join(timer.by_shape(timed_steps)), file=sys.stderr)
        # kernel families = device kernels (a family = the C entries that launch the same kernel template), each priced against
        # the MFMA roof of the arithmetic it runs:
        #   f32    exact-f32 MFMA kernels: 157.3 TFLOP/s
        #   emu    fp32 emulated with 3-way bf16 splits: six bf16 products per product -> 2500 / 6 TFLOP/s of fp32-equivalent work
        #   split  f16 hi + lo pairs (the configs[4] eval attention): three f16 products per product -> 2500 / 3
        sq = ["hoisdf_sdf_query_fwd"]           # its six GEMMs follow the library's emulation switch
        sq_emu = bool(_lib.lib().hoisdf_get_gemm_emu())
        FAMS = [
            ("attn_delta + attn_bwd_fused (dK, dV, dQ in one pass)", ["hoisdf_attention_bwd"], "f32"),
            ("attn_fwd_kernel", ["hoisdf_attention_fwd"], "f32"),
            ("gemm_f32_kernel (linear fwd + grad-input + grad-weight)",
             ["hoisdf_linear_fwd", "hoisdf_linear_bwd_input", "hoisdf_linear_bwd_weight"] + ([] if sq_emu else sq), "f32"),
            ("emu_h2_kernel (linear fwd + grad-input, f16x2 form: 256 x 128 tile, 256 x 256 for masked grad-input over >= 768)" if h2_form else
             "emu_kc2_kernel (linear fwd + grad-input; emu_kc_kernel with HOISDF_EMU_KC=1)",
             ["hoisdf_linear_fwd_emu", "hoisdf_linear_bwd_input_emu", "hoisdf_linear_fwd_emu_mag", "hoisdf_linear_bwd_input_emu_mag"] + (sq if sq_emu else []),
             "h2" if h2_form else "emu"),
            ("emu_dw2_kernel (linear grad-weight, bf16x3, + ordered reduce; emu_dw_kernel<128> for K <= 128)", ["hoisdf_linear_bwd_weight_emu"], "emu"),
            ("emu_dw2h_kernel (linear grad-weight, f16x2 form, + ordered reduce)", ["hoisdf_linear_bwd_weight_emu_mag"], "h2" if h2_form else "emu"),
            ("emu_small_kernel / emu_small_dw_kernel (linear layers of < 2048 rows: decoder stack, heads; latency-bound)",
             ["hoisdf_linear_fwd_emu_small", "hoisdf_linear_bwd_input_emu_small", "hoisdf_linear_bwd_weight_emu_small"], "emu"),
            ("emu_attn_fwd2_kernel (+ bf16x3 conversion passes)", ["hoisdf_attention_fwd_emu"], "emu"),
            ("emu_attn_fwd2_kernel<DROP, 2, true> (f16x2 form: scaled hi + lo f16 planes, three products; + magnitude / conversion passes)",
             ["hoisdf_attention_fwd_emu_mag"], "h2"),
            ("emu_attn_bwd4h_kernel (f16x2 form, fused dK, dV, dQ: 76 MFMAs per tile = 3.8 products per product; + magnitude / dO conversion / delta / dQ reduce passes)",
             ["hoisdf_attention_bwd_emu_mag"], "h2b"),
            ("emu_attn_bwd4_kernel (fused dK, dV, dQ; + dO conversion / delta / dQ reduce passes; emu_attn_bwd_stag_kernel with HOISDF_EMU_ATTN_BWD=3)",
             ["hoisdf_attention_bwd_emu"], "emu"),
            ("emu_attn_fwd2_kernel<NPL = 2> (bf16 hi + lo operands, three products; + conversion passes)", ["hoisdf_attention_fwd_bf16x2"], "split"),
            ("attn_fwd_f16_kernel (round 2, HOISDF_ATTN16=f16; + operand split pass)", ["hoisdf_attention_fwd_f16"], "split"),
        ]
        PEAK = {"f32": (PEAK_F32_TFLOPS, "f32 MFMA peak (= f32 vector peak), MI355X_MICROARCH.md:41"),
                "h2b": (round(PEAK_F16_TFLOPS / 3.8, 1), "dense f16 MFMA peak 2500 TFLOP/s / 3.8 products per product (S, dP, dV: 3; dQ, dK: 5 - dS carries three f16 pieces)"),
                "h2": (round(PEAK_F16_TFLOPS / 3.0, 1), "dense f16 MFMA peak 2500 TFLOP/s / 3 products per fp32-equivalent product (scaled hi + lo f16 pieces; MI355X_MICROARCH.md:42)"),
                "emu": (round(PEAK_F16_TFLOPS / 6.0, 1), "dense bf16 MFMA peak 2500 TFLOP/s / 6 products per fp32-equivalent product (MI355X_MICROARCH.md:42)"),
                "split": (round(PEAK_F16_TFLOPS / 3.0, 1), "dense 16-bit MFMA peak 2500 TFLOP/s / 3 products per product (hi + lo operand pairs)")}
        # what a bare MFMA stream sustains on such operands under this board's power cap, per fp32-equivalent product
        CEIL = {"emu": MEASURED_MFMA_CEILING_TFLOPS / 6.0, "split": MEASURED_MFMA_CEILING_TFLOPS / 3.0, "h2": MEASURED_MFMA_CEILING_F16_TFLOPS / 3.0,
                "h2b": MEASURED_MFMA_CEILING_F16_TFLOPS / 3.8}
        agg = {}
        for fam, members, cls in FAMS:
            ms = sum(ks[m]["total_ms"] for m in members if m in ks)
            if ms > 0:
                gf = sum(ks[m]["gflop"] for m in members if m in ks)
                n = sum(ks[m]["launches"] for m in members if m in ks)
                agg[fam] = dict(total_ms=ms, gflop=gf, launches=n, tflops=gf / ms, members=[m for m in members if m in ks], cls=cls)
        dom = max(agg, key=lambda f: agg[f]["total_ms"])
        d = agg[dom]
        peak, peak_note = PEAK[d["cls"]]
        res["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": round(d["tflops"], 2),
                           "peak": peak, "unit": "TFLOP/s",
                           "frac": round(d["tflops"] / peak, 4), "traffic": None,
                           "peak_note": peak_note,
                           # the same achieved rate against the roof the exact-f32 kernels (dtype "f32" on the f32 MFMA) are bound by
                           "frac_of_f32_mfma_peak": round(d["tflops"] / PEAK_F32_TFLOPS, 4),
                           "avg_launch_us": round(1e3 * d["total_ms"] / d["launches"], 2),
                           "launches_per_step": d["launches"] / timed_steps,
                           "ms_per_step": round(d["total_ms"] / timed_steps, 3),
                           "algorithmic_gflop_per_launch": round(d["gflop"] / d["launches"], 3),
                           "events_on_steps": f"{timed_steps} of {args.steps} (those steps single-stream)"}
        # PMC evidence (rocprofv3 --pmc, separate passes per counter group, tools/pmc_collect.sh -> profiles/<PMC_FILE>): a list of
        # records keyed by (C entry, shape), never merged across shapes.  The record of the dominant family's entry at the shape
        # with the most time in this run is quoted: HBM bytes per launch (FETCH_SIZE x 2 [gfx950] + WRITE_SIZE), MFMA-busy fraction
        # and the effective clock under that kernel; the achieved HBM rate uses THIS run's HIP-event duration at that shape.
        try:
            pmc = json.load(open(os.path.join(REPO, "profiles", PMC_FILE)))
            by_shape = {}
            for m in d["members"]:
                for s0, e0, fl0, shape in timer.records.get(m, []):
                    a0 = by_shape.setdefault((m, shape), [0, 0.0])
                    a0[0] += 1
                    a0[1] += s0.elapsed_time(e0)
            # the family's DOMINANT entry first (the C entry with the most time: for the f16x2 linear family the forward,
            # emu_h2_kernel<false, false, 2>), within it the shape with the most time
            per_entry = {}
            for (m, shape), (cnt, ms) in by_shape.items():
                per_entry[m] = per_entry.get(m, 0.0) + ms
            for (m, shape), (cnt, ms) in sorted(by_shape.items(), key=lambda kv: (-per_entry[kv[0][0]], -kv[1][1])):
                rec = [r for r in pmc if r["entry"] == m and tuple(r["shape"]) == tuple(shape) and "hbm_bytes_per_launch" in r]
                if rec:
                    r0 = rec[0]
                    us = 1e3 * ms / cnt
                    res["roofline"]["traffic"] = int(r0["hbm_bytes_per_launch"])
                    res["roofline"]["traffic_record"] = {"file": f"profiles/{PMC_FILE}", "case": r0["case"], "entry": m, "shape": list(shape),
                                                         "launch_us_this_run": round(us, 1)}
                    res["roofline"]["hbm_gbps"] = round(r0["hbm_bytes_per_launch"] / (us * 1e-6) / 1e9, 1)
                    res["roofline"]["hbm_peak_gbps"] = 8000.0
                    # algorithmic bytes of that launch: the row operand read once, the output written once (f32),
                    # the weight image read once (4 bytes per weight: two f16 planes) - what `traffic` is to be compared with
                    if "linear" in m and len(shape) == 3:
                        Mr, a1, a2 = shape
                        res["roofline"]["algorithmic_bytes"] = int(4 * (Mr * a1 + Mr * a2 + a1 * a2))
                        res["roofline"]["traffic_over_algorithmic"] = round(r0["hbm_bytes_per_launch"] / res["roofline"]["algorithmic_bytes"], 3)
                    for k_ in ("mfma_busy", "mfma_busy_ghz", "mfma_busy_of_peak_clock", "clock_note"):
                        if k_ in r0:
                            res["roofline"][k_] = r0[k_]
                    clk = r0.get("effective_clock_ghz")
                    res["roofline"]["effective_clock_ghz"] = clk if (clk is not None and clk <= 2.45) else None     # (never above the part's 2.4 GHz)
                    break
        except Exception as ex:                 # the PMC file is evidence, not a dependency of the measurement
            res["roofline"]["traffic_note"] = f"no PMC record: {ex}"
        res["families"] = [{"kernel": fam, "arithmetic": v["cls"], "ms_per_step": round(v["total_ms"] / timed_steps, 3),
                            "launches_per_step": v["launches"] / timed_steps, "achieved_tflops": round(v["tflops"], 2),
                            "peak_tflops": PEAK[v["cls"]][0], "frac": round(v["tflops"] / PEAK[v["cls"]][0], 4),
                            # against what a bare MFMA stream sustains on such operands under this board's power cap (None: f32 MFMA)
                            "frac_of_measured_mfma_ceiling": (round(v["tflops"] / CEIL[v["cls"]], 4) if v["cls"] in CEIL else None)}
                           for fam, v in sorted(agg.items(), key=lambda kv: -kv[1]["total_ms"])]
        tot_ms = sum(v["total_ms"] for v in agg.values())
        tot_gf = sum(v["gflop"] for v in agg.values())
        # the whole HIP hot path's MFMA-shaped work: algorithmic TFLOP per step / the HIP-event time of those launches
        res["hot_path"] = {"algorithmic_tflop_per_step": round(tot_gf / timed_steps / 1e3, 3),
                           "mfma_kernel_ms_per_step": round(tot_ms / timed_steps, 3),
                           "tflops": round(tot_gf / tot_ms, 2) if tot_ms > 0 else 0.0,
                           "tflops_over_the_whole_step": round(tot_gf / timed_steps / 1e3 / (ms_per_step * 1e-3), 2)}
        # the roof of this board for these operands (see MEASURED_MFMA_CEILING_TFLOPS): products per fp32-equivalent product as in PEAK
        if d["cls"] in CEIL:
            res["roofline"]["measured_mfma_ceiling"] = {"tflops": round(CEIL[d["cls"]], 1),
                                                         "what": ("bare v_mfma_f32_32x32x16_f16 stream on the hi / lo f16 pieces of N(0,1) operands" if d["cls"] in ("h2", "h2b") else
                                                                  "bare v_mfma_f32_32x32x16_bf16 stream on bf16x3 pieces / random operands") +
                                                                 " under the 1400 W cap, / products per product",
                                                         "file": "profiles/r05_mfma_rate_vs_operand_data_f16.txt" if d["cls"] in ("h2", "h2b") else "profiles/r05_mfma_rate_vs_operand_data.txt"}
            res["roofline"]["frac_of_measured_mfma_ceiling"] = round(d["tflops"] / CEIL[d["cls"]], 4)
        # Amdahl: where the kernel time of a step goes (HIP hot path / the PyTorch-ROCm image encoder's libraries / ATen glue), from
        # the committed rocprofv3 trace of this same command (single stream, 5 timed steps) - evidence quoted, not measured in this run
        try:
            import csv
            split = {"hip_hot_path": 0.0, "hip_encoder_side": 0.0, "encoder_libraries": 0.0, "aten_glue": 0.0}
            with open(os.path.join(REPO, "profiles", STEP_TRACE), newline="") as f:
                for r in csv.DictReader(f):
                    n, ms = r["Name"], float(r["TotalNsPerStep"]) * 1e-6
                    # (hip_encoder_side: this library's (f4) kernels inside the image encoder - BatchNorm + residual + ReLU, csrc/bnact.hip)
                    k = (("hip_encoder_side" if "::bn_" in n else "hip_hot_path") if "hoisdf" in n else
                         ("aten_glue" if ("at::native" in n or "rocclr" in n) else "encoder_libraries"))
                    split[k] += ms
            res["hot_path"]["kernel_ms_per_step_by_owner"] = {k: round(v, 2) for k, v in split.items()}
            res["hot_path"]["kernel_ms_per_step_by_owner"]["file"] = f"profiles/{STEP_TRACE}"
            enc_ms = split["encoder_libraries"] + split["aten_glue"] + split["hip_encoder_side"]
            res["hot_path"]["amdahl_note"] = ("the image encoder (MIOpen / CK convolutions, ATen, this library's fused BatchNorm passes; out of the hot path by "
                                              "north_star) is %.0f %% of the step's kernel time: hot-path work can still buy at most %.2fx" %
                                              (100 * enc_ms / max(sum(split.values()), 1e-9), sum(split.values()) / max(enc_ms, 1e-9)))
        except Exception as ex:
            res["hot_path"]["kernel_ms_per_step_by_owner"] = f"no trace file: {ex}"
        res["kernels"] = {n: {"ms_per_step": round(v["total_ms"] / timed_steps, 3), "tflops": round(v["tflops"], 2),
                              "launches_per_step": v["launches"] / timed_steps, "avg_us": round(v["avg_us"], 2)}
                          for n, v in ks.items()}
    if exact is not None:
        res["exact_f32"] = exact
    # ---- the same step in the bf16x3 form (exact three-way operand split, six products: the round-4 arithmetic),
    # OUTSIDE the timed region.  The form is fixed per process (it is the weight-image format): a child process on this GPU, rank 0 at
    # N = 1 only.
    if train and world == 1 and args.bf16x3_leg and h2_form and args.gemm == "emu" and not args.branch_mix and args.config == 1:
        import subprocess
        env = dict(os.environ, HOISDF_EMU_FORM="b3")
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--exact-f32", "0",
               "--no-kernel-timing", "--bf16x3-leg", "0", "--batch", str(args.batch), "--n-hand", str(args.n_hand), "--n-obj", str(args.n_obj),
               "--resnet", str(args.resnet)]
        try:
            out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
            res["bf16x3"] = {"what": "the same train step with HOISDF_EMU_FORM=b3 (linear layers and attention in the bf16x3 form: exact 3-way split, "
                                     "6 products - the round-4 arithmetic), 10 steps in a child process after the timed region", "dtype": line["dtype"],
                                           "value": line["value"], "unit": "samples/s", "ms_per_step": line["ms_per_step"]}
        except Exception as ex:
            res["bf16x3"] = f"child run failed: {ex}"
    if comm is not None:
        res["comm"] = comm
    if world == 1 and not args.no_cpu_baseline:
        from oracle.cpu_step import time_cpu_baseline
        threads = min(args.cpu_threads, os.cpu_count() or 1)
        cb = time_cpu_baseline(args.n_hand, args.n_obj, args.cpu_batch if train else min(args.cpu_batch, 2), iters=3,
                               warmup=1, resnet_type=args.resnet, train=train, threads=threads,
                               setting={1: "dexycb", 3: "ho3d_render", 4: "dexycb"}[args.config])
        cb["value"] = round(cb["value"], 4)
        cb.pop("seconds_per_step", None)
        cb["host_cores"] = os.cpu_count()
        res["cpu_baseline"] = cb
        res["speedup_vs_cpu"] = round(value / cb["value"], 1)
    print(json.dumps(res))
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
