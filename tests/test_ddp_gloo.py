"""CPU, world_size 2 over gloo: the bucketed gradient reducer (hoisdf_amd/ddp.py) gives every rank
the mean of the per-rank gradients, launches its collectives from autograd hooks, tolerates
parameters that receive no gradient in a step, and leaves out the reference's never-used groups."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = torch.nn.Linear(16, 32)
        self.b = torch.nn.Linear(32, 8)
        self.sometimes = torch.nn.Linear(8, 8)          # used on even steps only
        self.norm1 = torch.nn.LayerNorm(8)              # reference: defined, never used -> not reduced
        self.linear_objvote = torch.nn.Linear(8, 8)

    def forward(self, x, use_extra):
        y = self.b(torch.relu(self.a(x)))
        return self.sometimes(y) if use_extra else y


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hoisdf_amd.ddp import GradReducer, reducible_parameters
    model = Toy()
    names = [n for n, _ in reducible_parameters(model)]
    assert not any(n.startswith(("norm1.", "linear_objvote.")) for n in names)
    red = GradReducer(reducible_parameters(model), bucket_mb=0.002)     # several small buckets
    assert len(red.buckets) >= 2
    ok = True
    for step in range(3):
        torch.manual_seed(100 * step + rank)
        x = torch.randn(4, 16)
        red.zero_grad()
        loss = model(x, step % 2 == 0).pow(2).mean()
        loss.backward()
        red.finish()
        # reference: single process, both shards, mean of the two per-rank losses
        ref = Toy()
        ref.load_state_dict(model.state_dict())
        tot = 0
        for r in range(world):
            torch.manual_seed(100 * step + r)
            tot = tot + ref(torch.randn(4, 16), step % 2 == 0).pow(2).mean() / world
        tot.backward()
        for (n, p), (_, pr) in zip(model.named_parameters(), ref.named_parameters()):
            if n.startswith(("norm1.", "linear_objvote.")):
                continue
            if pr.grad is None:               # unused on every rank this step -> grad None, as in the reference (AdamW skips it)
                ok &= p.grad is None
            else:
                ok &= p.grad is not None and torch.allclose(p.grad, pr.grad, atol=1e-6)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_grad_reducer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(ok for _, ok in res), res


def test_grad_reducer_single_process_is_identity():
    from hoisdf_amd.ddp import GradReducer, reducible_parameters
    model = Toy()
    red = GradReducer(reducible_parameters(model), bucket_mb=64)
    red.zero_grad()
    model(torch.ones(2, 16), True).sum().backward()
    g = model.a.weight.grad.clone()
    red.finish()
    assert torch.equal(g, model.a.weight.grad)
    lo = red.buckets[0].data_ptr()
    assert lo <= model.a.weight.grad.data_ptr() < lo + red.buckets[0].numel() * 4      # grads live inside the bucket


def _worker_real(rank, world, port, q):
    try:
        _worker_real_body(rank, world, port, q)
    except Exception as e:                              # surface the failure instead of a gloo "connection reset"
        import traceback
        q.put((rank, ["exception: " + traceback.format_exc()[-1500:]]))


def _worker_real_body(rank, world, port, q):
    """the REAL model's reducer inputs: every reducible parameter of get_model (hot path + ResNet-50 encoder in
    channels_last), synthetic per-rank gradients, a step where one rank / both ranks miss a parameter."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hoisdf_amd.config import Config
    from hoisdf_amd.ddp import UNUSED_PREFIXES, GradReducer, reducible_parameters
    from hoisdf_amd.model import get_model
    c = Config()
    c.resnet_type = 50
    c.apply_setting("dexycb")
    torch.manual_seed(0)
    model = get_model("train", cfg=c)
    model.backbone_net.to(memory_format=torch.channels_last)
    model.decoder_net.to(memory_format=torch.channels_last)
    named = reducible_parameters(model)
    names = [n for n, _ in named]
    msgs = []
    # excluded groups: the reference's never-used modules (main/model.py:55,86-87) and the frozen BN affine (:100-105)
    if any(n.startswith(UNUSED_PREFIXES) for n in names):
        msgs.append("unused group in the reducer")
    if any(("bn" in n and n.startswith("backbone_net")) for n in names):
        msgs.append("frozen BN parameter in the reducer")
    total = sum(p.numel() for _, p in named)
    red = GradReducer(named, bucket_mb=64.0)
    cap = 64 * 1024 * 1024 // 4
    sizes = [sum(p.numel() for p in m) for m in red._members]
    if not (len(red.buckets) == -(-total // cap) or len(red.buckets) == -(-total // cap) + 1):
        msgs.append(f"bucket count {len(red.buckets)} for {total} elements")
    if any(sz > cap for sz in sizes) or sum(sizes) != total or red.total_bytes() != 4 * total:
        msgs.append(f"bucket sizes {sizes}")
    # bucket order follows backward: the LAST parameters of the module list come first
    if red._members[0][0] is not named[-1][1]:
        msgs.append("buckets are not in reverse parameter order")
    n_cl = sum(1 for _, p in named if p.dim() == 4 and not p.is_contiguous())
    if n_cl < 10:
        msgs.append(f"only {n_cl} channels_last conv weights seen")
    some = named[len(named) // 2][1]                    # missing on rank 1 only in step 1, on both ranks in step 2

    def has(step, r, p):
        return not (p is some and ((step == 1 and r == 1) or step == 2))

    def grad_of(step, r, idx, p):                       # cheap to regenerate for any (step, rank, parameter)
        t = torch.randn(p.shape, generator=torch.Generator().manual_seed(7919 * step + 104729 * r + idx))
        return t.contiguous(memory_format=torch.channels_last) if p.dim() == 4 else t

    for step in range(3):
        red.zero_grad()
        for idx in reversed(range(len(named))):         # gradients arrive in reverse parameter order, via the hooks
            p = named[idx][1]
            if has(step, rank, p):
                p.grad = grad_of(step, rank, idx, p)
                for h in p._post_accumulate_grad_hooks.values():
                    h(p)
        red.finish()
        for idx in (0, 1, len(named) // 3, len(named) // 2, len(named) - 2, len(named) - 1):
            n, p = named[idx]
            parts = [grad_of(step, r, idx, p) for r in range(world) if has(step, r, p)]
            if not parts:
                if p.grad is not None:
                    msgs.append(f"step {step}: {n} unused on every rank but grad is not None")
            elif p.grad is None or not torch.allclose(p.grad, sum(parts) / world, atol=1e-6):
                msgs.append(f"step {step}: {n} reduced gradient differs")
            elif p.grad.stride() != p.stride():
                msgs.append(f"step {step}: {n} gradient strides {p.grad.stride()} != parameter strides {p.stride()}")
    q.put((rank, msgs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_grad_reducer_world2_gloo_real_model_parameters():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_real, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(not msgs for _, msgs in res), res
