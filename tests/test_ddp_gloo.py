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
            want = pr.grad if pr.grad is not None else torch.zeros_like(pr)
            ok &= torch.allclose(p.grad, want, atol=1e-6)
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
