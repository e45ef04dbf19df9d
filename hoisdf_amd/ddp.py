"""Data-parallel gradient reduction for one-process-per-GPU training (RCCL over xGMI).

The reference trains under single-process ``torch.nn.DataParallel`` (common/base.py:103): each
step it re-broadcasts 208 MB of parameters, scatters inputs and reduce-adds gradients onto GPU 0.
Here parameters stay resident per rank and the only collective of a step is a bucketed
all-reduce of the gradients, launched bucket-by-bucket from autograd hooks so it overlaps the
rest of backward.  Semantics preserved: loss = mean over the global batch of per-replica means
(main/train.py:113) <=> gradient = mean over ranks of per-rank gradients.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce is per-link bound, so
buckets are large (default 64 MB) to amortise RCCL's launch/latency floor - the whole 52 M-parameter
model is 4 buckets.  Parameters that never receive a gradient on this path (``norm1``,
``linear_objvote``, ``linear_objcls``; frozen BN affine) are left out of the reducer.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

UNUSED_PREFIXES = ("norm1.", "linear_objvote.", "linear_objcls.")


def reducible_parameters(model: torch.nn.Module) -> List[Tuple[str, torch.nn.Parameter]]:
    return [(n, p) for n, p in model.named_parameters()
            if p.requires_grad and not n.startswith(UNUSED_PREFIXES)]


class GradReducer:
    """Flat gradient buckets + asynchronous all-reduce.

    ``p.grad`` of every managed parameter is a view into its bucket's flat buffer, so the
    collective runs in place on what the optimizer reads (no pack/unpack copies).  Buckets are
    filled in reverse registration order (= roughly the order backward produces gradients)."""

    def __init__(self, params: Sequence[Tuple[str, torch.nn.Parameter]], bucket_mb: float = 64.0,
                 group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.buckets: List[torch.Tensor] = []
        self._members: List[List[torch.nn.Parameter]] = []
        self._pending: List[int] = []
        self._handles = []
        self._launched: List[bool] = []
        cap = int(bucket_mb * 1024 * 1024 / 4)
        cur: List[torch.nn.Parameter] = []
        size = 0
        groups: List[List[torch.nn.Parameter]] = []
        for _, p in reversed(list(params)):
            if cur and size + p.numel() > cap:
                groups.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += p.numel()
        if cur:
            groups.append(cur)
        for bi, members in enumerate(groups):
            n = sum(p.numel() for p in members)
            flat = torch.zeros(n, dtype=members[0].dtype, device=members[0].device)
            off = 0
            for p in members:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                p.register_post_accumulate_grad_hook(self._make_hook(bi))
            self.buckets.append(flat)
            self._members.append(members)
            self._pending.append(len(members))
            self._launched.append(False)

    def _make_hook(self, bi: int):
        def hook(_p):
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                self._launch(bi)
        return hook

    def _launch(self, bi: int):
        if self._launched[bi]:
            return
        self._launched[bi] = True
        if self.world > 1:
            self._handles.append(dist.all_reduce(self.buckets[bi], op=dist.ReduceOp.SUM, group=self.group,
                                                 async_op=True))

    def zero_grad(self):
        """call instead of optimizer.zero_grad(): keeps the bucket views alive"""
        for bi, flat in enumerate(self.buckets):
            flat.zero_()
            self._pending[bi] = len(self._members[bi])
            self._launched[bi] = False
        self._handles = []

    def finish(self):
        """after backward(): reduce buckets whose hooks did not all fire (a parameter unused this
        step keeps a zero gradient), wait, and average."""
        for bi in range(len(self.buckets)):
            self._launch(bi)
        for h in self._handles:
            h.wait()
        self._handles = []
        if self.world > 1:
            for flat in self.buckets:
                flat.div_(self.world)

    def total_bytes(self) -> int:
        return sum(b.numel() * 4 for b in self.buckets)
