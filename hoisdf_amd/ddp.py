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

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

UNUSED_PREFIXES = ("norm1.", "linear_objvote.", "linear_objcls.")


def reducible_parameters(model: torch.nn.Module) -> List[Tuple[str, torch.nn.Parameter]]:
    return [(n, p) for n, p in model.named_parameters()
            if p.requires_grad and not n.startswith(UNUSED_PREFIXES)]


class GradReducer:
    """Flat gradient buckets + asynchronous all-reduce.

    Autograd produces each parameter's gradient normally (``p.grad`` starts as None every step, so
    AccumulateGrad just adopts the tensor - no per-parameter add kernel).  When the last gradient
    of a bucket has arrived (post-accumulate-grad hooks, buckets ordered like backward), the
    bucket is packed with a multi-tensor copy (``torch._foreach_copy_``), its all-reduce is
    launched asynchronously, and ``p.grad`` is re-pointed at views of the flat buffer, which is
    what the optimizer then reads.  ~450 tiny kernels per step become ~2 per bucket."""

    def __init__(self, params: Sequence[Tuple[str, torch.nn.Parameter]], bucket_mb: float = 64.0,
                 group: Optional[dist.ProcessGroup] = None, always_reduce: bool = False, average: bool = True):
        self.group = group
        # average=False leaves the SUM over ranks in p.grad: the optimizer folds 1/world into its own pass
        # (hoisdf_amd.optim.FusedAdamW(grad_scale=1/world)) and one 208 MB read-modify-write per step disappears
        self.average = average
        # always_reduce: issue the collective even with one rank (exercises RCCL on a single-GPU box)
        self.always_reduce = bool(always_reduce) and dist.is_initialized()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.buckets: List[torch.Tensor] = []
        self._members: List[List[torch.nn.Parameter]] = []
        self._views: List[List[torch.Tensor]] = []
        self._pending: List[int] = []
        self._handles = []
        self._launched: List[bool] = []
        self._ready: List[bool] = []
        self._next = 0
        self._side_events: List[list] = []
        self._main = None
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
        self._flags: List[torch.Tensor] = []          # per bucket: one "this rank produced a gradient" float per member
        self._ones: List[List[torch.Tensor]] = []
        self._check_unused: List[bool] = []
        for bi, members in enumerate(groups):
            n = sum(p.numel() for p in members)
            # the bucket carries one extra float per member behind the gradients: 1.0 if this rank produced a gradient
            # for it this step.  After the SUM all-reduce a zero there means NO rank used the parameter, and its
            # ``grad`` goes back to None (the reference's behaviour: AdamW then applies neither weight decay nor
            # moment decay to it); the flags ride in the same collective, so the common path costs nothing extra.
            flat = torch.zeros(n + len(members), dtype=members[0].dtype, device=members[0].device)
            self._flags.append(flat[n:])
            self._ones.append([torch.ones(1, dtype=flat.dtype, device=flat.device) for _ in members])
            self._check_unused.append(False)
            views, off = [], 0
            for p in members:
                chunk = flat[off:off + p.numel()]
                if p.dim() == 4 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last):
                    o, i, kh, kw = p.shape           # same physical layout as the channels_last parameter
                    views.append(chunk.view(o, kh, kw, i).permute(0, 3, 1, 2))
                else:
                    views.append(chunk.view_as(p))
                off += p.numel()
                p.register_post_accumulate_grad_hook(self._make_hook(bi))
            self.buckets.append(flat)
            self._members.append(members)
            self._views.append(views)
            self._pending.append(len(members))
            self._launched.append(False)
            self._ready.append(False)
            self._side_events.append([])

    def _make_hook(self, bi: int):
        def hook(p):
            # A gradient may be produced on a side stream (the model overlaps the object stack with the hand stack on
            # a second HIP stream; autograd replays each op's backward on its forward stream).  Buckets are packed on
            # the stream that was current at zero_grad(); gradients that arrive on another stream leave an event for it.
            if p.is_cuda and self._main is not None:
                cur = torch.cuda.current_stream(p.device)
                if cur != self._main:
                    ev = torch.cuda.Event()
                    ev.record(cur)
                    self._side_events[bi].append(ev)
                    if p.grad is not None:
                        p.grad.record_stream(self._main)      # freed after the pack on _main, not before

            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                self._launch(bi)
        return hook

    def _launch(self, bi: int):
        """Collectives must be issued in the SAME order on every rank: a bucket whose last gradient arrived is only
        marked ready, and buckets go out strictly in index order (bucket i waits for buckets < i).  A rank on which some
        parameter got no gradient this step (its bucket is then flushed by finish()) therefore still matches the others."""
        self._ready[bi] = True
        while self._next < len(self.buckets) and self._ready[self._next]:
            self._launch_now(self._next)
            self._next += 1

    def _launch_now(self, bi: int):
        if self._launched[bi]:
            return
        self._launched[bi] = True
        if self._main is not None:
            for ev in self._side_events[bi]:
                self._main.wait_event(ev)
            self._side_events[bi] = []
            with torch.cuda.stream(self._main):
                self._pack_and_reduce(bi)
        else:
            self._pack_and_reduce(bi)

    def _pack_and_reduce(self, bi: int):
        members, views, flat = self._members[bi], self._views[bi], self.buckets[bi]
        flags = self._flags[bi]
        if all(p.grad is not None for p in members):
            torch._foreach_copy_(views + [flags[i:i + 1] for i in range(len(members))],
                                 [p.grad for p in members] + self._ones[bi])       # multi-tensor packed copy (+ flags)
        else:                                   # some parameter got no gradient this step: zero + per-tensor copy
            flat.zero_()
            for i, (p, v) in enumerate(zip(members, views)):
                if p.grad is not None:
                    v.copy_(p.grad)
                    flags[i:i + 1].copy_(self._ones[bi][i])
            self._check_unused[bi] = True
        for p, v in zip(members, views):
            p.grad = v
        if self.world > 1 or self.always_reduce:
            self._handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def zero_grad(self):
        """call instead of optimizer.zero_grad()"""
        dev = self.buckets[0].device if self.buckets else None
        self._main = torch.cuda.current_stream(dev) if dev is not None and dev.type == "cuda" else None
        if self._main is not None:
            # every gradient this reducer handles is copied into its buckets before the optimizer reads it, so the
            # backward's zero-initialised scratch can come from the per-step arena (one memset per step)
            from . import ops
            ops.zero_arena_begin_step(dev)
        for bi, members in enumerate(self._members):
            for p in members:
                p.grad = None
            self._pending[bi] = len(members)
            self._launched[bi] = False
            self._ready[bi] = False
            self._side_events[bi] = []
            self._check_unused[bi] = False
        self._next = 0
        self._handles = []

    def finish(self):
        """after backward(): pack + reduce buckets whose hooks did not all fire (a parameter unused this
        step keeps a zero gradient), wait, and average."""
        for bi in range(len(self.buckets)):
            self._launch(bi)
        for h in self._handles:
            h.wait()
        self._handles = []
        if self._main is not None:
            from . import ops
            ops.zero_arena_end_step()        # every arena-backed gradient now lives in a bucket
        if self.world > 1 and self.average:
            for flat, flags in zip(self.buckets, self._flags):
                flat[:flat.numel() - flags.numel()].div_(self.world)
        # parameters that NO rank used this step get grad = None back.  Reading the flags synchronises the host, so it is
        # done only for buckets that had a locally-unused member: a flag can only sum to zero if every rank, this one
        # included, produced no gradient for that member, so every rank takes this branch together.
        for bi, members in enumerate(self._members):
            if not self._check_unused[bi]:
                continue
            fl = self._flags[bi].cpu()
            unused = [i for i in range(len(members)) if float(fl[i]) == 0.0]
            for i in unused:
                members[i].grad = None

    def total_bytes(self) -> int:
        return sum((b.numel() - f.numel()) * 4 for b, f in zip(self.buckets, self._flags))
