"""AdamW for the training step as one HIP launch (``hoisdf_adamw_step``, csrc/optim.hip).

Drop-in for ``torch.optim.AdamW`` (the reference's optimizer, common/base.py:64-73): same hyper-parameters, same
update rule, same ``state_dict()`` layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter) so checkpoints in the
reference format load either way.  ``grad_scale`` lets the data-parallel reducer skip its own divide-by-world pass."""
from __future__ import annotations

import torch

from . import ops
from ._lib import call

CHUNK = 16384


class FusedAdamW(torch.optim.AdamW):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale: float = 1.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, foreach=False, fused=False)
        self.grad_scale = float(grad_scale)
        self._table = {}            # group index -> (key, device chunk table)

    @staticmethod
    def _dense_same_layout(*ts) -> bool:
        def eff(t):                 # strides of size-1 dims are arbitrary (1x1 conv weights in channels_last)
            return tuple(st for sz, st in zip(t.shape, t.stride()) if sz > 1)
        s0 = eff(ts[0])
        same = all(eff(t) == s0 and t.shape == ts[0].shape and t.dtype == torch.float32 and t.is_cuda for t in ts)
        dense = ts[0].is_contiguous() or (ts[0].dim() == 4 and ts[0].is_contiguous(memory_format=torch.channels_last))
        return same and dense

    def load_state_dict(self, state_dict):
        """torch's loader casts per-parameter state to the parameter's device; the scalar ``step`` counters stay on the
        host so that reading them never synchronises the stream."""
        super().load_state_dict(state_dict)
        for st in self.state.values():
            if torch.is_tensor(st.get("step")):
                st["step"] = st["step"].detach().to("cpu", torch.float32)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("maximize"):
                raise RuntimeError("FusedAdamW: amsgrad / maximize are not supported")
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            # bias correction uses each parameter's OWN step count (torch.optim.AdamW semantics): parameters are grouped
            # by step value - one launch per distinct value, i.e. one launch unless a parameter first received a
            # gradient later than the others.  ``step`` tensors live on the host (see load_state_dict), so reading them
            # does not synchronise with the device.
            by_step = {}
            for p in ps:
                by_step.setdefault(float(self.state[p]["step"]), []).append(p)
            b1, b2 = group["betas"]
            for si, (step0, sub) in enumerate(sorted(by_step.items())):
                key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(),
                             self.state[p]["exp_avg_sq"].data_ptr()) for p in sub)
                cached = self._table.get((gi, si))
                if cached is None or cached[0] != key:
                    rows = []
                    for p in sub:
                        m, v = self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"]
                        if not self._dense_same_layout(p, p.grad, m, v):
                            raise RuntimeError(f"FusedAdamW: parameter of shape {tuple(p.shape)} / its gradient / moments "
                                               "must be dense float32 CUDA tensors with identical strides")
                        n = p.numel()
                        for o in range(0, n, CHUNK):
                            rows.append((p.data_ptr() + 4 * o, p.grad.data_ptr() + 4 * o, m.data_ptr() + 4 * o,
                                         v.data_ptr() + 4 * o, min(CHUNK, n - o)))
                    table = torch.tensor(rows, dtype=torch.int64).to(sub[0].device)
                    cached = (key, table)
                    self._table[(gi, si)] = cached
                for p in sub:
                    self.state[p]["step"] += 1
                call("hoisdf_adamw_step", ops._p(cached[1]), cached[1].shape[0], float(group["lr"]), float(b1), float(b2),
                     float(group["eps"]), float(group["weight_decay"]), int(step0 + 1.0), self.grad_scale, ops._st())
        ops.bump_weight_generation()         # parameters changed without bumping torch's version counters
        return loss
