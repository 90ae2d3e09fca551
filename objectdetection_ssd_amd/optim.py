"""`torch.optim.SGD` for the step that follows the hot path, as one fused launch per parameter group.

The reference builds its optimizer as (train.py:44-55)

    torch.optim.SGD(params=[{'params': biases, 'lr': 2*lr}, {'params': not_biases}], lr=lr, momentum=0.9, weight_decay=5e-4)

and `train_function.py` then uses `.zero_grad()`, `.step()`, `.param_groups[*]['lr']`, `.state_dict()` and
`.load_state_dict()` (train_function.py:8-9,27-30,76,95,116).  `SGD` here keeps all of that -- same constructor, same
param-group keys, same `state[p]['momentum_buffer']` entries, so checkpoints move between the two -- and runs
`ssd_sgd_momentum` (include/ssd_gfx950.h) once per group on flat storage: each group's parameters and momentum buffers
are re-seated as views of one buffer the first time `step()` sees them, gradients are gathered with one foreach copy.
The kernel reproduces torch's rounding sequence, so the parameters after k steps are bit-identical to torch.optim.SGD's.

Only what the reference uses is implemented: `dampening`, `nesterov`, `maximize` must stay at their defaults.  GPU only,
like the rest of the package.
"""
from __future__ import annotations

from typing import List

import torch

from . import ops


class _GroupBuffers:
    """Flat storage of one parameter group."""

    def __init__(self, params: List[torch.nn.Parameter], state) -> None:
        dev = params[0].device
        sizes = [p.numel() for p in params]
        slots = [(s + 3) // 4 * 4 for s in sizes]                  # every parameter starts 16-byte aligned
        self.offsets, off = [], 0
        for s in slots:
            self.offsets.append(off)
            off += s
        self.total = off
        self.sizes = sizes
        self.param = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        self.mom = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad_views = []
        with torch.no_grad():
            for p, o, sz in zip(params, self.offsets, sizes):
                view = self.param[o:o + sz].view_as(p)
                view.copy_(p.data)
                p.data = view
                self.grad_views.append(self.grad[o:o + sz].view_as(p))
                st = state.get(p)
                if st is not None and st.get("momentum_buffer") is not None:
                    mv = self.mom[o:o + sz].view_as(p)
                    mv.copy_(st["momentum_buffer"])
                    st["momentum_buffer"] = mv
        self.ptrs = [p.data_ptr() for p in params]

    def seated(self, params) -> bool:
        return len(params) == len(self.ptrs) and all(p.data_ptr() == q for p, q in zip(params, self.ptrs))


class SGD(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, momentum: float = 0.0, dampening: float = 0.0, weight_decay: float = 0.0,
                 nesterov: bool = False, *, maximize: bool = False, foreach=None, differentiable: bool = False, fused=None):
        if lr < 0.0 or momentum < 0.0 or weight_decay < 0.0:
            raise ValueError("SGD: lr, momentum and weight_decay must be non-negative")
        if dampening != 0.0 or nesterov or maximize or differentiable:
            raise ValueError("SGD: dampening / nesterov / maximize / differentiable are not part of the reference's step "
                             "(train.py:53-55) and are not implemented")
        defaults = dict(lr=lr, momentum=momentum, dampening=0.0, weight_decay=weight_decay, nesterov=False, maximize=False,
                        foreach=foreach, differentiable=False, fused=fused)
        super().__init__(params, defaults)
        self._buffers: List[_GroupBuffers | None] = [None] * len(self.param_groups)

    def add_param_group(self, param_group) -> None:
        super().add_param_group(param_group)
        if hasattr(self, "_buffers"):
            self._buffers.append(None)

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        self._buffers = [None] * len(self.param_groups)           # restored momentum buffers are re-seated at the next step

    def _group_buffers(self, gi: int, params) -> _GroupBuffers:
        buf = self._buffers[gi]
        if buf is None or not buf.seated(params):                 # first step, or the caller moved / replaced the tensors
            for p in params:
                if p.dtype != torch.float32 or p.device.type != "cuda" or not p.is_contiguous():
                    raise ValueError("SGD: parameters must be contiguous float32 tensors on the GPU")
            buf = self._buffers[gi] = _GroupBuffers(params, self.state)
        return buf

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            if group["dampening"] != 0 or group["nesterov"] or group.get("maximize", False):
                raise ValueError("SGD: dampening / nesterov / maximize are not implemented")
            params = group["params"]
            if not params:
                continue
            buf = self._group_buffers(gi, params)
            lr, mom, wd = float(group["lr"]), float(group["momentum"]), float(group["weight_decay"])
            # torch skips parameters without a gradient and starts a momentum buffer at the first gradient it sees: maximal runs of
            # neighbouring parameters in the same situation are one launch (the whole group, in the reference's loop)
            runs, cur = [], None
            for i, p in enumerate(params):
                if p.grad is None:
                    cur = None
                    continue
                if p.grad.is_sparse:
                    raise ValueError("SGD: sparse gradients are not supported")
                fresh = mom != 0.0 and self.state[p].get("momentum_buffer") is None
                if cur is not None and cur[2] == fresh:
                    cur[1] = i + 1
                else:
                    cur = [i, i + 1, fresh]
                    runs.append(cur)
            have = [i for i, p in enumerate(params) if p.grad is not None]
            if not have:
                continue
            torch._foreach_copy_([buf.grad_views[i] for i in have], [params[i].grad for i in have])
            for lo, hi, fresh in runs:
                a = buf.offsets[lo]
                b = buf.offsets[hi - 1] + (buf.sizes[hi - 1] + 3) // 4 * 4
                # momentum 0: torch keeps no buffer and steps with the gradient itself = the "first step" form of the kernel
                ops.sgd_momentum_(buf.param[a:b], buf.grad[a:b], buf.mom[a:b], lr, mom, wd, None, fresh or mom == 0.0)
                for i in range(lo, hi):
                    p = params[i]
                    if mom != 0.0 and fresh:
                        o, sz = buf.offsets[i], buf.sizes[i]
                        self.state[p]["momentum_buffer"] = buf.mom[o:o + sz].view_as(p)
                    torch.autograd.graph.increment_version(p)     # written outside autograd: invalidate re-laid weight caches
        return loss
