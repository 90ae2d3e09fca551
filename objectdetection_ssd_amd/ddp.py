"""Data-parallel train step around the hot path: flat parameter / gradient buffers,
ONE gradient all-reduce per step over RCCL (torch.distributed backend "nccl"), fused SGD.

The reference has no distributed code; this is the build
addition BASELINE.json's north_star asks for.  Images are sharded by rank, weights
replicated.  Both losses normalise by the batch-GLOBAL number of positive priors
(reference Losses.py:182,197), so each rank back-propagates the un-normalised sums
(`Losses.ssd(..., norm_mode=1)`), appends its n_pos to the flat gradient buffer,
the buffer is all-reduced once, and the fused SGD kernel multiplies the gradient by
1 / n_pos_global -- the same arithmetic as the reference at the global batch size,
with a single collective.

SGD follows train.py:44-55: momentum 0.9, weight decay 5e-4, biases at 2x lr (weight
decay applies to both groups, as torch.optim.SGD does there).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from . import ops


class FlatSGDDataParallel(torch.optim.Optimizer):
    """The data-parallel optimizer of the step.  It IS a `torch.optim.Optimizer` with the two parameter groups of reference
    train.py:53-55 in the same order (`param_groups[0]` = biases at 2x lr, `param_groups[1]` = the rest), so the caller's
    `StepLR(optimizer, ...)` (train.py:57), `for g in optimizer.param_groups: g['lr'] = lr` (train_function.py:29-30),
    `optimizer.state_dict()` / `load_state_dict()` (train_function.py:27,116) work on it: lr / momentum / weight decay are read
    from the groups at every step, and the state dict carries one `momentum_buffer` per parameter (views of the flat momentum
    buffer), so a resumed run continues with its momentum instead of restarting it."""

    def __init__(self, model, lr: float = 1e-4, momentum: float = 0.9, weight_decay: float = 5e-4,
                 bias_lr_mult: float = 2.0, process_group=None, overlap: bool = False, bucket_bytes: int = 16 << 20):
        """overlap: start the all-reduce of a slice of the flat gradient buffer as soon as the backward has produced all of it
        (asynchronous collectives on the process group's stream, `bucket_bytes` per slice), instead of one all-reduce after the
        backward.  Same sums, same result; opt-in until it has been timed on a multi-GPU node."""
        self.model = model
        self.group = process_group
        named = dict(model.named_parameters())
        names = [n for n in model._engine.names if named[n].requires_grad]
        # weights first, biases second: two contiguous segments = two SGD launches (train.py:46-51 groups)
        self.w_names = [n for n in names if not n.endswith(".bias")]
        self.b_names = [n for n in names if n.endswith(".bias")]
        self.names = self.w_names + self.b_names
        self.params: List[torch.nn.Parameter] = [named[n] for n in self.names]
        n_wn = len(self.w_names)
        super().__init__([{"params": self.params[n_wn:], "lr": lr * bias_lr_mult}, {"params": self.params[:n_wn]}],
                         dict(lr=lr, momentum=momentum, dampening=0.0, weight_decay=weight_decay, nesterov=False))
        dev = self.params[0].device      # flat buffers live where the model lives (the SGD kernel itself is GPU-only)
        sizes = [p.numel() for p in self.params]
        slots = [(s + 3) // 4 * 4 for s in sizes]              # every parameter starts 16-byte aligned
        self.n_w = sum(slots[:len(self.w_names)])
        self.n = sum(slots)
        pad = (-(self.n + 1)) % 64
        self.flat_param = torch.zeros(self.n, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(self.n + 1 + pad, device=dev, dtype=torch.float32)     # [grads | n_pos | pad]
        self.flat_mom = torch.zeros(self.n, device=dev, dtype=torch.float32)
        self.inv_npos = torch.ones(1, device=dev, dtype=torch.float32)
        self.grad_views = []
        self.mom_views = []
        off = 0
        with torch.no_grad():
            for p, sz, slot in zip(self.params, sizes, slots):
                view = self.flat_param[off:off + sz].view_as(p)
                view.copy_(p.data)
                p.data = view                                  # parameters now live in the flat buffer
                self.grad_views.append(self.flat_grad[off:off + sz].view_as(p))
                self.mom_views.append(self.flat_mom[off:off + sz].view_as(p))
                off += slot
        self._has_momentum = False       # False: the next step starts the momentum buffer from the gradient, as torch.optim.SGD does
        self.steps = 0
        model._engine._wcache.clear()
        # -- overlapped exchange: contiguous slices of the weight segment, filled from the back of the network first ------------
        self.overlap = bool(overlap)
        self._handles: list = []
        if self.overlap:
            self._view = dict(zip(self.names, self.grad_views))
            self._bucket_of, self._bucket_rng, self._need = {}, [], []
            lo, cur = 0, []
            n_wn = len(self.w_names)
            offs = [0]
            for sl in slots:
                offs.append(offs[-1] + sl)
            for i in range(n_wn):
                cur.append(self.names[i])
                last = i + 1 == n_wn
                if (offs[i + 1] - lo) * 4 >= bucket_bytes or last:
                    b = len(self._bucket_rng)
                    for nm in cur:
                        self._bucket_of[nm] = b
                    self._bucket_rng.append((lo, offs[i + 1]))
                    self._need.append(len(cur))
                    lo, cur = offs[i + 1], []
            self._left = list(self._need)
            self._seen = 0
            model._engine.grad_sink = self._sink

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def broadcast_parameters(self, src: int = 0) -> None:
        if self.world > 1:
            dist.broadcast(self.flat_param, src, group=self.group)
            self.model._engine._wcache.clear()

    # the values of train.py:53-55 as the step reads them (lr schedulers and the caller's lr reset write the groups)
    @property
    def lr(self) -> float:
        return float(self.param_groups[1]["lr"])

    @lr.setter
    def lr(self, value: float) -> None:
        ratio = self.param_groups[0]["lr"] / self.param_groups[1]["lr"] if self.param_groups[1]["lr"] else 2.0
        self.param_groups[1]["lr"] = float(value)
        self.param_groups[0]["lr"] = float(value) * ratio

    def load_state_dict(self, state_dict) -> None:
        """Restores lr / momentum / weight decay of both groups and every parameter's momentum buffer INTO the flat momentum
        buffer; the step after a restore continues the momentum (a restore without buffers starts it afresh, like torch)."""
        super().load_state_dict(state_dict)
        have = [self.state.get(p, {}).get("momentum_buffer") for p in self.params]
        if any(h is not None for h in have) and not all(h is not None for h in have):
            raise ValueError("FlatSGDDataParallel.load_state_dict: momentum buffers for only some of the parameters")
        self._has_momentum = all(h is not None for h in have)
        with torch.no_grad():
            if self._has_momentum:
                for p, h, mv in zip(self.params, have, self.mom_views):
                    mv.copy_(h.to(mv.device, torch.float32).reshape(mv.shape))
                    self.state[p]["momentum_buffer"] = mv
            else:
                self.flat_mom.zero_()
        self.steps = int(state_dict.get("flat_steps", self.steps)) if isinstance(state_dict, dict) else self.steps

    def state_dict(self):
        sd = super().state_dict()
        sd["flat_steps"] = self.steps
        return sd

    def step(self, closure=None):
        """`optimizer.step()` of the caller's loop = `apply_sgd()`; the gradient exchange (`reduce_gradients`) comes first."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.apply_sgd()
        return loss

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.params:
            p.grad = None
        if self.overlap:                               # a step that died half-way must not leak its bookkeeping into the next
            self._handles, self._left, self._seen = [], list(self._need), 0

    def _sink(self, name: str, grad: torch.Tensor) -> None:
        """Engine callback during backward: park the gradient in its slot; when a weight slice is complete, start its all-reduce."""
        view = self._view.get(name)
        if view is None:
            return
        view.copy_(grad.reshape(view.shape))
        self._seen += 1
        b = self._bucket_of.get(name)
        if b is None:                                  # biases travel with n_pos in the closing collective
            return
        self._left[b] -= 1
        if self._left[b] == 0 and self.world > 1:
            lo, hi = self._bucket_rng[b]
            self._handles.append(dist.all_reduce(self.flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _finish_overlapped(self, n_pos: torch.Tensor) -> None:
        if self._seen != len(self.names) or any(self._left):
            raise RuntimeError("overlapped gradient exchange: the backward did not deliver every parameter's gradient "
                               f"({self._seen} of {len(self.names)})")
        self.flat_grad[self.n:self.n + 1].copy_(n_pos.reshape(1))
        if self.world > 1:
            tail = dist.all_reduce(self.flat_grad[self.n_w:self.n + 1], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            for h in self._handles + [tail]:
                h.wait()                                # the compute stream waits for the collectives, the host does not
        self._handles = []
        self._left = list(self._need)
        self._seen = 0
        torch.reciprocal(self.flat_grad[self.n:self.n + 1], out=self.inv_npos)

    def reduce_gradients(self, n_pos: torch.Tensor) -> None:
        """Pack this rank's gradients (of the UN-normalised loss sums) and its positive-prior count
        (`Losses.last_match['n_pos']`) into the flat buffer and all-reduce it once; afterwards
        `flat_grad[:n] * inv_npos` is the gradient of the reference loss at the global batch."""
        if self.overlap:
            return self._finish_overlapped(n_pos)
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads):
            missing = [n for n, g in zip(self.names, grads) if g is None]
            raise RuntimeError(f"parameters without gradient: {missing[:4]}...")
        torch._foreach_copy_(self.grad_views, grads)
        self.flat_grad[self.n:self.n + 1].copy_(n_pos.reshape(1))
        if self.world > 1:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)    # the one collective per step
        torch.reciprocal(self.flat_grad[self.n:self.n + 1], out=self.inv_npos)

    def apply_sgd(self) -> None:
        first = not self._has_momentum
        gb, gw = self.param_groups
        for g in (gb, gw):
            if g.get("dampening", 0) != 0 or g.get("nesterov", False):
                raise ValueError("FlatSGDDataParallel: dampening / nesterov are not part of the reference's step (train.py:53-55)")
        ops.sgd_momentum_(self.flat_param[:self.n_w], self.flat_grad[:self.n_w], self.flat_mom[:self.n_w],
                          float(gw["lr"]), float(gw["momentum"]), float(gw["weight_decay"]), self.inv_npos,
                          first or gw["momentum"] == 0)
        ops.sgd_momentum_(self.flat_param[self.n_w:], self.flat_grad[self.n_w:self.n], self.flat_mom[self.n_w:],
                          float(gb["lr"]), float(gb["momentum"]), float(gb["weight_decay"]), self.inv_npos,
                          first or gb["momentum"] == 0)
        if first and (gw["momentum"] != 0 or gb["momentum"] != 0):
            for p, mv in zip(self.params, self.mom_views):
                self.state[p]["momentum_buffer"] = mv          # torch.optim.SGD's state layout: checkpoints carry the momentum
            self._has_momentum = True
        self.steps += 1
        self.model._engine._wcache.clear()        # parameters changed in place: re-lay weights next forward

    def reduce_and_step(self, n_pos: torch.Tensor) -> None:
        self.reduce_gradients(n_pos)
        self.apply_sgd()
