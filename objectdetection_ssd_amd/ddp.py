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

import weakref
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
                 bias_lr_mult: float = 2.0, process_group=None, overlap: Optional[bool] = None, bucket_bytes: int = 16 << 20,
                 grad_dtype: torch.dtype = torch.float32, time_exchange: bool = False):
        """overlap: start the all-reduce of a slice of the flat gradient buffer as soon as the backward has produced all of it
        (asynchronous collectives on the process group's stream, `bucket_bytes` per slice), instead of one all-reduce after the
        backward.  Same sums, same result (bitwise: tests/test_ddp_gloo.py at world 2 / 4 / 8).  None (default) = overlapped whenever
        the process group has more than one rank: against a 10 - 20 ms step the 105 MB exchange is no longer negligible when it
        is left exposed behind the backward.

        grad_dtype = torch.bfloat16: the WEIGHT gradients travel as bf16 (52.6 MB instead of 105 MB): each slice is rounded once to
        bf16, summed by the collective in bf16, and widened back into the f32 buffer the optimizer reads (f32 master weights and f32
        momentum unchanged).  The bias segment and the positive-prior count stay f32 -- the count must be exact.  Meant for the
        bf16-tensor mode (BASELINE configs[2]), whose gradients already carry bf16-sized rounding; the f32 default is bit-identical
        to the single all-reduce.

        time_exchange: record HIP events around the points where the COMPUTE stream waits for the collectives, so that
        `exposed_exchange_ms()` reports how long a step's gradient exchange was not hidden behind the backward."""
        self.model = model
        self.group = process_group
        if grad_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("grad_dtype must be torch.float32 or torch.bfloat16")
        self.grad_dtype = grad_dtype
        self.time_exchange = bool(time_exchange)
        self._exch_events: list = []
        if overlap is None:
            overlap = dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1
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
        # The backward writes every gradient straight into its slot of the flat buffer: the engine asks `_grad_out` where the
        # gradient of a parameter (or of the two parameters a fused head convolution produces) goes and reports it through `_sink`;
        # the parameters' .grad are the views.  A gradient that arrives as a tensor of its own (conv1_1's permuted rows; a caller
        # that sets p.grad itself) is copied into its slot.
        self._sizes, self._slots = sizes, slots
        self._offs = [0]
        for sl in slots:
            self._offs.append(self._offs[-1] + sl)
        self._index = {n: i for i, n in enumerate(self.names)}
        self._view = dict(zip(self.names, self.grad_views))
        self._arrived: set = set()
        for p, v in zip(self.params, self.grad_views):
            p.grad = v
        eng = model._engine
        # The engine must not keep this optimizer (and its three 105 MB buffers) alive, and must not keep reporting to it once it is
        # gone: the hooks hold weak references, and when the last outside reference to the optimizer goes -- or `close()` is called --
        # they are taken off again, unless another optimizer has replaced them meanwhile (the token tells whose hooks are installed).
        self._closed = False
        self._token = object()
        eng.grad_sink, eng.grad_out = _WeakMethod(self, "_sink", self._token), _WeakMethod(self, "_grad_out", self._token)
        eng.sink_owns_grads, eng.sink_early = True, bool(overlap)
        try:
            eng_ref = weakref.ref(eng)
        except TypeError:                              # (test stand-ins for the engine that cannot be weakly referenced)
            eng_ref = (lambda e=eng: e)
        self._finalizer = weakref.finalize(self, FlatSGDDataParallel._detach, eng_ref, self._token)
        self.flat_grad16 = (torch.zeros(self.n_w, device=dev, dtype=torch.bfloat16) if grad_dtype == torch.bfloat16 else None)
        # -- overlapped exchange: contiguous slices of the weight segment, filled from the back of the network first ------------
        self.overlap = bool(overlap)
        self._handles: list = []
        if self.overlap:
            self._bucket_of, self._bucket_rng, self._need = {}, [], []
            lo, cur = 0, []
            n_wn = len(self.w_names)
            offs = self._offs
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

    def _from_reference_layout(self, state_dict):
        """A checkpoint written by the single-GPU loop (train_function.py:116: `torch.optim.SGD` -- or this package's `optim.SGD` --
        over train.py:44-55's groups) lists EVERY `requires_grad` parameter in `named_parameters()` order, biases in group 0 and the
        rest in group 1: that includes the never-updated `model.classifier.*` of the VGG trunk, which have no slot here, and orders
        the scale parameter first and the heads last.  Re-index such a state dict by parameter NAME into this optimizer's groups
        (engine order, 35 + 36 entries).  Returns the state dict unchanged when it already has this optimizer's layout."""
        groups = state_dict.get("param_groups", []) if isinstance(state_dict, dict) else []
        mine = [len(g["params"]) for g in self.param_groups]
        if len(groups) != 2 or [len(g["params"]) for g in groups] == mine:
            return state_dict
        named = [(n, p) for n, p in self.model.named_parameters() if p.requires_grad]
        ref_b = [n for n, _ in named if n.endswith(".bias")]
        ref_w = [n for n, _ in named if not n.endswith(".bias")]
        if [len(g["params"]) for g in groups] != [len(ref_b), len(ref_w)]:
            raise ValueError(f"optimizer state with groups of {[len(g['params']) for g in groups]} parameters fits neither this optimizer "
                             f"({mine}) nor train.py's groups over this model ({[len(ref_b), len(ref_w)]})")
        by_name = {}
        for names, g in zip((ref_b, ref_w), groups):
            for n, pid in zip(names, g["params"]):
                by_name[n] = pid
        n_wn = len(self.w_names)
        order = self.names[n_wn:] + self.names[:n_wn]              # this optimizer's packed order: group 0 = biases, group 1 = weights
        new_state, new_groups, k = {}, [], 0
        for g, names in zip(groups, (self.names[n_wn:], self.names[:n_wn])):
            ids = []
            for n in names:
                if n not in by_name:
                    raise ValueError(f"checkpoint has no entry for parameter {n}")
                if by_name[n] in state_dict["state"]:
                    new_state[k] = state_dict["state"][by_name[n]]
                ids.append(k)
                k += 1
            ng = {kk: vv for kk, vv in g.items() if kk not in ("params", "param_names")}
            ng["params"] = ids
            new_groups.append(ng)
        assert k == len(order)
        out = {kk: vv for kk, vv in state_dict.items() if kk not in ("state", "param_groups")}
        out["state"], out["param_groups"] = new_state, new_groups
        return out

    def load_state_dict(self, state_dict) -> None:
        """Restores lr / momentum / weight decay of both groups and every parameter's momentum buffer INTO the flat momentum
        buffer; the step after a restore continues the momentum (a restore without buffers starts it afresh, like torch).
        Accepts this optimizer's own state dicts and those of a train.py-style optimizer over the same model (re-indexed by
        parameter name: `_from_reference_layout`)."""
        state_dict = self._from_reference_layout(state_dict)
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
        """Nothing is zeroed or freed: every slot of the flat buffer is overwritten by the next backward (and `reduce_gradients`
        refuses a step whose backward did not deliver every gradient).  The parameters' .grad stay the views of the flat buffer."""
        for p, v in zip(self.params, self.grad_views):
            p.grad = v
        self._arrived.clear()
        if self.overlap:                               # a step that died half-way must not leak its bookkeeping into the next
            self._handles, self._left = [], list(self._need)

    def _grad_out(self, names):
        """Engine callback: the flat slice the gradient of these CONSECUTIVE parameters is to be written to (one parameter, or the
        loc + conf halves of a fused head convolution), or None when they are not adjacent, unpadded slots of the buffer."""
        idx = [self._index.get(n) for n in names]
        if any(i is None for i in idx):
            return None
        for a, b in zip(idx, idx[1:]):
            if b != a + 1 or self._slots[a] != self._sizes[a]:
                return None
        lo = self._offs[idx[0]]
        return self.flat_grad[lo:self._offs[idx[-1]] + self._sizes[idx[-1]]]

    def _sink(self, name: str, grad: torch.Tensor) -> None:
        """Engine callback during backward, on the caller's stream, the gradient's kernel ordered before it: note the arrival (copy the
        gradient into its slot if it was not written there); overlapped mode: when a weight slice is complete, start its all-reduce."""
        view = self._view.get(name)
        if view is None:
            return
        if grad.data_ptr() != view.data_ptr():
            view.copy_(grad.reshape(view.shape))
        if name in self._arrived:
            raise RuntimeError(f"gradient of {name} delivered twice in one step: the backward writes (not adds) every gradient into the flat "
                               "buffer, so one step is ONE backward pass -- zero_grad() starts a step; accumulating micro-batches is "
                               "not supported by this optimizer (use a larger per-GPU batch: activations are 3.3 GB at 32 images)")
        self._arrived.add(name)
        b = self._bucket_of.get(name) if self.overlap else None
        if b is None:                                  # biases travel with n_pos in the closing collective
            return
        self._left[b] -= 1
        if self._left[b] == 0 and self.world > 1:
            lo, hi = self._bucket_rng[b]
            self._handles.append(self._start_weight_slice(lo, hi))

    def _start_weight_slice(self, lo: int, hi: int):
        """Asynchronous all-reduce of flat_grad[lo:hi] (inside the weight segment) -> (work handle, lo, hi)."""
        if self.flat_grad16 is not None:
            self.flat_grad16[lo:hi].copy_(self.flat_grad[lo:hi])          # one rounding to bf16; the collective sums in bf16
            return dist.all_reduce(self.flat_grad16[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True), lo, hi
        return dist.all_reduce(self.flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True), lo, hi

    def _wait_weight_slice(self, handle) -> None:
        work, lo, hi = handle
        work.wait()                                     # the compute stream waits for the collective, the host does not
        if self.flat_grad16 is not None:
            self.flat_grad[lo:hi].copy_(self.flat_grad16[lo:hi])          # widened back: the optimizer reads f32

    def _mark(self):
        if not self.time_exchange or not self.flat_grad.is_cuda:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def exposed_exchange_ms(self, last: int = 0) -> Optional[float]:
        """Mean time per step the compute stream spent between reaching the gradient exchange's wait point and the moment the last
        collective (and the bf16 widening, if any) had finished -- the part of the exchange the backward did not hide.  Call after a
        device synchronisation; needs `time_exchange=True`.  last = only the most recent `last` steps (0 = all recorded)."""
        ev = self._exch_events[-last:] if last else self._exch_events
        if not ev:
            return None
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev)

    def _finish_overlapped(self, n_pos: torch.Tensor) -> None:
        if len(self._arrived) != len(self.names) or any(self._left):
            raise RuntimeError("overlapped gradient exchange: the backward did not deliver every parameter's gradient "
                               f"({len(self._arrived)} of {len(self.names)})")
        self.flat_grad[self.n:self.n + 1].copy_(n_pos.reshape(1))
        if self.world > 1:
            tail = dist.all_reduce(self.flat_grad[self.n_w:self.n + 1], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            e0 = self._mark()
            for h in self._handles:
                self._wait_weight_slice(h)
            tail.wait()
            e1 = self._mark()
            if e0 is not None:
                self._exch_events.append((e0, e1))
        self._handles = []
        self._left = list(self._need)
        self._arrived.clear()
        torch.reciprocal(self.flat_grad[self.n:self.n + 1], out=self.inv_npos)

    def reduce_gradients(self, n_pos: torch.Tensor) -> None:
        """All-reduce the flat buffer once: this rank's gradients of the UN-normalised loss sums (written there by the backward) and
        its positive-prior count (`Losses.ssd(..., with_n_pos=True)`); afterwards `flat_grad[:n] * inv_npos` is the gradient of the
        reference loss at the global batch."""
        if self.overlap and (self._arrived or self._handles):
            return self._finish_overlapped(n_pos)
        # (an overlapped optimizer whose engine reported nothing this step: the caller filled .grad itself -- one all-reduce, as below)
        late = [i for i, n in enumerate(self.names) if n not in self._arrived]
        if late:                                         # gradients the caller put into .grad itself (no engine callback)
            gs = [self.params[i].grad for i in late]
            bad = [self.names[i] for i, g in zip(late, gs) if g is None or g.data_ptr() == self.grad_views[i].data_ptr()]
            if bad:
                raise RuntimeError(f"parameters without gradient: {bad[:4]}...")
            torch._foreach_copy_([self.grad_views[i] for i in late], [g.reshape(self.grad_views[i].shape) for i, g in zip(late, gs)])
        self._arrived.clear()
        self.flat_grad[self.n:self.n + 1].copy_(n_pos.reshape(1))
        if self.world > 1:
            e0 = self._mark()
            if self.flat_grad16 is not None:             # bf16 weight payload + the small f32 tail (biases, n_pos)
                h = self._start_weight_slice(0, self.n_w)
                tail = dist.all_reduce(self.flat_grad[self.n_w:self.n + 1], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self._wait_weight_slice(h)
                tail.wait()
            else:
                dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)    # the one collective per step
            e1 = self._mark()
            if e0 is not None:
                self._exch_events.append((e0, e1))
        torch.reciprocal(self.flat_grad[self.n:self.n + 1], out=self.inv_npos)

    # -- lifetime -----------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _detach(eng_ref, token) -> None:
        eng = eng_ref()
        if eng is not None and getattr(eng.grad_sink, "_token", None) is token:
            eng.grad_sink, eng.grad_out, eng.sink_owns_grads, eng.sink_early = None, None, False, False

    def close(self) -> None:
        """Take this optimizer's hooks off the model's engine: later backward passes hand their gradients to autograd again (a plain
        optimizer can follow).  The parameters keep living in the flat buffer -- they are ordinary tensors viewing it -- and their
        .grad stay views of the flat gradient buffer until the caller resets them (`model.zero_grad(set_to_none=True)`)."""
        if not self._closed:
            self._closed = True
            self._finalizer()

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


class _WeakMethod:
    """Bound method through a weak reference to its object: the engine's hooks must not own the optimizer."""

    def __init__(self, obj, name: str, token=None):
        self._ref, self._name, self._token = weakref.ref(obj), name, token

    def __call__(self, *a, **k):
        obj = self._ref()
        if obj is None:
            raise RuntimeError("the FlatSGDDataParallel this engine reports to no longer exists")
        return getattr(obj, self._name)(*a, **k)


class GraphedTrainStep:
    """One train step of the hot path -- forward + MultiBox loss + backward (+ the fused SGD when there is one rank) -- captured once into a
    HIP graph and replayed: the eager step issues ~250 launches from one Python thread (4-5 ms of host time per step against a
    10-20 ms step); the replay is one `hipGraphLaunch` plus the handful of copies that bring the batch into the static buffers.
    The serving loop it replaces is train_function.py:80-95 (`outputs = model(inputs)`, `ssd(outputs, ...)`, `loss.backward()`,
    `optimizer.step()`).

    Everything the library launches only enqueues on the stream it is given (no allocation, no synchronisation), so the capture
    needs nothing special from the kernels; what it needs from the host side is fixed shapes:

      * the image batch has the shape of the first call;
      * the ground truth lives in static buffers of `max_boxes` rows (default 8 per image, VOC's practical maximum is 42 for one
        image and ~2.4 on average); the per-image offsets `img_start` are data, so any split of <= max_boxes boxes over the images
        replays correctly -- rows beyond the last offset are never read by the matching, the loss or the gradients;
      * the number of positive priors stays on the device (it is written into the flat gradient buffer's count slot inside the graph).

    The first `warmup` calls run the step eagerly (they are ordinary training steps: allocator pools, workspaces, weight tables and the
    momentum buffers come into being), the next call captures and replays.  The graph bakes in lr / momentum / weight decay: it is
    captured again when the optimizer's groups change (StepLR).  With more than one rank the gradient exchange is ONE all-reduce
    issued after the graph (a collective started from inside the backward cannot be part of a replay), followed by the two SGD
    launches; `FlatSGDDataParallel(overlap=...)` is switched off for the life of this object.

    two_streams (default): the tiny-map group (c_8, seq9 ... c_11) is captured on the engine's second stream, as the eager step runs it
    -- the graph then has two parallel branches (fork / join by events inside the capture); False puts every node on one chain.
    Measured interleaved with the eager step on one device (tools/ab_graph.py, batch 32, f32): eager 20.41 ms, replay with two branches
    20.57, replay on one chain 21.03 -- the step is GPU-bound, so the replay buys HOST time (5.1 -> 0.3 ms per step), not device time.

    `__call__(x, classes, boxes)` -> (loc_sum, conf_sum, n_pos): 0-dim views of a static device tensor holding this rank's
    un-normalised loss sums and positive count (`Losses.ssd(..., norm_mode=1, with_n_pos=True)`), overwritten by the next call."""

    def __init__(self, net, trainer: FlatSGDDataParallel, max_boxes_per_image: int = 8, warmup: int = 2, two_streams: bool = True):
        self.net, self.trainer = net, trainer
        self.max_per_image, self.warmup = int(max_boxes_per_image), int(warmup)
        self.two_streams = bool(two_streams)
        self.graph = None
        self._sig = None
        self._calls = 0
        self.kernel_nodes = None          # launches inside the captured graph (hipGraphGetNodes), filled at capture
        self._static = None
        trainer.overlap = False           # the exchange of a replayed step is one all-reduce behind the graph
        net._engine.sink_early = False
        self._stream = None
        self._ring, self._ring_pos = [], 0

    # -- static buffers ------------------------------------------------------------------------------------------------------------
    def _make_static(self, x: torch.Tensor):
        dev, bs = x.device, x.shape[0]
        cap = bs * self.max_per_image
        self._static = dict(x=torch.empty_like(x), gt=torch.zeros((cap, 4), device=dev), cls=torch.zeros((cap,), device=dev),
                            start=torch.zeros((bs + 1,), device=dev, dtype=torch.int32))
        # pinned staging for ground truth that arrives on the host, four steps deep (the host may run that far ahead of the device)
        self._ring = [dict(gt=torch.zeros((cap, 4)).pin_memory(), cls=torch.zeros((cap,)).pin_memory(),
                           start=torch.zeros((bs + 1,), dtype=torch.int32).pin_memory(), done=None) for _ in range(4)]

    def _load_batch(self, x, classes, boxes):
        S = self._static
        if tuple(x.shape) != tuple(S["x"].shape) or x.dtype != S["x"].dtype:
            raise ValueError(f"graph step built for images {tuple(S['x'].shape)}, got {tuple(x.shape)}")
        counts = [int(b.shape[0]) for b in boxes]
        if len(counts) != x.shape[0] or any(c == 0 for c in counts) or any(int(c.shape[0]) != n for c, n in zip(classes, counts)):
            raise ValueError("every image needs at least one ground-truth box, and classes / boxes must agree")   # Losses.py:153
        n = sum(counts)
        if n > S["gt"].shape[0]:
            raise ValueError(f"{n} ground-truth boxes in the batch, the graph step holds {S['gt'].shape[0]} (max_boxes_per_image)")
        S["x"].copy_(x, non_blocking=True)
        slot = self._ring[self._ring_pos]
        self._ring_pos = (self._ring_pos + 1) % len(self._ring)
        if slot["done"] is not None:
            slot["done"].synchronize()                       # the copy that last read this staging slot (four steps ago)
        st = slot["start"]
        st[0] = 0
        torch.cumsum(torch.tensor(counts, dtype=torch.int32), 0, out=st[1:])
        if boxes[0].is_cuda:
            torch.cat([b.reshape(-1, 4).to(torch.float32) for b in boxes], out=S["gt"][:n])
            torch.cat([c.reshape(-1).to(torch.float32) for c in classes], out=S["cls"][:n])
        else:
            torch.cat([b.reshape(-1, 4).to(torch.float32) for b in boxes], out=slot["gt"][:n])
            torch.cat([c.reshape(-1).to(torch.float32) for c in classes], out=slot["cls"][:n])
            S["gt"][:n].copy_(slot["gt"][:n], non_blocking=True)
            S["cls"][:n].copy_(slot["cls"][:n], non_blocking=True)
        S["start"].copy_(st, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        slot["done"] = ev

    # -- the step, written against the static buffers (what the capture records) ------------------------------------------------------
    def _body(self, with_sgd: bool):
        from . import Losses
        S, net, tr = self._static, self.net, self.trainer
        eng = net._engine
        P = net._forward_params()
        loc, conf, saved = eng.forward(S["x"], P, save=True)
        pri, pri_xyxy = Losses._priors_on(loc.device, loc.shape[1])
        out = ops.multibox_loss(loc, conf, S["gt"], S["cls"], S["start"], pri, pri_xyxy, Losses.IOU_THRESHOLD, Losses.NEG_POS_RATIO, 1,
                                want_grads=True)
        need = {n: bool(P[n].requires_grad) for n in eng.names}
        eng.backward(saved, out["dloc"], out["dconf"], P, need)
        tr.flat_grad[tr.n:tr.n + 1].copy_(out["losses"][2:3])
        tr._arrived.clear()
        if with_sgd:
            torch.reciprocal(tr.flat_grad[tr.n:tr.n + 1], out=tr.inv_npos)
            tr.apply_sgd()
        return out["losses"], out["obj"], out["cls"]

    def _signature(self):
        tr = self.trainer
        eng = self.net._engine
        return tuple((g["lr"], g["momentum"], g["weight_decay"]) for g in tr.param_groups) + (
            tr.world, eng.bf16, eng.bf16_tensors, eng.x3, eng.wino, eng.WINO_TILE, ops.wino_x3(4, 256))

    def _eager(self, x, classes, boxes):
        from . import Losses
        tr = self.trainer
        tr.zero_grad()
        loc, conf = self.net(x)
        l1, l2, n_pos = Losses.ssd((loc, conf), classes, boxes, norm_mode=1, with_n_pos=True)
        (l1 + l2).backward()
        tr.reduce_and_step(n_pos)
        return l1.detach(), l2.detach(), n_pos

    def _capture(self):
        eng, tr = self.net._engine, self.trainer
        dev = self._static["x"].device
        single = tr.world == 1
        keep_tail = eng.overlap_tail
        if not self.two_streams:
            eng.overlap_tail = False                          # one stream inside the graph (bit-identical to the two-stream schedule)
        try:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=dev)
            s = self._stream
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s), torch.no_grad():       # one untimed pass on the capture stream: its workspaces, the static shapes
                self._body(with_sgd=False)
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize(dev)
            # (that pass wrote gradients and n_pos only: no parameter moved)
            try:
                g = torch.cuda.CUDAGraph(keep_graph=True)
            except TypeError:
                g = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(g, stream=s):
                self.losses, self.obj, self.cls = self._body(with_sgd=single)
            self.graph = g
            self.kernel_nodes = _graph_kernel_nodes(g)
            if single:
                tr.steps -= 1                                 # apply_sgd's bookkeeping ran once during capture without a step being executed
        finally:
            eng.overlap_tail = keep_tail
        self._sig = self._signature()

    def __call__(self, x, classes, boxes):
        tr = self.trainer
        self._calls += 1
        if self._calls <= self.warmup or not tr._has_momentum:
            return self._eager(x, classes, boxes)
        if self._static is None:
            self._make_static(x)
        self._load_batch(x, classes, boxes)
        if self.graph is None or self._sig != self._signature():
            self._capture()
        self.graph.replay()
        if tr.world > 1:
            dist.all_reduce(tr.flat_grad, op=dist.ReduceOp.SUM, group=tr.group)      # the one collective per step
            torch.reciprocal(tr.flat_grad[tr.n:tr.n + 1], out=tr.inv_npos)
            tr.apply_sgd()
        else:
            tr.steps += 1
            self.net._engine._wcache.clear()
        return self.losses[0], self.losses[1], self.losses[2]


def _graph_kernel_nodes(graph) -> Optional[int]:
    """Kernel nodes of a captured torch graph (the launches one replay stands for), through the library's `ssd_graph_node_counts`."""
    try:
        raw = graph.raw_cuda_graph()
    except Exception:
        return None
    import ctypes as C
    from . import _lib
    kernels, total = C.c_int(0), C.c_int(0)
    if _lib.load().ssd_graph_node_counts(C.c_void_p(int(raw)), C.byref(kernels), C.byref(total)) != 0:
        return None
    return int(kernels.value)
