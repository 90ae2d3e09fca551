"""`Model.SSD_300` with the reference's constructor, attribute / parameter names and
`forward(x) -> (loc (bs,8732,4), conf (bs,8732,21))` contract (reference
Model.py:128-235), computed by the gfx950 kernels of libssd_gfx950.so.

The module tree exists to own the `nn.Parameter`s under the reference's
state-dict names (so `train.py`'s optimizer groups, `.to(device)`,
`state_dict()` and checkpoints keep working); none of the `nn.Conv2d` /
`nn.MaxPool2d` modules is ever called.  `forward` hands the parameters to
`_Engine`, which runs the network as a flat list of NHWC ops through the C ABI
and, under autograd, supplies an explicit backward (dgrad / wgrad / pool /
L2-norm kernels) through one `torch.autograd.Function`.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from .Util import ANCHORS_PER_CELL, ANCHORS_PER_CELL_512, subsampling

N_CLASSES = 21
_VGG_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")


class _VGG16(nn.Module):
    """Layer list of VGG-16 'D' (what `torchvision.models.vgg16()` builds and reference
    Model.py:131-157 slices by index).  Pretrained weights are a network download in
    the reference; here the layers are initialised like torchvision's untrained model
    and `load_state_dict` accepts the torchvision key names (`features.N.weight`, ...)."""

    def __init__(self):
        super().__init__()
        layers: List[nn.Module] = []
        cin = 3
        for v in _VGG_CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        self.classifier = nn.Sequential(nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), nn.Dropout(),
                                        nn.Linear(4096, 4096), nn.ReLU(True), nn.Dropout(), nn.Linear(4096, 1000))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)


# ------------------------------------------------------------------------------------------
# op list
# ------------------------------------------------------------------------------------------
def _conv(p, x, y, ci, co, k=3, s=1, pad=1, dil=1, relu=True):
    return dict(op="conv", p=p, x=x, y=y, ci=ci, co=co, k=k, s=s, pad=pad, dil=dil, relu=relu)


def _pool(x, y, k=2, s=2, pad=0, ceil=False):
    return dict(op="pool", x=x, y=y, k=k, s=s, pad=pad, ceil=ceil)


def _head(p, x, scale, ci, anchors=ANCHORS_PER_CELL):
    return dict(op="head", p=p, x=x, scale=scale, ci=ci, a=anchors[scale])


# aux blocks (name, cin, mid, cout, stride, pad) after fc7
_AUX_300 = (("seq8", 1024, 256, 512, 2, 1), ("seq9", 512, 128, 256, 2, 1), ("seq10", 256, 128, 256, 1, 0), ("seq11", 256, 128, 256, 1, 0))
_AUX_512 = (("seq8", 1024, 256, 512, 2, 1), ("seq9", 512, 128, 256, 2, 1), ("seq10", 256, 128, 256, 2, 1), ("seq11", 256, 128, 256, 2, 1),
            ("seq12", 256, 128, 256, 2, 1))


def build_ops(variant: int = 300) -> List[dict]:
    """SSD300 as a flat op list (reference Model.py:203-235).  Order matters only where a
    tensor has two consumers: in reverse order the consumer that cannot accumulate
    (L2-norm backward) must deliver its gradient first."""
    if variant not in (300, 512):
        raise ValueError("variant must be 300 or 512")
    anchors = ANCHORS_PER_CELL if variant == 300 else ANCHORS_PER_CELL_512
    f = "model.features."
    o = [dict(op="conv_first", p=f + "0", x="x", y="a1_1"),
         _conv(f + "2", "a1_1", "a1_2", 64, 64), _pool("a1_2", "p1"),
         _conv(f + "5", "p1", "a2_1", 64, 128), _conv(f + "7", "a2_1", "a2_2", 128, 128), _pool("a2_2", "p2"),
         _conv(f + "10", "p2", "a3_1", 128, 256), _conv(f + "12", "a3_1", "a3_2", 256, 256),
         _conv(f + "14", "a3_2", "a3_3", 256, 256), _pool("a3_3", "p3", ceil=True),            # Model.py:137
         _conv(f + "17", "p3", "a4_1", 256, 512), _conv(f + "19", "a4_1", "a4_2", 512, 512),
         _conv(f + "21", "a4_2", "a4_3", 512, 512),
         _pool("a4_3", "p4"),
         dict(op="l2norm", p="rescaling_conv_4_3", x="a4_3", y="n4_3"),                         # Model.py:206-209
         _head("c_4", "n4_3", 0, 512, anchors),
         _conv(f + "24", "p4", "a5_1", 512, 512), _conv(f + "26", "a5_1", "a5_2", 512, 512),
         _conv(f + "28", "a5_2", "a5_3", 512, 512), _pool("a5_3", "p5", k=3, s=1, pad=1, ceil=True),   # Model.py:142
         _conv("conv_fc6", "p5", "a6", 512, 1024, k=3, pad=4, dil=4),                            # Model.py:149
         _conv("conv_fc7", "a6", "a7", 1024, 1024, k=1, pad=0),
         _head("c_7", "a7", 1, 1024, anchors)]
    prev = "a7"
    for i, (name, cin, mid, cout, s, pad) in enumerate(_AUX_300 if variant == 300 else _AUX_512):
        n = 8 + i
        o.append(_conv(f"{name}.0", prev, f"a{n}a", cin, mid, k=1, pad=0))
        o.append(_conv(f"{name}.2", f"a{n}a", f"a{n}", mid, cout, k=3, s=s, pad=pad))
        o.append(_head(f"c_{n}", f"a{n}", 2 + i, cout, anchors))
        prev = f"a{n}"
    return o


def param_names(op_list: List[dict]) -> List[str]:
    names = []
    for op in op_list:
        if op["op"] in ("conv", "conv_first"):
            names += [op["p"] + ".weight", op["p"] + ".bias"]
        elif op["op"] == "l2norm":
            names.append(op["p"])
        elif op["op"] == "head":
            names += [op["p"] + "_bb.weight", op["p"] + "_bb.bias", op["p"] + "_cl.weight", op["p"] + "_cl.bias"]
    return names


def _cat_flat(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """cat((a, b)) of two 1-D parameters -- without a copy when b starts where a ends in one storage (ddp.py lays the loc and conf
    biases of a head side by side in its flat parameter buffer)."""
    if (a.dim() == 1 and b.dim() == 1 and a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and a.storage_offset() + a.numel() == b.storage_offset()):
        return torch.as_strided(a, (a.numel() + b.numel(),), (1,), a.storage_offset())
    return torch.cat((a, b))


class _Elided:
    """Stands in the activation table for a tensor that a fused kernel consumed without writing it: shape only.  In the backward it
    takes the ReLU-mask slot of its pool, which then gates by the pooled output (`maxpool_bwd(..., y_gate=)`)."""

    dtype = torch.float32

    def __init__(self, shape):
        self.shape = tuple(shape)


class _SinkDict(dict):
    """Gradient table of one backward pass that tells a listener about every entry as soon as it exists (ddp.py starts the
    all-reduce of a bucket when its last gradient has been computed, while the rest of the backward is still running).

    The listener is only ever called with the CALLER's stream current and with the producing kernel ordered before that stream's
    position: gradients produced on the engine's side stream are held (`hold()`) until the caller's stream has waited for the side
    stream's event (`release()`), so a collective the listener starts can never read a gradient that is still being written."""

    def __init__(self, sink):
        super().__init__()
        self._sink = sink
        self._held = None

    def __setitem__(self, name, tensor):
        super().__setitem__(name, tensor)
        if self._held is not None:
            self._held.append((name, tensor))
        else:
            self._sink(name, tensor)

    def hold(self):
        if self._held is None:
            self._held = []

    def release(self):
        held, self._held = self._held or [], None
        for name, tensor in held:
            self._sink(name, tensor)


class _Engine:
    """Runs the op list on the current HIP stream.  Holds only caches (re-laid-out weights)."""

    def __init__(self, variant: int = 300):
        ops_ = build_ops(variant)
        self.names = param_names(ops_)            # parameter order of the autograd Function: the reference's forward order
        # Schedule: [backbone ... seq8] then [L2-norm, c_4, c_7: big MFMA-bound head convolutions] then [c_8, seq9 ... c_11: a dozen tiny
        # maps (10x10 ... 1x1) whose ~60 launches per direction are latency-bound].  The last group runs on a second HIP stream beside
        # the middle one, forward and backward: it depends only on a8, and nothing but its own ops reads what it produces.
        late = [o for o in ops_ if o["op"] == "l2norm" or (o["op"] == "head" and o["p"] in ("c_4", "c_7"))]
        k8 = next(i for i, o in enumerate(ops_) if o["op"] == "conv" and o["y"] == "a8")
        side = [o for i, o in enumerate(ops_) if i > k8 and o not in late]
        early = [o for o in ops_ if o not in late and o not in side]
        self.ops = early + late + side
        self._side_ids = {id(o) for o in side}
        self._late_first = id(late[0])
        self.defer_tail_wgrad = True              # backward: the side group's weight gradients after its data-gradient chain, behind the join
        self.overlap_tail = True                  # False: everything on the caller's stream (bit-identical; interleaved A/B: 26.09 -> 25.85 ms)
        self._side_stream = None
        # Backward: the Winograd weight-gradient GEMMs (MFMA-bound, nothing waits for them) on their own stream beside the chain
        # dy transform -> dgrad GEMM -> output transform -> next layer's dy transform, whose transforms are HBM-bound.  Opt-in: measured
        # +-0.2 ms (the GEMMs end up beside other GEMMs, the transform kernels leave no room on the CUs), and the planes that cross the
        # streams (GBs per layer, held by record_stream until the other stream has passed) make torch's caching allocator reserve
        # 180 GB instead of 19.
        self.overlap_wgrad = False
        self._wgrad_stream = None
        self.batch_weights = True     # training forward: all ~56 filter transforms / re-layouts of the step in ONE launch (ops.WeightTable)
        self._wtable = None           # (signature, WeightTable, [(cache key, kind, layer signature fn, buffers)])
        self._wcache: Dict[str, tuple] = {}
        self.consumers: Dict[str, int] = {}
        for op in self.ops:
            self.consumers[op["x"]] = self.consumers.get(op["x"], 0) + 1
        self.relu_out = {op["y"] for op in self.ops if op["op"] == "conv_first" or (op["op"] == "conv" and op["relu"])}
        # conv -> ReLU -> MaxPool2d(2, 2) where nothing else reads the convolution (conv1_2, conv2_2, conv3_3): one fused pass
        self.pool_after: Dict[str, dict] = {}
        for op in self.ops:
            readers = [o for o in self.ops if o["x"] == op["y"]] if op["op"] == "conv" else []
            if (len(readers) == 1 and readers[0]["op"] == "pool" and (readers[0]["k"], readers[0]["s"], readers[0]["pad"]) == (2, 2, 0)
                    and op["relu"] and op["co"] % 4 == 0):
                self.pool_after[op["y"]] = readers[0]
        self._conv_of = {o["y"]: o for o in self.ops if o["op"] == "conv"}      # producer of an activation, where that is a convolution
        self.grad_sink = None     # callable(name, gradient): called during backward the moment a parameter's gradient is ready
        self.grad_out = None      # callable(names) -> flat f32 tensor or None: where the gradient of these consecutive parameters is to be written
        self.sink_owns_grads = False   # True (ddp.py): gradients live in the listener's buffer, autograd is handed None for them
        self.sink_early = False        # True: the listener acts on a gradient at once (overlapped all-reduce), so side-stream gradients are joined early
        self._test_side_delay = None   # tests: callable run on the side stream in front of the deferred weight gradients
        self.grad_tap = None           # tests: dict that receives every activation's finished gradient tensor (name -> dX as the backward stored it)
        self.dual_dy = True       # backward: one pass over dy writes the planes of the weight gradient AND of the data gradient
        self.keep_planes = True   # training forward keeps the F(4x4) input planes of the layers whose weight gradient is Winograd
        self.fuse_pool = True     # Winograd F(4x4) layers in pool_after write the pooled map + argmax only ...
        self.lazy_pool_grad = True     # ... and in the backward their dy pass reads the pooled gradient: the pool's dx is never written
        self.first_fused = True        # conv1_1 forward and weight gradient straight from the NCHW batch (csrc/conv_first.hip), no im2col rows
        self.wino_dilated = True       # fc6 (3x3, dilation 4) in the Winograd domain too: 49 tiles x 36 products per image instead of 361 x 9
        self.relu_bits = True     # training forward: the input transform also leaves the ReLU mask of its input as bits for the dgrad epilogue
        # Round 4: data gradient of the F(4x4) layers with >= 256 output channels in the ADJOINT Winograd form -- it multiplies the planes
        # A dy A^T the weight gradient already has (no second plane set B^T dy B), and between two such layers of one resolution
        # (conv3_3 -> 3_2 -> 3_1, conv4_3 -> 4_2 -> 4_1, conv5_3 -> 5_2 -> 5_1) the gradient tensor itself is never written: the output
        # transform of the upper layer's data gradient writes the lower layer's dy planes (csrc/winograd.hip wino4_adj_out_kernel)
        self.adjoint_dgrad = True
        # Round 4 experiment (SSD_EXPERIMENTAL builds): conv1_1 writes conv1_2's Winograd input planes (+ ReLU bits) itself; its 64-channel
        # activation -- read by nothing else -- is never stored (csrc/conv_first.hip conv_first_wino_kernel; bit-identical planes).  Measured
        # SLOWER than the two kernels it replaces (0.96 vs 0.25 + 0.52 ms; step 20.16 vs 19.97): off
        self.first_wino = False
        self.adjoint_chain = True      # False: every adjoint data gradient is written out as a tensor (A/B aid)
        self.prof = None          # bench.py: list collecting (label, kernel tag, flops, start event, end event)
        self.bf16 = False         # True: forward / dgrad / fused-wgrad convolutions multiply bf16-rounded operands (f32 accumulate)
        # bf16 mode: the VGG trunk (conv1_1 ... pool5, the L2-norm and the c_4 head's input) keeps its activations and gradients in bf16 in
        # HBM -- written once by the producing kernel -- and runs csrc/conv_bf16.hip (LDS-DMA halo kernel), the bf16-input nine-tap weight
        # gradient and the bf16 pools / L2-norm; fc6 onwards (19x19 maps and smaller) stays on the f32-tensor bf16-operand kernels.
        # False: the round-2 form (every tensor f32 in HBM, operands rounded on their way into LDS).
        self.bf16_tensors = True
        self.x3 = False           # True: forward / dgrad convolutions form f32 products from three bf16 limbs per operand
        self.wino = True          # f32 mode: Winograd F(4x4,3x3) (WINO_TILE) for the 3x3 / stride-1 layers with >= WINO_MIN_CI input channels

    def _timed(self, label, tag, flops, fn):
        """Run fn(); when profiling, bracket it with HIP events on the current stream."""
        if self.prof is None:
            return fn()
        executed = flops
        if isinstance(flops, tuple):          # (direct-convolution FLOPs, FLOPs the Winograd GEMMs execute)
            flops, executed = flops
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.prof.append((label, tag, flops, e0, e1, executed))
        return out

    # -- weights ----------------------------------------------------------------------------
    def _layouts(self, key: str, tensors, co_pad: int, need_bwd: bool):
        """Cached [Co_pad][T][Ci] / [Ci][T][Co_pad] copies, refreshed when a parameter changes."""
        sig = tuple((t.data_ptr(), t._version) for t in tensors)
        ent = self._wcache.get(key)
        if ent is None or ent[0] != sig:
            w = tensors[0] if len(tensors) == 1 else torch.cat(list(tensors), 0)
            ent = [sig, w.detach().contiguous(), None, None]
            ent[2] = ops.weight_ohwi(ent[1], co_pad)
            self._wcache[key] = ent
        if need_bwd and ent[3] is None:
            ent[3] = ops.weight_ihwo(ent[1], co_pad)
        if self.x3 or (self.bf16 and not self.bf16_tensors):      # pre-split limb planes of the layouts in use
            if len(ent) == 4:
                ent += [None, None]
            if ent[4] is None:
                ent[4] = ops.weight_split3(ent[2])
            if need_bwd and ent[5] is None:
                ent[5] = ops.weight_split3(ent[3])
            if self.x3:
                return ent[4], ent[5]
        return ent[2], ent[3]

    WINO_TILE = 4                 # forward / dgrad output tile: F(4x4,3x3) (36 multiplies per 16 outputs; ~1e-5 of the output scale) or 2
    WINO_WGRAD_MAX_HW = 512       # Winograd weight gradient on maps up to this size and from this many input channels.  Steps measured in
    WINO_WGRAD_MIN_CI = 64        # the train step: (80, 256) 985 -> (150, 128) 1018 images/s; once the forward planes were kept and one pass
                                  # over dy fed both gradients, the 64-channel layers paid too: + conv2_1 +0.7 %, + conv1_2 +3.8 %
    WINO_MIN_HW = 8               # maps below this (heads c_9 .. c_11: 5x5, 3x3, 1x1) take the direct kernel: one launch instead of five
    WINO_MIN_CI = 64              # measured in the step with F(4x4): 256 -> 836, 128 -> 872, 64 -> 879 images/s (F(2x2): only >= 256 paid)

    def _wino_ok(self, g) -> bool:
        # dilation d = padding (fc6, Model.py:149): d x d plain 3x3 convolutions on the sub-lattices of the map, F(4x4) tiles on each
        dil_ok = g.dil == 1 or (self.wino_dilated and self.WINO_TILE == 4 and 2 <= g.dil <= 4)
        return (self.wino and not self.bf16 and not self.x3 and g.R == 3 and g.S == 3 and g.stride == 1 and dil_ok and g.pad == g.dil
                and g.Ci % 32 == 0 and g.Ci >= self.WINO_MIN_CI and g.H >= self.WINO_MIN_HW)

    X31_MIN_PIXELS = 8192         # 1x1 layers on the limb GEMMs from this many output pixels (fc7, seq8.0 at batch 32: 11 552; below, the
                                  # 128 x 128 tiles leave most CUs idle and the split-K f32 kernel wins)

    def _x31_ok(self, g) -> bool:
        """f32 mode: this 1x1 / stride-1 convolution runs the three-limb GEMM kernels (forward, data gradient, weight gradient)."""
        return (self.wino and not self.bf16 and not self.x3 and ops.wino_x3(4, 256) and g.R == 1 and g.S == 1 and g.stride == 1 and g.pad == 0
                and g.Ci % 32 == 0 and g.Ci >= 256 and g.Co % 32 == 0 and g.Co >= 128 and g.N * g.H * g.W >= self.X31_MIN_PIXELS)

    def _x31_weights(self, key: str, tensors, co_pad: int):
        """Cached limb planes of a 1x1 filter and of its transpose, refreshed when the parameter changes."""
        sig = tuple((t.data_ptr(), t._version) for t in tensors)
        ent = self._wcache.get("x31:" + key)
        if ent is None or ent[0] != sig:
            wf, wb = ops.conv1x1_weights_x3(tensors[0].detach().contiguous(), co_pad)
            ent = (sig, wf, wb)
            self._wcache["x31:" + key] = ent
        return ent[1], ent[2]

    def _wino_wgrad_ok(self, g, head: bool) -> bool:
        """Weight gradient of this layer in the Winograd domain (else: the fused direct kernels)."""
        return (self._wino_ok(g) and g.H <= self.WINO_WGRAD_MAX_HW and g.Ci >= self.WINO_WGRAD_MIN_CI
                and (g.H >= 19 if head else g.Co % 4 == 0))

    def _adj_ok(self, g) -> bool:
        """This convolution's data gradient takes the adjoint Winograd form (needs the kept forward planes: its weight gradient is the
        Winograd one and shares the dy planes; reduction long enough for the plane GEMM kernels, i.e. not the fused K <= 128 kernel)."""
        return (self.adjoint_dgrad and self.keep_planes and self.dual_dy and not self.overlap_wgrad and self.WINO_TILE == 4 and self._wino_ok(g) and g.dil == 1
                and self._wino_wgrad_ok(g, False) and g.Co % 32 == 0 and 256 <= g.Co <= 1024 and g.Ci % 4 == 0)

    def _first_wino_ok(self, op, bs, x) -> bool:
        """conv1_1's output goes straight into the input planes of the layer behind it: that layer is its only reader and an F(4x4) one."""
        if not (self.first_wino and ops.has_experimental() and self.wino and self.WINO_TILE == 4 and self.consumers.get(op["y"], 0) == 1):
            return False
        nxt = next((o for o in self.ops if o["op"] == "conv" and o["x"] == op["y"]), None)
        if nxt is None or nxt["ci"] != 64:
            return False
        g = ops.make_geom(bs, x.shape[2], x.shape[3], nxt["ci"], nxt["co"], nxt["k"], nxt["s"], nxt["pad"], nxt["dil"])
        return self._wino_ok(g) and g.dil == 1 and g.Co % 4 == 0 and not ops.wino_uses_full(g, 0)

    def _wino_weights(self, key: str, tensors, co_pad: int, adj: bool = False):
        """Cached Winograd-domain filters (U_fwd [16][Co][Ci], U_bwd [16][Ci][co_pad]), refreshed when a parameter changes.
        adj: U_bwd in the adjoint form (the forward transform laid out [Ci][co_pad]) instead of the rotated filter's transform."""
        sig = tuple((t.data_ptr(), t._version) for t in tensors) + (ops.wino_x3(4, 256), bool(adj))     # + the GEMM form the filters were laid out for
        ent = self._wcache.get("wino:" + key)
        if ent is None or ent[0] != sig:
            w = tensors[0] if len(tensors) == 1 else torch.cat(list(tensors), 0)
            uf, ub = ops.wino_weights(w.detach().contiguous(), co_pad, want_bwd=not adj, mo=self.WINO_TILE)
            if adj:
                ub = ops.wino_adj_weights(w.detach().contiguous(), co_pad)
            ent = (sig, uf, ub)
            self._wcache["wino:" + key] = ent
        return ent[1], ent[2]

    def _planes(self, key: str, bwd: bool):
        """bf16 mode: the limb planes of a cached layout (plane 0 feeds the halo-tile kernel); None otherwise."""
        if not self.bf16:
            return None
        ent = self._wcache.get(key)
        return None if ent is None or len(ent) < 6 else ent[5 if bwd else 4]

    def _t16(self, g, xin) -> bool:
        """bf16-tensor mode: this convolution runs csrc/conv_bf16.hip (3x3 / stride 1 / pad 1 on a bf16 input, Ci a multiple of 64)."""
        return (self.bf16 and self.bf16_tensors and g.R == 3 and g.S == 3 and g.stride == 1 and g.dil == 1 and g.pad == 1 and g.Ci % 64 == 0
                and (xin is None or xin.dtype == torch.bfloat16))

    def _head16(self, op, g) -> bool:
        """bf16-tensor mode: this head, although its input is an f32 tensor, runs csrc/conv_bf16.hip on a bf16 copy of it -- where the map is
        large enough to pay (19x19) and the head is the FIRST to deliver its input's gradient (so that gradient needs neither += nor a
        ReLU mask from the kernel, which writes it as f32)."""
        if not (self._t16(g, None) and g.H >= 16):
            return False
        readers = [o for o in self.ops if o["x"] == op["x"]]
        return readers[-1] is op

    def _bf16_weights(self, key: str, tensors):
        """bf16 OHWI (co, 9, ci) and IHWO (ci, 9, pad64(co)) copies of a 3x3 filter (+ the bias padded to a multiple of 4), cached."""
        sig = tuple((t.data_ptr(), t._version) for t in tensors)
        ent = self._wcache.get("b16:" + key)
        if ent is None or ent[0] != sig:
            w = (tensors[0] if len(tensors) == 1 else torch.cat(list(tensors), 0)).detach()
            co, ci = int(w.shape[0]), int(w.shape[1])
            wf = w.permute(0, 2, 3, 1).reshape(co, 9, ci).contiguous().to(torch.bfloat16)
            wb = torch.zeros((ci, 9, ops.pad64(co)), device=w.device, dtype=torch.bfloat16)
            wb[:, :, :co] = w.permute(1, 2, 3, 0).reshape(ci, 9, co)
            ent = (sig, wf, wb)
            self._wcache["b16:" + key] = ent
        return ent[1], ent[2]

    def _prepare_weights_batched(self, x: torch.Tensor, P: Dict[str, torch.Tensor]) -> None:
        """Fill the weight cache for a training step with one launch.  The output buffers persist across steps (rewritten in place, on the
        caller's stream, after the previous step's last use); the job table is rebuilt only when a parameter's storage or the input
        size changes."""
        sig = (x.shape[2], x.shape[3], self.wino, self.WINO_TILE, self.WINO_MIN_CI, self.WINO_MIN_HW, self.bf16, self.bf16_tensors,
               ops.wino_x3(4, 256), self.adjoint_dgrad, self.keep_planes, self.dual_dy, self.overlap_wgrad) + tuple(P[n].data_ptr() for n in self.names)
        if self._wtable is None or self._wtable[0] != sig:
            jobs, entries = [], []
            bs, hw, dev = x.shape[0], {"x": (x.shape[2], x.shape[3])}, x.device
            b16 = {"x": False}                     # which tensors the forward will hold in bf16 (mirrors the dtype flow of `forward`)
            for op in self.ops:
                kind = op["op"]
                if kind in ("pool", "l2norm"):
                    b16[op["y"]] = b16[op["x"]]
                if kind == "conv_first":
                    b16[op["y"]] = self.bf16 and self.bf16_tensors
                    hw[op["y"]] = hw[op["x"]]
                    w = P[op["p"] + ".weight"]
                    rows = torch.empty((64, 1, 32), device=dev, dtype=torch.float32)
                    jobs.append(dict(kind=2, w0=w.detach(), co0=64, co=64, ci=3, taps=9, co_pad=64, out_fwd=rows))
                    entries.append((op["p"], "first", (w,), (rows,)))
                elif kind == "pool":
                    h, w_ = hw[op["x"]]
                    hw[op["y"]] = (ops.pool_out(h, op["k"], op["s"], op["pad"], op["ceil"]), ops.pool_out(w_, op["k"], op["s"], op["pad"], op["ceil"]))
                elif kind == "l2norm":
                    hw[op["y"]] = hw[op["x"]]
                elif kind in ("conv", "head"):
                    h, w_ = hw[op["x"]]
                    if kind == "conv":
                        g = ops.make_geom(bs, h, w_, op["ci"], op["co"], op["k"], op["s"], op["pad"], op["dil"])
                        hw[op["y"]] = (g.Ho, g.Wo)
                        tensors, co_pad = (P[op["p"] + ".weight"],), op["co"]
                    else:
                        co = op["a"] * (4 + N_CLASSES)
                        g = ops.make_geom(bs, h, w_, op["ci"], co, 3, 1, 1, 1)
                        tensors, co_pad = (P[op["p"] + "_bb.weight"], P[op["p"] + "_cl.weight"]), ops.pad32(co)
                    t16 = self._t16(g, None) and (b16[op["x"]] or (kind == "head" and self._head16(op, g)))
                    if kind == "conv":
                        b16[op["y"]] = t16
                    co_all = sum(t.shape[0] for t in tensors)
                    job = dict(w0=tensors[0].detach(), w1=tensors[1].detach() if len(tensors) > 1 else None, co0=tensors[0].shape[0], co=co_all,
                               ci=g.Ci, taps=g.R * g.S, co_pad=co_pad)
                    # bf16-tensor mode: which layers see a bf16 input is decided by the trunk's structure (everything up to pool5, the c_4 head)
                    if t16:
                        wf = torch.empty((co_all, 9, g.Ci), device=dev, dtype=torch.bfloat16)
                        wb = torch.empty((g.Ci, 9, ops.pad64(co_all)), device=dev, dtype=torch.bfloat16)
                        jobs.append(dict(job, kind=3, co_pad=co_all, pad1=ops.pad64(co_all), out_fwd=wf, out_bwd=wb))
                        entries.append((op["p"], "b16", tensors, (wf, wb)))
                    elif kind == "conv" and self._x31_ok(g):
                        wf = ops.x3_filter_alloc(co_all, g.Ci, dev)
                        wb = ops.x3_filter_alloc(g.Ci, co_pad, dev)
                        jobs.append(dict(job, kind=4, out_fwd=wf, out_bwd=wb))
                        entries.append((op["p"], "x31", tensors, (wf, wb)))
                    elif self._wino_ok(g) and self.WINO_TILE == 4:
                        uf = ops.wino_filter_alloc(4, co_all, g.Ci, dev)
                        ub = ops.wino_filter_alloc(4, g.Ci, co_pad, dev)
                        adj = kind == "conv" and self._adj_ok(g)
                        jobs.append(dict(job, kind=0, out_fwd=uf, out_bwd=ub, adj=adj))
                        entries.append((op["p"], "wino_adj" if adj else "wino", tensors, (uf, ub)))
                    elif self._wino_ok(g):
                        continue                               # F(2x2) (a tuning aid): transformed per layer as before
                    else:
                        wf = torch.empty((co_pad, g.R * g.S, g.Ci), device=dev, dtype=torch.float32)
                        wb = torch.empty((g.Ci, g.R * g.S, co_pad), device=dev, dtype=torch.float32)
                        jobs.append(dict(job, kind=1, out_fwd=wf, out_bwd=wb))
                        entries.append((op["p"], "layout", tensors, (wf, wb)))
            self._wtable = (sig, ops.WeightTable(jobs, dev), entries)
        _, table, entries = self._wtable
        table.run()
        for key, what, tensors, bufs in entries:
            lsig = tuple((t.data_ptr(), t._version) for t in tensors)
            if what == "b16":
                self._wcache["b16:" + key] = (lsig, bufs[0], bufs[1])
            elif what in ("wino", "wino_adj"):
                self._wcache["wino:" + key] = (lsig + (ops.wino_x3(4, 256), what == "wino_adj"), bufs[0], bufs[1])
            elif what == "x31":
                self._wcache["x31:" + key] = (lsig, bufs[0], bufs[1])
            elif what == "layout":
                self._wcache[key] = [lsig, None, bufs[0], bufs[1]]
            else:
                self._wcache[key] = [lsig[0], None, bufs[0], None]

    # -- forward ----------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, P: Dict[str, torch.Tensor], save: bool):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"SSD_300 expects (bs,3,H,W) NCHW input, got {tuple(x.shape)}")
        if x.dtype != torch.float32:
            raise ValueError("SSD_300 expects float32 input")
        if not x.is_cuda:
            raise RuntimeError("SSD_300 runs on the gfx950 HIP kernels only: move the model and the input to the GPU "
                               "(there is no CPU fallback)")
        x = x.contiguous()
        bs = x.shape[0]
        if save:
            # A training forward re-lays every weight: the parameters change each step anyway, and the cache key
            # (data_ptr, _version) cannot see writes through `.data` (p.data.mul_(), dist.broadcast(p.data), EMA swaps).
            # The backward of this step reads what this forward stored.
            self._wcache.clear()
            if self.batch_weights and not self.x3 and (not self.bf16 or self.bf16_tensors):
                self._prepare_weights_batched(x, P)
        T = {"x": x}
        aux = {}
        heads = []
        main = torch.cuda.current_stream(x.device)
        overlap = self.overlap_tail and bool(self._side_ids)
        if overlap and self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=x.device)
        side, fork = self._side_stream, None
        side_ctx = [None]              # the stream context of the side group, left in `finally`: an exception must not leave the side stream current
        try:
            self._forward_ops(x, P, save, T, aux, heads, main, side, side_ctx, overlap, bs)
        finally:
            if side_ctx[0] is not None:
                side_ctx[0].__exit__(None, None, None)
        if side_ctx[0] is not None:
            for op, packed, _ in heads:
                if id(op) in self._side_ids:
                    packed.record_stream(main)
            main.wait_event(side.record_event())
        heads.sort(key=lambda h: h[0]["scale"])                   # prior order: c_4, c_7, c_8, ... (Model.py:235)
        P_total = sum(g.Ho * g.Wo * op["a"] for op, _, g in heads)
        loc = torch.empty((bs, P_total, 4), device=x.device, dtype=torch.float32)
        conf = torch.empty((bs, P_total, N_CLASSES), device=x.device, dtype=torch.float32)
        off = 0
        offs = {}
        for op, packed, g in heads:
            ops.heads_scatter(packed, ops.pad32(g.Co), loc, conf, bs, g.Ho * g.Wo, op["a"], off)
            offs[op["p"]] = (off, g)
            off += g.Ho * g.Wo * op["a"]
        saved = dict(T=T, aux=aux, offs=offs, bs=bs) if save else None
        return loc, conf, saved

    def _forward_ops(self, x, P, save, T, aux, heads, main, side, side_ctx, overlap, bs):
        fork = None
        for op in self.ops:
            kind = op["op"]
            if overlap and id(op) == self._late_first:
                fork = main.record_event()                        # a8 (and everything before it) is enqueued
            if overlap and side_ctx[0] is None and id(op) in self._side_ids:
                side.wait_event(fork)
                T[op["x"]].record_stream(side)
                side_ctx[0] = torch.cuda.stream(side)
                side_ctx[0].__enter__()
            if kind == "conv_first":
                g = ops.make_geom(bs, x.shape[2], x.shape[3], 32, 64, 1, 1, 0, 1)
                wkey = P[op["p"] + ".weight"]
                sig = (wkey.data_ptr(), wkey._version)
                ent = self._wcache.get(op["p"])
                if ent is None or ent[0] != sig:
                    ent = [sig, None, ops.first_weight_rows(wkey), None]
                    self._wcache[op["p"]] = ent
                bias = P[op["p"] + ".bias"].detach()
                flops1 = 2.0 * bs * g.Ho * g.Wo * 64 * 27
                if self.bf16 and self.bf16_tensors:
                    # bf16-tensor mode: operands rounded to bf16 inside the kernel, bf16 NHWC out; the weight gradient reads x itself
                    T[op["y"]] = self._timed("fwd " + op["p"], "conv_first_fwd_kernel", flops1,
                                             lambda: ops.conv1_first_fwd_bf16(x, ent[2], bias, True))
                    col = None
                elif (self.first_fused and not self.bf16 and not self.x3 and self._first_wino_ok(op, bs, x)
                      and (not save or (self.relu_bits and self.dual_dy and self.keep_planes))):
                    # ... and, where its only reader is a Winograd layer, straight into that layer's input planes: no activation tensor at all
                    pre = self._timed("fwd " + op["p"], "conv_first_fwd_kernel", flops1,
                                      lambda: ops.conv1_first_wino_fwd(x, ent[2], bias, want_bits=save and self.relu_bits))
                    T[op["y"]] = _Elided((bs, x.shape[2], x.shape[3], 64))
                    aux["preplanes:" + op["y"]] = pre
                    col = None
                elif self.first_fused and not self.bf16 and not self.x3:
                    # one kernel from the NCHW batch: halo tile in LDS, K = 27 on the MFMA, bias + ReLU; the weight gradient's rows ride along
                    T[op["y"]], col = self._timed("fwd " + op["p"], "conv_first_fwd_kernel", flops1,
                                                  lambda: ops.conv1_first_fwd(x, ent[2], bias, True, want_col=False))
                else:
                    # conv1_1 as im2col (K = 27 -> 32) + the 1x1 MFMA convolution
                    col = ops.im2col_first(x)
                    T[op["y"]] = self._timed("fwd " + op["p"], ops.igemm_tile(g, 0, self.bf16, self.x3) if self.prof is not None else "", flops1,
                                             lambda: ops.conv2d_fwd(col, ent[2], bias, g, True, bf16=self.bf16))
                T["x_col"] = col
                aux[op["y"]] = g
            elif kind == "conv":
                xin = T[op["x"]]
                g = ops.make_geom(bs, xin.shape[1], xin.shape[2], op["ci"], op["co"], op["k"], op["s"], op["pad"], op["dil"])
                bias = P[op["p"] + ".bias"].detach()
                if self._t16(g, xin):
                    wf16, _ = self._bf16_weights(op["p"], (P[op["p"] + ".weight"],))
                    T[op["y"]] = self._timed("fwd " + op["p"], "conv3x3_bf16_kernel", ops.conv_flops(g),
                                             lambda: ops.conv3x3_bf16(xin, wf16, bias, op["co"], op["relu"]))
                    aux[op["y"]] = g
                    continue
                if xin.dtype == torch.bfloat16:                   # boundary of the bf16 trunk (pool5 -> fc6): the kernels below take f32 tensors
                    xin = T[op["x"] + ":f32"] = ops.cast_f32(xin)
                if self._wino_ok(g):
                    uf, _ = self._wino_weights(op["p"], (P[op["p"] + ".weight"],), op["co"], adj=save and self._adj_ok(g))
                    pl = self.pool_after.get(op["y"]) if (self.fuse_pool and self.WINO_TILE == 4) else None
                    # training: the transformed input stays for the weight gradient, which multiplies the same planes
                    keep = save and self.keep_planes and self.WINO_TILE == 4 and self._wino_wgrad_ok(g, False)
                    # the mask is only ever applied to a post-ReLU input (deliver() below): pool outputs and the image are not gated here
                    wb = keep and self.relu_bits and self.dual_dy and op["x"] in self.relu_out and g.Co % 32 == 0
                    pre = aux.pop("preplanes:" + op["x"], None)
                    if pre is not None:
                        # the producer of this layer's input left the input planes (and the ReLU bits): GEMMs + output transform only
                        planes, pbits = pre
                        if pl is not None:
                            yp, am = self._timed("fwd " + op["p"], "winograd_3x3", ops.wino_flops(g),
                                                 lambda: ops.conv2d_fwd_wino_from_planes(planes, uf, bias, g, True, pool_ceil=pl["ceil"], want_argmax=save))
                            T[op["y"]] = _Elided((bs, g.H, g.W, op["co"]))
                            T[pl["y"]], aux[pl["y"]] = yp, am
                        else:
                            T[op["y"]] = self._timed("fwd " + op["p"], "winograd_3x3", ops.wino_flops(g),
                                                     lambda: ops.conv2d_fwd_wino_from_planes(planes, uf, bias, g, op["relu"]))
                        aux[op["y"]] = g
                        if keep:
                            aux["planes:" + op["p"]] = planes
                        if wb and pbits is not None:
                            aux["bits:" + op["p"]] = pbits
                        continue
                    if pl is not None:
                        res = self._timed("fwd " + op["p"], "winograd_3x3", ops.wino_flops(g),
                                          lambda: ops.conv2d_fwd_wino_pool(xin, uf, bias, g, pl["ceil"], want_argmax=save, keep_planes=keep,
                                                                           want_bits=wb))
                        T[op["y"]] = _Elided((bs, g.H, g.W, op["co"]))      # never materialised: its only reader is the pool
                        T[pl["y"]], aux[pl["y"]], aux[op["y"]] = res[0], res[1], g
                        if keep:
                            aux["planes:" + op["p"]] = res[2]
                        if wb:
                            aux["bits:" + op["p"]] = res[3]
                        continue
                    res = self._timed("fwd " + op["p"], "winograd_3x3", ops.wino_flops(g),
                                      lambda: ops.conv2d_fwd_wino(xin, uf, bias, g, op["relu"], keep_planes=keep, want_bits=wb))
                    T[op["y"]] = res[0] if keep else res
                    if keep:
                        aux["planes:" + op["p"]] = res[1]
                    if wb:
                        aux["bits:" + op["p"]] = res[2]
                    aux[op["y"]] = g
                    continue
                if self._x31_ok(g):
                    w3f, _ = self._x31_weights(op["p"], (P[op["p"] + ".weight"],), op["co"])
                    T[op["y"]] = self._timed("fwd " + op["p"], "gemm_planes_x3_kernel (1x1)", ops.conv_flops(g),
                                             lambda: ops.conv1x1_fwd_x3(xin, w3f, bias, g, op["relu"]))
                    aux[op["y"]] = g
                    continue
                wf, _ = self._layouts(op["p"], (P[op["p"] + ".weight"],), op["co"], False)
                T[op["y"]] = self._timed("fwd " + op["p"], ops.igemm_tile(g, 0, self.bf16, self.x3) if self.prof is not None else "", ops.conv_flops(g),
                                         lambda: ops.conv2d_fwd_x3(xin, wf, bias, g, op["relu"]) if self.x3 else
                                         ops.conv2d_fwd(xin, wf, bias, g, op["relu"], bf16=self.bf16, w3=self._planes(op["p"], False)))
                aux[op["y"]] = g
            elif kind == "pool":
                if op["y"] in T:                                  # produced by the convolution before it
                    continue
                y, am = ops.maxpool_fwd(T[op["x"]], op["k"], op["s"], op["pad"], op["ceil"], want_argmax=save)
                T[op["y"]] = y
                aux[op["y"]] = am
            elif kind == "l2norm":
                T[op["y"]] = ops.l2norm_fwd(T[op["x"]], P[op["p"]].detach().reshape(-1))
            elif kind == "head":
                xin = T[op["x"]]
                a = op["a"]
                co = a * (4 + N_CLASSES)
                g = ops.make_geom(bs, xin.shape[1], xin.shape[2], op["ci"], co, 3, 1, 1, 1)
                pre = op["p"]
                bias = _cat_flat(P[pre + "_bb.bias"].detach(), P[pre + "_cl.bias"].detach())
                if self._head16(op, g) and xin.dtype == torch.float32:
                    # a head on an f32 tensor outside the bf16 trunk (c_7 on fc7's output): one cast, then the LDS-DMA kernel
                    xin = T[op["x"] + ":b16"] = ops.cast_bf16(xin)
                if self._t16(g, xin):
                    wf16, _ = self._bf16_weights(pre, (P[pre + "_bb.weight"], P[pre + "_cl.weight"]))
                    ld, co4 = ops.pad32(co), (co + 3) // 4 * 4
                    out = torch.empty((bs, g.H, g.W, ld), device=xin.device, dtype=torch.float32)      # pad columns are never read
                    packed = self._timed("fwd " + pre, "conv3x3_bf16_kernel", ops.conv_flops(g),
                                         lambda: ops.conv3x3_bf16(xin, wf16, bias, co4, False, out=out, out_f32=True, ldo=ld))
                    heads.append((op, packed, g))
                    continue
                if xin.dtype == torch.bfloat16:
                    xin = T[op["x"] + ":f32"] = T.get(op["x"] + ":f32") if T.get(op["x"] + ":f32") is not None else ops.cast_f32(xin)
                if self._wino_ok(g):
                    uf, _ = self._wino_weights(pre, (P[pre + "_bb.weight"], P[pre + "_cl.weight"]), ops.pad32(co))
                    keep = save and self.keep_planes and self.WINO_TILE == 4 and self._wino_wgrad_ok(g, True)
                    res = self._timed("fwd " + pre, "winograd_3x3", ops.wino_flops(g),
                                      lambda: ops.conv2d_fwd_wino(xin, uf, bias, g, False, ld=ops.pad32(co), keep_planes=keep))
                    if keep:
                        aux["planes:" + pre] = res[1]
                    heads.append((op, res[0] if keep else res, g))
                    continue
                wf, _ = self._layouts(pre, (P[pre + "_bb.weight"], P[pre + "_cl.weight"]), ops.pad32(co), False)
                packed = self._timed("fwd " + pre, ops.igemm_tile(g, 0, self.bf16, self.x3) if self.prof is not None else "", ops.conv_flops(g),
                                     lambda: ops.conv2d_fwd_x3(xin, wf, bias, g, False, ld=ops.pad32(co)) if self.x3 else
                                     ops.conv2d_fwd(xin, wf, bias, g, False, ld=ops.pad32(co), bf16=self.bf16, w3=self._planes(pre, False)))
                heads.append((op, packed, g))

    def _wgrad_gemm_async(self, main, Y, kept, part, g, ldy, assign):
        """Second half of a Winograd weight gradient on the weight-gradient stream: waits for what `main` has enqueued so far (the dy
        transform), multiplies, and hands (dw, db) to `assign` inside the stream context."""
        ws = self._wgrad_stream
        ws.wait_event(main.record_event())
        for t in (Y, kept, part):
            t.record_stream(ws)
        with torch.cuda.stream(ws):
            dw, db = ops.wino_wgrad_gemm(Y, kept, part, g, ldy)
            dw.record_stream(main)
            db.record_stream(main)
            assign(dw, db)

    # -- backward ---------------------------------------------------------------------------
    def backward(self, saved, dloc: torch.Tensor, dconf: torch.Tensor, P: Dict[str, torch.Tensor], need: Dict[str, bool]):
        dloc = dloc.contiguous()
        dconf = dconf.contiguous()
        G: Dict[str, torch.Tensor] = {}
        arrived: Dict[str, int] = {}
        grads: Dict[str, torch.Tensor] = _SinkDict(self.grad_sink) if self.grad_sink is not None else {}
        side_ctx = [None]              # left in `finally`: an exception must not leave the side stream current
        try:
            return self._backward_ops(saved, dloc, dconf, P, need, G, arrived, grads, side_ctx)
        finally:
            if side_ctx[0] is not None:
                side_ctx[0].__exit__(None, None, None)

    def _gout(self, *names):
        """Where the gradient of these consecutive parameters is to be written (a flat view of the listener's buffer), or None."""
        return None if self.grad_out is None else self.grad_out(names)

    def _backward_ops(self, saved, dloc, dconf, P, need, G, arrived, grads, side_ctx):
        T, aux, offs, bs = saved["T"], saved["aux"], saved["offs"], saved["bs"]

        def deliver(name, fn):
            """fn(dx, accumulate, mask) -> dx; mask only when this is the last contribution to a post-ReLU tensor."""
            k = arrived.get(name, 0)
            last = k + 1 == self.consumers[name]
            mask = T[name] if (last and name in self.relu_out) else None
            G[name] = fn(G.get(name), k > 0, mask)
            arrived[name] = k + 1
            if self.grad_tap is not None and torch.is_tensor(G[name]):
                if last:
                    self.grad_tap[name] = G[name]
                else:                                        # a partial sum, as stored before the next reader adds to it in place
                    self.grad_tap[f"{name}:{k + 1}"] = G[name].clone()

        main = torch.cuda.current_stream(dloc.device)
        # (with a gradient listener every gradient is reported on the caller's stream: no third stream then)
        async_wgrad = self.overlap_wgrad and self.prof is None and self.dual_dy and self.WINO_TILE == 4 and self.grad_sink is None
        if async_wgrad and self._wgrad_stream is None:
            self._wgrad_stream = torch.cuda.Stream(device=dloc.device)
        wgrad_used = False
        overlap = self.overlap_tail and bool(self._side_ids) and self._side_stream is not None
        side, joined = self._side_stream, not overlap
        held = isinstance(grads, _SinkDict)
        if overlap:                                               # the reversed list starts with the side group
            side.wait_event(main.record_event())
            dloc.record_stream(side)
            dconf.record_stream(side)
            side_ctx[0] = torch.cuda.stream(side)
            side_ctx[0].__enter__()
            if held:
                grads.hold()                                      # the listener hears of side-stream gradients once the caller's stream has waited for them
        join_event = side_done = None
        # The side group's weight gradients are not on the path to a8's gradient (the only thing the caller's stream needs from the group):
        # they are enqueued AFTER the group's data-gradient chain, behind the join event.
        deferred = []
        defer = self.defer_tail_wgrad and self.prof is None
        for op in reversed(self.ops):
            kind = op["op"]
            if side_ctx[0] is not None and id(op) not in self._side_ids:   # the side group is enqueued: back to the caller's stream
                join_event = side.record_event()                           # every data gradient the caller's stream will read exists
                for t in G.values():
                    t.record_stream(main)
                if self._test_side_delay is not None:
                    self._test_side_delay()
                for wg in deferred:                                        # the side group's weight gradients: nothing waits for them
                    wg()
                deferred.clear()
                side_ctx[0].__exit__(None, None, None)
                side_ctx[0] = None
                side_done = side.record_event()
                for t in grads.values():
                    t.record_stream(main)
                if held and self.sink_early:
                    # the listener starts a collective the moment a bucket is complete: order the side stream's gradients before the
                    # caller's stream first (a listener that only collects hears of them at the end of the backward, below)
                    main.wait_event(side_done)
                    side_done = None
                    grads.release()
            if not joined and side_ctx[0] is None and kind in ("conv", "pool", "conv_first"):      # first op that needs what the side group produced (a8's gradient)
                main.wait_event(join_event)
                joined = True
            if kind == "head":
                pre = op["p"]
                off, g = offs[pre]
                co_pad = ops.pad32(g.Co)
                xin = T[op["x"]]
                a4 = 4 * op["a"]
                x16 = T.get(op["x"] + ":b16")
                if x16 is not None or self._t16(g, xin):
                    # bf16 trunk: packed bf16 dy (K of the data gradient padded to 64), nine-tap bf16 weight gradient, LDS-DMA data gradient
                    f32_dx = x16 is not None                  # the head ran on a bf16 copy of an f32 tensor: its dx goes back as f32
                    if f32_dx:
                        xin = x16
                    ld = ops.pad64(g.Co)
                    dy = ops.heads_gather_bf16(dloc, dconf, ld, bs, g.Ho * g.Wo, op["a"], off).view(bs, g.Ho, g.Wo, ld)
                    if any(need[pre + s] for s in ("_bb.weight", "_bb.bias", "_cl.weight", "_cl.bias")):
                        dw, db = self._timed("wgrad " + pre, "wgrad3x3_bf16_kernel", ops.conv_flops(g),
                                             lambda: ops.conv3x3_wgrad_bf16t(xin, dy, g, ld, True, dw_out=self._gout(pre + "_bb.weight", pre + "_cl.weight"),
                                                                             db_out=self._gout(pre + "_bb.bias", pre + "_cl.bias")))
                        grads[pre + "_bb.weight"], grads[pre + "_cl.weight"] = dw[:a4], dw[a4:]
                        grads[pre + "_bb.bias"], grads[pre + "_cl.bias"] = db[:a4], db[a4:]
                    _, wb16 = self._bf16_weights(pre, (P[pre + "_bb.weight"], P[pre + "_cl.weight"]))
                    def dgrad16(dx, acc, mask, dy=dy, wb16=wb16, g=g, pre=pre, f32_dx=f32_dx):
                        if f32_dx and (acc or mask is not None or dx is not None):
                            raise RuntimeError("a head on a bf16 copy must be the first to deliver its input's gradient")
                        return self._timed("dgrad " + pre, "conv3x3_bf16_kernel", ops.conv_flops(g),
                                           lambda: ops.conv3x3_bf16(dy, wb16, None, g.Ci, False, flip=True, out=dx, out_f32=f32_dx, relu_mask=mask,
                                                                    accumulate=acc))
                    deliver(op["x"], dgrad16)
                    continue
                if xin.dtype == torch.bfloat16:
                    xin = T[op["x"] + ":f32"]
                dy = ops.heads_gather(dloc, dconf, co_pad, bs, g.Ho * g.Wo, op["a"], off)
                dyp = None
                dw = db = None
                if any(need[pre + s] for s in ("_bb.weight", "_bb.bias", "_cl.weight", "_cl.bias")):
                    if self._wino_wgrad_ok(g, True) and async_wgrad and side_ctx[0] is None and aux.get("planes:" + pre) is not None \
                            and co_pad <= 1024:
                        kept = aux.pop("planes:" + pre)
                        Y, dyp, part = ops.wino_dy_transform(dy, g, co_pad, True, True)

                        def put(dw, db, pre=pre, a4=a4):
                            grads[pre + "_bb.weight"], grads[pre + "_cl.weight"] = dw[:a4], dw[a4:]
                            grads[pre + "_bb.bias"], grads[pre + "_cl.bias"] = db[:a4], db[a4:]
                        self._wgrad_gemm_async(main, Y, kept, part, g, co_pad, put)
                        wgrad_used = True
                    elif self._wino_wgrad_ok(g, True):
                        kept = aux.pop("planes:" + pre, None)
                        dual = kept is not None and self.dual_dy        # one pass over dy feeds the weight and the data gradient
                        res = self._timed("wgrad " + pre, "winograd_3x3", ops.wino_flops(g),
                                          lambda: ops.conv2d_wgrad_wino(xin, dy, g, co_pad, True, mo=self.WINO_TILE, planes=kept,
                                                                        dgrad_planes=dual, dw_out=self._gout(pre + "_bb.weight", pre + "_cl.weight"),
                                                                        db_out=self._gout(pre + "_bb.bias", pre + "_cl.bias")))
                        dw, db = res[0], res[1]
                        dyp = res[2] if dual else None
                    else:
                        def wg(pre=pre, a4=a4, xin=xin, dy=dy, g=g, co_pad=co_pad):
                            dw_, db_ = self._timed("wgrad " + pre, ops.wgrad_tile(g, self.bf16) if self.prof is not None else "", ops.conv_flops(g),
                                                   lambda: ops.conv2d_wgrad(xin, dy, g, co_pad, True, bf16=self.bf16,
                                                                            dw_out=self._gout(pre + "_bb.weight", pre + "_cl.weight"),
                                                                            db_out=self._gout(pre + "_bb.bias", pre + "_cl.bias")))
                            grads[pre + "_bb.weight"], grads[pre + "_cl.weight"] = dw_[:a4], dw_[a4:]
                            grads[pre + "_bb.bias"], grads[pre + "_cl.bias"] = db_[:a4], db_[a4:]
                        if side_ctx[0] is not None and defer:
                            deferred.append(wg)
                        else:
                            wg()
                    if dw is not None:
                        grads[pre + "_bb.weight"], grads[pre + "_cl.weight"] = dw[:a4], dw[a4:]
                        grads[pre + "_bb.bias"], grads[pre + "_cl.bias"] = db[:a4], db[a4:]
                if self._wino_ok(g):
                    _, ub = self._wino_weights(pre, (P[pre + "_bb.weight"], P[pre + "_cl.weight"]), co_pad)
                    deliver(op["x"], lambda dx, acc, mask: self._timed("dgrad " + pre, "winograd_3x3", ops.wino_flops(g),
                                                                       lambda: ops.conv2d_dgrad_wino(dy, ub, g, dx, mask, acc, planes=dyp)))
                    continue
                _, wb = self._layouts(pre, (P[pre + "_bb.weight"], P[pre + "_cl.weight"]), co_pad, True)
                deliver(op["x"], lambda dx, acc, mask: self._timed(
                    "dgrad " + pre, ops.igemm_tile(g, 1, self.bf16, self.x3) if self.prof is not None else "", ops.conv_flops(g),
                    lambda: ops.conv2d_dgrad_x3(dy, wb, g, dx, mask, acc) if self.x3 else
                    ops.conv2d_dgrad(dy, wb, g, dx, mask, acc, bf16=self.bf16, w3=self._planes(pre, True))))
            elif kind == "conv":
                dy = G.pop(op["y"])
                g = aux[op["y"]]
                xin = T[op["x"]]
                dyp = None
                if self._t16(g, xin) and dy.dtype == torch.bfloat16:
                    name = op["p"]
                    if need[name + ".weight"] or need[name + ".bias"]:
                        grads[name + ".weight"], grads[name + ".bias"] = self._timed(
                            "wgrad " + name, "wgrad3x3_bf16_kernel", ops.conv_flops(g),
                            lambda: ops.conv3x3_wgrad_bf16t(xin, dy, g, g.Co, True, dw_out=self._gout(name + ".weight"), db_out=self._gout(name + ".bias")))
                    _, wb16 = self._bf16_weights(name, (P[name + ".weight"],))
                    deliver(op["x"], lambda dx, acc, mask: self._timed(
                        "dgrad " + name, "conv3x3_bf16_kernel", ops.conv_flops(g),
                        lambda: ops.conv3x3_bf16(dy, wb16, None, g.Ci, False, flip=True, out=dx, relu_mask=mask, accumulate=acc)))
                    continue
                to_bf16 = xin.dtype == torch.bfloat16         # boundary of the bf16 trunk: this layer ran on the f32 copy, its dx goes back as bf16
                if to_bf16:
                    xin = T[op["x"] + ":f32"]
                if isinstance(dy, ops.PooledGrad):
                    # dy exists only behind its pool; the Winograd dy pass of this layer can form it on the fly when that pass feeds
                    # both the weight gradient and the data gradient -- otherwise it is scattered to memory after all
                    if not ((need[op["p"] + ".weight"] or need[op["p"] + ".bias"]) and self._wino_wgrad_ok(g, False) and self.WINO_TILE == 4
                            and self.dual_dy and aux.get("planes:" + op["p"]) is not None and g.Co % 32 == 0 and g.Co <= 1024
                            and not ops.wino_uses_full(g, 1) and self._wino_ok(g)):
                        dy = dy.materialize()
                adj = (self._adj_ok(g) and not to_bf16 and aux.get("planes:" + op["p"]) is not None and not (async_wgrad and side_ctx[0] is None)
                       and not ops.wino_uses_full(g, 1))
                if isinstance(dy, ops.AdjPlanesGrad) and not adj:
                    dy = dy.materialize()
                if adj:
                    # Adjoint Winograd form: ONE set of dy planes (A dy A^T) feeds the weight gradient's TN GEMMs and the data gradient's
                    # plane GEMMs; where dy itself is the adjoint data gradient of the layer above (same map, this layer its only reader)
                    # the planes come straight from that layer's product planes and dy is never a tensor.
                    name = op["p"]
                    kept = aux.pop("planes:" + name)
                    need_w = need[name + ".weight"] or need[name + ".bias"]

                    def dy_planes(dy=dy, g=g, need_w=need_w):
                        if isinstance(dy, ops.AdjPlanesGrad):
                            return ops.wino_adj_output_to_planes(dy.md, dy.g, g, dy.relu_mask, dy.bits, want_bias=need_w)
                        Yp, _, part = ops.wino_dy_transform(dy, g, g.Co, False, need_w)
                        return Yp, part
                    if need_w:
                        def wg_adj(name=name, g=g, kept=kept):
                            Yp, part = dy_planes()
                            dw_, db_ = ops.wino_wgrad_gemm(Yp, kept, part, g, g.Co, dw_out=self._gout(name + ".weight"), db_out=self._gout(name + ".bias"))
                            return Yp, dw_, db_
                        Y, dw, db = self._timed("wgrad " + name, "winograd_3x3", ops.wino_flops(g), wg_adj)
                        grads[name + ".weight"], grads[name + ".bias"] = dw, db
                    else:
                        Y, _ = dy_planes()
                    del kept
                    _, uadj = self._wino_weights(name, (P[name + ".weight"],), op["co"], adj=True)
                    bits = aux.pop("bits:" + name, None)
                    below = self._conv_of.get(op["x"]) if self.adjoint_chain else None
                    gb = aux.get(op["x"]) if below is not None else None
                    chain = (below is not None and gb is not None and self.consumers[op["x"]] == 1 and op["x"] in self.relu_out
                             and self._adj_ok(gb) and aux.get("planes:" + below["p"]) is not None and gb.Co == g.Ci
                             and (gb.N, gb.H, gb.W) == (g.N, g.H, g.W) and T[op["x"]].dtype == torch.float32)

                    def dgrad_adj(dx, acc, mask, Y=Y, uadj=uadj, g=g, bits=bits, chain=chain):
                        md = ops.wino_dgrad_adj_gemm(Y, uadj, g, g.Co)
                        use_bits = bits if mask is not None else None
                        fmask = None if use_bits is not None else mask
                        if chain and dx is None and not acc and mask is not None:
                            return ops.AdjPlanesGrad(md, g, fmask, use_bits)         # the layer below reads the planes, not a tensor
                        return ops.wino_adj_output(md, g, dx, relu_mask=fmask, bits=use_bits, accumulate=acc)
                    deliver(op["x"], lambda dx, acc, mask: self._timed("dgrad " + name, "winograd_3x3", ops.wino_flops(g),
                                                                       lambda: dgrad_adj(dx, acc, mask)))
                    continue
                if need[op["p"] + ".weight"] or need[op["p"] + ".bias"]:
                    if self._wino_wgrad_ok(g, False) and async_wgrad and side_ctx[0] is None and aux.get("planes:" + op["p"]) is not None \
                            and g.Co % 32 == 0 and g.Co <= 1024:
                        kept = aux.pop("planes:" + op["p"])
                        # the one-kernel data gradient reads dy itself: no B^T dy B planes to write
                        Y, dyp, part = ops.wino_dy_transform(dy, g, g.Co, not ops.wino_uses_full(g, 1), True)

                        def put(dw, db, name=op["p"]):
                            grads[name + ".weight"], grads[name + ".bias"] = dw, db
                        self._wgrad_gemm_async(main, Y, kept, part, g, g.Co, put)
                        wgrad_used = True
                        dw = None
                    elif self._wino_wgrad_ok(g, False):
                        kept = aux.pop("planes:" + op["p"], None)
                        dual = kept is not None and self.dual_dy and g.Co % 32 == 0 and not ops.wino_uses_full(g, 1)
                        res = self._timed("wgrad " + op["p"], "winograd_3x3", ops.wino_flops(g),
                                          lambda: ops.conv2d_wgrad_wino(xin, dy, g, g.Co, True, mo=self.WINO_TILE, planes=kept,
                                                                        dgrad_planes=dual, dw_out=self._gout(op["p"] + ".weight"),
                                                                        db_out=self._gout(op["p"] + ".bias")))
                        dw, db = res[0], res[1]
                        dyp = res[2] if dual else None
                    else:
                        def wg(name=op["p"], xin=xin, dy=dy, g=g):
                            if self._x31_ok(g):
                                grads[name + ".weight"], grads[name + ".bias"] = self._timed(
                                    "wgrad " + name, "gemm_tn_x3_kernel (1x1)", ops.conv_flops(g),
                                    lambda: ops.conv1x1_wgrad_x3(xin, dy, g, g.Co, True, dw_out=self._gout(name + ".weight"),
                                                                 db_out=self._gout(name + ".bias")))
                                return
                            grads[name + ".weight"], grads[name + ".bias"] = self._timed(
                                "wgrad " + name, ops.wgrad_tile(g, self.bf16) if self.prof is not None else "", ops.conv_flops(g),
                                lambda: ops.conv2d_wgrad(xin, dy, g, g.Co, True, bf16=self.bf16, dw_out=self._gout(name + ".weight"),
                                                         db_out=self._gout(name + ".bias")))
                        if side_ctx[0] is not None and defer:
                            deferred.append(wg)
                        else:
                            wg()
                        dw = None
                    if dw is not None:
                        grads[op["p"] + ".weight"], grads[op["p"] + ".bias"] = dw, db
                if self._wino_ok(g):
                    _, ub = self._wino_weights(op["p"], (P[op["p"] + ".weight"],), op["co"], adj=self._adj_ok(g))
                    if self._adj_ok(g):
                        raise RuntimeError(f"{op['p']}: its filter planes are laid out for the adjoint data gradient, which needs the forward's kept planes "
                                           "(the engine's keep_planes / dual_dy / adjoint_dgrad switches must not change between a forward and its backward)")
                    bits = aux.pop("bits:" + op["p"], None) if (dyp is not None or ops.wino_uses_full(g, 1) or isinstance(T[op["x"]], _Elided)) else None
                    def dgrad_rot(dx, acc, mask, dy=dy, ub=ub, g=g, dyp=dyp, bits=bits, name=op["p"]):
                        if isinstance(mask, _Elided):            # the gated activation was never stored (conv1_1 -> planes): only its bits exist
                            if bits is None:
                                raise RuntimeError(f"{name}: its input was not stored and no ReLU bits were kept for its data gradient")
                            mask_t = None
                        else:
                            mask_t = mask
                        return ops.conv2d_dgrad_wino(None if isinstance(dy, ops.PooledGrad) else dy, ub, g, dx, mask_t, acc, planes=dyp,
                                                     bits=bits if mask is not None else None)
                    deliver(op["x"], lambda dx, acc, mask: self._timed("dgrad " + op["p"], "winograd_3x3", ops.wino_flops(g),
                                                                       lambda: dgrad_rot(dx, acc, mask)))
                    continue
                if self._x31_ok(g) and not to_bf16:
                    _, w3b = self._x31_weights(op["p"], (P[op["p"] + ".weight"],), op["co"])
                    deliver(op["x"], lambda dx, acc, mask: self._timed(
                        "dgrad " + op["p"], "gemm_planes_x3_kernel (1x1)", ops.conv_flops(g),
                        lambda: ops.conv1x1_dgrad_x3(dy, w3b, g, dx, mask, acc)))
                    continue
                _, wb = self._layouts(op["p"], (P[op["p"] + ".weight"],), op["co"], True)
                if to_bf16:
                    def boundary(dx, acc, mask, dy=dy, wb=wb, g=g, name=op["p"]):
                        if acc or mask is not None or dx is not None:
                            raise RuntimeError("the bf16 trunk's boundary tensor must have one consumer and no ReLU of its own")
                        d32 = self._timed("dgrad " + name, ops.igemm_tile(g, 1, self.bf16, self.x3) if self.prof is not None else "", ops.conv_flops(g),
                                          lambda: ops.conv2d_dgrad(dy, wb, g, None, None, False, bf16=self.bf16, w3=self._planes(name, True)))
                        return ops.cast_bf16(d32)
                    deliver(op["x"], boundary)
                    continue
                deliver(op["x"], lambda dx, acc, mask: self._timed(
                    "dgrad " + op["p"], ops.igemm_tile(g, 1, self.bf16, self.x3) if self.prof is not None else "", ops.conv_flops(g),
                    lambda: ops.conv2d_dgrad_x3(dy, wb, g, dx, mask, acc) if self.x3 else
                    ops.conv2d_dgrad(dy, wb, g, dx, mask, acc, bf16=self.bf16, w3=self._planes(op["p"], True))))
            elif kind == "pool":
                dy = G.pop(op["y"])
                xin = T[op["x"]]
                am = aux[op["y"]]
                yout = T[op["y"]]
                deliver(op["x"], lambda dx, acc, mask: (
                    # behind a fused conv -> ReLU -> pool layer: that layer's dy pass reads (dy, argmax, gate) itself
                    ops.PooledGrad(dy, am, yout, xin.shape)
                    if (self.lazy_pool_grad and isinstance(xin, _Elided) and mask is not None and not acc and dx is None
                        and (op["k"], op["s"], op["pad"]) == (2, 2, 0)) else
                    ops.maxpool_bwd(dy, am, tuple(xin.shape), op["k"], op["s"], op["pad"], dx, y_gate=yout)
                    if (mask is not None and not acc) else          # sole consumer of a ReLU output: gate by the pooled output
                    ops.maxpool_bwd(dy, am, tuple(xin.shape), op["k"], op["s"], op["pad"], dx, mask, acc)))
            elif kind == "l2norm":
                dy = G.pop(op["y"])
                xin = T[op["x"]]
                gamma = P[op["p"]].detach().reshape(-1)
                box = {}

                def run(dx, acc, mask, xin=xin, gamma=gamma, dy=dy, box=box):
                    if acc or mask is not None:
                        raise RuntimeError("L2-norm backward must deliver the first, non-final gradient of its input")
                    dx, box["dg"] = ops.l2norm_bwd(xin, gamma, dy, dg_out=self._gout(op["p"]))
                    return dx
                deliver(op["x"], run)
                grads[op["p"]] = box["dg"].reshape(P[op["p"]].shape)
            elif kind == "conv_first":
                dy = G.pop(op["y"])
                if need[op["p"] + ".weight"] or need[op["p"] + ".bias"]:
                    g = aux[op["y"]]
                    col = T["x_col"]
                    if col is None:                                    # the one-kernel forward left no [pixel][32] rows: gradient from x itself
                        dw, db = self._timed("wgrad " + op["p"], "conv_first_wgrad_kernel", 2.0 * dy.numel() * 27,
                                             lambda: (ops.conv1_first_wgrad_bf16 if dy.dtype == torch.bfloat16 else ops.conv1_first_wgrad)(T["x"], dy, True))
                    else:
                        dw, db = self._timed("wgrad " + op["p"], ops.wgrad_tile(g, self.bf16) if self.prof is not None else "", 2.0 * dy.numel() * 27,
                                             lambda: ops.conv2d_wgrad(col, dy, g, g.Co, True))
                    grads[op["p"] + ".weight"], grads[op["p"] + ".bias"] = ops.first_weight_grad(dw), db
        if not joined:
            main.wait_event(join_event)
        if side_done is not None:
            main.wait_event(side_done)
        if wgrad_used:
            main.wait_event(self._wgrad_stream.record_event())
        if held:
            grads.release()
        return grads


class _SSD300Function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, engine, *params):
        P = dict(zip(engine.names, params))
        loc, conf, saved = engine.forward(x, P, save=True)
        ctx.engine = engine
        ctx.saved = saved
        ctx.params = params
        return loc, conf

    @staticmethod
    def backward(ctx, dloc, dconf):
        eng = ctx.engine
        P = dict(zip(eng.names, ctx.params))
        need = {n: bool(f) for n, f in zip(eng.names, ctx.needs_input_grad[2:])}
        grads = eng.backward(ctx.saved, dloc, dconf, P, need)
        ctx.saved = None
        if eng.sink_owns_grads and eng.grad_sink is not None:
            # ddp.py: every gradient already sits in the flat buffer the all-reduce reads (the parameters' .grad are views of it);
            # handing the views to autograd would make it clone 105 MB into fresh .grad tensors
            return (None, None) + (None,) * len(eng.names)
        return (None, None) + tuple(grads.get(n) if need[n] else None for n in eng.names)


class SSD_300(nn.Module):
    """Drop-in for reference Model.py:128-235 (same no-argument constructor, same parameter names)."""
    _VARIANT = 300

    def __init__(self):
        super().__init__()
        self.model = _VGG16()                                              # Model.py:131 (no download here)
        self.rescaling_conv_4_3 = nn.Parameter(torch.full((1, 512, 1, 1), 20.))   # Model.py:132-133
        feats = self.model.features
        self.conv_4_3 = nn.Sequential(*feats[0:16], nn.MaxPool2d(2, 2, 0, 1, ceil_mode=True), *feats[17:23])
        self.seq5 = nn.Sequential(*feats[23:30], nn.MaxPool2d(3, 1, 1, 1, ceil_mode=True))
        # fc6 / fc7 as convolutions by sub-sampling the classifier weights (Model.py:145-161)
        cls0, cls3 = self.model.classifier[0], self.model.classifier[3]
        self.fc6 = subsampling(cls0.weight.detach().view(4096, 512, 7, 7), [4, None, 3, 3])
        self.fc6_b = subsampling(cls0.bias.detach(), [4])
        self.fc7 = subsampling(cls3.weight.detach().view(4096, 4096, 1, 1), [4, 4, None, None])
        self.fc7_b = subsampling(cls3.bias.detach(), [4])
        self.conv_fc6 = nn.Conv2d(512, 1024, 3, padding=4, dilation=4)
        self.conv_fc6.weight = nn.Parameter(self.fc6.clone())
        self.conv_fc6.bias = nn.Parameter(self.fc6_b.clone())
        self.conv_fc7 = nn.Conv2d(1024, 1024, 1)
        self.conv_fc7.weight = nn.Parameter(self.fc7.clone())
        self.conv_fc7.bias = nn.Parameter(self.fc7_b.clone())
        self.seq7 = nn.Sequential(self.conv_fc6, nn.ReLU(), self.conv_fc7, nn.ReLU())
        self.seq8 = nn.Sequential(nn.Conv2d(1024, 256, 1), nn.ReLU(), nn.Conv2d(256, 512, 3, 2, padding=1), nn.ReLU())
        self.seq9 = nn.Sequential(nn.Conv2d(512, 128, 1), nn.ReLU(), nn.Conv2d(128, 256, 3, 2, padding=1), nn.ReLU())
        aux = _AUX_300 if self._VARIANT == 300 else _AUX_512
        for name, cin, mid, cout, stride, pad in aux[2:]:
            setattr(self, name, nn.Sequential(nn.Conv2d(cin, mid, 1), nn.ReLU(), nn.Conv2d(mid, cout, 3, stride, padding=pad), nn.ReLU()))
        anchors = ANCHORS_PER_CELL if self._VARIANT == 300 else ANCHORS_PER_CELL_512
        self._aux_names = tuple(a[0] for a in aux)
        self._head_names = ("c_4", "c_7") + tuple(f"c_{8 + i}" for i in range(len(aux)))
        for name, cin, a in zip(self._head_names, (512, 1024, 512) + (256,) * (len(aux) - 1), anchors):
            setattr(self, name + "_bb", nn.Conv2d(cin, 4 * a, 3, padding=1))
            setattr(self, name + "_cl", nn.Conv2d(cin, N_CLASSES * a, 3, padding=1))
        self.initialization()
        self._engine = _Engine(self._VARIANT)

    @property
    def conv_dtype(self) -> str:
        """"f32" (default: exact-f32 MFMA, the reference's precision) or "bf16" (BASELINE configs[2]: forward and
        data-gradient convolutions on the bf16 MFMA with f32 accumulation; weight gradients, loss and everything
        else stay f32)."""
        return "bf16" if self._engine.bf16 else ("f32x3" if self._engine.x3 else "f32")

    @conv_dtype.setter
    def conv_dtype(self, value: str) -> None:
        """"f32x3": forward / dgrad products formed from three exact bf16 limbs per f32 operand (six bf16 MFMAs per
        block, f32 accumulate): f32-accurate (measured against f64: not worse than the exact-f32 MFMA kernels)."""
        if value not in ("f32", "bf16", "f32x3"):
            raise ValueError("conv_dtype must be 'f32', 'f32x3' or 'bf16'")
        self._engine.bf16 = value == "bf16"
        self._engine.x3 = value == "f32x3"
        self._engine._wcache.clear()

    @property
    def winograd(self) -> bool:
        """f32 mode: Winograd F(4x4,3x3) for forward / dgrad of the 3x3 stride-1 layers and the deep weight gradients (default on:
        ~1e-5 of the output scale from the direct sum, 1.5x the step rate).  False = exact-f32 MFMA direct kernels everywhere."""
        return self._engine.wino

    @winograd.setter
    def winograd(self, value: bool) -> None:
        self._engine.wino = bool(value)
        self._engine._wcache.clear()

    def invalidate_weight_cache(self) -> None:
        """Drop the re-laid (OHWI / IHWO / Winograd-domain) copies of the weights.  A training forward always rebuilds them;
        an inference forward reuses them while every parameter's `(data_ptr, _version)` is unchanged.  In-place updates through
        autograd-visible ops (`p.mul_()`, `p.copy_()` under `no_grad`, optimizers, `load_state_dict`, `.to()`) change that key;
        writes through `p.data` (`p.data.copy_()`, `dist.broadcast(p.data, 0)`, EMA weight swaps) do NOT -- call this after
        them before the next `.eval()` forward (and re-capture any `graphed_forward`)."""
        self._engine._wcache.clear()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if hasattr(self, "_engine"):
            self._engine._wcache.clear()
        return out

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._engine._wcache.clear()
        return out

    def get_norm(self):
        return torch.norm(self.fc6) + torch.norm(self.fc6_b) + torch.norm(self.fc7) + torch.norm(self.fc7_b)

    def initialization(self):
        """Xavier-uniform weights / zero biases for the aux and head convolutions (Model.py:190-200)."""
        for name in self._aux_names:
            seq = getattr(self, name)
            self.initialize(seq[0])
            self.initialize(seq[2])
        for name in self._head_names:
            self.initialize(getattr(self, name + "_bb"))
            self.initialize(getattr(self, name + "_cl"))

    def initialize(self, c):
        nn.init.xavier_uniform_(c.weight)
        nn.init.constant_(c.bias, 0.)

    def graphed_forward(self, example_input: torch.Tensor) -> "GraphedForward":
        """Capture the inference forward for this input shape into a HIP graph (see GraphedForward)."""
        return GraphedForward(self, example_input)

    def _forward_params(self) -> Dict[str, torch.Tensor]:
        named = dict(self.named_parameters())
        return {n: named[n] for n in self._engine.names}

    def forward(self, x):
        P = self._forward_params()
        eng = self._engine
        if torch.is_grad_enabled() and any(p.requires_grad for p in P.values()):
            return _SSD300Function.apply(x, eng, *[P[n] for n in eng.names])
        loc, conf, _ = eng.forward(x, P, save=False)
        return loc, conf


class GraphedForward:
    """Inference forward of a fixed input shape captured once into a HIP graph and replayed: at small batches the
    ~80 kernel launches of a forward cost more host time than GPU time, the replay is one launch.  Every kernel of the
    path only enqueues on the stream it is given (no allocation, no synchronisation inside the library), so the capture
    needs nothing special.  The graph bakes in the weight layouts of the moment of capture: capture again after the
    parameters change.  `__call__(x)` copies x into the static input and returns the static (loc, conf) tensors, which
    the next call overwrites."""

    def __init__(self, net: "SSD_300", example: torch.Tensor):
        if net.training and any(p.requires_grad for p in net.parameters()) and torch.is_grad_enabled():
            pass                                               # captured under no_grad below either way
        self.net = net
        self.x = example.detach().clone().contiguous()
        side = torch.cuda.Stream(device=self.x.device)
        side.wait_stream(torch.cuda.current_stream(self.x.device))
        with torch.no_grad(), torch.cuda.stream(side):        # warm-up: weight layouts, workspaces, allocator pools
            for _ in range(2):
                net(self.x)
        torch.cuda.current_stream(self.x.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.loc, self.conf = net(self.x)

    def __call__(self, x: torch.Tensor):
        if tuple(x.shape) != tuple(self.x.shape) or x.dtype != self.x.dtype:
            raise ValueError(f"graph captured for {tuple(self.x.shape)} {self.x.dtype}, got {tuple(x.shape)} {x.dtype}")
        self.x.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.loc, self.conf


class SSD_512(SSD_300):
    """Build-defined SSD512 (NOT in the reference): 512x512 input, seven maps
    64/32/16/8/4/2/1 (every aux block 1x1 -> 3x3 stride 2 pad 1, plus `seq12` / `c_12_*`), 24564 priors
    (`Util.create_priors_ssd512`).  Same kernels, same parameter naming scheme; parity is against the oracle's own
    restatement only -- there is no reference implementation to compare with."""
    _VARIANT = 512


from .ModelResnet import SSD_resnet34  # noqa: E402,F401  (reference Model.py:12-126 lives in the same module)
