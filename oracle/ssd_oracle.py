"""CPU oracle for the SSD300 hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (numpy for the box / index arithmetic, plain
torch-CPU ops for the floating-point network and the log-softmax) of the
algorithms on the reference's hot path.  It exists so that the HIP kernels in
``objectdetection_ssd_amd/csrc`` can be checked on the GPU box, where
``/root/reference`` does not exist.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package never does.

Pinning: ``oracle/gen_golden.py`` runs the reference's own ``Util.py`` /
``Losses.py`` / ``Model.py`` (imported unmodified in the build container) on
seeded inputs and commits the input/output vectors under ``tests/golden``;
``tests/test_oracle_golden.py`` checks every function here against them.
The reference has no tests or golden vectors of its own.

Every function cites the reference file:line it restates (paths relative to
the reference repository root).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np

P_SSD300 = 8732
N_CLASSES = 21          # 20 VOC classes + background
BG_CLASS = 20           # Losses.py:171,179 -- background is the LAST class

# ---------------------------------------------------------------------------
# priors  (Util.py:105-137)
# ---------------------------------------------------------------------------
_GRIDS = (38, 19, 10, 5, 3, 1)
_SCALES = (0.1, 0.2, 0.375, 0.55, 0.725, 0.9)
# literal .333, not 1/3  (Util.py:114-116)
_RATIOS = ((1., 2., .5), (1., 2., 3., .5, .333), (1., 2., 3., .5, .333),
           (1., 2., 3., .5, .333), (1., 2., .5), (1., 2., .5))
ANCHORS_PER_CELL = (4, 6, 6, 6, 4, 4)
SCALE_OFFSETS = (0, 5776, 7942, 8542, 8692, 8728)


def create_priors_ssd300() -> np.ndarray:
    """(8732,4) f32 cx,cy,w,h.  Util.py:105-137.

    Row-major over the grid (cy outer, cx inner, Util.py:122-126); for every
    ratio ``a`` a box (s*sqrt(a), s/sqrt(a)); immediately after a == 1 the
    extra square box of scale sqrt(s_k*s_{k+1}) (1.0 for the last grid).
    Computed in python doubles, rounded to f32, then clamped to [0,1] on
    cx,cy,w,h (Util.py:135-136).
    """
    rows = []
    for k, g in enumerate(_GRIDS):
        s = _SCALES[k]
        s_next = math.sqrt(s * _SCALES[k + 1]) if k + 1 < len(_SCALES) else 1.0
        for i in range(g):
            for j in range(g):
                cx = (j + 0.5) / float(g)
                cy = (i + 0.5) / float(g)
                for a in _RATIOS[k]:
                    rows.append((cx, cy, s * math.sqrt(a), s / math.sqrt(a)))
                    if a == 1.:
                        rows.append((cx, cy, s_next, s_next))
    pri = np.asarray(rows, dtype=np.float64).astype(np.float32)
    np.clip(pri, 0.0, 1.0, out=pri)
    assert pri.shape == (P_SSD300, 4)
    return pri


# ---- SSD512: NOT in the reference.  Build-defined geometry, restated here only so
# that the kernels can be checked at this size too; nothing below is pinned by the reference.
_GRIDS_512 = (64, 32, 16, 8, 4, 2, 1)
_SCALES_512 = (0.07, 0.15, 0.30, 0.45, 0.60, 0.75, 0.90)
_RATIOS_512 = ((1., 2., .5),) + ((1., 2., 3., .5, .333),) * 4 + ((1., 2., .5),) * 2
P_SSD512 = 24564


def create_priors_ssd512() -> np.ndarray:
    rows = []
    for k, g in enumerate(_GRIDS_512):
        s = _SCALES_512[k]
        s_next = math.sqrt(s * _SCALES_512[k + 1]) if k + 1 < len(_SCALES_512) else 1.0
        for i in range(g):
            for j in range(g):
                cx, cy = (j + 0.5) / float(g), (i + 0.5) / float(g)
                for a in _RATIOS_512[k]:
                    rows.append((cx, cy, s * math.sqrt(a), s / math.sqrt(a)))
                    if a == 1.:
                        rows.append((cx, cy, s_next, s_next))
    pri = np.asarray(rows, dtype=np.float64).astype(np.float32)
    np.clip(pri, 0.0, 1.0, out=pri)
    assert pri.shape == (P_SSD512, 4)
    return pri


def xywh_to_xyxy(b: np.ndarray) -> np.ndarray:
    """Util.py:93-96 (f32: c - wh/2, c + wh/2)."""
    b = np.asarray(b, dtype=np.float32)
    half = b[:, 2:] / np.float32(2.)
    return np.concatenate([b[:, :2] - half, b[:, :2] + half], axis=1).astype(np.float32)


def xyxy_to_xywh(b: np.ndarray) -> np.ndarray:
    """Util.py:57-63: ((x2+x1)/2, (y2+y1)/2, x2-x1, y2-y1) in f32."""
    b = np.asarray(b, dtype=np.float32)
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    two = np.float32(2.)
    return np.stack([(x2 + x1) / two, (y2 + y1) / two, x2 - x1, y2 - y1], axis=1).astype(np.float32)


def encode_offsets(cxcywh: np.ndarray, pri: np.ndarray) -> np.ndarray:
    """Util.py:98-102 get_offsets_coords: (c - pc)/(pwh/10), log(wh/pwh)*5."""
    c = np.asarray(cxcywh, np.float32)
    p = np.asarray(pri, np.float32)
    g_c = (c[:, :2] - p[:, :2]) / (p[:, 2:] / np.float32(10))
    g_wh = np.log(c[:, 2:] / p[:, 2:]).astype(np.float32) * np.float32(5)
    return np.concatenate([g_c, g_wh], axis=1).astype(np.float32)


def decode_offsets(g: np.ndarray, pri: np.ndarray) -> np.ndarray:
    """Util.py:86-91 gcxgcy_to_cxcy: g_c*pwh/10 + pc, exp(g_wh/5)*pwh."""
    g = np.asarray(g, np.float32)
    p = np.asarray(pri, np.float32)
    c = g[:, :2] * p[:, 2:] / np.float32(10) + p[:, :2]
    wh = np.exp(g[:, 2:] / np.float32(5)).astype(np.float32) * p[:, 2:]
    return np.concatenate([c, wh], axis=1).astype(np.float32)


# ---------------------------------------------------------------------------
# IoU (Util.py:252-265 find_intersection, Util.py:288-301 get_jaccard_tensor1)
# ---------------------------------------------------------------------------
def iou_matrix(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """(n1,4),(n2,4) xyxy -> (n1,n2) f32.

    inter = clamp(min(hi) - max(lo), 0) product; area = (x2-x1)*(y2-y1);
    iou = inter / ((area1 + area2) - inter); no epsilon (0/0 -> NaN).
    Only +,-,*,/,min,max: reproducible bit-for-bit in IEEE f32 as long as no
    fused multiply-add is formed.
    """
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    lo = np.maximum(a[:, None, :2], b[None, :, :2])
    hi = np.minimum(a[:, None, 2:], b[None, :, 2:])
    d = np.maximum(hi - lo, np.float32(0))
    inter = d[:, :, 0] * d[:, :, 1]
    a1 = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    a2 = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    union = (a1[:, None] + a2[None, :]) - inter
    with np.errstate(invalid="ignore", divide="ignore"):
        return (inter / union).astype(np.float32)


# ---------------------------------------------------------------------------
# matching (Losses.py:119-134 ssd, Losses.py:150-179 ssd1_)
# ---------------------------------------------------------------------------
def match_priors(boxes: Sequence[np.ndarray], classes: Sequence[np.ndarray],
                 pri_xyxy: np.ndarray):
    """Batched prior<->GT matching.

    boxes[i]: (n_i,4) f32 xyxy in 0..1, classes[i]: (n_i,) values 0..19.
    Returns obj (bs,P) int64 = GLOBAL index into cat(boxes); cls (bs,P) int64
    (20 = background); overlap (bs,P) f32 (after the forced 1.0).

    * best GT per prior is taken over the image's own rows only, first index
      on ties (Losses.py:152-155);
    * best prior per GT is taken over all priors, first index on ties
      (Losses.py:157);
    * forced matches are written in GT order so on collisions the LAST GT of
      the image wins (Losses.py:164-167);
    * threshold: overlap < 0.5 -> background (Losses.py:171).
    An image without GT makes the reference raise (max over an empty dim); so
    do we.
    """
    bs = len(boxes)
    counts = [int(np.asarray(b).shape[0]) for b in boxes]
    if any(c == 0 for c in counts):
        raise ValueError("every image needs at least one ground-truth box")
    start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    allb = np.concatenate([np.asarray(b, np.float32).reshape(-1, 4) for b in boxes], 0)
    allc = np.concatenate([np.asarray(c, np.float32).reshape(-1) for c in classes], 0)
    iou = iou_matrix(allb, pri_xyxy)                       # (N,P)
    P = pri_xyxy.shape[0]
    obj = np.zeros((bs, P), np.int64)
    overlap = np.zeros((bs, P), np.float32)
    prior_for_obj = np.argmax(iou, axis=1)                 # first index on ties
    for i in range(bs):
        s, e = start[i], start[i + 1]
        sub = iou[s:e]
        obj[i] = np.argmax(sub, axis=0) + s                # first index on ties
        overlap[i] = sub.max(axis=0)
        for k in range(s, e):                              # sequential: last k wins
            obj[i, prior_for_obj[k]] = k
            overlap[i, prior_for_obj[k]] = np.float32(1.)
    cls = allc[obj].astype(np.float32)
    cls[overlap < np.float32(0.5)] = BG_CLASS
    return obj, cls.astype(np.int64), overlap, allb, start


def _log_softmax_rows(x: np.ndarray) -> np.ndarray:
    import torch
    return torch.log_softmax(torch.from_numpy(np.ascontiguousarray(x, np.float32)), dim=-1).numpy()


def multibox_loss(loc: np.ndarray, conf: np.ndarray, boxes, classes,
                  pri_cxcywh: np.ndarray | None = None, neg_pos_ratio: int = 3,
                  want_grads: bool = True):
    """Losses.py:119-199.  Returns dict with loc_loss, conf_loss (f32 scalars),
    and -- restated analytically, checked against the reference's autograd in
    gen_golden.py -- dloc, dconf = d(loc_loss + conf_loss)/d(loc|conf).

    loc_loss  = mean over ALL n_pos*4 elements of |pred - g|   (nn.L1Loss, :147,:182)
    conf_loss = (sum of top-(3*n_pos_i) negative CE per image + sum positive CE) / n_pos_total  (:184-197)
    """
    loc = np.asarray(loc, np.float32)
    conf = np.asarray(conf, np.float32)
    bs, P, C = conf.shape
    if pri_cxcywh is None:
        pri_cxcywh = create_priors_ssd300()
    pri_xyxy = xywh_to_xyxy(pri_cxcywh)
    obj, cls, overlap, allb, _ = match_priors(boxes, classes, pri_xyxy)
    pos = cls != BG_CLASS
    n_pos = int(pos.sum())
    gt_cxcywh = xyxy_to_xywh(allb)[obj]                   # (bs,P,4)
    pri_rep = np.broadcast_to(pri_cxcywh[None], (bs, P, 4))
    g = encode_offsets(gt_cxcywh[pos], pri_rep[pos])     # (n_pos,4)
    diff = loc[pos] - g
    loc_loss = np.float32(np.abs(diff).astype(np.float64).sum() / (n_pos * 4))

    logp = _log_softmax_rows(conf.reshape(-1, C)).reshape(bs, P, C)
    cce = -np.take_along_axis(logp, cls[..., None], axis=2)[..., 0]   # (bs,P)
    neg = cce.copy()
    neg[pos] = 0.
    k = neg_pos_ratio * pos.sum(axis=1)
    order = np.argsort(-neg, axis=1, kind="stable")
    hn_mask = np.zeros_like(pos)
    hn_sum = 0.0
    for i in range(bs):
        sel = order[i, :min(int(k[i]), P)]
        hn_mask[i, sel] = True
        hn_sum += float(neg[i, sel].astype(np.float64).sum())
    conf_loss = np.float32((hn_sum + float(cce[pos].astype(np.float64).sum())) / n_pos)
    out = dict(loc_loss=loc_loss, conf_loss=conf_loss, obj=obj, cls=cls, pos=pos,
               n_pos=n_pos, cce=cce, hn_mask=hn_mask, enc=g, overlap=overlap)
    if want_grads:
        dloc = np.zeros_like(loc)
        dloc[pos] = np.sign(diff) / np.float32(n_pos * 4)
        sel = pos | hn_mask
        sm = np.exp(logp)
        onehot = np.zeros_like(sm)
        np.put_along_axis(onehot, cls[..., None], 1.0, axis=2)
        dconf = ((sm - onehot) * sel[..., None] / np.float32(n_pos)).astype(np.float32)
        out.update(dloc=dloc.astype(np.float32), dconf=dconf)
    return out


# ---------------------------------------------------------------------------
# decode + per-class NMS + top-k (Losses.py:11-98 inference)
# ---------------------------------------------------------------------------
def decode_nms(l_: np.ndarray, c_: np.ndarray, w: float, h: float, top_k: int = 200,
               min_score: float = 0.2, iou_threshold: float = 0.45,
               pri_cxcywh: np.ndarray | None = None):
    """Returns (boxes (K,4) f32 pixels xyxy, classes (K,) int64, probs (K,) f32,
    prior_ids (K,) int64).  K == 0 -> empty arrays (the reference returns
    three empty lists, Losses.py:62-63).

    Per class 0..19 (never background, :27): candidates prob >= min_score (:32);
    sorted by prob descending, lower prior index first on ties (:38, CPU
    behaviour); greedy suppression with IoU >= iou_threshold (:44-55); kept
    boxes concatenated class-major (:71-73); only if more than top_k survive,
    a global descending sort keeps the first top_k (:77-81); boxes scaled by
    (w,h,w,h) (:89).
    """
    if pri_cxcywh is None:
        pri_cxcywh = create_priors_ssd300()
    l_ = np.asarray(l_, np.float32)
    c_ = np.asarray(c_, np.float32)
    boxes_cxcywh = decode_offsets(l_, pri_cxcywh)
    import torch
    probs = torch.softmax(torch.from_numpy(c_), dim=1).numpy()
    kb, kc, kp, ki = [], [], [], []
    for c in range(N_CLASSES - 1):
        pc = probs[:, c]
        cand = np.nonzero(pc >= np.float32(min_score))[0]
        if cand.size == 0:
            continue
        order = cand[np.argsort(-pc[cand], kind="stable")]
        bx = xywh_to_xyxy(boxes_cxcywh[order])
        iou = iou_matrix(bx, bx)
        n = order.size
        suppressed = np.zeros(n, bool)
        for i in range(n):
            if suppressed[i]:
                continue
            suppressed |= iou[i] >= np.float32(iou_threshold)
            suppressed[i] = False
        keep = ~suppressed
        kb.append(bx[keep]); kp.append(pc[order][keep]); ki.append(order[keep])
        kc.append(np.full(int(keep.sum()), c, np.int64))
    if not kb:
        z = np.zeros
        return z((0, 4), np.float32), z((0,), np.int64), z((0,), np.float32), z((0,), np.int64)
    kb = np.concatenate(kb); kc = np.concatenate(kc); kp = np.concatenate(kp); ki = np.concatenate(ki)
    if kb.shape[0] > top_k:
        o = np.argsort(-kp, kind="stable")[:top_k]
        kb, kc, kp, ki = kb[o], kc[o], kp[o], ki[o]
    scale = np.asarray([w, h, w, h], np.float32)[None]
    return (kb * scale).astype(np.float32), kc, kp.astype(np.float32), ki


# ---------------------------------------------------------------------------
# SSD_300 network (Model.py:128-235) as a functional torch-CPU forward
# ---------------------------------------------------------------------------
# (name, cin, cout, k, stride, pad, dil, relu) in execution order; pools are
# interleaved by name.  VGG indices follow torchvision's vgg16 'D' features
# list that Model.py:136-141 slices.
VGG_CONV_IDX = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)
VGG_CH = (3, 64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512)
HEADS = (("c_4", 512, 4), ("c_7", 1024, 6), ("c_8", 512, 6),
         ("c_9", 256, 6), ("c_10", 256, 4), ("c_11", 256, 4))
AUX = (("seq8", 1024, 256, 512, 2, 1), ("seq9", 512, 128, 256, 2, 1),
       ("seq10", 256, 128, 256, 1, 0), ("seq11", 256, 128, 256, 1, 0))
# build-defined SSD512 (not in the reference): every aux block is 1x1 -> 3x3 stride 2 pad 1, one more block
HEADS_512 = (("c_4", 512, 4), ("c_7", 1024, 6), ("c_8", 512, 6), ("c_9", 256, 6), ("c_10", 256, 6), ("c_11", 256, 4),
             ("c_12", 256, 4))
AUX_512 = (("seq8", 1024, 256, 512, 2, 1), ("seq9", 512, 128, 256, 2, 1), ("seq10", 256, 128, 256, 2, 1),
           ("seq11", 256, 128, 256, 2, 1), ("seq12", 256, 128, 256, 2, 1))


def ssd300_param_shapes(variant: int = 300) -> Dict[str, Tuple[int, ...]]:
    """Names follow the reference's named_parameters() for the tensors the
    forward actually uses."""
    sh: Dict[str, Tuple[int, ...]] = {"rescaling_conv_4_3": (1, 512, 1, 1)}
    for li, idx in enumerate(VGG_CONV_IDX):
        sh[f"model.features.{idx}.weight"] = (VGG_CH[li + 1], VGG_CH[li], 3, 3)
        sh[f"model.features.{idx}.bias"] = (VGG_CH[li + 1],)
    sh["conv_fc6.weight"] = (1024, 512, 3, 3); sh["conv_fc6.bias"] = (1024,)
    sh["conv_fc7.weight"] = (1024, 1024, 1, 1); sh["conv_fc7.bias"] = (1024,)
    for name, cin, mid, cout, _, _ in (AUX if variant == 300 else AUX_512):
        sh[f"{name}.0.weight"] = (mid, cin, 1, 1); sh[f"{name}.0.bias"] = (mid,)
        sh[f"{name}.2.weight"] = (cout, mid, 3, 3); sh[f"{name}.2.bias"] = (cout,)
    for name, cin, a in (HEADS if variant == 300 else HEADS_512):
        sh[f"{name}_bb.weight"] = (4 * a, cin, 3, 3); sh[f"{name}_bb.bias"] = (4 * a,)
        sh[f"{name}_cl.weight"] = (21 * a, cin, 3, 3); sh[f"{name}_cl.bias"] = (21 * a,)
    return sh


def ssd300_random_params(seed: int = 0, variant: int = 300):
    """Seeded parameters with O(1) activations (He-normal fan-in for the ReLU
    stack, Xavier-uniform + zero bias for aux/head convs as Model.py:190-200,
    rescale 20 as Model.py:133).  Pretrained VGG weights are not available
    offline, so parity is on these."""
    import torch
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in ssd300_param_shapes(variant).items():
        if name == "rescaling_conv_4_3":
            t = torch.full(shape, 20.0)
        elif name.endswith(".bias"):
            backbone = name.startswith("model.") or name.startswith("conv_fc")
            t = (torch.randn(shape, generator=g) * 0.05) if backbone else torch.zeros(shape)
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            fan_out = shape[0] * shape[2] * shape[3]
            if name.startswith("model.") or name.startswith("conv_fc"):
                t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
            else:
                bound = math.sqrt(6.0 / (fan_in + fan_out))
                t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        out[name] = t.float()
    return out


def bf16_round(t):
    """f32 -> nearest-even bf16 -> f32 (what `v_cvt_pk_bf16_f32` does to a convolution operand on its way into LDS)."""
    import torch
    return t.to(torch.bfloat16).to(t.dtype)


def _round_store():
    """Identity that rounds to bf16 in BOTH directions: the value on the way forward (an activation stored in bf16) and its gradient on
    the way back (the gradient tensor stored in bf16).  The bf16-tensor mode of the build (csrc/conv_bf16.hip & co.) has exactly these
    rounding points on the VGG trunk."""
    import torch

    class RoundStore(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return bf16_round(x)

        @staticmethod
        def backward(ctx, g):
            return bf16_round(g)
    return RoundStore.apply


def bf16_spacing(t):
    """Distance between neighbouring bf16 values at |t| (8 significant bits: 2^(floor(log2|t|) - 7)), as a tensor of t's dtype."""
    import torch
    a = t.abs().clamp_min(2.0 ** -126)
    return torch.pow(torch.full_like(a, 2.0), torch.floor(torch.log2(a)) - 7.0)


def _pin_store(name, fwd, bwd, report, kind, gate):
    """TESTS ONLY (the rounding-pinned comparison of the bf16-tensor mode).  An identity that REPLACES what flows through it by the
    values ANOTHER evaluation of the same network stored at this point: on the way forward the activation tensor `fwd`, on the way
    back the gradient tensor `bwd` (None: pass the gradient through).  Every layer of this evaluation then sees exactly the other
    evaluation's inputs -- the stored bf16 / f32 values, i.e. its rounding DIRECTIONS are pinned like its ReLU / arg-max decisions --
    and `report[name + ':fwd' | ':bwd']` records how far this evaluation's own value was from the one it is replaced by:

        max over elements of (|own - stored| - 2e-5 * rms(own)) / spacing,   spacing = the bf16 spacing at |stored| (kind 'bf16')
                                                                              or 1e-5 * max|stored|            (kind 'f32')

    -- for a correct layer <= 0.5 (one rounding to nearest) on a bf16 tensor, <= 1 on an f32 tensor; the rms term covers sums that
    cancel to (almost) nothing, whose f32 accumulation error is set by the size of the terms, not of the result.
    gate: bool tensor or None -- the gradient is compared where gate is set only (a post-ReLU tensor's stored gradient is already
    masked by the ReLU that follows on the way back)."""
    import torch

    def dev(own, stored, mask):
        own, stored = own.detach().double(), stored.double()
        rms = float(own.pow(2).mean().sqrt())
        space = bf16_spacing(stored) if kind == "bf16" else torch.full_like(stored, 1e-5 * max(float(stored.abs().max()), 1e-30))
        d = ((own - stored).abs() - 2e-5 * rms) / space
        if mask is not None:
            d = d * mask.to(d.dtype)
        return float(d.max()) if d.numel() else 0.0

    class PinStore(torch.autograd.Function):
        @staticmethod
        def forward(ctx, z):
            if fwd is None:                      # a branch pin: only the gradient flowing back through THIS reader is replaced
                return z.view_as(z)
            report[name + ":fwd"] = dev(z, fwd, None)
            return fwd.to(z.dtype)

        @staticmethod
        def backward(ctx, g):
            if bwd is None:
                return g
            report[name + ":bwd"] = dev(g if gate is None else g * gate.to(g.dtype), bwd, gate)
            return bwd.to(g.dtype)
    return PinStore.apply


def _conv_bf16_operands():
    """conv2d of BASELINE.json configs[2] ("bf16 convs"): operands rounded to bf16, products accumulated in f32, bias in f32.
    Forward y = conv(r(x), r(w)) + b; data gradient dx = conv^T(r(dy), r(w)); weight gradient corr(r(x), r(dy)) for the geometries the
    bf16 patch kernel takes (stride 1 and: 3x3 with padding = dilation in {1, 4}, or 1x1 without padding -- the VGG layers, the heads on
    the large maps, fc6, fc7 and the aux blocks' first convolutions) and corr(x, dy) in plain f32 for the others (strided / unpadded 3x3:
    those weight gradients stay on the f32 kernels).  The bias gradient is the f32 sum of dy as it is STORED: unrounded where dy is an
    f32 tensor, rounded (`dy_bf16=True`) where the build holds dy in bf16 only (the c_4 head's packed gradient in the bf16-tensor mode).
    `wgrad_f32=True` forces the f32 weight gradient (conv1_1: K = 27 from the f32 image)."""
    import torch
    import torch.nn.functional as F

    def patch_geom(w, stride, padding, dilation):
        k = tuple(w.shape[2:])
        return stride == 1 and ((k == (3, 3) and padding == dilation and dilation in (1, 4)) or (k == (1, 1) and padding == 0))

    class ConvBf16Operands(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, b, stride, padding, dilation, wgrad_f32, dy_bf16):
            ctx.save_for_backward(x, w)
            ctx.cfg = (stride, padding, dilation, wgrad_f32, dy_bf16)
            return F.conv2d(bf16_round(x), bf16_round(w), b, stride=stride, padding=padding, dilation=dilation)

        @staticmethod
        def backward(ctx, dy):
            x, w = ctx.saved_tensors
            stride, padding, dilation, wgrad_f32, dy_bf16 = ctx.cfg
            dx = dw = db = None
            if ctx.needs_input_grad[0]:
                dx = torch.nn.grad.conv2d_input(x.shape, bf16_round(w), bf16_round(dy), stride=stride, padding=padding, dilation=dilation)
            if ctx.needs_input_grad[1]:
                fused = patch_geom(w, stride, padding, dilation) and not wgrad_f32
                xa, da = (bf16_round(x), bf16_round(dy)) if fused else (x, dy)
                dw = torch.nn.grad.conv2d_weight(xa, w.shape, da, stride=stride, padding=padding, dilation=dilation)
            if ctx.needs_input_grad[2]:
                db = (bf16_round(dy) if dy_bf16 else dy).sum(dim=(0, 2, 3))
            return dx, dw, db, None, None, None, None, None

    def conv(x, w, b=None, stride=1, padding=0, dilation=1, wgrad_f32=False, dy_bf16=False):
        return ConvBf16Operands.apply(x, w, b, stride, padding, dilation, wgrad_f32, dy_bf16)
    return conv


_VGG_ACT = ("a1_1", "a1_2", "a2_1", "a2_2", "a3_1", "a3_2", "a3_3", "a4_1", "a4_2", "a4_3", "a5_1", "a5_2", "a5_3")


def pinned_max_pool(z, code, k: int, stride: int, pad: int):
    """Max pool with the arg-max GIVEN: y[n,c,oh,ow] = z[n,c, oh*stride - pad + code // k, ow*stride - pad + code % k], `code` (N,C,Ho,Wo)
    the window index r*k + s the HIP pools store (csrc/elementwise.hip maxpool_fwd_kernel).  Differentiable in z (a gather)."""
    import torch
    N, C, H, W = z.shape
    Ho, Wo = code.shape[2], code.shape[3]
    code = code.long()
    ih = torch.arange(Ho).view(1, 1, Ho, 1) * stride - pad + code // k
    iw = torch.arange(Wo).view(1, 1, 1, Wo) * stride - pad + code % k
    if bool((ih < 0).any() or (ih >= H).any() or (iw < 0).any() or (iw >= W).any()):
        raise ValueError("pinned_max_pool: an arg-max code points outside the map")
    return z.flatten(2).gather(2, (ih * W + iw).flatten(2)).view(N, C, Ho, Wo)


def ssd300_forward(x, params, return_features: bool = False, variant: int = 300, operand_round: str = None, acts: dict = None,
                   decisions: dict = None, store_round: bool = False, pinned: dict = None):
    """x (bs,3,300,300) f32 NCHW torch tensor -> loc (bs,8732,4), conf (bs,8732,21).
    store_round (with operand_round="bf16"): the bf16-TENSOR mode -- the VGG trunk's tensors (conv1_1 .. conv5_3 outputs, the pools'
    outputs, the L2-norm's output) are stored in bf16, and so are their gradients (`_round_store`); fc6 onwards keeps f32 tensors.
    decisions (tests only): the discrete choices of ANOTHER evaluation of the same network, which this one then follows instead of
    making its own -- {"relu": {activation name: bool mask NCHW}, "pool": {pool name p1..p5: (arg-max codes (N,C,Ho,Wo), gate or
    None)}}.  ReLU becomes `z * mask`; a max pool becomes a gather at the given arg-max (times `gate` = "pooled output > 0" where the
    other evaluation fused conv -> ReLU -> pool and left no full-resolution mask).  With the decisions pinned the network is one
    fixed linear-in-pieces function: two evaluations differ by arithmetic only, not by a ReLU / arg-max that flipped on a last-bit
    difference (which is a discrete jump of one gradient path).
    pinned (tests only, with decisions): {"fwd": {tensor name: NCHW values}, "bwd": {tensor name: NCHW gradient}, "bf16": set of the
    names stored in bf16, "report": {}} -- every named tensor (a1_1 .. a11, p1 .. p5, n4_3) and its gradient are REPLACED by the
    given values of another evaluation as they pass (`_pin_store`), and `report` receives this evaluation's distance to them,
    tensor by tensor.  Replaces `store_round`'s own rounding: the other evaluation's rounding directions are followed.
    operand_round="bf16": every convolution multiplies bf16-rounded operands with f32 accumulation (`_conv_bf16_operands`:
    BASELINE.json configs[2]); pools, L2-norm, biases, ReLU and the tensors between layers stay f32.
    acts: optional dict, filled with every post-ReLU activation (NCHW, detached) under the build's tensor names
    (a1_1 .. a5_3, a6, a7, a8a, a8, ... , n4_3): lets a test follow the two computations layer by layer.

    Model.py:203-235: conv1_1..conv4_3 with 2x2/s2 pools (third one
    ceil_mode, :137); L2-norm over channels * gamma, no epsilon (:206-209);
    pool4 2x2/s2, conv5_x, pool5 3x3/s1/p1 (:140-143); fc6 3x3 dilation 4
    padding 4, fc7 1x1 (:149,:159); aux blocks 1x1 -> 3x3 (s2p1,s2p1,s1p0,s1p0)
    (:163-166); six head pairs, NHWC-flattened and concatenated in order
    4,7,8,9,10,11 (:212-235).
    """
    import torch
    import torch.nn.functional as F
    if operand_round not in (None, "bf16"):
        raise ValueError("operand_round must be None or 'bf16'")
    if store_round and operand_round != "bf16":
        raise ValueError("store_round needs operand_round='bf16'")
    conv2d = F.conv2d if operand_round is None else _conv_bf16_operands()
    first = {} if operand_round is None else {"wgrad_f32": True}
    rs_plain = _round_store() if store_round else (lambda t: t)

    def rs(t, name, trunk=True):
        """the store of tensor `name`: pinned to another evaluation's stored values, rounded to bf16 (trunk tensors of the bf16-tensor
        mode), or left alone"""
        if pinned is not None and name in pinned["fwd"]:
            relu_gate = (pinned["fwd"][name] > 0) if name[0] == "a" else None        # a*: post-ReLU activations; p*, n4_3: no ReLU of their own
            return _pin_store(name, pinned["fwd"][name], pinned["bwd"].get(name), pinned["report"],
                              "bf16" if name in pinned["bf16"] else "f32", relu_gate)(t)
        return rs_plain(t) if trunk else t
    feats = {}
    h = x
    pin_relu = None if decisions is None else decisions["relu"]
    pin_pool = None if decisions is None else decisions["pool"]

    def relu(z, name):
        if pin_relu is None:
            return F.relu(z)
        m = pin_relu.get(name)
        return z if m is None else z * m.to(z.dtype)       # no mask: the pool behind this layer carries the gate

    def pool(z, name, k, stride, pad=0, ceil=False):
        if pin_pool is None:
            return F.max_pool2d(z, k, stride, padding=pad, ceil_mode=ceil)
        code, gate = pin_pool[name]
        y = pinned_max_pool(z, code, k, stride, pad)
        return y if gate is None else y * gate.to(z.dtype)

    pools_after = {2: False, 4: False, 7: True, 10: False}   # conv ordinal -> ceil_mode
    n_pool = 0
    for li, idx in enumerate(VGG_CONV_IDX):
        h = rs(relu(conv2d(h, params[f"model.features.{idx}.weight"],
                           params[f"model.features.{idx}.bias"], padding=1, **(first if li == 0 else {})), _VGG_ACT[li]), _VGG_ACT[li])
        n = li + 1
        if acts is not None:
            acts[_VGG_ACT[li]] = h.detach()
        if n == 10:
            feats["conv4_3"] = h
        if n in pools_after:
            n_pool += 1
            h = rs(pool(h, f"p{n_pool}", 2, 2, ceil=pools_after[n]), f"p{n_pool}")
    h = rs(pool(h, "p5", 3, 1, pad=1), "p5")
    c43 = feats["conv4_3"]
    if pinned is not None and "a4_3:1" in pinned["bwd"]:
        # conv4_3's output has two readers; the other evaluation stored the L2-norm's contribution to its gradient (rounded to bf16) before
        # the pool's was added to it: follow that intermediate value too, so that each of the two roundings is compared on its own
        c43 = _pin_store("a4_3:1", None, pinned["bwd"]["a4_3:1"], pinned["report"], "bf16" if "a4_3" in pinned["bf16"] else "f32", None)(c43)
    norm = c43.pow(2).sum(dim=1, keepdim=True).sqrt()
    c43n = rs(c43 / norm * params["rescaling_conv_4_3"], "n4_3")
    h = rs(relu(conv2d(h, params["conv_fc6.weight"], params["conv_fc6.bias"], padding=4, dilation=4), "a6"), "a6", trunk=False)
    if acts is not None:
        acts["n4_3"], acts["a6"] = c43n.detach(), h.detach()
    h = rs(relu(conv2d(h, params["conv_fc7.weight"], params["conv_fc7.bias"]), "a7"), "a7", trunk=False)
    if acts is not None:
        acts["a7"] = h.detach()
    srcs = [c43n, h]
    for name, _, _, _, stride, pad in (AUX if variant == 300 else AUX_512):
        h = rs(relu(conv2d(h, params[f"{name}.0.weight"], params[f"{name}.0.bias"]), "a" + name[3:] + "a"), "a" + name[3:] + "a", trunk=False)
        if acts is not None:
            acts["a" + name[3:] + "a"] = h.detach()
        h = rs(relu(conv2d(h, params[f"{name}.2.weight"], params[f"{name}.2.bias"], stride=stride, padding=pad), "a" + name[3:]), "a" + name[3:],
               trunk=False)
        if acts is not None:
            acts["a" + name[3:]] = h.detach()
        srcs.append(h)
    bs = x.shape[0]
    locs, confs = [], []
    for (name, _, _), s in zip(HEADS if variant == 300 else HEADS_512, srcs):
        # the heads that run the bf16-tensor kernels (c_4 on the bf16 trunk, c_7 on a bf16 copy of fc7's output): their packed gradient
        # exists in bf16 only, so the bias gradient sums the rounded values
        hk = {"dy_bf16": True} if (store_round and name in ("c_4", "c_7")) else {}
        bb = conv2d(s, params[f"{name}_bb.weight"], params[f"{name}_bb.bias"], padding=1, **hk)
        cl = conv2d(s, params[f"{name}_cl.weight"], params[f"{name}_cl.bias"], padding=1, **hk)
        locs.append(bb.permute(0, 2, 3, 1).reshape(bs, -1, 4))
        confs.append(cl.permute(0, 2, 3, 1).reshape(bs, -1, 21))
    loc, conf = torch.cat(locs, 1), torch.cat(confs, 1)
    if return_features:
        return loc, conf, srcs
    return loc, conf


def multibox_loss_torch(loc, conf, boxes, classes, pri_cxcywh=None, neg_select=None):
    """Differentiable torch-CPU form of multibox_loss (same selection logic,
    matching done by match_priors) used as the CPU train-step baseline.
    neg_select (tests only): bool (bs, P), the hard negatives ANOTHER evaluation picked; used instead of this one's own top-k
    (the decision-pinned comparison: a negative that enters or leaves the top-k on a last-bit difference of its cross-entropy is a
    discrete jump of the conf gradient)."""
    import torch
    import torch.nn.functional as F
    if pri_cxcywh is None:
        pri_cxcywh = create_priors_ssd300()
    bs, P, C = conf.shape
    obj, cls, _, allb, _ = match_priors([b.detach().cpu().numpy() for b in boxes],
                                        [c.detach().cpu().numpy() for c in classes],
                                        xywh_to_xyxy(pri_cxcywh))
    pos_np = cls != BG_CLASS
    g = encode_offsets(xyxy_to_xywh(allb)[obj][pos_np],
                       np.broadcast_to(pri_cxcywh[None], (bs, P, 4))[pos_np])
    pos = torch.from_numpy(pos_np)
    cls_t = torch.from_numpy(cls)
    loc_loss = (loc[pos] - torch.from_numpy(g)).abs().mean()
    cce = F.cross_entropy(conf.reshape(-1, C), cls_t.reshape(-1), reduction="none").view(bs, P)
    neg = cce.clone()
    neg[pos] = 0.
    if neg_select is not None:
        sel = torch.as_tensor(neg_select, dtype=torch.bool)
        if tuple(sel.shape) != (bs, P) or bool((sel & pos).any()) or not bool((sel.sum(1) == torch.clamp(3 * pos.sum(1), max=P - pos.sum(1))).all()):
            raise ValueError("neg_select must pick min(3 * n_pos, n_neg) negatives per image")
        conf_loss = (cce[sel].sum() + cce[pos].sum()) / pos.sum().to(cce.dtype)
        return loc_loss, conf_loss
    neg_sorted, _ = neg.sort(dim=1, descending=True)
    k = 3 * pos.sum(dim=1, keepdim=True)
    hn = torch.arange(P)[None, :] < k
    conf_loss = (neg_sorted[hn].sum() + cce[pos].sum()) / pos.sum().float()
    return loc_loss, conf_loss


# ---------------------------------------------------------------------------
# SSD_resnet34 (Model.py:12-126) -- eval-mode forward only (train mode draws
# dropout masks from the global RNG stream and cannot be matched).
# ---------------------------------------------------------------------------
RESNET34_LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))      # (channels, blocks, first stride)
BN_EPS = 1e-5
_BN_FIELDS = (("weight", None), ("bias", None), ("running_mean", None), ("running_var", None), ("num_batches_tracked", ()))


def _bn_shapes(prefix: str, c: int, sh: Dict[str, Tuple[int, ...]]):
    for f, shape in _BN_FIELDS:
        sh[f"{prefix}.{f}"] = (c,) if shape is None else shape


def resnet34_trunk_shapes(prefix: str = "resnet.") -> Dict[str, Tuple[int, ...]]:
    """state_dict entries of the torchvision ResNet-34 layer list the reference slices (Model.py:21-30):
    conv1 7x7/s2/p3 (no bias), bn1, four stages of BasicBlocks (conv3x3 -> bn -> relu -> conv3x3 -> bn, identity or
    1x1-stride conv + bn shortcut, relu), fc 512 -> 1000 (dead in the reference's forward)."""
    sh: Dict[str, Tuple[int, ...]] = {prefix + "conv1.weight": (64, 3, 7, 7)}
    _bn_shapes(prefix + "bn1", 64, sh)
    cin = 64
    for li, (c, nblk, stride) in enumerate(RESNET34_LAYERS, start=1):
        for b in range(nblk):
            p = f"{prefix}layer{li}.{b}."
            sh[p + "conv1.weight"] = (c, cin, 3, 3)
            _bn_shapes(p + "bn1", c, sh)
            sh[p + "conv2.weight"] = (c, c, 3, 3)
            _bn_shapes(p + "bn2", c, sh)
            if b == 0 and (stride != 1 or cin != c):
                sh[p + "downsample.0.weight"] = (c, cin, 1, 1)
                _bn_shapes(p + "downsample.1", c, sh)
            cin = c
    sh[prefix + "fc.weight"] = (1000, 512)
    sh[prefix + "fc.bias"] = (1000,)
    return sh


def ssd_resnet34_state_shapes(k: int = 3, n_classes: int = 20) -> Dict[str, Tuple[int, ...]]:
    """The reference module's state_dict layout minus the seq1..seq5 aliases of the trunk (Model.py:26-54)."""
    sh = resnet34_trunk_shapes("resnet.")
    for name, cin in (("conv2d_0", 512), ("conv2d_01", 256), ("conv2d_02", 256), ("conv2d_03", 256)):
        sh[f"{name}.0.weight"] = (256, cin, 3, 3); sh[f"{name}.0.bias"] = (256,)
        _bn_shapes(f"{name}.2", 256, sh)
    for s in ("4", "2", "1"):
        sh[f"conv2d_02_bb{s}.0.weight"] = (4 * k, 256, 3, 3); sh[f"conv2d_02_bb{s}.0.bias"] = (4 * k,)
        _bn_shapes(f"conv2d_02_bb{s}.1", 4 * k, sh)
        sh[f"conv2d_02_c{s}.weight"] = ((n_classes + 1) * k, 256, 3, 3); sh[f"conv2d_02_c{s}.bias"] = ((n_classes + 1) * k,)
    for s in ("4", "2", "1"):
        _bn_shapes(f"bn{s}", (n_classes + 1) * k, sh)
    return sh


def ssd_resnet34_aliases() -> Dict[str, str]:
    """alias prefix -> trunk prefix for the seq1..seq5 views (Model.py:26-30): children() order is conv1, bn1, relu,
    maxpool, layer1..4, avgpool, fc, so seq1 = [conv1, bn1, relu], seq2 = [maxpool, layer1], seq3/4/5 = the blocks of
    layer2/3/4 unpacked."""
    al = {"seq1.0.": "resnet.conv1.", "seq1.1.": "resnet.bn1.", "seq2.1.": "resnet.layer1."}
    for s, li in ((3, 2), (4, 3), (5, 4)):
        al[f"seq{s}."] = f"resnet.layer{li}."
    return al


def ssd_resnet34_random_state(seed: int = 0, k: int = 3):
    """Seeded state (pretrained weights are not available offline): He-normal convs (second conv of a block scaled down
    so the residual sum stays O(1)), BN gamma in [.5,1.5], beta/mean ~ N(0,.1), var in [.5,1.5], conf-head bias -2
    (Model.py:39,43,47)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    out = {}
    shapes = ssd_resnet34_state_shapes(k)
    bn_prefixes = {n[:-len("running_mean")] for n in shapes if n.endswith(".running_mean")}
    for name, shape in shapes.items():
        prefix, leaf = name.rsplit(".", 1)
        is_bn = prefix + "." in bn_prefixes
        if leaf == "num_batches_tracked":
            t = torch.zeros((), dtype=torch.int64)
        elif leaf == "running_mean":
            t = torch.randn(shape, generator=g) * 0.1
        elif leaf == "running_var":
            t = torch.rand(shape, generator=g) + 0.5
        elif is_bn and leaf == "weight":
            t = torch.rand(shape, generator=g) + 0.5
        elif is_bn and leaf == "bias":
            t = torch.randn(shape, generator=g) * 0.1
        elif leaf == "bias":
            t = torch.full(shape, -2.0) if "_c" in name else torch.randn(shape, generator=g) * 0.05
        elif len(shape) == 2:
            t = torch.randn(shape, generator=g) * 0.01
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in) * (0.25 if name.endswith("conv2.weight") else 1.0)
        out[name] = t if t.dtype == torch.int64 else t.float()
    return out


def ssd_resnet34_forward(x, st, k: int = 3):
    """Eval-mode forward of Model.py:72-126.  x (bs,3,224,224) -> (bs, 21k, 4), (bs, 21k, 21).
    Quirks kept: conv2d_01 is applied twice (:91,:96) and conv2d_03 / bn4 / bn2 / bn1 are never used; blocks are
    Conv -> ReLU -> BN (:56-62); loc heads Conv -> BN (:64-70); conf heads plain convs (:38)."""
    import torch.nn.functional as F

    def bn(h, p):
        return F.batch_norm(h, st[p + ".running_mean"], st[p + ".running_var"], st[p + ".weight"], st[p + ".bias"], False, 0.0, BN_EPS)

    h = F.relu(bn(F.conv2d(x, st["resnet.conv1.weight"], None, stride=2, padding=3), "resnet.bn1"))
    h = F.max_pool2d(h, 3, 2, padding=1)
    for li, (c, nblk, stride) in enumerate(RESNET34_LAYERS, start=1):
        for b in range(nblk):
            p = f"resnet.layer{li}.{b}."
            s = stride if b == 0 else 1
            idt = h
            o = F.relu(bn(F.conv2d(h, st[p + "conv1.weight"], None, stride=s, padding=1), p + "bn1"))
            o = bn(F.conv2d(o, st[p + "conv2.weight"], None, padding=1), p + "bn2")
            if p + "downsample.0.weight" in st:
                idt = bn(F.conv2d(h, st[p + "downsample.0.weight"], None, stride=s), p + "downsample.1")
            h = F.relu(o + idt)

    def block(h, name, stride):
        return bn(F.relu(F.conv2d(h, st[name + ".0.weight"], st[name + ".0.bias"], stride=stride, padding=1)), name + ".2")

    h = F.relu(h)                                 # :88 (dropout is the identity in eval)
    x6 = block(h, "conv2d_0", 1)
    x7 = block(x6, "conv2d_01", 2)
    x8 = block(x7, "conv2d_01", 2)
    x9 = block(x8, "conv2d_02", 2)
    bs = x.shape[0]
    locs, confs = [], []
    for s, f in (("4", x7), ("2", x8), ("1", x9)):
        bb = bn(F.conv2d(f, st[f"conv2d_02_bb{s}.0.weight"], st[f"conv2d_02_bb{s}.0.bias"], padding=1), f"conv2d_02_bb{s}.1")
        cl = F.conv2d(f, st[f"conv2d_02_c{s}.weight"], st[f"conv2d_02_c{s}.bias"], padding=1)
        locs.append(bb.permute(0, 2, 3, 1).reshape(bs, -1, 4))
        confs.append(cl.permute(0, 2, 3, 1).reshape(bs, -1, 21))
    import torch
    return torch.cat(locs, 1), torch.cat(confs, 1)


def create_ancs_xywh_zoom_ratio() -> np.ndarray:
    """(189,4) f32 cx,cy,w,h  (Util.py:142-164): grids 4/2/1, nine (zoom x ratio) shapes per cell, centres at
    linspace(1/(2g), 1-1/(2g), g); the x column repeats and the y column tiles, and the function returns (y, x, w, h)
    order -- i.e. the first returned coordinate varies fastest."""
    grids = (4, 2, 1)
    zooms = (0.75, 1., 1.3)
    ratios = ((1., 1.), (1., 0.5), (0.5, 1.))
    shapes = [(z * i, z * j) for z in zooms for (i, j) in ratios]
    rows = []
    for gsz in grids:
        ctr = np.linspace(1 / (gsz * 2), 1 - 1 / (gsz * 2), gsz)
        for slow in ctr:
            for fast in ctr:
                for o, p in shapes:
                    rows.append([fast, slow, o / gsz, p / gsz])
    return np.asarray(rows, np.float64).astype(np.float32)


# ---------------------------------------------------------------------------
# mAP  (Util.py:783-885 get_map) -- SURVEY.md section 8(f) row 4
# ---------------------------------------------------------------------------
def ap_recall_thresholds() -> np.ndarray:
    """The eleven recall levels `torch.arange(0, 1.1, 0.1)` (Util.py:874) as float64 values of its float32 entries."""
    import torch
    return torch.arange(0, 1.1, 0.1).double().numpy()


def get_map(det_boxes, det_classes, det_scores, gt_boxes, gt_classes, return_details: bool = False):
    """Per-class 11-point interpolated AP over a list of images (Util.py:783-885).

    Per class: that class's detections of all images in descending score order (ties: lower flat index first --
    the reference's torch.sort is unstable, fixtures avoid ties); a detection is a true positive when its best-IoU
    ground truth of the same class in the same image (first index on ties, `:854`) has IoU > 0.5 (strict, `:855`)
    and is still unclaimed (`:856-859`); every other detection is a false positive, including those in images with
    no GT of that class (`:845-848`).  No 'difficult' handling.  precision = cumTP/(cumTP+cumFP) in float64, recall =
    float64(float32(1/n_gt)) * cumTP (see the note in the body); AP = mean over the eleven levels of max precision at recall >= level, 0 where no
    position reaches it (`:869-881`); a class without detections or without ground truth scores 0."""
    n_img = len(det_boxes)
    db = np.concatenate([np.asarray(b, np.float32).reshape(-1, 4) for b in det_boxes]) if n_img else np.zeros((0, 4), np.float32)
    dc = np.concatenate([np.asarray(c).reshape(-1).astype(np.int64) for c in det_classes]) if n_img else np.zeros(0, np.int64)
    ds = np.concatenate([np.asarray(s, np.float32).reshape(-1) for s in det_scores]) if n_img else np.zeros(0, np.float32)
    di = np.concatenate([np.full(len(np.asarray(b).reshape(-1, 4)), i, np.int64) for i, b in enumerate(det_boxes)]) if n_img else np.zeros(0, np.int64)
    gb = np.concatenate([np.asarray(b, np.float32).reshape(-1, 4) for b in gt_boxes]) if n_img else np.zeros((0, 4), np.float32)
    gc = np.concatenate([np.asarray(c).reshape(-1).astype(np.int64) for c in gt_classes]) if n_img else np.zeros(0, np.int64)
    gi = np.concatenate([np.full(len(np.asarray(b).reshape(-1, 4)), i, np.int64) for i, b in enumerate(gt_boxes)]) if n_img else np.zeros(0, np.int64)
    avail = np.ones(gb.shape[0], bool)
    tp_flag = np.zeros(db.shape[0], np.uint8)
    thr = ap_recall_thresholds()
    table = np.zeros((20, 11), np.float64)
    for cls in range(20):
        sel = np.nonzero(dc == cls)[0]
        if sel.size == 0:
            continue
        order = sel[np.lexsort((sel, -ds[sel].astype(np.float64)))]        # score descending, flat index ascending
        n_gt = int((gc == cls).sum())
        tps = []
        for d in order:
            cand = np.nonzero((gi == di[d]) & (gc == cls))[0]
            hit = False
            if cand.size:
                iou = iou_matrix(db[d:d + 1], gb[cand])[0]
                if np.isnan(iou).any():
                    best = -1                                           # torch.max propagates NaN; NaN > .5 is False
                else:
                    best = int(np.argmax(iou))                          # first index on ties
                if best >= 0 and iou[best] > np.float32(0.5) and avail[cand[best]]:
                    hit = True
                    avail[cand[best]] = False
            tps.append(1.0 if hit else 0.0)
            tp_flag[d] = 1 if hit else 0
        tps = np.asarray(tps, np.float64)
        cum_tp, cum_fp = tps.cumsum(), (1.0 - tps).cumsum()
        prec = cum_tp / (cum_tp + cum_fp)
        # `cum_TP / Objs[cls]` (Util.py:872) divides a numpy array by a 0-dim LONG tensor: torch answers through
        # Tensor.__rtruediv__ = reciprocal() * other, and the reciprocal of an integer tensor is float32 -- so the
        # recall is float64(float32(1/n_gt)) * cumTP, not cumTP/n_gt (n_gt = 0: inf * 0 = NaN, never >= a level)
        with np.errstate(divide="ignore", invalid="ignore"):
            rec = np.float64(np.float32(1.0) / np.float32(n_gt)) * cum_tp
        for t in range(11):
            m = rec >= thr[t]
            if m.any():
                table[cls, t] = prec[m].max()
    aps = {cls: np.float64(table[cls].mean()) for cls in range(20)}
    if return_details:
        return aps, tp_flag, table
    return aps


# ---------------------------------------------------------------------------
# input pipeline, deterministic part  (Dataset.py:10-13,24-39; Util.py:610-750) -- SURVEY.md section 8(f) row 3
#
# Third-party arithmetic restated here (neither ships inside the reference):
#   * Pillow (12.2.0 in this image; unpinned by the reference)  Image.resize(size, BILINEAR) on 8-bit images =
#     ImagingResample: per axis, scale = in/out, support = max(scale, 1), window [int(c - s + .5), int(c + s + .5))
#     around c = (o + .5) * scale, triangle weights in double normalised by their running sum, converted to 22-bit
#     fixed point (round half up), accumulated from 1 << 21, shifted and clamped to 0..255; horizontal pass first,
#     the vertical pass reads its 8-bit result.  Pinned against Pillow itself in tests (it is installed here and on
#     the GPU box).
#   * torchvision (absent; version unpinned) transforms.Resize -> the call above; ToTensor -> CHW float32 / 255;
#     Normalize -> (x - mean) / std in float32.  Restated from their documented behaviour: "parity unpinned" for
#     these two one-line formulas, Pillow-pinned for the resize.
# ---------------------------------------------------------------------------
PRECISION_BITS = 32 - 8 - 2
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def resample_coefficients(in_size: int, out_size: int):
    """-> (xmin (out,), count (out,), fixed-point weights (out, ksize) int32) of Pillow's bilinear precompute_coeffs."""
    scale = float(in_size) / float(out_size)
    filterscale = scale if scale > 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32)
    cnt = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for o in range(out_size):
        center = (o + 0.5) * scale
        lo = int(center - support + 0.5)
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        w = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(n):
            v = (x + lo - center + 0.5) * ss
            v = -v if v < 0 else v
            wx = 1.0 - v if v < 1.0 else 0.0
            w[x] = wx
            ww += wx
        for x in range(n):
            if ww != 0.0:
                w[x] /= ww
        for x in range(ksize):
            kk[o, x] = int(w[x] * (1 << PRECISION_BITS) - 0.5) if w[x] < 0 else int(w[x] * (1 << PRECISION_BITS) + 0.5)
        xmin[o], cnt[o] = lo, n
    return xmin, cnt, kk


def _resample_axis0(img: np.ndarray, out_size: int) -> np.ndarray:
    """8-bit pass along axis 0 of an (L, ..., C) uint8 array."""
    xmin, cnt, kk = resample_coefficients(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    src = img.astype(np.int64)
    for o in range(out_size):
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(int(cnt[o])):
            acc += src[xmin[o] + x] * int(kk[o, x])
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """(H, W, C) uint8 -> (out_h, out_w, C) uint8, Pillow's two-pass order (a pass is skipped when the size is kept)."""
    h, w = img.shape[:2]
    tmp = img if w == out_w else _resample_axis0(img.transpose(1, 0, 2), out_w).transpose(1, 0, 2)
    return tmp if h == out_h else _resample_axis0(tmp, out_h)


def compose_input(img: np.ndarray, canvas=None, crop=None, flip: bool = False, filler=None) -> np.ndarray:
    """The image the reference hands to Resize after its geometric augmentations, on 8-bit pixels:
    expand (Util.py:610-645): canvas = (canvas_h, canvas_w, top, left), filled with `filler` (the ImageNet mean after
    to_pil_image's `mul(255).byte()` truncation: 123, 116, 103); random_crop (:648-729): crop = (top, left, h, w) of the
    canvas; flip (:732-749): mirrored columns.  to_tensor -> to_pil_image is the identity on 8-bit values."""
    out = img
    if canvas is not None:
        ch, cw, top, left = canvas
        f = np.asarray(filler if filler is not None else mean_filler_u8(), np.uint8)
        big = np.empty((ch, cw, 3), np.uint8)
        big[:] = f
        big[top:top + img.shape[0], left:left + img.shape[1]] = img
        out = big
    if crop is not None:
        t, l, h, w = crop
        out = out[t:t + h, l:l + w]
    if flip:
        out = out[:, ::-1]
    return np.ascontiguousarray(out)


def mean_filler_u8() -> np.ndarray:
    import torch
    return torch.tensor(IMAGENET_MEAN, dtype=torch.float32).mul(255).byte().numpy()


def preprocess_image(img: np.ndarray, out_h: int = 300, out_w: int = 300, canvas=None, crop=None, flip: bool = False) -> np.ndarray:
    """(H, W, 3) uint8 -> (3, out_h, out_w) float32: compose_input, Resize, ToTensor, Normalize (Dataset.py:10-13,37)."""
    import torch
    u8 = resize_bilinear_u8(compose_input(img, canvas, crop, flip), out_h, out_w)
    t = torch.from_numpy(u8).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=torch.float32).view(3, 1, 1)
    return t.sub_(mean).div_(std).numpy()


# ---------------------------------------------------------------------------
# photometric_distort  (Util.py:752-780) -- SURVEY.md section 8(f) row 3, first stage of the training transform
#
# The reference calls torchvision's functional adjust_brightness / adjust_contrast / adjust_saturation / adjust_hue on a PIL
# image.  torchvision is absent; its PIL back end is four thin wrappers (restated below from its documented behaviour --
# "parity unpinned" for the wrappers) around Pillow code, and Pillow IS installed, so the arithmetic is pinned against Pillow
# itself: ImageEnhance.{Brightness,Contrast,Color}(img).enhance(f) = Image.blend(degenerate, img, f) with degenerate = black /
# solid int(mean(L) + .5) / the L image; Image.convert("L") = (19595 R + 38470 G + 7471 B + 0x8000) >> 16; Image.blend in
# float32 with truncation (clipped when f is outside [0,1]); adjust_hue = RGB -> HSV (Pillow's 8-bit conversion), H += the
# 8-bit wrap of int(hue_factor * 255), HSV -> RGB.  tests check every formula against Pillow (the two colour-space
# conversions exhaustively over all 2^24 inputs).
# ---------------------------------------------------------------------------
def rgb_to_l(a: np.ndarray) -> np.ndarray:
    r, g, b = (a[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend_u8(in1: np.ndarray, in2: np.ndarray, alpha: float) -> np.ndarray:
    """Pillow Image.blend(im1, im2, alpha) on uint8 arrays: float32 arithmetic, truncation, clipping outside [0,1]."""
    al = np.float32(alpha)
    i1 = in1.astype(np.int32)
    t = i1.astype(np.float32) + al * (in2.astype(np.int32) - i1).astype(np.float32)
    if 0.0 <= float(al) <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def rgb_to_hsv_u8(a: np.ndarray) -> np.ndarray:
    r = a[..., 0].astype(np.int32); g = a[..., 1].astype(np.int32); b = a[..., 2].astype(np.int32)
    maxc = np.maximum(r, np.maximum(g, b)); minc = np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(np.float32)
    with np.errstate(all="ignore"):
        s = cr / maxc.astype(np.float32)
        rc = (maxc - r).astype(np.float32) / cr; gc = (maxc - g).astype(np.float32) / cr; bc = (maxc - b).astype(np.float32) / cr
        h = np.where(r == maxc, (bc - gc).astype(np.float32),
                     np.where(g == maxc, (2.0 + rc.astype(np.float64) - bc.astype(np.float64)).astype(np.float32),
                              (4.0 + gc.astype(np.float64) - rc.astype(np.float64)).astype(np.float32)))
        hh = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
        uh = np.clip((hh.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
        us = np.clip((s.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    gray = minc == maxc
    return np.stack([np.where(gray, 0, uh), np.where(gray, 0, us), maxc], -1).astype(np.uint8)


def hsv_to_rgb_u8(a: np.ndarray) -> np.ndarray:
    hf = a[..., 0].astype(np.float32).astype(np.float64) * 6.0 / 255.0
    i = np.floor(hf)
    f = (hf - i).astype(np.float32).astype(np.float64)
    fs = (a[..., 1].astype(np.float64) / 255.0).astype(np.float32).astype(np.float64)
    v = a[..., 2].astype(np.int32)
    vf = v.astype(np.float64)
    rnd = lambda x: np.clip(np.floor(x + 0.5), 0, 255).astype(np.int32)          # noqa: E731  (C round(), then CLIP8)
    p, q, t = rnd(vf * (1.0 - fs)), rnd(vf * (1.0 - fs * f)), rnd(vf * (1.0 - fs * (1.0 - f)))
    k = i.astype(np.int32) % 6
    r = np.choose(k, [v, q, p, p, t, v]); g = np.choose(k, [t, v, v, q, p, p]); b = np.choose(k, [p, p, t, v, v, q])
    gray = a[..., 1] == 0
    return np.stack([np.where(gray, v, r), np.where(gray, v, g), np.where(gray, v, b)], -1).astype(np.uint8)


def hue_delta_u8(hue_factor: float) -> int:
    """torchvision adjust_hue: the uint8 added to the H plane (int(hue_factor * 255) wrapped to 8 bits)."""
    return int(hue_factor * 255) & 0xFF


def photometric_apply(img: np.ndarray, ops) -> np.ndarray:
    """ops: sequence of (kind, factor) applied in order; kind 0 brightness, 1 contrast, 2 saturation, 3 hue (Util.py:762-778)."""
    out = img
    for kind, factor in ops:
        if kind == 0:
            out = blend_u8(np.zeros_like(out), out, factor)
        elif kind == 1:
            lum = rgb_to_l(out)
            mean = int(float(lum.astype(np.int64).sum()) / float(lum.size) + 0.5)
            out = blend_u8(np.full_like(out, mean), out, factor)
        elif kind == 2:
            out = blend_u8(np.repeat(rgb_to_l(out)[..., None], 3, -1), out, factor)
        else:
            hsv = rgb_to_hsv_u8(out)
            hsv[..., 0] = (hsv[..., 0].astype(np.int32) + hue_delta_u8(factor)).astype(np.uint8)
            out = hsv_to_rgb_u8(hsv)
    return out
