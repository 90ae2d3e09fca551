"""`Losses.ssd` / `Losses.inference` with the reference's call signatures
(reference Losses.py:119-134 and :11-98), computed by libssd_gfx950.so.

`ssd(outputs, tr_classes, tr_bboxs) -> (loc_loss, conf_loss)`: matching, L1 +
cross-entropy MultiBox loss with 3:1 hard-negative mining and the gradients w.r.t.
(loc, conf), all on the device: no python loop over the batch, no host round trip.
`inference(l_, c_, index, ...)`: decode + softmax + per-class NMS + top-k.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from . import ops
from .Util import all_images, class_to_label, create_priors_ssd300, create_priors_ssd512, device, xywh_to_xyxy  # noqa: F401

# module-level priors, as reference Losses.py:6-7 (immutable after import)
ancs_xywh = create_priors_ssd300()
ancs_xyxy = xywh_to_xyxy(ancs_xywh)

IOU_THRESHOLD = 0.5      # Losses.py:171
NEG_POS_RATIO = 3        # Losses.py:189

_dev_priors: Dict[str, tuple] = {}
# Matching of the most recent ssd() call: dict(obj (bs,P) int32 global GT index, cls (bs,P) int32).
# (The reference leaves the per-prior classes in a module global too: Losses.py:172-173.)
last_match: Optional[dict] = None


def _priors_on(dev: torch.device, n_priors: int = 8732):
    """device copies of the prior set with `n_priors` boxes: 8732 = the reference's SSD300 set (module globals
    above), 24564 = the build-defined SSD512 set"""
    key = (str(dev), n_priors)
    if key not in _dev_priors:
        if n_priors == ancs_xywh.shape[0]:
            cx, xy = ancs_xywh, ancs_xyxy
        else:
            cx = create_priors_ssd512()
            if n_priors != cx.shape[0]:
                raise ValueError(f"no prior set with {n_priors} boxes (8732 = SSD300, {cx.shape[0]} = SSD512)")
            xy = xywh_to_xyxy(cx)
        _dev_priors[key] = (cx.to(dev).contiguous(), xy.to(dev).contiguous())
    return _dev_priors[key]


class _MultiBoxLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loc, conf, gt, gt_cls, img_start, norm_mode):
        pri, pri_xyxy = _priors_on(loc.device, loc.shape[1])
        want = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        out = ops.multibox_loss(loc.detach().contiguous(), conf.detach().contiguous(), gt, gt_cls, img_start, pri, pri_xyxy,
                                IOU_THRESHOLD, NEG_POS_RATIO, norm_mode, want_grads=want)
        ctx.dloc, ctx.dconf = out["dloc"], out["dconf"]
        global last_match
        last_match = dict(obj=out["obj"], cls=out["cls"], n_pos=out["losses"][2])
        losses = out["losses"]
        return losses[0].clone(), losses[1].clone(), losses[2].clone()

    @staticmethod
    def backward(ctx, g_loc, g_conf, _g_npos):
        # loc_loss depends on loc only, conf_loss on conf only
        return ctx.dloc * g_loc, ctx.dconf * g_conf, None, None, None, None


def _image_starts(start, dev: torch.device) -> torch.Tensor:
    """Device copy of the per-image box offsets.  A plain `.to(dev)` of a pageable host tensor makes the host wait for everything
    queued on the stream -- in the train step that drains the whole forward and leaves the loss and the small head / aux backward
    kernels launch-bound behind it -- so the offsets go up from pinned memory without blocking."""
    host = torch.tensor(start, dtype=torch.int32)
    if dev.type == "cuda":
        return host.pin_memory().to(dev, non_blocking=True)
    return host.to(dev)


def _pack_targets(tr_classes: Sequence[torch.Tensor], tr_bboxs: Sequence[torch.Tensor], dev: torch.device):
    counts = [int(b.shape[0]) for b in tr_bboxs]
    if len(counts) == 0 or any(c == 0 for c in counts):
        # the reference raises here too (max over an empty dimension, Losses.py:153)
        raise ValueError("ssd(): every image needs at least one ground-truth box")
    if len(tr_classes) != len(counts) or any(int(c.shape[0]) != n for c, n in zip(tr_classes, counts)):
        raise ValueError("ssd(): tr_classes and tr_bboxs disagree")
    start = [0]
    for c in counts:
        start.append(start[-1] + c)
    gt = torch.cat([b.reshape(-1, 4) for b in tr_bboxs]).to(device=dev, dtype=torch.float32).contiguous()
    cls = torch.cat([c.reshape(-1) for c in tr_classes]).to(device=dev, dtype=torch.float32).contiguous()
    return gt, cls, _image_starts(start, dev)


def ssd(outputs, tr_classes, tr_bboxs, norm_mode: int = 0, with_n_pos: bool = False):
    """outputs = (loc (bs,8732,4), conf (bs,8732,21)); tr_classes: list of (n_i,) float tensors with values
    0..19; tr_bboxs: list of (n_i,4) xyxy fractional boxes.  Returns (loc_loss, conf_loss) as 0-dim tensors
    that support `+`, `.item()` and `.backward()` (train_function.py:82-94).
    with_n_pos: also return this call's number of positive priors (0-dim device tensor) -- the data-parallel step
    (ddp.py) takes it from here, not from the module global `last_match`, so two models / threads cannot mix theirs up."""
    loc, conf = outputs
    if loc.dim() != 3 or conf.dim() != 3 or loc.shape[0] != len(tr_bboxs) or loc.shape[1] not in (ancs_xywh.shape[0], 24564):
        raise ValueError(f"ssd(): outputs {tuple(loc.shape)}, {tuple(conf.shape)} do not match {len(tr_bboxs)} images "
                         f"x {ancs_xywh.shape[0]} priors")
    gt, cls, img_start = _pack_targets(tr_classes, tr_bboxs, loc.device)
    if not loc.is_cuda:
        raise RuntimeError("ssd() runs on the gfx950 HIP kernels only (no CPU fallback): outputs must be device tensors")
    l_loc, l_conf, n_pos = _MultiBoxLoss.apply(loc, conf, gt, cls, img_start, norm_mode)
    if with_n_pos:
        return l_loc, l_conf, n_pos.detach()
    return l_loc, l_conf


def _image_size(index, phase):
    if isinstance(index, (tuple, list)) and len(index) == 2:
        return float(index[0]), float(index[1])
    path = all_images[phase][index]                       # reference Losses.py:87 reads the image file
    from PIL import Image
    with Image.open(path) as im:
        return float(im.size[0]), float(im.size[1])


draw_hook = None      # optional callable(image_path_or_size, boxes, labels, probs); drawing itself is out of scope


def inference(l_, c_, index, top_k=200, phase='train', toDraw=True, min_score=0.2, iou_threshold=0.45):
    """l_ (8732,4) predicted offsets, c_ (8732,21) class scores of ONE image.  `index` is either the
    reference's dataset index (image size read from `all_images[phase][index]`) or an (img_w, img_h)
    pair.  Returns (boxes (K,4) pixel xyxy, classes (K,) int64, probs (K,)) with K <= top_k, or
    ([], [], []) when nothing reaches min_score (reference Losses.py:62-63).  Drawing (Losses.py:92-97) is
    out of scope: with toDraw the optional module-level `draw_hook` is called, nothing otherwise."""
    if not l_.is_cuda:
        raise RuntimeError("inference() runs on the gfx950 HIP kernels only (no CPU fallback)")
    w, h = _image_size(index, phase)
    pri, _ = _priors_on(l_.device, l_.shape[0])
    boxes, classes, probs, ids, count = ops.decode_nms(l_.detach().float().contiguous(), c_.detach().float().contiguous(), pri,
                                                      w, h, top_k, min_score, iou_threshold)
    k = int(count.item())
    if k == 0:
        return [], [], []
    inference.last_prior_ids = ids[:k]
    if toDraw and draw_hook is not None:
        draw_hook(index, boxes[:k], [class_to_label[int(i)] for i in classes[:k].tolist()], probs[:k])
    return boxes[:k], classes[:k], probs[:k]


inference.last_prior_ids = None


def inference_batch_padded(l, c, sizes, top_k=200, min_score=0.2, iou_threshold=0.45):
    """The batched decode as it leaves the device: (boxes (B,top_k,4) pixel xyxy, classes (B,top_k) int64, probs (B,top_k),
    prior_ids (B,top_k) int32, count (B,) int32), rows i >= count[b] zero.  Four kernels + one memset, NO host synchronisation:
    the primary form for a serving loop (slice on the host only when the detections are consumed there).  `sizes`: a (B,2) device
    tensor of (img_w, img_h) is used as it is; anything else is uploaded."""
    if not l.is_cuda:
        raise RuntimeError("inference_batch_padded() runs on the gfx950 HIP kernels only (no CPU fallback)")
    if torch.is_tensor(sizes) and sizes.is_cuda and sizes.dtype == torch.float32 and sizes.is_contiguous():
        wh = sizes.reshape(-1, 2)
    else:
        wh = torch.as_tensor(sizes, dtype=torch.float32).reshape(-1, 2).to(l.device).contiguous()
    pri, _ = _priors_on(l.device, l.shape[1])
    return ops.decode_nms_batch(l.detach().float().contiguous(), c.detach().float().contiguous(), pri, wh, top_k, min_score, iou_threshold)


def inference_batch(l, c, sizes, top_k=200, min_score=0.2, iou_threshold=0.45):
    """Batched `inference`: l (B,8732,4), c (B,8732,21), sizes = B (img_w, img_h)
    pairs or a (B,2) tensor.  `inference_batch_padded` + ONE host sync (the counts) for the whole batch.  Returns a list of
    B tuples (boxes (K_i,4), classes (K_i,), probs (K_i,)); an image without detections gives ([], [], [])."""
    boxes, classes, probs, ids, count = inference_batch_padded(l, c, sizes, top_k, min_score, iou_threshold)
    out = []
    for i, k in enumerate(count.tolist()):
        out.append(([], [], []) if k == 0 else (boxes[i, :k], classes[i, :k], probs[i, :k]))
    inference_batch.last_prior_ids = [ids[i, :k] for i, k in enumerate(count.tolist())]
    return out


inference_batch.last_prior_ids = None
