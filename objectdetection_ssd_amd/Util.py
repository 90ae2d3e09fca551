"""Host-side mirror of the box utilities the hot path imports from the
reference's Util.py (names and argument meaning kept; bodies are ours).

What `Losses.py` / `train_function.py` pull in through `from Util import *` and what
`train.py:6` / `Dataset.py:4` import by name lives here: prior boxes, box coders, the
class table, `device`, the (empty) VOC lists, `transform` (as a geometry plan: the
pixels are made on the GPU) and `get_map`.  VOC XML parsing and drawing are out of
scope.
"""
from __future__ import annotations

from math import sqrt

import torch

use_cuda = torch.cuda.is_available()
device = torch.device("cuda" if use_cuda else "cpu")

# reference Util.py:26-27 -- background is the last entry
class_to_label = ['aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow',
                  'diningtable', 'dog', 'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train',
                  'tvmonitor', 'bg']
label_to_class = {name: i for i, name in enumerate(class_to_label)}

# The VOC lists reference Util.py:16 re-exports from DataLists.py (train.py:6 imports all four): filled by callers that have a
# dataset; empty here (the VOC XML walk is host I/O outside the path -- SURVEY.md section 2 row 13)
all_images = {"train": [], "test": []}
all_multi_bboxes = {"train": [], "test": []}
all_multi_labels = {"train": [], "test": []}
all_difficulties = {"train": [], "test": []}

_GRID = (38, 19, 10, 5, 3, 1)
_SCALE = (0.1, 0.2, 0.375, 0.55, 0.725, 0.9)
_RATIO = ((1., 2., .5), (1., 2., 3., .5, .333), (1., 2., 3., .5, .333), (1., 2., 3., .5, .333), (1., 2., .5), (1., 2., .5))
ANCHORS_PER_CELL = (4, 6, 6, 6, 4, 4)


def create_priors_ssd300() -> torch.Tensor:
    """(8732,4) f32 cx,cy,w,h  (reference Util.py:105-137): per grid cell, row-major,
    boxes (s*sqrt(a), s/sqrt(a)) for each ratio with the sqrt(s_k*s_k+1) square right
    after a == 1; double arithmetic, rounded to f32, clamped to [0,1]."""
    out = []
    for k, g in enumerate(_GRID):
        s = _SCALE[k]
        extra = sqrt(s * _SCALE[k + 1]) if k + 1 < len(_SCALE) else 1.
        for row in range(g):
            cy = (row + 0.5) / float(g)
            for col in range(g):
                cx = (col + 0.5) / float(g)
                for a in _RATIO[k]:
                    out.append([cx, cy, s * sqrt(a), s / sqrt(a)])
                    if a == 1.:
                        out.append([cx, cy, extra, extra])
    return torch.tensor(out, dtype=torch.float64).to(torch.float32).clamp_(0, 1)


# SSD512 is NOT in the reference: build-defined extension in the reference's
# style -- seven maps 64..1, the standard SSD512 scales, the reference's per-cell ratio lists and rounding.
_GRID512 = (64, 32, 16, 8, 4, 2, 1)
_SCALE512 = (0.07, 0.15, 0.30, 0.45, 0.60, 0.75, 0.90)
_RATIO512 = ((1., 2., .5),) + ((1., 2., 3., .5, .333),) * 4 + ((1., 2., .5),) * 2
ANCHORS_PER_CELL_512 = (4, 6, 6, 6, 6, 4, 4)


def create_priors_ssd512() -> torch.Tensor:
    """(24564,4) f32 cx,cy,w,h: create_priors_ssd300's construction on the SSD512 grids (build-defined)."""
    out = []
    for k, g in enumerate(_GRID512):
        s = _SCALE512[k]
        extra = sqrt(s * _SCALE512[k + 1]) if k + 1 < len(_SCALE512) else 1.
        for row in range(g):
            cy = (row + 0.5) / float(g)
            for col in range(g):
                cx = (col + 0.5) / float(g)
                for a in _RATIO512[k]:
                    out.append([cx, cy, s * sqrt(a), s / sqrt(a)])
                    if a == 1.:
                        out.append([cx, cy, extra, extra])
    return torch.tensor(out, dtype=torch.float64).to(torch.float32).clamp_(0, 1)


def xywh_to_xyxy(box: torch.Tensor) -> torch.Tensor:
    """reference Util.py:93-96"""
    return torch.cat((box[:, :2] - box[:, 2:] / 2., box[:, :2] + box[:, 2:] / 2.), dim=1)


def xyxy_to_xywh(box: torch.Tensor) -> torch.Tensor:
    """reference Util.py:57-63 (without its host round trip)"""
    return torch.stack(((box[:, 2] + box[:, 0]) / 2., (box[:, 3] + box[:, 1]) / 2.,
                        box[:, 2] - box[:, 0], box[:, 3] - box[:, 1]), dim=1)


def gcxgcy_to_cxcy(gcxgcy: torch.Tensor, priors_cxcy: torch.Tensor) -> torch.Tensor:
    """reference Util.py:86-91"""
    priors_cxcy = priors_cxcy.to(gcxgcy.device)
    return torch.cat([gcxgcy[:, :2] * priors_cxcy[:, 2:] / 10 + priors_cxcy[:, :2],
                      torch.exp(gcxgcy[:, 2:] / 5) * priors_cxcy[:, 2:]], 1)


def get_offsets_coords(cxcy: torch.Tensor, priors_cxcy: torch.Tensor) -> torch.Tensor:
    """reference Util.py:98-102"""
    priors_cxcy = priors_cxcy.to(cxcy.device)
    return torch.cat([(cxcy[:, :2] - priors_cxcy[:, :2]) / (priors_cxcy[:, 2:] / 10),
                      torch.log(cxcy[:, 2:] / priors_cxcy[:, 2:]) * 5], 1)


def subsampling(x: torch.Tensor, step) -> torch.Tensor:
    """keep every step[d]-th entry along dim d (None = keep all); reference Util.py:555-560"""
    for d, s in enumerate(step):
        if s is not None:
            x = x.index_select(d, torch.arange(0, x.shape[d], s, device=x.device))
    return x


def transform(image, boxes, labels):
    """Reference Util.py:566-607 (`from Util import transform`, Dataset.py:4): photometric distortion, expand, random crop,
    flip, with the reference's `random` draws in its order and its box arithmetic -- as a PLAN.  `image` is a PIL image, an HWC
    uint8 array or a `Dataset.RawImage`; the returned image is a `Dataset.RawImage` (source pixels + plan, `.size` = the
    augmented (width, height)): the pixels are produced later, on the GPU, by `Dataset.RawBatch.to(device)`."""
    from .Dataset import RawImage, plan_transform
    raw = RawImage.of(image)
    h, w = raw.pixels.shape[:2]
    plan, new_boxes, new_labels = plan_transform(w, h, boxes, labels)
    return RawImage(raw.pixels, plan), new_boxes, new_labels


def create_ancs_xywh_zoom_ratio() -> torch.Tensor:
    """(189,4) f32 anchors of the SSD_resnet34 variant (reference Util.py:142-164): grids 4/2/1, nine zoom x ratio
    shapes per cell, centres at linspace(1/(2g), 1-1/(2g), g); the first returned coordinate varies fastest."""
    import numpy as np
    shapes = [(z * i, z * j) for z in (0.75, 1., 1.3) for (i, j) in ((1., 1.), (1., 0.5), (0.5, 1.))]
    rows = []
    for g in (4, 2, 1):
        ctr = np.linspace(1 / (g * 2), 1 - 1 / (g * 2), g)
        rows += [[fast, slow, o / g, p / g] for slow in ctr for fast in ctr for o, p in shapes]
    return torch.tensor(np.asarray(rows, np.float64), dtype=torch.float32)


def get_map(det_boxes, det_classes, det_scores, gt_boxes, gt_classes):
    """Per-class 11-point interpolated AP (reference Util.py:783-885; same arguments: per-image lists of (n,4) boxes,
    (n,) classes, (n,) scores and the ground-truth boxes / classes).  Matching, the per-class sort and the
    precision/recall scan run on the GPU (csrc/map_eval.hip); returns {class: numpy.float64 AP} like the reference
    (which also prints each value).  Score ties are ordered lower flat index first."""
    import numpy as np
    from . import ops
    if not torch.cuda.is_available():
        raise RuntimeError("get_map() runs on the gfx950 HIP kernels only (no CPU fallback)")
    n_img = len(det_boxes)
    if not (n_img == len(det_classes) == len(det_scores) == len(gt_boxes) == len(gt_classes)) or n_img == 0:
        raise ValueError("get_map expects five lists with one entry per image")
    dev = next((t.device for t in list(det_boxes) + list(gt_boxes) if torch.is_tensor(t) and t.is_cuda), device)

    def flat(items, dtype, width):
        parts = [torch.as_tensor(t).reshape((-1, width) if width else (-1,)).to(device=dev, dtype=dtype) for t in items]
        start = torch.tensor([0] + [int(p.shape[0]) for p in parts], dtype=torch.int64).cumsum(0).to(device=dev, dtype=torch.int32)
        return torch.cat(parts).contiguous(), start

    db, d_start = flat(det_boxes, torch.float32, 4)
    dc, _ = flat(det_classes, torch.int32, 0)
    ds, _ = flat(det_scores, torch.float32, 0)
    gb, g_start = flat(gt_boxes, torch.float32, 4)
    gc, _ = flat(gt_classes, torch.int32, 0)
    levels = torch.arange(0, 1.1, 0.1).double().numpy()                # Util.py:874, float32 levels compared in float64
    table, _, _ = ops.map_eval(db, dc, ds, d_start, gb, gc, g_start, levels, 20)
    t = table.cpu().numpy()
    return {cls: np.float64(np.mean(t[cls])) for cls in range(20)}
