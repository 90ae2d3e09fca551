"""Host-side mirror of the reference's `Dataset.py` / `Util.transform` geometry with the pixel work on the GPU
(SURVEY.md section 8(f) row 3).

The reference resizes and normalises every image on the CPU with PIL (Dataset.py:10-13,37) after its augmentations
(Util.py:566-607).  Here the host only decides the geometry -- the same `random` draws, in the same order, as
`expand` (Util.py:610-645), `random_crop` (:648-729) and `flip` (:732-749), and the same box arithmetic -- and the
images travel as raw 8-bit pixels; one kernel set (csrc/preprocess.hip) then produces the normalised
(B,3,300,300) batch, the resize being bit-identical to PIL's.

`photometric_distort` (Util.py:752-780) is drawn here too (`plan_photometric`: the same shuffle and factors as the
reference, pinned on its own function) and applied by csrc/photometric.hip.  Its arithmetic is torchvision's PIL back end,
i.e. Pillow's ImageEnhance / blend / HSV conversions, reproduced bit for bit against Pillow; torchvision itself is absent,
so its four thin wrappers are restated from their documented behaviour (see oracle/ssd_oracle.py).
"""
from __future__ import annotations

import ctypes as C
import random
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, ops
from .Util import label_to_class

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)
FILLER_U8 = (123, 116, 103)          # the ImageNet mean after to_pil_image's mul(255).byte() (Util.py:593,601)


@dataclass
class GeomPlan:
    """Geometry of one image: source (h, w) placed on a canvas, a crop window of the canvas, a mirror flag."""
    src_h: int
    src_w: int
    canvas: Tuple[int, int, int, int]          # canvas_h, canvas_w, place_top, place_left
    crop: Tuple[int, int, int, int]            # top, left, h, w (of the canvas)
    flip: bool = False
    photo: Tuple[Tuple[int, float], ...] = ()  # (kind, factor) in application order; kind 0 brightness, 1 contrast, 2 saturation, 3 hue

    @property
    def size(self) -> Tuple[int, int]:
        """(width, height) of the image handed to Resize -- what Dataset.py:35 reads as `image.size`"""
        return self.crop[3], self.crop[2]


def identity_plan(h: int, w: int) -> GeomPlan:
    return GeomPlan(h, w, (h, w, 0, 0), (0, 0, h, w), False)


def plan_photometric(rng=random) -> Tuple[Tuple[int, float], ...]:
    """The draws of reference Util.py:752-780: shuffle [brightness, contrast, saturation, hue], then per op a coin and one
    factor (hue: uniform(-18/255, 18/255), the others uniform(0.5, 1.5))."""
    order = [0, 1, 2, 3]
    rng.shuffle(order)
    out = []
    for kind in order:
        if rng.random() < 0.5:
            out.append((kind, rng.uniform(-18 / 255., 18 / 255.) if kind == 3 else rng.uniform(0.5, 1.5)))
    return tuple(out)


def _iou_1xn(crop: torch.Tensor, boxes: torch.Tensor) -> torch.Tensor:
    """reference Util.py:303-317 get_jaccard_tensor11 for one crop box"""
    lo = torch.max(crop[:2].unsqueeze(0), boxes[:, :2])
    hi = torch.min(crop[2:].unsqueeze(0), boxes[:, 2:])
    wh = torch.clamp(hi - lo, min=0)
    inter = wh[:, 0] * wh[:, 1]
    a1 = (crop[2] - crop[0]) * (crop[3] - crop[1])
    a2 = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    return inter / (a1 + a2 - inter)


def plan_transform(width: int, height: int, boxes: torch.Tensor, labels: torch.Tensor, photometric: bool = True,
                   rng=random):
    """The draws and box arithmetic of reference Util.py:566-607 `transform` for an image of the given size.
    Returns (GeomPlan, new_boxes, new_labels); boxes are pixel xyxy float32 like the reference's."""
    boxes = boxes.clone().float()
    labels = labels.clone()
    photo = plan_photometric(rng) if photometric else ()
    h, w = height, width
    canvas = (h, w, 0, 0)
    if rng.random() < .5:                      # expand, Util.py:610-645
        scale = rng.uniform(1, 4)
        new_h, new_w = int(scale * h), int(scale * w)
        left = rng.randint(0, new_w - w)
        top = rng.randint(0, new_h - h)
        canvas = (new_h, new_w, top, left)
        boxes = boxes + torch.FloatTensor([left, top, left, top]).unsqueeze(0)
    ch, cw = canvas[0], canvas[1]
    crop = (0, 0, ch, cw)
    done = False
    while not done:                            # random_crop, Util.py:648-729
        min_overlap = rng.choice([0., .1, .3, .5, .7, .9, None])
        if min_overlap is None:
            break
        for _ in range(50):
            scale_h = rng.uniform(0.3, 1)
            scale_w = rng.uniform(0.3, 1)
            new_h, new_w = int(scale_h * ch), int(scale_w * cw)
            if not 0.5 < new_h / new_w < 2:
                continue
            left = rng.randint(0, cw - new_w)
            right = left + new_w
            top = rng.randint(0, ch - new_h)
            bottom = top + new_h
            cb = torch.FloatTensor([left, top, right, bottom])
            if _iou_1xn(cb, boxes).max().item() < min_overlap:
                continue
            centers = (boxes[:, :2] + boxes[:, 2:]) / 2.
            inside = (centers[:, 0] > left) * (centers[:, 0] < right) * (centers[:, 1] > top) * (centers[:, 1] < bottom)
            if not inside.any():
                continue
            boxes = boxes[inside, :]
            labels = labels[inside]
            boxes[:, :2] = torch.max(boxes[:, :2], cb[:2])
            boxes[:, :2] -= cb[:2]
            boxes[:, 2:] = torch.min(boxes[:, 2:], cb[2:])
            boxes[:, 2:] -= cb[:2]
            crop = (top, left, new_h, new_w)
            done = True
            break
    flip = False
    if rng.random() < .5:                      # flip, Util.py:732-749
        flip = True
        img_w = crop[3]
        x0 = img_w - boxes[:, 0] - 1
        x2 = img_w - boxes[:, 2] - 1
        boxes = torch.stack((x2, boxes[:, 1], x0, boxes[:, 3]), dim=1)
    return GeomPlan(height, width, canvas, crop, flip, photo), boxes, labels


class RawBatch:
    """A batch as it leaves a DataLoader worker: the images' 8-bit pixels packed into ONE CPU uint8 tensor plus each image's
    geometry / photometric plan -- plain host data, picklable, no GPU touched (reference train.py:29,40 runs `collate_fn`
    inside `num_workers=2` forked workers, which must not initialise HIP).  The caller's own `inputs = inputs.to(device)`
    (train_function.py:61) is where the work happens, in the main process: one pinned upload, then the photometric and
    resize / normalise kernels -> the (B,3,H,W) float32 device tensor `cnn(inputs)` expects."""

    def __init__(self, arena: torch.Tensor, offsets: Sequence[int], plans: Sequence[GeomPlan], size: Tuple[int, int] = (300, 300)):
        if arena.dtype != torch.uint8 or arena.dim() != 1 or arena.is_cuda:
            raise ValueError("RawBatch: arena must be a 1-D CPU uint8 tensor")
        if len(offsets) != len(plans) or not plans:
            raise ValueError("RawBatch: one offset and one plan per image, at least one image")
        self.arena, self.offsets, self.plans, self.out_hw = arena, list(offsets), list(plans), tuple(size)

    # what train_function.py reads from `inputs` (`.shape[0]`, `.size(0)`) also works before `.to(device)`
    @property
    def shape(self) -> torch.Size:
        return torch.Size((len(self.plans), 3) + self.out_hw)

    def size(self, dim: Optional[int] = None):
        return self.shape if dim is None else self.shape[dim]

    def __len__(self) -> int:
        return len(self.plans)

    def pin_memory(self) -> "RawBatch":          # DataLoader(pin_memory=True) calls this in its pinning thread (main process)
        return RawBatch(self.arena.pin_memory(), self.offsets, self.plans, self.out_hw)

    def cuda(self, device=None) -> torch.Tensor:
        return self.to(torch.device("cuda", torch.cuda.current_device()) if device is None else device)

    def to(self, device, *args, **kwargs) -> torch.Tensor:
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("RawBatch.to(): the resize / normalise kernels run on the gfx950 GPU only (no CPU fallback); "
                               f"got device {device}")
        descs = (_lib.ImageDesc * len(self.plans))()
        for d, p, o in zip(descs, self.plans, self.offsets):
            d.src_offset, d.src_h, d.src_w = o, p.src_h, p.src_w
            d.canvas_h, d.canvas_w, d.place_top, d.place_left = p.canvas
            d.crop_top, d.crop_left, d.crop_h, d.crop_w = p.crop
            d.flip = int(p.flip)
        host = self.arena if self.arena.is_pinned() else self.arena.pin_memory()
        with torch.cuda.device(device):
            arena = host.to(device, non_blocking=True)
            if any(p.photo for p in self.plans):       # photometric_distort comes first (Util.py:586), on the source pixels
                photo = (_lib.PhotoDesc * len(self.plans))()
                for i, p in enumerate(self.plans):
                    if len(p.photo) > 4:
                        raise ValueError("at most four photometric ops per image")
                    photo[i].n_ops = len(p.photo)
                    for k, (kind, factor) in enumerate(p.photo):
                        photo[i].kind[k] = int(kind)
                        photo[i].alpha[k] = float(factor)
                        photo[i].hue_delta[k] = (int(factor * 255) & 0xFF) if kind == 3 else 0
                ops.photometric_u8(arena, descs, photo)
            return ops.preprocess_u8(arena, descs, self.out_hw, MEAN, STD, FILLER_U8)


def pack_batch(images: Sequence, plans: Optional[Sequence[GeomPlan]] = None, size: Tuple[int, int] = (300, 300)) -> RawBatch:
    """images: HWC uint8 RGB arrays / tensors / PIL images of any sizes -> RawBatch (host only; safe in a DataLoader worker)."""
    arrs = []
    for im in images:
        a = im.numpy() if torch.is_tensor(im) else np.asarray(im)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("images must be HWC uint8 RGB")
        arrs.append(np.ascontiguousarray(a))
    if not arrs:
        raise ValueError("empty batch")
    plans = [identity_plan(a.shape[0], a.shape[1]) for a in arrs] if plans is None else list(plans)
    if len(plans) != len(arrs):
        raise ValueError("one plan per image")
    offs, total = [], 0
    for a in arrs:
        offs.append(total)
        total += (a.size + 255) & ~255
    host = torch.zeros(total, dtype=torch.uint8)
    hv = host.numpy()
    for a, p, o in zip(arrs, plans, offs):
        if (p.src_h, p.src_w) != a.shape[:2]:
            raise ValueError("plan does not belong to this image")
        hv[o:o + a.size] = a.reshape(-1)
    return RawBatch(host, offs, plans, size)


def preprocess_batch(images: Sequence, plans: Optional[Sequence[GeomPlan]] = None, size: Tuple[int, int] = (300, 300),
                     device=None) -> torch.Tensor:
    """images: HWC uint8 RGB arrays / tensors / PIL images of any sizes -> (B,3,H,W) float32 on the GPU
    (Resize + ToTensor + Normalize of Dataset.py:10-13 after the plans' geometry).  One host->device copy."""
    if not torch.cuda.is_available():
        raise RuntimeError("preprocess_batch() runs on the gfx950 HIP kernels only (no CPU fallback)")
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    return pack_batch(images, plans, size).to(device)


class MultiImageMultiBBoxDataset(torch.utils.data.Dataset):
    """Reference Dataset.py:7-39 with the same constructor.  `__getitem__` returns the RAW 8-bit image and its
    geometry plan instead of a normalised tensor: `(RawImage, classes, standardized_bbox, index)` -- host work only, as it
    runs in DataLoader workers; `collate_fn` packs a batch's RawImages into a `RawBatch` whose `.to(device)` (the caller's
    own line train_function.py:61, main process) makes the normalised device tensor."""

    def __init__(self, all_images, all_multi_bboxes, all_multi_labels, all_difficulties, all_indices, isTest=False,
                 keep_difficult=False):
        self.images_list = all_images
        self.multi_labels_list = all_multi_labels
        self.isTest = isTest
        self.multi_bbox_list = all_multi_bboxes
        self.all_difficulties = [torch.tensor(i) for i in all_difficulties]
        self.keep_difficult = keep_difficult
        self.all_indices = all_indices

    def __len__(self):
        return len(self.images_list)

    def __getitem__(self, index):
        from PIL import Image
        from .Util import transform
        image = RawImage.of(Image.open(self.images_list[index]).convert("RGB"))
        c = torch.Tensor([label_to_class[i] for i in self.multi_labels_list[index]])
        bboxes = torch.Tensor(self.multi_bbox_list[index])
        if self.keep_difficult is False:
            keep = self.all_difficulties[index] == 0
            bboxes, c = bboxes[keep], c[keep]
        if self.isTest is False:
            image, bboxes, c = transform(image, bboxes, c)            # Dataset.py:33
        w, h = image.size                                             # Dataset.py:35
        standardized_bbox = bboxes / torch.FloatTensor([w, h, w, h]).unsqueeze(0)
        return image, c, standardized_bbox, self.all_indices[index]


@dataclass
class RawImage:
    """8-bit HWC pixels of one image and the geometry / photometric plan drawn for it; `.size` = (width, height) of the
    augmented image, what the reference reads from its PIL image at Dataset.py:35."""
    pixels: np.ndarray
    plan: GeomPlan

    @staticmethod
    def of(image) -> "RawImage":
        if isinstance(image, RawImage):
            return image
        a = np.ascontiguousarray(np.asarray(image.convert("RGB") if hasattr(image, "convert") else image))
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("image must be a PIL image or an HWC uint8 RGB array")
        return RawImage(a, identity_plan(a.shape[0], a.shape[1]))

    @property
    def size(self) -> Tuple[int, int]:
        return self.plan.size


def collate_fn(batch):
    """Reference Dataset.py:41-53: (images, classes, boxes, indices) of a list of samples.  Host work only (it runs in the
    DataLoader's workers): RawImage entries are packed into one `RawBatch`; already-normalised tensors are stacked like the
    reference does."""
    images, classes, boxes, indices = [], [], [], []
    for b in batch:
        images.append(b[0]); classes.append(b[1]); boxes.append(b[2]); indices.append(b[3])
    if images and isinstance(images[0], RawImage):
        x = pack_batch([r.pixels for r in images], [r.plan for r in images])
    else:
        x = torch.stack(images, dim=0)
    return x, classes, boxes, indices
