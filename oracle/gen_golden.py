"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

TEST INFRASTRUCTURE.  Imports the reference's unmodified Util.py / Losses.py /
Model.py from /root/reference on CPU and records input/output vectors.  The
reference never travels to the GPU box; only the vectors written here do.

Two import-time obstacles are bridged before the import, exactly as recorded in
SURVEY.md section 8(c):
  * ``torchvision`` is not installed -> an in-memory module object whose
    ``transforms.*`` are no-ops and whose ``models.vgg16`` returns the standard
    VGG-16 'D' layer list built from ``torch.nn`` (seeded random weights; the
    pretrained weights are a network download and are not available).  No
    arithmetic on the hot path comes from torchvision: convolution, pooling,
    ReLU are ``torch.nn``.
  * ``DataLists.call_on_load`` reads VOC files that do not exist -> a module
    object with empty lists (plus one generated PNG so that
    ``Losses.inference`` can read an image size, Losses.py:87).
Consequently network parity is pinned on seeded random weights and
"torchvision VGG-16 layout" itself is unpinned.

Usage (build container):  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import ssd_oracle as O  # noqa: E402

IMG_W, IMG_H = 500, 375
IMG_PATH = "/tmp/_ssd_golden_blank.png"


def _install_stand_ins():
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    ft = types.ModuleType("torchvision.transforms.functional")
    md = types.ModuleType("torchvision.models")

    class _NoOp:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x
    for n in ("Compose", "Resize", "ToTensor", "Normalize"):
        setattr(tr, n, _NoOp)

    def vgg16(pretrained=False, **kw):
        cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
        layers, cin = [], 3
        for v in cfg:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        m = nn.Module()
        m.features = nn.Sequential(*layers)
        m.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        m.classifier = nn.Sequential(nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), nn.Dropout(),
                                     nn.Linear(4096, 4096), nn.ReLU(True), nn.Dropout(),
                                     nn.Linear(4096, 1000))
        return m
    md.vgg16 = vgg16
    ft.hflip = lambda im: im          # Util.flip (Util.py:732-749) is called for its BOX arithmetic only; see write_augment

    def resnet34(pretrained=False, **kw):
        """The standard ResNet-34 layer list built from torch.nn (torchvision is not installed): only the module tree /
        children() order matters to the reference (Model.py:21-30); weights are overwritten with the seeded state."""
        class BasicBlock(nn.Module):
            def __init__(self, cin, c, stride):
                super().__init__()
                self.conv1 = nn.Conv2d(cin, c, 3, stride, 1, bias=False)
                self.bn1 = nn.BatchNorm2d(c)
                self.relu = nn.ReLU(inplace=True)
                self.conv2 = nn.Conv2d(c, c, 3, 1, 1, bias=False)
                self.bn2 = nn.BatchNorm2d(c)
                self.downsample = None
                if stride != 1 or cin != c:
                    self.downsample = nn.Sequential(nn.Conv2d(cin, c, 1, stride, bias=False), nn.BatchNorm2d(c))

            def forward(self, x):
                idt = x if self.downsample is None else self.downsample(x)
                out = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
                return self.relu(out + idt)

        m = nn.Module()
        m.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        m.bn1 = nn.BatchNorm2d(64)
        m.relu = nn.ReLU(inplace=True)
        m.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for li, (c, nblk, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), start=1):
            blocks = []
            for b in range(nblk):
                blocks.append(BasicBlock(cin, c, stride if b == 0 else 1))
                cin = c
            setattr(m, f"layer{li}", nn.Sequential(*blocks))
        m.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        m.fc = nn.Linear(512, 1000)
        return m
    md.resnet34 = resnet34
    tr.functional = ft
    tv.transforms = tr
    tv.models = md
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tr,
                        "torchvision.transforms.functional": ft, "torchvision.models": md})
    from PIL import Image
    Image.new("RGB", (IMG_W, IMG_H)).save(IMG_PATH)
    dl = types.ModuleType("DataLists")
    dl.call_on_load = lambda: None
    for n in ("all_images", "all_multi_labels", "all_multi_bboxes", "all_difficulties"):
        setattr(dl, n, {"train": [IMG_PATH], "test": [IMG_PATH]})
    sys.modules["DataLists"] = dl
    sys.path.insert(0, REF)


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


# ---------------------------------------------------------------------------
# synthetic ground truth -- numpy PCG64 is stable across
# platforms, so tests regenerate inputs from the seed instead of storing them.
# ---------------------------------------------------------------------------
def synth_gt(rng, bs, max_extra=7):
    boxes, classes = [], []
    for _ in range(bs):
        n = 1 + min(int(rng.poisson(1.4)), max_extra)
        x1 = rng.uniform(0, .6, n); y1 = rng.uniform(0, .6, n)
        w = rng.uniform(.08, .6, n); h = rng.uniform(.08, .6, n)
        b = np.stack([x1, y1, np.minimum(x1 + w, 1.), np.minimum(y1 + h, 1.)], 1).astype(np.float32)
        boxes.append(b)
        classes.append(rng.integers(0, 20, n).astype(np.float32))
    return boxes, classes


def edge_cases(pri_xyxy):
    """Hand-made matching cases for the tie rules."""
    f = np.float32
    cases = []
    # 1. a tiny GT that overlaps nothing by >= .5, next to a large one (forced match only)
    cases.append(([np.array([[.30, .30, .33, .34], [.1, .1, .8, .9]], f)], [np.array([3., 7.], f)]))
    # 2. duplicate GT boxes with different classes (ties: first index for per-prior argmax,
    #    last GT wins the forced match)
    cases.append(([np.array([[.2, .2, .6, .7], [.2, .2, .6, .7], [.5, .5, .9, .9]], f)],
                  [np.array([1., 2., 3.], f)]))
    # 3. two different GT whose best prior is the same prior (collision -> last wins)
    p = pri_xyxy[8000]
    cases.append(([np.stack([p + f(.004), p - f(.003)]).astype(f).clip(0, 1)], [np.array([5., 9.], f)]))
    # 4. boxes touching / spanning the image border, and a whole-image box
    cases.append(([np.array([[0., 0., .25, .3], [.7, .65, 1., 1.], [0., 0., 1., 1.]], f)],
                  [np.array([0., 19., 10.], f)]))
    # 5. GT exactly equal to prior boxes (IoU == 1 exactly) in a 2-image batch;
    #    second image's GT is disjoint from most priors (all-zero IoU columns)
    cases.append(([pri_xyxy[[100, 6000]].clip(0, 1).astype(f), np.array([[.9, .9, .95, .95]], f)],
                  [np.array([4., 6.], f), np.array([11.], f)]))
    # 6. many GT in one image, one in the other (ragged batch)
    rng = np.random.default_rng(77)
    b, c = synth_gt(rng, 1, max_extra=7)
    many = np.concatenate([b[0]] + [synth_gt(rng, 1)[0][0] for _ in range(4)])[:12]
    cases.append(([many, np.array([[.4, .4, .6, .6]], f)],
                  [rng.integers(0, 20, many.shape[0]).astype(f), np.array([2.], f)]))
    return cases


def map_cases():
    """Detections / ground truth for the mAP fixture: jittered copies of the GT (true positives at varying IoU),
    duplicates of the same GT (the second is a false positive), random boxes, classes without detections and classes
    without ground truth; scores are distinct so the reference's unstable sort has one answer."""
    cases = []
    for seed, n_img, p_det, n_fp in ((11, 6, .8, 3), (12, 24, .6, 6), (13, 3, 1., 0), (14, 40, .7, 10)):
        rng = np.random.default_rng(seed)
        gtb, gtc = synth_gt(rng, n_img)
        gtc = [c.astype(np.int64) % (20 if seed != 13 else 4) for c in gtc]
        dets_b, dets_c, dets_s = [], [], []
        for b, c in zip(gtb, gtc):
            bb, cc = [], []
            for k in range(len(b)):
                if rng.uniform() < p_det:
                    jit = rng.normal(0, rng.choice([.01, .05, .12]), 4).astype(np.float32)
                    bb.append(np.clip(b[k] + jit, 0, 1)); cc.append(c[k])
                    if rng.uniform() < .3:                       # duplicate detection of the same object
                        bb.append(np.clip(b[k] + rng.normal(0, .01, 4).astype(np.float32), 0, 1)); cc.append(c[k])
                    if rng.uniform() < .2:                       # right box, wrong class
                        bb.append(b[k].copy()); cc.append((c[k] + 1) % 20)
            for _ in range(int(rng.integers(0, n_fp + 1))):
                x1, y1 = rng.uniform(0, .7, 2)
                bb.append(np.asarray([x1, y1, x1 + rng.uniform(.05, .3), y1 + rng.uniform(.05, .3)], np.float32))
                cc.append(int(rng.integers(0, 20)))
            bb = np.asarray(bb, np.float32).reshape(-1, 4)
            bb = np.stack([np.minimum(bb[:, 0], bb[:, 2]), np.minimum(bb[:, 1], bb[:, 3]),
                           np.maximum(bb[:, 0], bb[:, 2]) + np.float32(1e-3), np.maximum(bb[:, 1], bb[:, 3]) + np.float32(1e-3)], 1) \
                if len(bb) else bb
            dets_b.append(bb.astype(np.float32)); dets_c.append(np.asarray(cc, np.int64))
            dets_s.append(np.zeros(len(cc), np.float32))
        total = sum(len(c) for c in dets_c)
        scores = rng.permutation(total).astype(np.float32) / np.float32(total + 1) + np.float32(.01)   # distinct
        o = 0
        for i in range(n_img):
            dets_s[i] = scores[o:o + len(dets_c[i])]; o += len(dets_c[i])
        if seed == 12:                                            # pixel coordinates, as `inference` returns them
            dets_b = [b * np.float32([500, 375, 500, 375]) for b in dets_b]
            gtb = [b * np.float32([500, 375, 500, 375]) for b in gtb]
        cases.append((dets_b, dets_c, dets_s, gtb, gtc))
    return cases


def write_photometric(RU):
    """The draws of Util.photometric_distort (Util.py:752-780): the reference's own function runs on seeded `random`
    streams with torchvision's four adjust_* functions replaced by recorders (torchvision is absent; the recorders keep
    the __name__ the reference tests at Util.py:770), so the fixture holds the order and the factors it would apply."""
    import random
    log = []

    def rec(kind, name):
        def fn(img, factor):
            log.append((kind, float(factor)))
            return img
        fn.__name__ = name
        return fn
    ft = RU.FT
    ft.adjust_brightness = rec(0, "adjust_brightness")
    ft.adjust_contrast = rec(1, "adjust_contrast")
    ft.adjust_saturation = rec(2, "adjust_saturation")
    ft.adjust_hue = rec(3, "adjust_hue")
    store = {}
    n = 64
    for ci in range(n):
        log.clear()
        random.seed(12000 + ci)
        RU.photometric_distort(object())
        tail = random.random()                     # the stream position after the call
        store[f"c{ci}_kinds"] = np.asarray([k for k, _ in log], np.int64)
        store[f"c{ci}_factors"] = np.asarray([f for _, f in log], np.float64)
        store[f"c{ci}_next"] = np.float64(tail)
    store["n_cases"] = np.int64(n)
    np.savez_compressed(os.path.join(GOLD, "photometric_draws.npz"), **store)


def write_degenerate(RL):
    """Losses.ssd on degenerate ground truth: zero-area and zero-height boxes.  Their IoU with
    every prior is 0, so they are matched only through the forced match (first prior on the all-zero row, Losses.py:157-160);
    the width/height target is log(0) = -inf, the L1 loss inf, the gradients stay finite (sign(+inf) = 1)."""
    f = np.float32
    cases = [
        ([np.array([[.4, .4, .4, .4], [.2, .3, .7, .8]], f)], [np.array([5., 2.], f)], 4000),
        ([np.array([[.1, .5, .6, .5]], f), np.array([[.3, .3, .9, .8]], f)], [np.array([7.], f), np.array([1.], f)], 4001),
        ([np.array([[.25, .25, .25, .25]], f)], [np.array([19.], f)], 4002),
    ]
    store = {"n_cases": np.int64(len(cases))}
    for ci, (boxes, classes, seed) in enumerate(cases):
        bs = len(boxes)
        r = np.random.default_rng(seed)
        loc = r.standard_normal((bs, 8732, 4), dtype=np.float32)
        conf = (r.standard_normal((bs, 8732, 21), dtype=np.float32) * np.float32(2.0))
        lt = torch.from_numpy(loc).requires_grad_(True)
        ct = torch.from_numpy(conf).requires_grad_(True)
        with quiet():
            l_loc, l_conf = RL.ssd((lt, ct), [torch.from_numpy(c) for c in classes], [torch.from_numpy(b) for b in boxes])
        cls = RL.obj_forEach_prior___.numpy().astype(np.int8)
        (l_loc + l_conf).backward()
        p = f"c{ci}_"
        store[p + "seed"] = np.int64(seed)
        store[p + "counts"] = np.asarray([b.shape[0] for b in boxes], np.int64)
        store[p + "boxes"] = np.concatenate(boxes).astype(np.float32)
        store[p + "classes"] = np.concatenate(classes).astype(np.float32)
        store[p + "cls"] = cls
        store[p + "loc_loss"] = np.float32(l_loc.item())
        store[p + "conf_loss"] = np.float32(l_conf.item())
        store[p + "dloc_pos"] = lt.grad.numpy()[cls != 20].astype(np.float32)
        dconf = ct.grad.numpy()
        store[p + "dconf_touched"] = np.nonzero(np.abs(dconf.reshape(-1, 21)).sum(1) > 0)[0].astype(np.int64)
        store[p + "dconf_abs_sum"] = np.float64(np.abs(dconf).astype(np.float64).sum())
    np.savez_compressed(os.path.join(GOLD, "degenerate.npz"), **store)


def write_augment(RU):
    """The geometric half of Util.transform (Util.py:566-607): `expand`, `random_crop`, `flip` of the reference, run on
    seeded `random` streams in transform's own order.  photometric_distort and the PIL<->tensor conversions are
    torchvision functions (absent): the script restates to_tensor as /255 and to_pil_image as mul(255).byte(), mirrors
    the pixels itself for the flip, and calls the reference's `flip` with a width-only stand-in image for the boxes."""
    import hashlib
    import random

    class _W:
        def __init__(self, w):
            self.width = w
    store = {}
    mean = [0.485, 0.456, 0.406]
    n = 40
    for ci in range(n):
        r = np.random.default_rng(700 + ci)
        h, w = int(r.integers(18, 60)), int(r.integers(18, 60))
        img = r.integers(0, 256, (h, w, 3), dtype=np.uint8)
        nb = int(r.integers(1, 5))
        x1 = r.uniform(0, w * .6, nb); y1 = r.uniform(0, h * .6, nb)
        bx = np.stack([x1, y1, np.minimum(x1 + r.uniform(3, w * .6, nb), w - 1), np.minimum(y1 + r.uniform(3, h * .6, nb), h - 1)], 1)
        boxes = torch.from_numpy(bx.astype(np.float32))
        labels = torch.from_numpy(r.integers(0, 20, nb).astype(np.float32))
        random.seed(9000 + ci)
        t = torch.from_numpy(img).permute(2, 0, 1).float().div(255)
        new_image, new_boxes, new_labels = t, boxes, labels
        if random.random() < .5:
            new_image, new_boxes = RU.expand(t, boxes, filler=mean)
        new_image, new_boxes, new_labels = RU.random_crop(new_image, new_boxes, new_labels)
        u8 = new_image.mul(255).byte().permute(1, 2, 0).contiguous().numpy()
        flipped = False
        if random.random() < .5:
            _, new_boxes = RU.flip(_W(u8.shape[1]), new_boxes)
            u8 = np.ascontiguousarray(u8[:, ::-1])
            flipped = True
        p = f"c{ci}_"
        store[p + "img"] = img; store[p + "boxes"] = bx.astype(np.float32); store[p + "labels"] = labels.numpy()
        store[p + "seed"] = np.int64(9000 + ci)
        store[p + "out_shape"] = np.asarray(u8.shape[:2], np.int64)
        store[p + "out_sha256"] = np.asarray(hashlib.sha256(u8.tobytes()).hexdigest())
        store[p + "out_boxes"] = new_boxes.numpy().astype(np.float32)
        store[p + "out_labels"] = new_labels.numpy().astype(np.float32)
        store[p + "flipped"] = np.bool_(flipped)
        if ci < 4:
            store[p + "out_img"] = u8
    store["n_cases"] = np.int64(n)
    np.savez_compressed(os.path.join(GOLD, "augment.npz"), **store)


class _MaskArray(np.ndarray):
    """ndarray whose indexing accepts a torch bool mask as the boolean mask it is (see write_map)."""

    def __getitem__(self, idx):
        if isinstance(idx, torch.Tensor):
            idx = idx.numpy()
        return super().__getitem__(idx)


def write_map(RU):
    """Util.get_map (Util.py:783-885) on the cases above.

    As shipped the function does not run under this image's numpy 2.2 / torch 2.10: `cum_precision[recalls_mask]`
    (Util.py:879) indexes a numpy array with a torch bool tensor, which numpy 2 takes as the integer indices 0/1
    (IndexError for a class with one detection, silently wrong values otherwise); under the numpy 1.x the code was
    written for it is a boolean mask -- and the `recalls_mask.sum() != 0` guard one line above shows that intent.
    The fixture is therefore generated with `Util.np.array` wrapped (not replaced) so that the arrays it returns
    accept a torch mask as a boolean mask; nothing else of the function is touched."""
    store = {}
    cases = map_cases()
    real_np = RU.np
    proxy = types.ModuleType("numpy_proxy")
    proxy.__dict__.update(real_np.__dict__)
    proxy.array = lambda *a, **k: real_np.array(*a, **k).view(_MaskArray)
    RU.np = proxy
    for ci, (db, dc, ds, gb, gc) in enumerate(cases):
        with quiet():
            aps = RU.get_map([torch.from_numpy(b) for b in db], [torch.from_numpy(c) for c in dc],
                             [torch.from_numpy(s) for s in ds], [torch.from_numpy(b) for b in gb],
                             [torch.from_numpy(c) for c in gc])
        p = f"c{ci}_"
        store[p + "ap"] = np.asarray([float(aps[c]) for c in range(20)], np.float64)
        store[p + "n_img"] = np.int64(len(db))
        store[p + "det_count"] = np.asarray([len(c) for c in dc], np.int64)
        store[p + "gt_count"] = np.asarray([len(c) for c in gc], np.int64)
        store[p + "det_boxes"] = np.concatenate(db); store[p + "det_classes"] = np.concatenate(dc)
        store[p + "det_scores"] = np.concatenate(ds)
        store[p + "gt_boxes"] = np.concatenate(gb); store[p + "gt_classes"] = np.concatenate(gc)
    RU.np = real_np
    store["n_cases"] = np.int64(len(cases))
    store["recall_levels"] = torch.arange(0, 1.1, 0.1).double().numpy()
    np.savez_compressed(os.path.join(GOLD, "map.npz"), **store)


def write_resnet34(RM, RU):
    """SSD_resnet34 (Model.py:12-126) in eval mode on the seeded state, plus the zoom/ratio anchors (Util.py:142-164)."""
    with quiet():
        net = RM.SSD_resnet34(20)
    state = O.ssd_resnet34_random_state(seed=0)
    sd = net.state_dict()
    full = dict(state)
    for alias, trunk in O.ssd_resnet34_aliases().items():
        for k in list(state):
            if k.startswith(trunk):
                full[alias + k[len(trunk):]] = state[k]
    assert set(full) == set(sd), (sorted(set(full) ^ set(sd))[:8])
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(full[k].shape), (k, v.shape, full[k].shape)
    net.load_state_dict(full)
    net.eval()
    x = np.random.default_rng(5151).standard_normal((2, 3, 224, 224), dtype=np.float32)
    with torch.no_grad():
        loc, conf = net(torch.from_numpy(x))
    np.savez_compressed(os.path.join(GOLD, "resnet34.npz"), x_seed=np.int64(5151), state_seed=np.int64(0),
                        loc=loc.numpy(), conf=conf.numpy(),
                        state_dict_keys=np.asarray(list(sd.keys())),
                        state_dict_shapes=np.asarray([",".join(str(d) for d in v.shape) for v in sd.values()]),
                        named_parameter_keys=np.asarray([n for n, _ in net.named_parameters()]),
                        ancs_zoom_ratio=RU.create_ancs_xywh_zoom_ratio().numpy().astype(np.float32))


def write_import_names():
    """tests/golden/import_names.json: every name the reference's callers (train.py, train_function.py, Dataset.py) take from the four
    modules the drop-in replaces, read from their source with `ast` (nothing is imported or executed).  For `from X import *` the
    names the file then USES without binding them itself are listed under "star_used"."""
    import ast
    import builtins
    import json
    mods = ("Model", "Losses", "Util", "Dataset")
    out = {}
    for fn in ("train.py", "train_function.py", "Dataset.py"):
        tree = ast.parse(open(os.path.join(REF, fn)).read(), fn)
        explicit, star, bound, loaded = {}, [], set(), set()
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom):
                if node.module in mods:
                    for a in node.names:
                        if a.name == "*":
                            star.append(node.module)
                        else:
                            explicit.setdefault(node.module, []).append(a.name)
                for a in node.names:
                    bound.add(a.asname or a.name)
            elif isinstance(node, ast.Import):
                for a in node.names:
                    bound.add((a.asname or a.name).split(".")[0])
            elif isinstance(node, (ast.FunctionDef, ast.ClassDef)):
                bound.add(node.name)
                if isinstance(node, ast.FunctionDef):
                    for a in node.args.args + node.args.kwonlyargs:
                        bound.add(a.arg)
            elif isinstance(node, ast.Name):
                (bound if isinstance(node.ctx, (ast.Store, ast.Del)) else loaded).add(node.id)
            elif isinstance(node, ast.arg):
                bound.add(node.arg)
        used = sorted(n for n in loaded - bound if not hasattr(builtins, n)) if star else []
        out[fn] = {"explicit": {k: sorted(v) for k, v in explicit.items()}, "star": star, "star_used": used}
    with open(os.path.join(GOLD, "import_names.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


def main():
    os.makedirs(GOLD, exist_ok=True)
    if sys.argv[1:] == ["imports"]:
        return write_import_names()
    if sys.argv[1:] in (["resnet34"], ["map"], ["augment"], ["degenerate"], ["photometric"]):     # add one fixture without rewriting the others
        _install_stand_ins()
        with quiet():
            import Util as RU
            import Model as RM
        torch.manual_seed(0)
        if sys.argv[1] == "map":
            write_map(RU)
        elif sys.argv[1] == "augment":
            write_augment(RU)
        elif sys.argv[1] == "photometric":
            write_photometric(RU)
        elif sys.argv[1] == "degenerate":
            import Losses as RL
            write_degenerate(RL)
        else:
            write_resnet34(RM, RU)
        return
    _install_stand_ins()
    with quiet():
        import Util as RU          # noqa: F401  (reference)
        import Losses as RL        # reference
        import Model as RM         # reference
    torch.manual_seed(0)
    torch.set_num_threads(8)

    # ---- 1. priors ---------------------------------------------------------
    pri = RL.ancs_xywh.numpy().astype(np.float32)
    pri_xyxy = RL.ancs_xyxy.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "priors_ssd300.npz"), cxcywh=pri, xyxy=pri_xyxy)

    # ---- 2. box coders + IoU -----------------------------------------------
    rng = np.random.default_rng(5)
    a = np.sort(rng.uniform(0, 1, (40, 2, 2)).astype(np.float32), axis=1).reshape(40, 4)[:, [0, 1, 2, 3]]
    a = np.stack([np.minimum(a[:, 0], a[:, 2]), np.minimum(a[:, 1], a[:, 3]),
                  np.maximum(a[:, 0], a[:, 2]) + np.float32(.01), np.maximum(a[:, 1], a[:, 3]) + np.float32(.01)], 1)
    sub = pri_xyxy[rng.integers(0, 8732, 300)]
    iou = RU.get_jaccard_tensor1(torch.from_numpy(a), torch.from_numpy(sub)).numpy()
    g = rng.standard_normal((300, 4)).astype(np.float32)
    psub = pri[rng.integers(0, 8732, 300)]
    dec = RU.gcxgcy_to_cxcy(torch.from_numpy(g), torch.from_numpy(psub)).numpy()
    enc = RU.get_offsets_coords(torch.from_numpy(dec), torch.from_numpy(psub)).numpy()
    np.savez_compressed(os.path.join(GOLD, "boxmath.npz"), a=a, b=sub, iou=iou, g=g, pri=psub, dec=dec,
                        enc=enc, a_xywh=RU.xyxy_to_xywh(torch.from_numpy(a)).numpy(),
                        a_back=RU.xywh_to_xyxy(RU.xyxy_to_xywh(torch.from_numpy(a))).numpy())

    # ---- 3. matching + loss cases --------------------------------------------
    captured = {}
    orig_enc = RL.get_offsets_coords

    def spy(cxcy, priors_cxcy):                    # wraps, does not replace, Util.py:98-102
        out = orig_enc(cxcy, priors_cxcy)
        captured["gt"] = cxcy.detach().clone().numpy()
        captured["enc"] = out.detach().clone().numpy()
        return out
    RL.get_offsets_coords = spy

    cases = [(b, c, 1000 + i) for i, (b, c) in enumerate(edge_cases(pri_xyxy))]
    for i, bs in enumerate([1, 2, 2, 3, 4, 4, 8, 8, 2, 3, 5, 6]):
        r = np.random.default_rng(2000 + i)
        b, c = synth_gt(r, bs)
        cases.append((b, c, 2000 + i))
    store = {"n_cases": np.int64(len(cases))}
    for ci, (boxes, classes, seed) in enumerate(cases):
        bs = len(boxes)
        r = np.random.default_rng(seed)
        loc = r.standard_normal((bs, 8732, 4), dtype=np.float32)
        conf = (r.standard_normal((bs, 8732, 21), dtype=np.float32) * np.float32(2.0))
        lt = torch.from_numpy(loc).requires_grad_(True)
        ct = torch.from_numpy(conf).requires_grad_(True)
        with quiet():
            l_loc, l_conf = RL.ssd((lt, ct), [torch.from_numpy(c) for c in classes],
                                   [torch.from_numpy(b) for b in boxes])
        cls = RL.obj_forEach_prior___.numpy().astype(np.int8)        # Losses.py:172-173
        (l_loc + l_conf).backward()
        dloc = lt.grad.numpy(); dconf = ct.grad.numpy()
        rows = np.random.default_rng(seed + 7).integers(0, bs * 8732, 400)
        touched = np.nonzero(np.abs(dconf.reshape(-1, 21)).sum(1) > 0)[0]
        p = f"c{ci}_"
        store[p + "seed"] = np.int64(seed)
        store[p + "counts"] = np.asarray([b.shape[0] for b in boxes], np.int64)
        store[p + "boxes"] = np.concatenate(boxes).astype(np.float32)
        store[p + "classes"] = np.concatenate(classes).astype(np.float32)
        store[p + "cls"] = cls
        store[p + "gt_pos"] = captured["gt"].astype(np.float32)
        store[p + "enc_pos"] = captured["enc"].astype(np.float32)
        store[p + "loc_loss"] = np.float32(l_loc.item())
        store[p + "conf_loss"] = np.float32(l_conf.item())
        store[p + "dloc_pos"] = dloc[cls != 20].astype(np.float32)
        store[p + "dloc_abs_sum"] = np.float64(np.abs(dloc).astype(np.float64).sum())
        store[p + "dconf_rows"] = rows
        store[p + "dconf_vals"] = dconf.reshape(-1, 21)[rows].astype(np.float32)
        store[p + "dconf_touched"] = touched.astype(np.int64)
        store[p + "dconf_abs_sum"] = np.float64(np.abs(dconf).astype(np.float64).sum())
    RL.get_offsets_coords = orig_enc
    np.savez_compressed(os.path.join(GOLD, "match_loss.npz"), **store)

    # ---- 4. decode + NMS cases -------------------------------------------------
    store = {}
    n_ok = 0
    seed = 3000
    want = [dict(scale=3.0, top_k=200), dict(scale=3.0, top_k=200), dict(scale=2.0, top_k=200),
            dict(scale=4.0, top_k=50), dict(scale=1.2, top_k=200), dict(scale=0.1, top_k=200)]
    while n_ok < len(want):
        cfg = want[n_ok]
        r = np.random.default_rng(seed)
        seed += 1
        l_ = (r.standard_normal((8732, 4), dtype=np.float32) * np.float32(0.5))
        c_ = (r.standard_normal((8732, 21), dtype=np.float32) * np.float32(cfg["scale"]))
        # margins: keep every prob away from the score threshold and every candidate-pair IoU away
        # from the NMS threshold, so ulp-level differences in exp() cannot flip a decision.
        pr = torch.softmax(torch.from_numpy(c_).double(), 1).numpy()
        if np.min(np.abs(pr[:, :20] - 0.2)) < 2e-6:
            continue
        bx = O.xywh_to_xyxy(O.decode_offsets(l_, pri))
        ok = True
        for c in range(20):
            idx = np.nonzero(pr[:, c] >= 0.2)[0]
            if idx.size < 2:
                continue
            sp = np.sort(pr[idx, c])
            if np.min(np.diff(sp)) < 1e-7:
                ok = False; break
            io = O.iou_matrix(bx[idx], bx[idx]).astype(np.float64)
            if np.min(np.abs(io - 0.45)) < 2e-6:
                ok = False; break
        if not ok:
            continue
        with quiet():
            out = RL.inference(torch.from_numpy(l_), torch.from_numpy(c_), 0, top_k=cfg["top_k"],
                               phase="train", toDraw=False)
        p = f"n{n_ok}_"
        store[p + "seed"] = np.int64(seed - 1)
        store[p + "scale"] = np.float32(cfg["scale"])
        store[p + "top_k"] = np.int64(cfg["top_k"])
        if isinstance(out[0], list):
            store[p + "boxes"] = np.zeros((0, 4), np.float32)
            store[p + "classes"] = np.zeros((0,), np.int64)
            store[p + "probs"] = np.zeros((0,), np.float32)
        else:
            store[p + "boxes"] = out[0].numpy().astype(np.float32)
            store[p + "classes"] = out[1].numpy().astype(np.int64)
            store[p + "probs"] = out[2].numpy().astype(np.float32)
        n_ok += 1
    store["n_cases"] = np.int64(n_ok)
    store["img_wh"] = np.asarray([IMG_W, IMG_H], np.int64)
    np.savez_compressed(os.path.join(GOLD, "nms.npz"), **store)

    # ---- 5. network forward / train step ----------------------------------------
    params = O.ssd300_random_params(seed=0)
    with quiet():
        net = RM.SSD_300()
    names = dict(net.named_parameters())
    missing = [k for k in params if k not in names]
    assert not missing, missing                     # our names == reference names
    with torch.no_grad():
        for k, v in params.items():
            assert tuple(names[k].shape) == tuple(v.shape), (k, names[k].shape, v.shape)
            names[k].copy_(v)
    net.train()
    r = np.random.default_rng(4242)
    bs = 2
    x = r.standard_normal((bs, 3, 300, 300), dtype=np.float32)
    boxes, classes = synth_gt(np.random.default_rng(4243), bs)
    loc, conf = net(torch.from_numpy(x))
    with quiet():
        l_loc, l_conf = RL.ssd((loc, conf), [torch.from_numpy(c) for c in classes],
                               [torch.from_numpy(b) for b in boxes])
    net.zero_grad()
    (l_loc + l_conf).backward()
    pidx = np.random.default_rng(4244).integers(0, 8732, 512)
    store = dict(x_seed=np.int64(4242), gt_seed=np.int64(4243), param_seed=np.int64(0), bs=np.int64(bs),
                 prior_idx=pidx, loc_s=loc.detach().numpy()[:, pidx], conf_s=conf.detach().numpy()[:, pidx],
                 loc_sum=np.float64(loc.detach().double().sum()), loc_abs=np.float64(loc.detach().double().abs().sum()),
                 conf_sum=np.float64(conf.detach().double().sum()), conf_abs=np.float64(conf.detach().double().abs().sum()),
                 loc_loss=np.float32(l_loc.item()), conf_loss=np.float32(l_conf.item()))
    gnames = sorted(k for k in params)
    store["grad_names"] = np.asarray(gnames)
    store["grad_l2"] = np.asarray([float(names[k].grad.double().norm()) for k in gnames], np.float64)
    store["grad_sum"] = np.asarray([float(names[k].grad.double().sum()) for k in gnames], np.float64)
    for k in ("model.features.0.weight", "model.features.21.bias", "c_11_cl.weight", "seq10.2.weight",
              "rescaling_conv_4_3", "c_4_bb.bias"):
        store["g_" + k] = names[k].grad.numpy().astype(np.float32)
    n_named = len(list(net.named_parameters()))
    store["ref_named_parameters"] = np.int64(n_named)
    # checkpoint layout of the reference (train_function.py:114-120 saves cnn.state_dict()): key names + shapes
    sd = net.state_dict()
    store["state_dict_keys"] = np.asarray(list(sd.keys()))
    store["state_dict_shapes"] = np.asarray([",".join(str(d) for d in v.shape) for v in sd.values()])
    store["named_parameter_keys"] = np.asarray([n for n, _ in net.named_parameters()])
    np.savez_compressed(os.path.join(GOLD, "network.npz"), **store)
    write_resnet34(RM, RU)
    write_map(RU)
    write_augment(RU)
    write_degenerate(RL)
    write_photometric(RU)
    write_import_names()
    print("golden fixtures written to", GOLD)
    for f in sorted(os.listdir(GOLD)):
        print(f"  {f}: {os.path.getsize(os.path.join(GOLD, f))} bytes")


if __name__ == "__main__":
    main()
