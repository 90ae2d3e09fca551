"""Shared input builders for the tests (numpy PCG64 seeds == oracle/gen_golden.py)."""
import numpy as np


def synth_gt(rng, bs, max_extra=7):
    """Same generator as oracle/gen_golden.py:synth_gt."""
    boxes, classes = [], []
    for _ in range(bs):
        n = 1 + min(int(rng.poisson(1.4)), max_extra)
        x1 = rng.uniform(0, .6, n); y1 = rng.uniform(0, .6, n)
        w = rng.uniform(.08, .6, n); h = rng.uniform(.08, .6, n)
        b = np.stack([x1, y1, np.minimum(x1 + w, 1.), np.minimum(y1 + h, 1.)], 1).astype(np.float32)
        boxes.append(b)
        classes.append(rng.integers(0, 20, n).astype(np.float32))
    return boxes, classes


def split_case(z, ci):
    """Unpack match/loss case ``ci`` of tests/golden/match_loss.npz."""
    p = f"c{ci}_"
    counts = z[p + "counts"]
    off = np.concatenate([[0], np.cumsum(counts)])
    boxes = [z[p + "boxes"][off[i]:off[i + 1]] for i in range(len(counts))]
    classes = [z[p + "classes"][off[i]:off[i + 1]] for i in range(len(counts))]
    seed = int(z[p + "seed"])
    bs = len(counts)
    r = np.random.default_rng(seed)
    loc = r.standard_normal((bs, 8732, 4), dtype=np.float32)
    conf = r.standard_normal((bs, 8732, 21), dtype=np.float32) * np.float32(2.0)
    return boxes, classes, loc, conf, p


def nms_case(z, ni):
    p = f"n{ni}_"
    r = np.random.default_rng(int(z[p + "seed"]))
    l_ = r.standard_normal((8732, 4), dtype=np.float32) * np.float32(0.5)
    c_ = r.standard_normal((8732, 21), dtype=np.float32) * np.float32(z[p + "scale"])
    return l_, c_, int(z[p + "top_k"]), p
