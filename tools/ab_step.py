"""Interleaved A/B of one tuning switch on the full train step at batch 32 (same process, same device, alternating repetitions): a
library switch (`ssd_tune_set_...`) or an attribute of the engine (`engine.<name>`).

    python tools/ab_step.py ssd_tune_set_batched_units 0 1
    python tools/ab_step.py engine.first_fused 0 1
    AB_CONV_DTYPE=bf16 python tools/ab_step.py ssd_tune_set_conv_bf16_k64 0 1        (the bf16-tensor mode)
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from objectdetection_ssd_amd import Losses, Model, _lib  # noqa: E402
from objectdetection_ssd_amd.ddp import FlatSGDDataParallel  # noqa: E402


def main():
    fn, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
    lib = _lib.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = Model.SSD_300().to(dev).train()
    net.conv_dtype = os.environ.get("AB_CONV_DTYPE", "f32")
    tr = FlatSGDDataParallel(net, lr=1e-4)
    x, classes, boxes = bench.synth_batch(32, 1234, dev)

    def steps(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            tr.zero_grad()
            loc, conf = net(x)
            l1, l2, n_pos = Losses.ssd((loc, conf), classes, boxes, norm_mode=1, with_n_pos=True)
            (l1 + l2).backward()
            tr.reduce_and_step(n_pos)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    steps(5)
    res = {v: [] for v in vals}
    for rep in range(5):
        for v in vals:
            if fn.startswith("engine."):
                setattr(net._engine, fn[7:], type(getattr(net._engine, fn[7:]))(v))
                net.invalidate_weight_cache()
            else:
                _lib.check(getattr(lib, fn)(*([v] if fn != "ssd_tune_set_wgrad" else [-1, -1, v])), "tune")
            steps(2)
            res[v].append(steps(15))
    for v in vals:
        r = sorted(res[v])
        print(f"{fn}({v}): median {r[len(r) // 2]:.3f} ms/step  (min {r[0]:.3f}, max {r[-1]:.3f})")


if __name__ == "__main__":
    main()
