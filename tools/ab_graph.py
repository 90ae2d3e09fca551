"""Interleaved A/B in one process: the eager train step against its HIP-graph replay (ddp.GraphedTrainStep), batch 32, alternating
repetitions; prints median / min ms per step and the shader clock each form ran at.

    python tools/ab_graph.py            [AB_CONV_DTYPE=bf16] [AB_TWO_STREAMS=1]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from objectdetection_ssd_amd import Losses, Model, ops  # noqa: E402
from objectdetection_ssd_amd.ddp import FlatSGDDataParallel, GraphedTrainStep  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = Model.SSD_300().to(dev).train()
    net.conv_dtype = os.environ.get("AB_CONV_DTYPE", "f32")
    tr = FlatSGDDataParallel(net, lr=1e-4)
    x, classes, boxes = bench.synth_batch(32, 1234, dev)

    def eager():
        tr.zero_grad()
        loc, conf = net(x)
        l1, l2, n_pos = Losses.ssd((loc, conf), classes, boxes, norm_mode=1, with_n_pos=True)
        (l1 + l2).backward()
        tr.reduce_and_step(n_pos)
    g = GraphedTrainStep(net, tr, warmup=0, two_streams=os.environ.get("AB_TWO_STREAMS", "0") == "1")
    for _ in range(5):
        eager()
    for _ in range(3):
        g(x, classes, boxes)

    def timed(fn, n):
        torch.cuda.synchronize()
        a = ops.clock_probe(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        b = ops.clock_probe(dev)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, ops.shader_mhz(a, b)
    res = {"eager": [], "graph": []}
    for rep in range(6):
        for tag, fn in (("eager", eager), ("graph", lambda: g(x, classes, boxes))):
            timed(fn, 2)
            res[tag].append(timed(fn, 15))
    for tag, r in res.items():
        r.sort()
        print(f"{tag}: median {r[len(r) // 2][0]:.3f} ms/step at {r[len(r) // 2][1]:.0f} MHz (min {r[0][0]:.3f}, max {r[-1][0]:.3f}); "
              f"graph kernel nodes {g.kernel_nodes}")


if __name__ == "__main__":
    main()
