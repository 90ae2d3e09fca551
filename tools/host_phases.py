"""Host-side time of each phase of the train step (no synchronisation inside the loop): a phase in which the host BLOCKS on the GPU
shows up as long host time.  python tools/host_phases.py"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from objectdetection_ssd_amd import Losses, Model
from objectdetection_ssd_amd.ddp import FlatSGDDataParallel
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = Model.SSD_300().to(dev).train()
tr = FlatSGDDataParallel(net, lr=1e-4)
x, classes, boxes = bench.synth_batch(32, 1234, dev)
names = ["zero_grad", "forward", "loss", "backward", "reduce+sgd"]
acc = np.zeros((0, 5))
for it in range(14):
    t = [time.perf_counter()]
    tr.zero_grad(); t.append(time.perf_counter())
    loc, conf = net(x); t.append(time.perf_counter())
    l1, l2, n_pos = Losses.ssd((loc, conf), classes, boxes, norm_mode=1, with_n_pos=True); t.append(time.perf_counter())
    (l1 + l2).backward(); t.append(time.perf_counter())
    tr.reduce_and_step(n_pos); t.append(time.perf_counter())
    if it >= 4:
        acc = np.vstack([acc, np.diff(t) * 1e3])
torch.cuda.synchronize()
print("host ms per phase (median over 10 steps):", {n: round(float(v), 3) for n, v in zip(names, np.median(acc, axis=0))}, "sum", round(float(np.median(acc.sum(1))), 3))
