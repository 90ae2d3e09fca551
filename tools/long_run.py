"""300 train steps over four alternating synthetic batches: step time, loss and allocator state every 50 steps (a stability check:
the step time must stay flat and the reserved memory must not grow -- 21.2 ms, 18.7 GB reserved on an MI355X).

    python tools/long_run.py
"""
import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from objectdetection_ssd_amd import Losses, Model
from objectdetection_ssd_amd.ddp import FlatSGDDataParallel
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = Model.SSD_300().to(dev).train()
tr = FlatSGDDataParallel(net, lr=1e-4, momentum=0.9, weight_decay=5e-4)
batches = [bench.synth_batch(32, 1000 + i, dev) for i in range(4)]
for rep in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(50):
        x, classes, boxes = batches[it % 4]
        tr.zero_grad()
        loc, conf = net(x)
        l1, l2, n_pos = Losses.ssd((loc, conf), classes, boxes, norm_mode=1, with_n_pos=True)
        (l1 + l2).backward()
        tr.reduce_and_step(n_pos)
    torch.cuda.synchronize()
    npos = float(Losses.last_match["n_pos"])
    print(f"rep {rep}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms/step  loss {(float(l1) + float(l2)) / max(npos, 1):.4f}  "
          f"allocated {torch.cuda.memory_allocated() / 1e9:.2f} GB reserved {torch.cuda.memory_reserved() / 1e9:.2f} GB", flush=True)
