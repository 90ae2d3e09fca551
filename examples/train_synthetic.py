"""Drop-in usage on synthetic data: the reference's training loop shape (train_function.py:59-95) on the gfx950 path.

    python examples/train_synthetic.py [--steps 20] [--batch 20]

With the reference checked out next to this repo, the same three lines (`install_dropin()`, then
`from train_function import train_model`) run its own `train_model` unchanged -- see INTEGRATION.md.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import objectdetection_ssd_amd as amd

amd.install_dropin()
from Losses import ssd            # noqa: E402  (reference import style)
from Model import SSD_300         # noqa: E402


def batches(n, bs, dev, seed=0):
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    for _ in range(n):
        x = torch.randn(bs, 3, 300, 300, generator=g)
        classes, boxes = [], []
        for _ in range(bs):
            k = 1 + min(int(rng.poisson(1.4)), 7)
            x1, y1 = rng.uniform(0, .6, k), rng.uniform(0, .6, k)
            w, h = rng.uniform(.08, .6, k), rng.uniform(.08, .6, k)
            boxes.append(torch.tensor(np.stack([x1, y1, np.minimum(x1 + w, 1), np.minimum(y1 + h, 1)], 1), dtype=torch.float32))
            classes.append(torch.tensor(rng.integers(0, 20, k), dtype=torch.float32))
        yield x, classes, boxes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=20)        # train.py:29
    ap.add_argument("--fused-sgd", action="store_true", help="objectdetection_ssd_amd.optim.SGD in place of torch.optim.SGD (same results)")
    a = ap.parse_args()
    device = torch.device("cuda")
    cnn = SSD_300().to(device)
    biases = [p for n, p in cnn.named_parameters() if p.requires_grad and n.endswith(".bias")]
    not_biases = [p for n, p in cnn.named_parameters() if p.requires_grad and not n.endswith(".bias")]
    lr = 1e-4
    if a.fused_sgd:
        from objectdetection_ssd_amd.optim import SGD
    else:
        SGD = torch.optim.SGD
    optimizer = SGD(params=[{"params": biases, "lr": 2 * lr}, {"params": not_biases}], lr=lr, momentum=0.9,
                    weight_decay=5e-4)                      # train.py:53-55
    cnn.train()
    t0 = time.time()
    for count, (inputs, classes, bboxes) in enumerate(batches(a.steps, a.batch, device)):
        inputs = inputs.to(device)
        classes = [c.to(device) for c in classes]
        bboxes = [b.to(device) for b in bboxes]
        optimizer.zero_grad()
        outputs = cnn(inputs)
        loss1, loss2 = ssd(outputs, classes, bboxes)
        loss = loss1 + loss2
        loss.backward()
        optimizer.step()
        if count % 5 == 0:
            print(f"it {count:3d}  l1 {loss1.item():.4f}  l2 {loss2.item():.4f}  {time.time() - t0:.1f}s")
    torch.cuda.synchronize()
    print(f"{a.steps * a.batch / (time.time() - t0):.1f} images/s incl. host-side data generation")


if __name__ == "__main__":
    main()
