"""Time of the MultiBox loss call (both forms) at the bench's batch: HIP events around 200 calls each, interleaved rounds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objectdetection_ssd_amd import _lib, ops, Util

dev = torch.device("cuda:0")
bs, P, C = 32, 8732, 21
g = torch.Generator().manual_seed(0)
pri = Util.get_priors().to(dev) if hasattr(Util, "get_priors") else None
if pri is None or pri.shape[0] != P:
    cxcy = torch.rand(P, 2, generator=g); wh = torch.rand(P, 2, generator=g) * 0.5 + 0.03
    pri = torch.cat([cxcy, wh], 1).to(dev)
pri_xyxy = torch.cat([pri[:, :2] - pri[:, 2:] / 2, pri[:, :2] + pri[:, 2:] / 2], 1).contiguous()
boxes, classes, start = [], [], [0]
for i in range(bs):
    n = int(torch.randint(1, 9, (1,), generator=g))
    c = torch.rand(n, 2, generator=g) * 0.6 + 0.2
    s = torch.rand(n, 2, generator=g) * 0.4 + 0.05
    boxes.append(torch.cat([c - s / 2, c + s / 2], 1)); classes.append(torch.randint(0, 20, (n,), generator=g).float())
    start.append(start[-1] + n)
args = [torch.randn(bs, P, 4, generator=g).to(dev), (torch.randn(bs, P, C, generator=g) * 2).to(dev), torch.cat(boxes).to(dev),
        torch.cat(classes).to(dev), torch.tensor(start, dtype=torch.int32, device=dev), pri.contiguous(), pri_xyxy]
lib = _lib.load()
for rnd in range(3):
    for form in (0, 1):
        _lib.check(lib.ssd_tune_set_loss_form(form), "tune")
        for _ in range(10):
            ops.multibox_loss(*args, norm_mode=1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(200):
            ops.multibox_loss(*args, norm_mode=1)
        b.record()
        torch.cuda.synchronize()
        print(f"round {rnd} form {form}: {a.elapsed_time(b) / 200 * 1e3:.1f} us per call (includes host launch gaps)", flush=True)
_lib.check(lib.ssd_tune_set_loss_form(1), "tune")
