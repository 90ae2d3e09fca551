"""conv1_1 forward at batch 32: im2col + 1x1 MFMA convolution against the one-kernel form (with and without the weight gradient's rows)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from objectdetection_ssd_amd import ops

dev = torch.device("cuda:0")
x = torch.randn(32, 3, 300, 300, device=dev)
w = torch.randn(64, 3, 3, 3, device=dev) * 0.2
b = torch.randn(64, device=dev)
rows = ops.first_weight_rows(w)
g = ops.make_geom(32, 300, 300, 32, 64, 1, 1, 0, 1)


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, c in ev:
        a.record(); fn(); c.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(c) for a, c in ev)
    return t[len(t) // 2]


print("im2col            ", timed(lambda: ops.im2col_first(x)))
col = ops.im2col_first(x)
print("1x1 conv on rows  ", timed(lambda: ops.conv2d_fwd(col, rows, b, g, True)))
print("one kernel + rows ", timed(lambda: ops.conv1_first_fwd(x, rows, b, True, want_col=True)))
print("one kernel        ", timed(lambda: ops.conv1_first_fwd(x, rows, b, True, want_col=False)))
dy = torch.randn(32, 300, 300, 64, device=dev)
print("wgrad on rows     ", timed(lambda: ops.conv2d_wgrad(col, dy, g, 64, True)))
print("wgrad from x      ", timed(lambda: ops.conv1_first_wgrad(x, dy, True)))
