"""Accuracy of one Winograd F(4x4) layer (forward, data gradient, weight gradient) with the plane GEMMs on the f32 MFMA vs from three bf16
limbs, against an f64 convolution.   python tools/x3_accuracy.py [n h w ci co]"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from objectdetection_ssd_amd import _lib, ops  # noqa: E402


def main():
    n, h, w, ci, co = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else (4, 38, 38, 512, 512)
    scale = float(sys.argv[6]) if len(sys.argv) > 6 else 1.0
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g_ = torch.Generator().manual_seed(5)
    x = torch.relu(torch.randn(n, ci, h, w, generator=g_))
    wt = torch.randn(co, ci, 3, 3, generator=g_) * (2.0 / (9 * ci)) ** 0.5
    b = torch.randn(co, generator=g_) * 0.1
    dy = torch.randn(n, co, h, w, generator=g_) * (torch.rand(n, co, h, w, generator=g_) < 0.3) * scale
    x64 = x.double().requires_grad_(True)
    w64 = wt.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, b.double(), padding=1)
    y64.backward(dy.double())
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    g = ops.make_geom(n, h, w, ci, co, 3, 1, 1, 1)
    xd, dyd = nhwc(x).to(dev), nhwc(dy).to(dev)
    rel = lambda a, r: float((a.double().cpu() - r).norm() / r.norm())
    for mode in (0, 1):
        _lib.check(lib.ssd_tune_set_wino_x3(mode), "tune")
        uf, ub = ops.wino_weights(wt.to(dev), co, mo=4)
        y, planes = ops.conv2d_fwd_wino(xd, uf, b.to(dev), g, False, keep_planes=True)
        dw, db, pd = ops.conv2d_wgrad_wino(None, dyd, g, co, True, mo=4, planes=planes, dgrad_planes=True)
        dx = ops.conv2d_dgrad_wino(None, ub, g, planes=pd)
        print(f"x3={mode}: fwd {rel(y, nhwc(y64.detach())):.2e}  dgrad {rel(dx, nhwc(x64.grad)):.2e}  wgrad {rel(dw, w64.grad):.2e}  "
              f"dbias {rel(db, dy.double().sum((0, 2, 3))):.2e}   mean(dx - ref)/mean|ref| "
              f"{float((dx.double().cpu() - nhwc(x64.grad)).mean() / nhwc(x64.grad).abs().mean()):.2e}")
    _lib.check(lib.ssd_tune_set_wino_x3(-1), "tune")


if __name__ == "__main__":
    main()
