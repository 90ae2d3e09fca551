"""(the kernels this tool forces -- gemm_nt.hip / wino4_full_kernel -- need a library built with SSD_EXPERIMENTAL=1)
Times the F(4x4,3x3) forward / dgrad entry points per bench layer at batch 32 with the fused GEMM + output-transform kernel
(ssd_tune_set_wino_fused 1) and without it (0), interleaved in one process; checks that the two forms agree.

    python tools/wino_bench.py [rounds]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from objectdetection_ssd_amd import _lib, ops  # noqa: E402

LAYERS = [("conv1_2", 300, 64, 64, False), ("conv2_1", 150, 64, 128, None), ("conv2_2", 150, 128, 128, False), ("conv3_1", 75, 128, 256, None),
          ("conv3_2", 75, 256, 256, None), ("conv3_3", 75, 256, 256, True), ("conv4_1", 38, 256, 512, None), ("conv4_2", 38, 512, 512, None),
          ("conv5_2", 19, 512, 512, None), ("c_4", 38, 512, 100, None)]


def timed(fn, rounds):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(rounds)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2], ts[0]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    dev = torch.device("cuda:0")
    lib = _lib.load()
    n = 32
    for name, h, ci, co, pool in LAYERS:
        gen = torch.Generator(device=dev).manual_seed(h + co)
        x = torch.randn(n, h, h, ci, device=dev, generator=gen)
        w = torch.randn(co, ci, 3, 3, device=dev, generator=gen) * (2.0 / (ci * 9)) ** 0.5
        b = torch.randn(co, device=dev, generator=gen) * 0.1
        g = ops.make_geom(n, h, h, ci, co, 3, 1, 1, 1)
        ld = ops.pad32(co)
        uf, ub = ops.wino_weights(w, ld, mo=4)
        dy = torch.zeros(n, h, h, ld, device=dev)
        dy[..., :co] = torch.randn(n, h, h, co, device=dev, generator=gen)
        mask = torch.randn(n, h, h, ci, device=dev, generator=gen).clamp_min(0)
        _, planes = ops.conv2d_fwd_wino(x, uf, b, g, True, ld=ld, keep_planes=True)
        _, _, dyp = ops.conv2d_wgrad_wino(None, dy, g, ld, True, mo=4, planes=planes, dgrad_planes=True)
        del planes

        def fwd():
            if pool is not None:
                return ops.conv2d_fwd_wino_pool(x, uf, b, g, pool, keep_planes=True)[0]
            return ops.conv2d_fwd_wino(x, uf, b, g, True, ld=ld, keep_planes=True)[0]

        def dgrad():
            if _lib.load().ssd_conv3x3_wino_uses_full(__import__("ctypes").byref(g), 1):
                return ops.conv2d_dgrad_wino(dy, ub, g, relu_mask=mask)            # one kernel, from dy (no planes to read)
            return ops.conv2d_dgrad_wino(None, ub, g, relu_mask=mask, planes=dyp)
        res = {}
        for what, fn in (("fwd", fwd), ("dgrad", dgrad)):
            outs = {}
            modes = (0, 1, 2) if (ci in (64, 128) if what == "fwd" else ld in (64, 128)) else (0, 1)

            def setmode(mode):
                _lib.check(lib.ssd_tune_set_wino_fused(min(mode, 1)), "tune")
                _lib.check(lib.ssd_tune_set_wino_full(1 if mode == 2 else 0), "tune")
            for mode in modes:
                setmode(mode)
                outs[mode] = fn().clone()
            err = max(float((outs[m] - outs[0]).abs().max() / outs[0].abs().max().clamp_min(1e-30)) for m in modes[1:])
            t = {2: (float("nan"), 0)}
            for rep in range(2):                          # interleaved
                for mode in modes:
                    setmode(mode)
                    fn()
                    t[mode] = timed(fn, rounds)
            res[what] = (t[0][0], t[1][0], err, t[2][0])
        _lib.check(lib.ssd_tune_set_wino_fused(-1), "tune")
        _lib.check(lib.ssd_tune_set_wino_full(-1), "tune")
        ex = ops.wino_flops(g)[1] / 1e9
        print(f"{name:8s} {h:3d} {ci:4d}->{co:4d}  fwd  two-kernel {res['fwd'][0]:.3f}  fused {res['fwd'][1]:.3f}  one-kernel {res['fwd'][3]:.3f} ms (max rel diff {res['fwd'][2]:.1e})"
              f" | dgrad {res['dgrad'][0]:.3f} -> {res['dgrad'][1]:.3f} -> {res['dgrad'][3]:.3f} ms ({res['dgrad'][2]:.1e}) | executed {ex:.1f} GF", flush=True)
        del x, dy, dyp, mask
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
