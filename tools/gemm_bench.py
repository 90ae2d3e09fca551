"""(the kernels this tool forces -- gemm_nt.hip / wino4_full_kernel -- need a library built with SSD_EXPERIMENTAL=1)
Times the plane GEMMs of the Winograd layers with K >= 256 by themselves (the library's own event pair around each launch), the generic
64 x 64 kernel (ssd_tune_set_gemm_nt 0) against the 128 x 128 LDS-DMA kernel (1), interleaved in one process at batch 32.

    python tools/gemm_bench.py [rounds]
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from objectdetection_ssd_amd import _lib, ops  # noqa: E402

# name, map, K (reduction = input planes' channels), N (output channels)
LAYERS = [("conv3_2 fwd", 75, 256, 256), ("conv4_1 fwd", 38, 256, 512), ("conv4_2 fwd", 38, 512, 512), ("conv5_2 fwd", 19, 512, 512),
          ("conv3_1 dgrad", 75, 256, 128), ("conv4_1 dgrad", 38, 512, 256), ("fc-like", 19, 1024, 1024)]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    dev = torch.device("cuda:0")
    lib = _lib.load()
    n = 32
    for name, h, ci, co in LAYERS:
        gen = torch.Generator(device=dev).manual_seed(h + co)
        x = torch.randn(n, h, h, ci, device=dev, generator=gen)
        w = torch.randn(co, ci, 3, 3, device=dev, generator=gen) * (2.0 / (ci * 9)) ** 0.5
        g = ops.make_geom(n, h, h, ci, co, 3, 1, 1, 1)
        uf, _ = ops.wino_weights(w, co, want_bwd=False, mo=4)
        res = {}
        for rep in range(2):
            for mode in (0, 1):
                _lib.check(lib.ssd_tune_set_gemm_nt(mode), "tune")
                ops.conv2d_fwd_wino(x, uf, None, g, False)
                _lib.check(lib.ssd_prof_gemm_begin(), "prof")
                for _ in range(rounds):
                    ops.conv2d_fwd_wino(x, uf, None, g, False)
                torch.cuda.synchronize()
                ms, fl, kd = (C.c_float * 64)(), (C.c_double * 64)(), (C.c_int * 64)()
                k = lib.ssd_prof_gemm_collect_kinds(ms, fl, kd, 64)
                ts = sorted(ms[i] for i in range(k))
                res[mode] = (ts[len(ts) // 2], fl[0], kd[0])
        _lib.check(lib.ssd_tune_set_gemm_nt(-1), "tune")
        (t0, f0, k0), (t1, f1, k1) = res[0], res[1]
        print(f"{name:14s} M={n * ((h + 3) // 4) ** 2:6d} K={ci:5d} N={co:5d}  64x64 {t0:.3f} ms {f0 / t0 / 1e9:6.1f} TF/s (kind {k0})   "
              f"128x128 dma {t1:.3f} ms {f1 / t1 / 1e9:6.1f} TF/s (kind {k1})", flush=True)


if __name__ == "__main__":
    main()
