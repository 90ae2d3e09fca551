"""Do an MFMA-bound kernel and an HBM-bound kernel share the chip when they sit on two streams?  Times n x (TN GEMM of a Winograd weight
gradient) on one stream, n x (dy transform pass) on another, alone and together.  together ~ max(...) = they overlap; ~ sum = they do not.

    python tools/overlap_probe.py
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from objectdetection_ssd_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    n = 32
    for name, h, ci, co, h2, c2 in (("conv4_2 GEMM | conv3_2 dy pass", 38, 512, 512, 75, 256), ("conv3_2 GEMM | conv2_2 dy pass", 75, 256, 256, 150, 128),
                                    ("conv4_2 GEMM | conv4_2 dy pass", 38, 512, 512, 38, 512)):
        g = ops.make_geom(n, h, h, ci, co, 3, 1, 1, 1)
        dy = torch.randn(n, h, h, co, device=dev)
        kept = torch.randn(ops.wino_planes_shape(g), device=dev)
        Y, _, part = ops.wino_dy_transform(dy, g, co, False, True)
        g2 = ops.make_geom(n, h2, h2, c2, c2, 3, 1, 1, 1)
        dy2 = torch.randn(n, h2, h2, c2, device=dev)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        reps = 20

        def gemm():
            with torch.cuda.stream(s1):
                for _ in range(reps):
                    ops.wino_wgrad_gemm(Y, kept, part, g, co)

        def xform():
            with torch.cuda.stream(s2):
                for _ in range(reps):
                    ops.wino_dy_transform(dy2, g2, c2, True, True)

        def run(fns):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for f in fns:
                f()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3
        for f in (gemm, xform):
            run([f])
        res = []
        for _ in range(3):
            res.append((run([gemm]), run([xform]), run([gemm, xform]), run([xform, gemm])))
        a, b, c, d = (sorted(r[i] for r in res)[1] for i in range(4))
        print(f"{name}: GEMM {a:.3f} ms  transform {b:.3f} ms  sum {a + b:.3f}  together {c:.3f} / {d:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
