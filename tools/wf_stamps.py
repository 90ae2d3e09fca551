"""(the kernels this tool forces -- gemm_nt.hip / wino4_full_kernel -- need a library built with SSD_EXPERIMENTAL=1)
In-kernel phase stamps of the fused Winograd kernel (diagnostic): shares of prologue / main loop / epilogue per workgroup."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from objectdetection_ssd_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")
FULL = len(sys.argv) > 1 and sys.argv[1] == "full"
for name, h, ci, co, pool, dgrad in (("conv1_2 fwd+pool", 300, 64, 64, False, False), ("conv1_2 dgrad", 300, 64, 64, None, True),
                                     ("conv2_2 fwd+pool", 150, 128, 128, False, False), ("conv3_2 fwd", 75, 256, 256, None, False),
                                     ("conv4_2 fwd", 38, 512, 512, None, False)):
    n = 32
    x = torch.randn(n, h, h, ci, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev) * (2.0 / (ci * 9)) ** 0.5
    b = torch.randn(co, device=dev)
    g = ops.make_geom(n, h, h, ci, co, 3, 1, 1, 1)
    uf, ub = ops.wino_weights(w, co, mo=4)
    mask = torch.randn(n, h, h, ci, device=dev).clamp_min(0)
    _lib.check(lib.ssd_tune_set_wino_fused(1))
    _lib.check(lib.ssd_tune_set_wino_full(1 if FULL else 0))
    if FULL and ci > 128:
        continue
    tiles = n * ((h + 3) // 4) ** 2
    nblk = ((tiles + 31) // 32) * ((co + 63) // 64)
    buf = torch.zeros((nblk // 16 + 2, 8), dtype=torch.int64, device=dev)
    def run():
        if dgrad:
            return ops.conv2d_dgrad_wino(x if ci == co else None, ub, g, relu_mask=mask)
        if pool is not None:
            return ops.conv2d_fwd_wino_pool(x, uf, b, g, pool, keep_planes=FULL)
        return ops.conv2d_fwd_wino(x, uf, b, g, True, keep_planes=FULL)
    run(); torch.cuda.synchronize()
    _lib.check(lib.ssd_tune_set_wino_fused_stamps(buf.data_ptr()))
    run(); torch.cuda.synchronize()
    _lib.check(lib.ssd_tune_set_wino_fused_stamps(None))
    t = buf.cpu().numpy()[: nblk // 16]
    t = t[t[:, 0] > 0]
    t = t[:, :6]
    d = np.diff(t, axis=1).astype(np.float64)
    names = ["T (input transform)", "M (MFMA)", "E transform", "phase 2", "store drain"] if FULL else ["prologue", "main loop", "transform", "phase 2 issue", "store drain"]
    med = np.median(d, axis=0)
    print(f"{name:18s} blocks {nblk:5d}  total/WG {np.median(t[:, 5] - t[:, 0]):8.0f} cyc: " + "  ".join(f"{k} {v:.0f}" for k, v in zip(names, med)))
    _lib.check(lib.ssd_tune_set_wino_fused(-1))
    if name.startswith("conv1_2"):                     # effect of the first-round start stagger on the same launch
        _lib.check(lib.ssd_tune_set_wino_fused(1))
        for stg in (0, 2000, 8000, 20000):
            _lib.check(lib.ssd_tune_set_wino_fused_stagger(stg))
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize()
            print(f"    stagger step {stg:6d} cycles: {e0.elapsed_time(e1):.3f} ms")
        _lib.check(lib.ssd_tune_set_wino_fused_stagger(-1))
        _lib.check(lib.ssd_tune_set_wino_fused(-1))
