"""Per-layer conv micro-benchmark over tile / stage variants (tuning aid; GPU only).
usage: python tools/conv_bench.py [fwd|dgrad|wgrad|all]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from objectdetection_ssd_amd import ops, _lib
lib = _lib.load()
BS = 32
LAYERS = [  # name, H, Ci, Co, k, s, pad, dil
    ("conv1_2", 300, 64, 64, 3, 1, 1, 1), ("conv2_1", 150, 64, 128, 3, 1, 1, 1), ("conv2_2", 150, 128, 128, 3, 1, 1, 1),
    ("conv3_1", 75, 128, 256, 3, 1, 1, 1), ("conv3_2", 75, 256, 256, 3, 1, 1, 1), ("conv4_1", 38, 256, 512, 3, 1, 1, 1),
    ("conv4_2", 38, 512, 512, 3, 1, 1, 1), ("conv5_1", 19, 512, 512, 3, 1, 1, 1), ("fc6", 19, 512, 1024, 3, 1, 4, 4),
    ("fc7", 19, 1024, 1024, 1, 1, 0, 1), ("c_4", 38, 512, 100, 3, 1, 1, 1), ("c_7", 19, 1024, 150, 3, 1, 1, 1),
    ("seq8.2", 19, 256, 512, 3, 2, 1, 1),
]
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
    vonly = sys.argv[3].split(",") if len(sys.argv) > 3 else None
    dev = torch.device("cuda")
    ig_var = [(-1, -1, "auto"), (1, 2, "128x128/2"), (1, 1, "128x128/1"), (0, 1, "256x64/1"), (0, 2, "256x64/2"),
              (2, 2, "128x64/2"), (2, 1, "128x64/1"), (3, 2, "64x64/2"), (3, 1, "64x64/1")]
    wg_var = [(-1, -1, -1, "auto"), (128, 1, 9, "128/1/9"), (64, 1, 18, "64/1/18"),
              (3, 1, 1, "f3/1"), (3, 1, 2, "f3/2"), (3, 1, 3, "f3/3")]
    if vonly:
        ig_var = [v for v in ig_var if v[2] in vonly]
        wg_var = [v for v in wg_var if v[3] in vonly]
    for name, H, ci, co, k, s, pad, dil in LAYERS:
        if only and name not in only:
            continue
        g = ops.make_geom(BS, H, H, ci, co, k, s, pad, dil)
        ld = ops.pad32(co)
        x = torch.randn(BS, H, H, ci, device=dev)
        w = torch.randn(co, ci, k, k, device=dev) * 0.05
        b = torch.randn(co, device=dev)
        wf, wb = ops.weight_ohwi(w, ld), ops.weight_ihwo(w, ld)
        dy = torch.randn(BS, g.Ho, g.Wo, ld, device=dev)
        if ld != co: dy[..., co:] = 0
        dx = torch.empty(BS, H, H, ci, device=dev)
        fl = ops.conv_flops(g)
        if which in ("fwd", "all"):
            row = []
            for t, nb, lab in ig_var:
                lib.ssd_tune_set_igemm(t, nb)
                ms = timeit(lambda: ops.conv2d_fwd(x, wf, b, g, True, ld=ld, out=dy))
                row.append(f"{lab}:{fl / ms / 1e9:6.1f}")
            print(f"fwd   {name:8s} " + "  ".join(row), flush=True)
        if which == "stamps":                   # in-kernel phase times of every 64th block (shader-clock cycles)
            nblk = ((BS * g.Ho * g.Wo + 63) // 64) * ((co + 63) // 64)
            buf = torch.zeros((nblk // 64 + 2, 4), device=dev, dtype=torch.int64)
            lib.ssd_tune_set_igemm_splitk(1)
            ops.conv2d_fwd(x, wf, b, g, True, ld=ld, out=dy)
            torch.cuda.synchronize()
            lib.ssd_tune_set_igemm_stamps(buf.data_ptr())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.conv2d_fwd(x, wf, b, g, True, ld=ld, out=dy); e1.record()
            torch.cuda.synchronize()
            lib.ssd_tune_set_igemm_stamps(None)
            lib.ssd_tune_set_igemm_splitk(-1)
            t = buf[: (nblk + 63) // 64].cpu().double()
            t = t[t[:, 3] > 0]
            life, pro, loop, epi = t[:, 3] - t[:, 0], t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
            kt = k * k * (ci // 32)
            ms = e0.elapsed_time(e1)
            print(f"stamps {name:8s} blocks {nblk} K-steps {kt}: kernel {ms:.3f} ms = {fl / ms / 1e9:.1f} TF/s | per sampled block (cycles): "
                  f"life {life.mean():.0f} (min {life.min():.0f} max {life.max():.0f})  prologue {pro.mean():.0f}  loop {loop.mean():.0f} "
                  f"= {loop.mean() / kt:.0f}/K-step  epilogue {epi.mean():.0f} | sum of block lives x64 / (256 CUs x 7) = "
                  f"{life.sum() * 64 / 1792 / 2.4e6:.3f} ms at 2.4 GHz", flush=True)
        if which == "patch" and (k, s, pad, dil) == (3, 1, 1, 1):     # fused f32 wgrad: patch shapes 4x8 / 1x38 / 2x19 vs the one-tap kernel
            row = []
            for shape, lab in ((-1, "auto"), (0, "4x8"), (1, "1x38"), (2, "2x19")):
                lib.ssd_tune_set_wgrad(3 if shape >= 0 else -1, 1, -1)
                lib.ssd_tune_set_wgrad_patch(shape)
                ms = timeit(lambda: ops.conv2d_wgrad(x, dy, g, ld, True))
                row.append(f"{lab}:{fl / ms / 1e9:6.1f}")
            lib.ssd_tune_set_wgrad(128, 1, -1)
            ms = timeit(lambda: ops.conv2d_wgrad(x, dy, g, ld, True))
            row.append(f"one-tap128:{fl / ms / 1e9:6.1f}")
            lib.ssd_tune_set_wgrad(-1, -1, -1); lib.ssd_tune_set_wgrad_patch(-1)
            print(f"wgrad patch {name:8s} " + "  ".join(row), flush=True)
        if which == "wino" and (k, s, pad, dil) == (3, 1, 1, 1) and ci % 32 == 0:
            uf, ub = ops.wino_weights(w, ld)
            print("wino", name, "ws", lib.ssd_conv3x3_wino_workspace(g, 0, 2) / 1e6, "MB", flush=True)
            ms0 = timeit(lambda: ops.conv2d_fwd(x, wf, b, g, True, ld=ld, out=dy))
            ms1 = timeit(lambda: ops.conv2d_fwd_wino(x, uf, b, g, True, ld=ld))
            ms2 = timeit(lambda: ops.conv2d_dgrad(dy, wb, g, dx, x, False))
            ms3 = timeit(lambda: ops.conv2d_dgrad_wino(dy, ub, g, dx, x, False))
            ms4 = timeit(lambda: ops.conv2d_wgrad(x, dy, g, ld, True))
            ms5 = timeit(lambda: ops.conv2d_wgrad_wino(x, dy, g, ld, True))
            print(f"wino {name:8s} wgrad direct {ms4:.3f} ms ({fl / ms4 / 1e9:.0f} TF/s)  winograd {ms5:.3f} ms ({fl / ms5 / 1e9:.0f} TF/s alg.)", flush=True)
            uf4, ub4 = ops.wino_weights(w, ld, mo=4)
            ms6 = timeit(lambda: ops.conv2d_fwd_wino(x, uf4, b, g, True, ld=ld))
            ms7 = timeit(lambda: ops.conv2d_dgrad_wino(dy, ub4, g, dx, x, False))
            ms8 = timeit(lambda: ops.conv2d_wgrad_wino(x, dy, g, ld, True, mo=4))
            print(f"wino {name:8s} F(4x4) wgrad {ms8:.3f} ms ({fl / ms8 / 1e9:.0f} TF/s alg.)", flush=True)
            print(f"wino {name:8s} F(4x4): fwd {ms6:.3f} ms ({fl / ms6 / 1e9:.0f} TF/s alg.)  dgrad {ms7:.3f} ms ({fl / ms7 / 1e9:.0f} TF/s alg.)", flush=True)
            print(f"wino {name:8s} fwd direct {ms0:.3f} ms ({fl / ms0 / 1e9:.0f} TF/s)  winograd {ms1:.3f} ms ({fl / ms1 / 1e9:.0f} TF/s alg.)   "
                  f"dgrad direct {ms2:.3f} ms  winograd {ms3:.3f} ms ({fl / ms3 / 1e9:.0f} TF/s alg.)", flush=True)
        if which == "occ":                      # blocks per CU capped through extra dynamic LDS (64x64 tile, 18 KB static)
            row = []
            for pad_kb, lab in ((0, "7/CU"), (8, "6"), (14, "5"), (22, "4"), (35, "3"), (62, "2")):
                lib.ssd_tune_set_igemm_lds_pad(pad_kb * 1024)
                ms = timeit(lambda: ops.conv2d_fwd(x, wf, b, g, True, ld=ld, out=dy))
                ms2 = timeit(lambda: ops.conv2d_dgrad(dy, wb, g, dx, x, False))
                row.append(f"{lab}:{fl / ms / 1e9:6.1f}/{fl / ms2 / 1e9:6.1f}")
            lib.ssd_tune_set_igemm_lds_pad(0)
            print(f"occ fwd/dgrad {name:8s} " + "  ".join(row), flush=True)
        if which in ("dgrad", "all"):
            row = []
            for t, nb, lab in ig_var:
                lib.ssd_tune_set_igemm(t, nb)
                ms = timeit(lambda: ops.conv2d_dgrad(dy, wb, g, dx, x, False))
                row.append(f"{lab}:{fl / ms / 1e9:6.1f}")
            print(f"dgrad {name:8s} " + "  ".join(row), flush=True)
        lib.ssd_tune_set_igemm(-1, -1)
        if which in ("bf16", "all"):
            for dirn in ("fwd", "dgrad"):
                row = []
                for t, lab in ((-1, "auto"), (0, "256x128"), (1, "128x128"), (2, "128x64"), (3, "64x64")):
                    lib.ssd_tune_set_igemm_bf16(t)
                    if dirn == "fwd":
                        ms = timeit(lambda: ops.conv2d_fwd(x, wf, b, g, True, ld=ld, out=dy, bf16=True))
                    else:
                        ms = timeit(lambda: ops.conv2d_dgrad(dy, wb, g, dx, x, False, bf16=True))
                    row.append(f"{lab}:{fl / ms / 1e9:6.1f}")
                print(f"bf16 {dirn:5s} {name:8s} " + "  ".join(row), flush=True)
            lib.ssd_tune_set_igemm_bf16(-1)
        if which in ("halobf16",):
            wf3, wb3 = ops.weight_split3(wf), ops.weight_split3(wb)
            import ctypes as CC
            for hm, lab in ((1, "halo8x8"), (2, "halo8x16")):
                lib.ssd_tune_set_halo(hm)
                st = torch.cuda.current_stream().cuda_stream
                r1 = lib.ssd_conv3x3_halo_fwd_bf16(x.data_ptr(), wf3.data_ptr(), int(wf3.shape[1]), b.data_ptr(), dy.data_ptr(), ld, CC.byref(g), 1, st)
                if r1 != 0:
                    continue
                ms = timeit(lambda: lib.ssd_conv3x3_halo_fwd_bf16(x.data_ptr(), wf3.data_ptr(), int(wf3.shape[1]), b.data_ptr(), dy.data_ptr(), ld, CC.byref(g), 1, st))
                ms2 = timeit(lambda: lib.ssd_conv3x3_halo_dgrad_bf16(dy.data_ptr(), ld, wb3.data_ptr(), ld, dx.data_ptr(), x.data_ptr(), 0, CC.byref(g), st))
                print(f"halobf16 {name:8s} {lab}: fwd {fl / ms / 1e9:6.1f}  dgrad {fl / ms2 / 1e9:6.1f}", flush=True)
            lib.ssd_tune_set_halo(-1)
        if which in ("x3", "all"):
            wf3, wb3 = ops.weight_split3(wf), ops.weight_split3(wb)
            for dirn in ("fwd", "dgrad"):
                row = []
                for t, hm, lab in ((-1, -1, "auto"), (1, 0, "128x128"), (2, 0, "128x64"), (3, 0, "64x64"), (-1, 1, "halo8x8"), (-1, 2, "halo8x16")):
                    lib.ssd_tune_set_igemm_x3(t)
                    lib.ssd_tune_set_halo(hm)
                    if dirn == "fwd":
                        ms = timeit(lambda: ops.conv2d_fwd_x3(x, wf3, b, g, True, ld=ld))
                    else:
                        ms = timeit(lambda: ops.conv2d_dgrad_x3(dy, wb3, g, dx, x, False))
                    row.append(f"{lab}:{fl / ms / 1e9:6.1f}")
                print(f"x3   {dirn:5s} {name:8s} " + "  ".join(row), flush=True)
            lib.ssd_tune_set_igemm_x3(-1)
            lib.ssd_tune_set_halo(-1)
        if which in ("wgrad", "all"):
            row = []
            for bt, nb, bpc, lab in wg_var:
                lib.ssd_tune_set_wgrad(bt, nb, bpc)
                ms = timeit(lambda: ops.conv2d_wgrad(x, dy, g, ld, True))
                row.append(f"{lab}:{fl / ms / 1e9:6.1f}")
            print(f"wgrad {name:8s} " + "  ".join(row), flush=True)
        lib.ssd_tune_set_wgrad(-1, -1, -1)
main()
