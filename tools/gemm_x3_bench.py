"""Plane GEMMs of the Winograd layers: f32 MFMA kernel vs the three-limb bf16 MFMA kernel (csrc/gemm_x3.hip), same operands.

    python tools/gemm_x3_bench.py            # the batch-32 shapes of conv3_2, conv4_2, conv5_2, fc6, c_4
Prints per shape: ms and TFLOP/s (f32-equivalent: 2*M*K*N*P) of both, and the relative L2 error of each against an f64 product
of plane 0.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from objectdetection_ssd_amd import _lib, ops  # noqa: E402

SHAPES = [("conv3_2", 32 * 19 * 19, 256, 256, 36), ("conv4_2", 32 * 10 * 10, 512, 512, 36), ("conv5_2", 32 * 5 * 5, 512, 512, 36),
          ("fc6", 32 * 8 * 8, 512, 1024, 36), ("c_4", 32 * 10 * 10, 512, 100, 36), ("dgrad conv3_1", 32 * 19 * 19, 256, 128, 36)]


def main():
    lib = _lib.load()
    if not lib.ssd_has_experimental():
        raise SystemExit("tools/gemm_x3_bench.py compares the shipped kernel with the experiments of csrc/gemm_x3v2.hip: build with SSD_EXPERIMENTAL=1")
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for name, M, K, N, P in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(1)
        a = torch.randn(P, M, K, generator=g).to(dev)
        w = (torch.randn(P, N, K, generator=g) / K ** 0.5).to(dev)
        o32 = torch.empty(P, M, N, device=dev)
        ox3 = torch.empty(P, M, N, device=dev)
        w3 = torch.empty(lib.ssd_gemm_x3_weights_bytes(N, K, P), dtype=torch.uint8, device=dev)
        _lib.check(lib.ssd_gemm_x3_split_weights(w.data_ptr(), w3.data_ptr(), N, K, P, st), "split")
        f32 = lambda: _lib.check(lib.ssd_gemm_planes_f32(a.data_ptr(), w.data_ptr(), o32.data_ptr(), M, K, N, N, P, st), "f32")   # noqa: E731
        x3 = lambda: _lib.check(lib.ssd_gemm_planes_x3(a.data_ptr(), w3.data_ptr(), ox3.data_ptr(), M, K, N, N, P, st), "x3")    # noqa: E731
        res = {}
        a3 = torch.empty(lib.ssd_gemm_x3_weights_bytes(M, K, P), dtype=torch.uint8, device=dev)     # the activation operand as limb planes too
        _lib.check(lib.ssd_gemm_x3_split_weights(a.data_ptr(), a3.data_ptr(), M, K, P, st), "split a")
        ov2 = torch.empty(P, M, N, device=dev)
        v2 = lambda: _lib.check(lib.ssd_gemm_planes_x3v2(a3.data_ptr(), w3.data_ptr(), ov2.data_ptr(), M, K, N, N, P, 0, st), "x3v2")   # noqa: E731

        def x3():
            _lib.check(lib.ssd_tune_set_x3_big(0), "tune")
            _lib.check(lib.ssd_gemm_planes_x3(a.data_ptr(), w3.data_ptr(), ox3.data_ptr(), M, K, N, N, P, st), "x3")

        def x3s():
            _lib.check(lib.ssd_tune_set_x3_big(2), "tune")
            try:
                _lib.check(lib.ssd_gemm_planes_x3(a.data_ptr(), w3.data_ptr(), os_.data_ptr(), M, K, N, N, P, st), "x3s")
            finally:
                _lib.check(lib.ssd_tune_set_x3_big(0), "tune")
        os_ = torch.empty(P, M, N, device=dev)

        def x3m16():
            _lib.check(lib.ssd_tune_set_x3_mfma(16), "tune")
            try:
                x3()
            finally:
                _lib.check(lib.ssd_tune_set_x3_mfma(32), "tune")
        runs = {"f32": [], "x3": [], "x3m16": [], "v2": [], "x3s": []}
        for rnd in range(3):                       # interleaved rounds in one process (variants ranked on one device, one minute)
            for tag, fn in (("f32", f32), ("x3", x3), ("x3m16", x3m16), ("v2", v2), ("x3s", x3s)):
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                pa = ops.clock_probe(dev)
                for _ in range(20):
                    fn()
                pb = ops.clock_probe(dev)
                e1.record()
                torch.cuda.synchronize()
                runs[tag].append((e0.elapsed_time(e1) / 20, ops.shader_mhz(pa, pb)))
        for tag, r in runs.items():
            r.sort()
            res[tag], res[tag + "_mhz"] = r[1]
        v2()
        torch.cuda.synchronize()
        ev2 = float((ov2[0].double() - a[0].double() @ w[0].double().T).norm() / (a[0].double() @ w[0].double().T).norm())
        ev2l = float((ov2[P - 1].double() - a[P - 1].double() @ w[P - 1].double().T).norm() / (a[P - 1].double() @ w[P - 1].double().T).norm())
        x3s()
        torch.cuda.synchronize()
        x3()
        torch.cuda.synchronize()
        es = float((os_[0].double() - a[0].double() @ w[0].double().T).norm() / (a[0].double() @ w[0].double().T).norm())
        same = bool(torch.equal(os_, ox3))
        x3m16()
        o16 = ox3.clone()
        x3()
        e16 = float((o16[0].double() - a[0].double() @ w[0].double().T).norm() / (a[0].double() @ w[0].double().T).norm())
        ref = a[0].double() @ w[0].double().T
        err = {t: float((o[0].double() - ref).norm() / ref.norm()) for t, o in (("f32", o32), ("x3", ox3))}
        last = P - 1
        ref2 = a[last].double() @ w[last].double().T
        err2 = float((ox3[last].double() - ref2).norm() / ref2.norm())
        fl = 2.0 * M * K * N * P
        if os.environ.get("X3_BRIEF"):
            print(f"{name:14s} x3 {res['x3']:.3f}  x3m16 {res['x3m16']:.3f}  v2 {res['v2']:.3f}  x3s {res['x3s']:.3f} ms   (executed x3s {6 * fl / res['x3s'] / 1e9:6.0f} TF/s, == x3: {same})",
                  flush=True)
            continue
        print(f"{name:14s} M={M:6d} K={K:4d} N={N:4d}  f32 {res['f32']:.3f} ms {fl / res['f32'] / 1e9:7.1f} TF/s   x3 {res['x3']:.3f} ms "
              f"{fl / res['x3'] / 1e9:7.1f} TF/s (executed bf16 {6 * fl / res['x3'] / 1e9:7.1f})   x3/16x16x32 {res['x3m16']:.3f} ms (executed {6 * fl / res['x3m16'] / 1e9:7.1f}, err {e16:.2e}, {res['x3m16_mhz']:.0f} MHz)   v2 256x256 ping-pong {res['v2']:.3f} ms (executed {6 * fl / res['v2'] / 1e9:7.1f}, err {ev2:.2e} / {ev2l:.2e}, {res['v2_mhz']:.0f} MHz)   x3s 256x256 in-kernel split {res['x3s']:.3f} ms (executed {6 * fl / res['x3s'] / 1e9:7.1f}, err {es:.2e}, bitwise == x3: {same})   err f32 {err['f32']:.2e} x3 {err['x3']:.2e} / {err2:.2e}  clock f32 {res['f32_mhz']:.0f} x3 {res['x3_mhz']:.0f} MHz",
              flush=True)


if __name__ == "__main__":
    main()
