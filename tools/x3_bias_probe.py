"""Signed error of the plane GEMM kernels against f64: is the bf16-MFMA accumulation biased?  python tools/x3_bias_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from objectdetection_ssd_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    M, N, P = 1024, 256, 1
    for K in (256, 512, 2048):
        for name in ("gauss", "positive", "negative", "a>0,w gauss"):
            g = torch.Generator().manual_seed(3)
            a = torch.randn(P, M, K, generator=g)
            w = torch.randn(P, N, K, generator=g) / K ** 0.5
            if name == "positive":
                a, w = a.abs(), w.abs()
            if name == "negative":
                a, w = a.abs(), -w.abs()
            if name == "a>0,w gauss":
                a = a.abs()
            ad, wd = a.to(dev), w.to(dev)
            w3 = torch.zeros(lib.ssd_gemm_x3_weights_bytes(N, K, P), dtype=torch.uint8, device=dev)
            _lib.check(lib.ssd_gemm_x3_split_weights(wd.data_ptr(), w3.data_ptr(), N, K, P, st), "split")
            o3 = torch.empty(P, M, N, device=dev)
            o32 = torch.empty(P, M, N, device=dev)
            _lib.check(lib.ssd_gemm_planes_x3(ad.data_ptr(), w3.data_ptr(), o3.data_ptr(), M, K, N, N, P, st), "x3")
            _lib.check(lib.ssd_gemm_planes_f32(ad.data_ptr(), wd.data_ptr(), o32.data_ptr(), M, K, N, N, P, st), "f32")
            ref = torch.bmm(a.double(), w.double().transpose(1, 2))
            sc = float(ref.abs().mean())
            for tag, o in (("f32", o32), ("x3 ", o3)):
                e = o.cpu().double() - ref
                print(f"K={K:5d} {name:12s} {tag}: rel L2 {float(e.norm() / ref.norm()):.2e}   mean signed err / mean|ref| {float(e.mean()) / sc:+.2e}   "
                      f"(in f32 ulps of mean|ref|: {float(e.mean()) / (sc * 2 ** -23):+.2f})")


if __name__ == "__main__":
    main()
