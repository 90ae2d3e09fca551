"""Calibration, not product: what the vendor library (hipBLASLt behind torch.matmul) reaches on THIS device for plain bf16 GEMMs on
random operands -- the ceiling against which the hand-written MFMA loops of csrc/ are read (cdna_hip_programming.md rule 10: never
infer a platform ceiling from your own kernels).  Prints TFLOP/s for a square GEMM and for the batched shapes of the Winograd plane
GEMMs (36 planes, M = tiles, K = N = channels), random normal and all-zero operands.
Measured (round 4, one MI355X): 8192^3 1 398 TFLOP/s on random normal operands (2 089 on zeros), 4096^3 1 434; the batched plane shapes
755 (conv4: 36 x [3200 x 512] x [512 x 512]), 384 (conv3) and 530 (conv5).  (A 36 x [3200 x 3072] x [3072 x 512] batch faulted inside the
library on this image and is left out.)"""
import torch

dev = torch.device("cuda:0")


def bench(a, b, iters=20):
    for _ in range(3):
        torch.matmul(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.matmul(a, b)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, (bt, m, k, n) in {"square 8192": (1, 8192, 8192, 8192), "square 4096": (1, 4096, 4096, 4096),
                            "conv4 planes": (36, 3200, 512, 512), "conv3 planes": (36, 11552, 256, 256),
                            "conv5 planes": (36, 800, 512, 512)}.items():
    for fill in ("randn", "zeros"):
        a = (torch.randn(bt, m, k, device=dev) if fill == "randn" else torch.zeros(bt, m, k, device=dev)).to(torch.bfloat16)
        b = (torch.randn(bt, n, k, device=dev) if fill == "randn" else torch.zeros(bt, n, k, device=dev)).to(torch.bfloat16)
        ms = bench(a, b.transpose(1, 2))
        print(f"{name:24s} {fill:6s} {bt}x[{m}x{k}]x[{k}x{n}]  {ms:8.4f} ms  {2.0 * bt * m * k * n / ms / 1e9:9.1f} TFLOP/s", flush=True)
