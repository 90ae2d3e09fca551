"""Kernel bench of csrc/conv_bf16.hip on the VGG / head shapes of the train step at batch 32 (GPU box):
    python tools/conv_bf16_bench.py [bs] [mode] [bn]
forward and data gradient of every 3x3 / stride-1 layer, TFLOP/s of direct-convolution work, with a spot check against torch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objectdetection_ssd_amd import _lib, ops

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mode = int(sys.argv[2]) if len(sys.argv) > 2 else -1
bn = int(sys.argv[3]) if len(sys.argv) > 3 else -1
_lib.check(_lib.load().ssd_tune_set_conv_bf16(mode, bn), "tune")
if len(sys.argv) > 4:
    _lib.check(_lib.load().ssd_tune_set_conv_bf16_k64(int(sys.argv[4])), "tune")
if len(sys.argv) > 5:
    _lib.check(_lib.load().ssd_tune_set_conv_bf16_mfma(int(sys.argv[5])), "tune")
dev = "cuda:0"
LAYERS = [("conv1_2", 300, 64, 64), ("conv2_1", 150, 64, 128), ("conv2_2", 150, 128, 128), ("conv3_1", 75, 128, 256),
          ("conv3_2", 75, 256, 256), ("conv4_1", 38, 256, 512), ("conv4_2", 38, 512, 512), ("conv5_1", 19, 512, 512),
          ("c_4", 38, 512, 128), ("c_7", 19, 1024, 192)]
g = torch.Generator().manual_seed(0)
tot = {0: [0.0, 0.0], 1: [0.0, 0.0]}
for name, hw, ci, co in LAYERS:
    x = torch.randn(bs, hw, hw, ci, generator=g).to(dev).bfloat16()
    w = (torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (9 * ci)) ** 0.5).to(dev)
    w_f = w.permute(0, 2, 3, 1).reshape(co, 9, ci).contiguous().bfloat16()
    w_b = w.permute(1, 2, 3, 0).reshape(ci, 9, co).contiguous().bfloat16()
    bias = torch.randn(co, generator=g).to(dev)
    dy = torch.randn(bs, hw, hw, co, generator=g).to(dev).bfloat16()
    flops = 2.0 * bs * hw * hw * co * ci * 9
    for flip, (a, wt, nout) in enumerate(((x, w_f, co), (dy, w_b, ci))):
        if flip == 1 and (name.startswith("c_")):
            pass
        fn = lambda: ops.conv3x3_bf16(a, wt, bias if flip == 0 else None, nout, relu=(flip == 0), flip=bool(flip),
                                      relu_mask=(x if flip == 1 else None))
        for _ in range(3):
            y = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n_it = 10
        for _ in range(n_it):
            y = fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n_it
        tot[flip][0] += ms; tot[flip][1] += flops
        # spot check: image 0 against torch on the same bf16 operands (f32 accumulate)
        with torch.no_grad():
            if flip == 0:
                ref = torch.nn.functional.conv2d(a[:1].float().permute(0, 3, 1, 2), wt.float().reshape(co, 3, 3, ci).permute(0, 3, 1, 2), bias, padding=1).relu()
            else:
                ref = torch.nn.functional.conv_transpose2d(a[:1].float().permute(0, 3, 1, 2), wt.float().reshape(ci, 3, 3, co).permute(3, 0, 1, 2), padding=1)
                ref = ref * (x[:1].float().permute(0, 3, 1, 2) > 0)
            err = float((y[:1].float().permute(0, 3, 1, 2) - ref).abs().max() / ref.abs().max())
        print(f"{name:8s} {'dgrad' if flip else 'fwd  '} {hw:4d}^2 {ci:5d}->{co:4d}  {ms:7.3f} ms  {flops / ms / 1e9:7.1f} TF/s   rel err {err:.1e}", flush=True)
for flip in (0, 1):
    print(f"{'dgrad' if flip else 'fwd'} total {tot[flip][0]:.3f} ms, {tot[flip][1] / tot[flip][0] / 1e9:.1f} TF/s")
