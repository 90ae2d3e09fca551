"""`SSD_resnet34` -- drop-in for reference Model.py:12-126 (BASELINE configs[4]), eval-mode forward on the gfx950 kernels.

The module tree mirrors the reference's so that `state_dict()` / `load_state_dict()` exchange checkpoints with it
(keys and shapes pinned by tests/golden/resnet34.npz): `resnet.*` is the torchvision ResNet-34 layer list (built from
torch.nn here -- torchvision is a download in the reference), `seq1..seq5` are views of it, and the SSD blocks keep the
reference's names including the unused `conv2d_03` / `bn4` / `bn2` / `bn1`.

Scope: the reference has no loss wired to this model (Losses.py:6-7 are the SSD300 priors)
and its train mode draws Dropout / Dropout2d masks from the global RNG stream, so only `.eval()` can be matched.
`forward` in train mode raises instead of silently computing something else.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops

_STAGES = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))       # (channels, blocks, stride of the first block)
_BN_EPS = 1e-5


class _BasicBlock(nn.Module):
    """Parameter container with torchvision's BasicBlock attribute names (conv1, bn1, conv2, bn2, downsample)."""

    def __init__(self, cin: int, c: int, stride: int):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, c, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(c)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(c, c, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(c)
        self.downsample = None
        if stride != 1 or cin != c:
            self.downsample = nn.Sequential(nn.Conv2d(cin, c, 1, stride, bias=False), nn.BatchNorm2d(c))
        self.stride = stride


class _ResNet34(nn.Module):
    """children() order conv1, bn1, relu, maxpool, layer1..4, avgpool, fc -- what Model.py:22-30 slices."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for li, (c, nblk, stride) in enumerate(_STAGES, start=1):
            blocks = []
            for b in range(nblk):
                blocks.append(_BasicBlock(cin, c, stride if b == 0 else 1))
                cin = c
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


def _bn_affine(bn: nn.BatchNorm2d) -> Tuple[torch.Tensor, torch.Tensor]:
    """eval-mode BatchNorm as y = x*scale + shift"""
    scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
    return scale, bn.bias.detach() - bn.running_mean * scale


class SSD_resnet34(nn.Module):
    """Reference Model.py:12-126: `SSD_resnet34(n_classes, dropout_p=0.4, k=3)`; input (bs,3,224,224) NCHW f32;
    returns (bs, 21k, 4), (bs, 21k, n_classes+1) with the reference's hard-coded `.view(..., 21)` (so n_classes = 20)."""

    def __init__(self, n_classes, dropout_p=0.4, k=3):
        super().__init__()
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.k = k
        self.n_classes = n_classes
        self.dropout_p = dropout_p
        if n_classes + 1 != 21:
            raise ValueError("the reference reshapes the class scores with a hard-coded 21 (Model.py:117): n_classes must be 20")
        self.resnet = _ResNet34()
        self.resnet_layers = list(self.resnet.children())
        self.relu = nn.ReLU()
        self.drop = nn.Dropout(p=0.4)
        L = self.resnet_layers
        self.seq1 = nn.Sequential(*L[0:3])
        self.seq2 = nn.Sequential(*L[3:5])
        self.seq3 = nn.Sequential(*L[5])
        self.seq4 = nn.Sequential(*L[6])
        self.seq5 = nn.Sequential(*L[7])
        self.conv2d_0 = self.conv2d(512, 256, kernel=3, stride=1, padding=1)
        self.conv2d_01 = self.conv2d(256, 256, kernel=3, stride=2, padding=1)
        self.conv2d_02 = self.conv2d(256, 256, kernel=3, stride=2, padding=1)
        self.conv2d_03 = self.conv2d(256, 256, kernel=3, stride=2, padding=1)
        for s in ("4", "2", "1"):
            setattr(self, f"conv2d_02_bb{s}", self.conv2d_final(256, 4 * k, kernel=3, stride=1, padding=1))
            c = nn.Conv2d(256, (n_classes + 1) * k, kernel_size=3, stride=1, padding=1)
            c.bias.data.zero_().add_(-2)                                   # Model.py:39,43,47
            setattr(self, f"conv2d_02_c{s}", c)
        self.bn4 = nn.BatchNorm2d((n_classes + 1) * k)
        self.bn2 = nn.BatchNorm2d((n_classes + 1) * k)
        self.bn1 = nn.BatchNorm2d((n_classes + 1) * k)
        self._cache: Dict[str, tuple] = {}
        self.conv_dtype = "f32"            # "bf16": bf16-operand MFMA kernels with f32 accumulation

    def conv2d(self, in_channels, out_channels, kernel=1, stride=1, padding=0):
        return nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=kernel, stride=stride, padding=padding),
                             nn.ReLU(), nn.BatchNorm2d(out_channels), nn.Dropout2d(p=0.4))

    def conv2d_final(self, in_channels, out_channels, kernel=1, stride=1, padding=0):
        return nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=kernel, stride=stride, padding=padding),
                             nn.BatchNorm2d(out_channels), nn.Dropout2d(p=0.4))

    def invalidate_weight_cache(self) -> None:
        """Drop the BatchNorm-folded kernel-layout weights.  They are rebuilt when any tensor's `(data_ptr, _version)` changes;
        writes through `.data` change neither, so call this after such a write (same contract as `SSD_300`)."""
        self._cache.clear()

    # -- prepared (BatchNorm-folded, kernel-layout) weights, rebuilt when any tensor of the module changes ---------------------
    def _signature(self):
        return tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    def _prepared(self):
        sig = self._signature()
        ent = self._cache.get("w")
        if ent is not None and ent[0] == sig:
            return ent[1]
        W: Dict[str, tuple] = {}

        def fold(conv: nn.Conv2d, bn: Optional[nn.BatchNorm2d]):
            w = conv.weight.detach()
            b = conv.bias.detach() if conv.bias is not None else torch.zeros(w.shape[0], device=w.device)
            if bn is not None:
                scale, shift = _bn_affine(bn)
                w = w * scale.view(-1, 1, 1, 1)
                b = b * scale + shift
            return w.contiguous(), b.contiguous()

        with torch.no_grad():
            w, b = fold(self.resnet.conv1, self.resnet.bn1)
            W["stem"] = (ops.stem_weight_rows(w), b)
            for li in range(1, 5):
                for bi, blk in enumerate(getattr(self.resnet, f"layer{li}")):
                    p = f"layer{li}.{bi}."
                    for cname, bname in (("conv1", "bn1"), ("conv2", "bn2")):
                        w, b = fold(getattr(blk, cname), getattr(blk, bname))
                        W[p + cname] = (ops.weight_ohwi(w), b)
                    if blk.downsample is not None:
                        w, b = fold(blk.downsample[0], blk.downsample[1])
                        W[p + "down"] = (ops.weight_ohwi(w), b)
            for name in ("conv2d_0", "conv2d_01", "conv2d_02"):
                seq = getattr(self, name)
                w, b = fold(seq[0], None)
                scale, shift = _bn_affine(seq[2])
                W[name] = (ops.weight_ohwi(w), b, scale.contiguous(), shift.contiguous())
            for s in ("4", "2", "1"):
                bb = getattr(self, f"conv2d_02_bb{s}")
                wb, bbias = fold(bb[0], bb[1])                     # loc head: Conv -> BN folds
                wc, cbias = fold(getattr(self, f"conv2d_02_c{s}"), None)
                w = torch.cat((wb, wc), 0)
                W["head" + s] = (ops.weight_ohwi(w, ops.pad32(w.shape[0])), torch.cat((bbias, cbias)).contiguous())
        self._cache["w"] = (sig, W)
        return W

    def forward(self, x):
        if self.training:
            raise RuntimeError("SSD_resnet34 runs in eval mode only: the reference's train-mode forward draws Dropout/Dropout2d "
                               "masks and BatchNorm batch statistics (Model.py:24,56-70,88) that cannot be reproduced; call .eval()")
        if x.dim() != 4 or x.shape[1] != 3 or x.dtype != torch.float32:
            raise ValueError(f"SSD_resnet34 expects float32 (bs,3,H,W) NCHW input, got {tuple(x.shape)} {x.dtype}")
        if not x.is_cuda:
            raise RuntimeError("SSD_resnet34 runs on the gfx950 HIP kernels only: move the model and the input to the GPU "
                               "(there is no CPU fallback)")
        if self.conv_dtype not in ("f32", "bf16"):
            raise ValueError("conv_dtype must be 'f32' or 'bf16'")
        bf16 = self.conv_dtype == "bf16"
        W = self._prepared()
        x = x.contiguous()
        bs = x.shape[0]

        def conv(h, key, ci, co, k, s, pad, relu, out=None, accumulate=False, ld=None):
            g = ops.make_geom(bs, h.shape[1], h.shape[2], ci, co, k, s, pad, 1)
            return ops.conv2d_fwd(h, W[key][0], W[key][1], g, relu, ld=ld, out=out, bf16=bf16, accumulate=accumulate)

        with torch.no_grad():
            col = ops.im2col_nchw3(x, 7, 2, 3)                                     # seq1: conv 7x7/s2 + BN + ReLU
            g = ops.make_geom(bs, col.shape[1], col.shape[2], col.shape[3], 64, 1, 1, 0, 1)
            h = ops.conv2d_fwd(col, W["stem"][0], W["stem"][1], g, True, bf16=bf16)
            h, _ = ops.maxpool_fwd(h, 3, 2, 1, False, want_argmax=False)           # seq2[0]
            cin = 64
            for li, (c, nblk, stride) in enumerate(_STAGES, start=1):              # seq2[1], seq3, seq4, seq5
                for bi in range(nblk):
                    p = f"layer{li}.{bi}."
                    s = stride if bi == 0 else 1
                    o = conv(h, p + "conv1", cin, c, 3, s, 1, True)
                    idt = conv(h, p + "down", cin, c, 1, s, 0, False) if (p + "down") in W else h
                    h = conv(o, p + "conv2", c, c, 3, 1, 1, True, out=idt, accumulate=True)    # relu(bn2(conv2) + identity)
                    cin = c
            # Model.py:88 relu (h is already >= 0) and dropout (identity in eval)
            feats = []
            for name, ci, s in (("conv2d_0", 512, 1), ("conv2d_01", 256, 2), ("conv2d_01", 256, 2), ("conv2d_02", 256, 2)):
                h = conv(h, name, ci, 256, 3, s, 1, True)
                h = ops.channel_affine(h, W[name][2], W[name][3], out=h)
                feats.append(h)
            k = self.k
            co = 25 * k
            total = sum(f.shape[1] * f.shape[2] for f in feats[1:]) * k
            loc = torch.empty((bs, total, 4), device=x.device, dtype=torch.float32)
            conf = torch.empty((bs, total, 21), device=x.device, dtype=torch.float32)
            off = 0
            for s, f in zip(("4", "2", "1"), feats[1:]):
                packed = conv(f, "head" + s, 256, co, 3, 1, 1, False, ld=ops.pad32(co))
                hw = f.shape[1] * f.shape[2]
                ops.heads_scatter(packed, ops.pad32(co), loc, conf, bs, hw, k, off)
                off += hw * k
        return loc, conf
