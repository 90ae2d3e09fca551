"""torch-tensor front ends of the C ABI (device memory + streams are torch's; the
arithmetic is libssd_gfx950.so's).  Every wrapper validates what the kernels
assume (device, dtype, contiguity, shapes) before enqueueing -- a wrong shape
must never reach a hand-written kernel.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ConvGeom, check


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


_current_device = torch.cuda.current_device


def _req(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the gfx950 path needs a device tensor (got {t.device}); there is no CPU fallback")
    if t.device.index != _current_device():
        # the kernels are enqueued on the CURRENT device's stream (`_stream()`): a tensor of another GPU would be a foreign pointer there
        raise RuntimeError(f"{name}: tensor lives on {t.device} but the current device is cuda:{_current_device()}; run the call under "
                           f"`torch.cuda.device({t.device.index})` (one process per GPU: `torch.cuda.set_device(local_rank)`)")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _grad_dst(dst: Optional[torch.Tensor], shape, device, name: str) -> torch.Tensor:
    """Destination of a parameter gradient: a fresh tensor, or the caller's buffer (ddp.py hands out views of its flat gradient
    buffer, so the kernel writes the gradient where the all-reduce reads it)."""
    if dst is None:
        return torch.empty(shape, device=device, dtype=torch.float32)
    _req(dst, name)
    n = 1
    for d in shape:
        n *= d
    if dst.numel() != n or dst.device != device:
        raise ValueError(f"{name}: destination has {dst.numel()} elements on {dst.device}, the gradient has {n} on {device}")
    return dst.view(shape)


def conv_out_hw(h: int, w: int, k: int, stride: int, pad: int, dil: int) -> Tuple[int, int]:
    return ((h + 2 * pad - dil * (k - 1) - 1) // stride + 1, (w + 2 * pad - dil * (k - 1) - 1) // stride + 1)


def make_geom(n, h, w, ci, co, k, stride, pad, dil) -> ConvGeom:
    ho, wo = conv_out_hw(h, w, k, stride, pad, dil)
    return ConvGeom(n, h, w, ci, ho, wo, co, k, k, stride, pad, dil)


def pad32(c: int) -> int:
    return (c + 31) // 32 * 32


def igemm_tile(g: ConvGeom, direction: int, bf16: bool = False, x3: bool = False) -> str:
    """Name prefix of the kernel the library picks for this geometry (bench.py groups per-launch timings by it)."""
    if (x3 or bf16) and g.R == 3 and g.S == 3 and g.stride == 1 and g.dil == 1 and g.pad == 1 and (direction == 1 or g.Ci % 32 == 0):
        ho, wo = (g.Ho, g.Wo) if direction == 0 else (g.H, g.W)
        if ho >= 30 and wo >= (64 if x3 else 30):
            return "conv3x3_halo_kernel"
    if x3:
        return "igemm_x3_kernel"
    if bf16:
        return "igemm_bf16_kernel"
    bm, bn = C.c_int(0), C.c_int(0)
    check(_lib.load().ssd_conv2d_igemm_tile(C.byref(g), direction, C.byref(bm), C.byref(bn)), "igemm_tile")
    return f"igemm_kernel<{bm.value}, {bn.value}"


def wgrad_tile(g: ConvGeom, bf16: bool = False) -> str:
    if bf16 and g.stride == 1 and g.R == g.S and ((g.R == 3 and g.pad == g.dil and g.dil in (1, 4)) or (g.R == 1 and g.pad == 0)):
        return "wgrad3x3_bf16_kernel"                   # the bf16 patch kernel (csrc/conv_wgrad.hip plan_wgrad)
    bt, ns = C.c_int(0), C.c_int(0)
    check(_lib.load().ssd_conv2d_wgrad_tile(C.byref(g), C.byref(bt), C.byref(ns)), "wgrad_tile")
    return "wgrad3x3_kernel" if bt.value == 3 else f"wgrad_kernel<{bt.value}"


def conv_flops(g: ConvGeom) -> float:
    """algorithmic FLOPs of one pass (fwd, dgrad or wgrad all cost 2*M*Co*K)"""
    return 2.0 * g.N * g.Ho * g.Wo * g.Co * g.R * g.S * g.Ci


def wino_tiles(g: ConvGeom) -> int:
    """4x4 output tiles of the F(4x4,3x3) form of a 3x3 / stride-1 convolution with padding = dilation D: the D x D sub-lattices of the
    map are tiled one by one (csrc/winograd.hip, struct Lat); D = 1 is the plain ceil(H/4) x ceil(W/4) grid."""
    d = g.dil
    th = sum((((g.H - a + d - 1) // d) + 3) // 4 for a in range(d) if a < g.H)
    tw = sum((((g.W - a + d - 1) // d) + 3) // 4 for a in range(d) if a < g.W)
    return g.N * th * tw


def wino_flops(g: ConvGeom):
    """(direct-convolution FLOPs, FLOPs the 36 batched GEMMs of the F(4x4,3x3) form execute: 36 multiplies per 4x4 output tile
    instead of 144, on the tile grid padded to multiples of 4, output channels padded to 32)"""
    return conv_flops(g), 2.0 * 36 * wino_tiles(g) * pad32(g.Co) * g.Ci


# ---- weights ---------------------------------------------------------------------------
def weight_ohwi(w_oihw: torch.Tensor, co_pad: Optional[int] = None) -> torch.Tensor:
    _req(w_oihw, "weight")
    co, ci, r, s = w_oihw.shape
    co_pad = co if co_pad is None else co_pad
    out = torch.empty((co_pad, r * s, ci), device=w_oihw.device, dtype=torch.float32)
    check(_lib.load().ssd_weight_oihw_to_ohwi(w_oihw.data_ptr(), out.data_ptr(), co, ci, r, s, co_pad, _stream()), "weight_ohwi")
    return out


def weight_ihwo(w_oihw: torch.Tensor, co_pad: Optional[int] = None) -> torch.Tensor:
    _req(w_oihw, "weight")
    co, ci, r, s = w_oihw.shape
    co_pad = pad32(co) if co_pad is None else co_pad
    out = torch.empty((ci, r * s, co_pad), device=w_oihw.device, dtype=torch.float32)
    check(_lib.load().ssd_weight_oihw_to_ihwo(w_oihw.data_ptr(), out.data_ptr(), co, ci, r, s, co_pad, _stream()), "weight_ihwo")
    return out


def weight_split3(w_layout: torch.Tensor) -> torch.Tensor:
    """f32 weights in a kernel layout (OHWI / IHWO) -> (3, *shape) bf16 limb planes hi, mid, lo with hi+mid+lo == w exactly"""
    _req(w_layout, "w_layout")
    out = torch.empty((3,) + tuple(w_layout.shape), device=w_layout.device, dtype=torch.bfloat16)
    check(_lib.load().ssd_weight_split_bf16x3(w_layout.data_ptr(), out.data_ptr(), w_layout.numel(), _stream()), "weight_split3")
    return out


def conv2d_fwd_x3(x: torch.Tensor, w3_ohwi: torch.Tensor, bias: Optional[torch.Tensor], g: ConvGeom, relu: bool,
                  ld: Optional[int] = None) -> torch.Tensor:
    _req(x, "x"); _req(w3_ohwi, "w3_ohwi", torch.bfloat16)
    if tuple(x.shape) != (g.N, g.H, g.W, g.Ci):
        raise ValueError("x shape does not match geometry")
    if w3_ohwi.dim() != 4 or w3_ohwi.shape[0] != 3 or w3_ohwi.shape[1] < g.Co or w3_ohwi.shape[2] != g.R * g.S or w3_ohwi.shape[3] != g.Ci:
        raise ValueError("w3_ohwi shape does not match geometry")
    if bias is not None:
        _req(bias, "bias")
    ld = g.Co if ld is None else ld
    out = torch.empty((g.N, g.Ho, g.Wo, ld), device=x.device, dtype=torch.float32) if ld == g.Co else \
        torch.zeros((g.N, g.Ho, g.Wo, ld), device=x.device, dtype=torch.float32)
    check(_lib.load().ssd_conv2d_fwd_x3(x.data_ptr(), w3_ohwi.data_ptr(), int(w3_ohwi.shape[1]), _ptr(bias), out.data_ptr(), ld,
                                        C.byref(g), int(relu), _stream()), "conv2d_fwd_x3")
    return out


def conv2d_dgrad_x3(dy: torch.Tensor, w3_ihwo: torch.Tensor, g: ConvGeom, dx: Optional[torch.Tensor] = None,
                    relu_mask: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    _req(dy, "dy"); _req(w3_ihwo, "w3_ihwo", torch.bfloat16)
    co_pad = w3_ihwo.shape[3]
    if dy.numel() != g.N * g.Ho * g.Wo * co_pad or w3_ihwo.shape[0] != 3 or w3_ihwo.shape[1] != g.Ci or w3_ihwo.shape[2] != g.R * g.S \
            or co_pad % 32 != 0:
        raise ValueError("dgrad_x3 shapes do not match geometry")
    if dx is None:
        if accumulate:
            raise ValueError("accumulate needs an existing dx")
        dx = torch.empty((g.N, g.H, g.W, g.Ci), device=dy.device, dtype=torch.float32)
    _req(dx, "dx")
    if dx.numel() != g.N * g.H * g.W * g.Ci:
        raise ValueError("dx size")
    if relu_mask is not None:
        _req(relu_mask, "relu_mask")
        if relu_mask.numel() != dx.numel():
            raise ValueError("relu_mask size")
    check(_lib.load().ssd_conv2d_dgrad_x3(dy.data_ptr(), co_pad, w3_ihwo.data_ptr(), co_pad, dx.data_ptr(), _ptr(relu_mask),
                                          int(accumulate), C.byref(g), _stream()), "conv2d_dgrad_x3")
    return dx


# ---- convolution -----------------------------------------------------------------------
def conv2d_fwd(x: torch.Tensor, w_ohwi: torch.Tensor, bias: Optional[torch.Tensor], g: ConvGeom, relu: bool,
               ld: Optional[int] = None, out: Optional[torch.Tensor] = None, bf16: bool = False,
               w3: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """x (N,H,W,Ci) -> (N,Ho,Wo,ld) (ld defaults to Co; columns Co..ld-1 are left untouched).
    bf16 with `w3` (weight_split3 of w_ohwi): 3x3/s1/p1 layers on large maps take the halo-tile kernel."""
    _req(x, "x"); _req(w_ohwi, "w_ohwi")
    if tuple(x.shape) != (g.N, g.H, g.W, g.Ci):
        raise ValueError(f"x shape {tuple(x.shape)} != geometry {(g.N, g.H, g.W, g.Ci)}")
    if w_ohwi.shape[0] < g.Co or w_ohwi.shape[1] != g.R * g.S or w_ohwi.shape[2] != g.Ci:
        raise ValueError("w_ohwi shape does not match geometry")
    if bias is not None:
        _req(bias, "bias")
        if bias.numel() != g.Co:
            raise ValueError("bias length")
    ld = g.Co if ld is None else ld
    if out is None:
        out = torch.empty((g.N, g.Ho, g.Wo, ld), device=x.device, dtype=torch.float32) if ld == g.Co else \
            torch.zeros((g.N, g.Ho, g.Wo, ld), device=x.device, dtype=torch.float32)
    else:
        _req(out, "out")
        if out.numel() != g.N * g.Ho * g.Wo * ld:
            raise ValueError("out size")
    if accumulate:                         # out += conv (+ bias), then ReLU: the residual tail of a ResNet BasicBlock
        if out is None:
            raise ValueError("accumulate needs an existing out")
        fn = _lib.load().ssd_conv2d_fwd_accum_bf16 if bf16 else _lib.load().ssd_conv2d_fwd_accum
        check(fn(x.data_ptr(), w_ohwi.data_ptr(), _ptr(bias), out.data_ptr(), ld, C.byref(g), int(relu), _stream()), "conv2d_fwd_accum")
        return out
    if bf16 and w3 is not None and g.Ci % 32 == 0:
        _req(w3, "w3", torch.bfloat16)
        if tuple(w3.shape) != (3,) + tuple(w_ohwi.shape):
            raise ValueError("w3 must be weight_split3(w_ohwi)")
        rc = _lib.load().ssd_conv3x3_halo_fwd_bf16(x.data_ptr(), w3.data_ptr(), int(w3.shape[1]), _ptr(bias), out.data_ptr(), ld,
                                                   C.byref(g), int(relu), _stream())
        if rc <= 0:
            check(rc, "conv3x3_halo_fwd_bf16")
            return out
    lib = _lib.load()
    nbytes = lib.ssd_conv2d_igemm_workspace(C.byref(g), 0)             # > 0 only for small grids with a deep K loop
    ws = workspace(nbytes, x.device, "igemm") if nbytes else None
    if bf16:
        check(lib.ssd_conv2d_fwd_bf16_ws(x.data_ptr(), w_ohwi.data_ptr(), _ptr(bias), out.data_ptr(), ld, C.byref(g), int(relu), _ptr(ws),
                                         ws.numel() if ws is not None else 0, _stream()), "conv2d_fwd_bf16")
        return out
    check(lib.ssd_conv2d_fwd_ws(x.data_ptr(), w_ohwi.data_ptr(), _ptr(bias), out.data_ptr(), ld, C.byref(g), int(relu), _ptr(ws),
                                ws.numel() if ws is not None else 0, _stream()), "conv2d_fwd")
    return out


def conv2d_dgrad(dy: torch.Tensor, w_ihwo: torch.Tensor, g: ConvGeom, dx: Optional[torch.Tensor] = None,
                 relu_mask: Optional[torch.Tensor] = None, accumulate: bool = False, bf16: bool = False,
                 w3: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(dy, "dy"); _req(w_ihwo, "w_ihwo")
    co_pad = w_ihwo.shape[2]
    if dy.numel() != g.N * g.Ho * g.Wo * co_pad:
        raise ValueError(f"dy has {dy.numel()} elements, geometry wants {g.N * g.Ho * g.Wo * co_pad}")
    if w_ihwo.shape[0] != g.Ci or w_ihwo.shape[1] != g.R * g.S or co_pad % 32 != 0 or co_pad < g.Co:
        raise ValueError("w_ihwo shape does not match geometry")
    if dx is None:
        if accumulate:
            raise ValueError("accumulate needs an existing dx")
        dx = torch.empty((g.N, g.H, g.W, g.Ci), device=dy.device, dtype=torch.float32)
    _req(dx, "dx")
    if dx.numel() != g.N * g.H * g.W * g.Ci:
        raise ValueError("dx size")
    if relu_mask is not None:
        _req(relu_mask, "relu_mask")
        if relu_mask.numel() != dx.numel():
            raise ValueError("relu_mask size")
    if bf16 and w3 is not None and g.Ci % 4 == 0:
        _req(w3, "w3", torch.bfloat16)
        if tuple(w3.shape) != (3,) + tuple(w_ihwo.shape):
            raise ValueError("w3 must be weight_split3(w_ihwo)")
        rc = _lib.load().ssd_conv3x3_halo_dgrad_bf16(dy.data_ptr(), co_pad, w3.data_ptr(), co_pad, dx.data_ptr(), _ptr(relu_mask),
                                                     int(accumulate), C.byref(g), _stream())
        if rc <= 0:
            check(rc, "conv3x3_halo_dgrad_bf16")
            return dx
    lib = _lib.load()
    nbytes = lib.ssd_conv2d_igemm_workspace(C.byref(g), 1)
    ws = workspace(nbytes, dy.device, "igemm") if nbytes else None
    if bf16:
        check(lib.ssd_conv2d_dgrad_bf16_ws(dy.data_ptr(), co_pad, w_ihwo.data_ptr(), co_pad, dx.data_ptr(), _ptr(relu_mask), int(accumulate),
                                           C.byref(g), _ptr(ws), ws.numel() if ws is not None else 0, _stream()), "conv2d_dgrad_bf16")
        return dx
    check(lib.ssd_conv2d_dgrad_ws(dy.data_ptr(), co_pad, w_ihwo.data_ptr(), co_pad, dx.data_ptr(), _ptr(relu_mask), int(accumulate),
                                  C.byref(g), _ptr(ws), ws.numel() if ws is not None else 0, _stream()), "conv2d_dgrad")
    return dx


def conv3x3_bf16(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], n_out: int, relu: bool, flip: bool = False,
                 out: Optional[torch.Tensor] = None, out_f32: bool = False, ldo: Optional[int] = None,
                 relu_mask: Optional[torch.Tensor] = None, accumulate: bool = False, K: Optional[int] = None) -> torch.Tensor:
    """3x3 / stride 1 / pad 1 convolution on bf16 tensors (csrc/conv_bf16.hip).  x (N,H,W,ldx) bf16; w (rows, 9, K) bf16 -- the
    OHWI copy of the weights for the forward, the IHWO copy with flip=True for the data gradient (x is then dy);
    -> out (N,H,W,ldo) bf16 (f32 with out_f32), columns 0..n_out-1 written: [accumulate: out +] conv (+ bias) -> [relu] -> [mask > 0]."""
    _req(x, "x", torch.bfloat16); _req(w, "w", torch.bfloat16)
    if x.dim() != 4 or w.dim() != 3 or w.shape[1] != 9:
        raise ValueError("conv3x3_bf16: x (N,H,W,ldx), w (rows,9,K)")
    n, h, wd, ldx = x.shape
    K = int(w.shape[2]) if K is None else K
    if K != w.shape[2] or K % 64 != 0 or K > ldx or n_out % 4 != 0:
        raise ValueError("conv3x3_bf16: K must be the weights' row length, a multiple of 64 and <= ldx; n_out % 4 == 0")
    ldo = n_out if ldo is None else ldo
    odt = torch.float32 if out_f32 else torch.bfloat16
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs an existing out")
        out = torch.empty((n, h, wd, ldo), device=x.device, dtype=odt) if ldo == n_out else torch.zeros((n, h, wd, ldo), device=x.device, dtype=odt)
    else:
        _req(out, "out", odt)
        if out.numel() != n * h * wd * ldo:
            raise ValueError("conv3x3_bf16: out size")
    if bias is not None:
        _req(bias, "bias")
        if bias.numel() < min(n_out, int(w.shape[0])):
            raise ValueError("conv3x3_bf16: bias needs one entry per weight row")
    if relu_mask is not None:
        _req(relu_mask, "relu_mask", torch.bfloat16)
        if relu_mask.numel() != out.numel() or out_f32:
            raise ValueError("conv3x3_bf16: relu_mask must have the layout of a bf16 out")
    check(_lib.load().ssd_conv3x3_bf16(x.data_ptr(), ldx, w.data_ptr(), int(w.shape[0]), K, _ptr(bias), out.data_ptr(), ldo, n_out,
                                       int(out_f32), _ptr(relu_mask), int(accumulate), int(relu), int(flip), n, h, wd, _stream()),
          "conv3x3_bf16")
    return out


def pad64(c: int) -> int:
    return (c + 63) // 64 * 64


def cast_bf16(x: torch.Tensor) -> torch.Tensor:
    """f32 -> bf16 (round to nearest even), same shape; numel % 8 == 0"""
    _req(x, "x")
    y = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    check(_lib.load().ssd_cast_f32_bf16(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "cast_f32_bf16")
    return y


def cast_f32(x: torch.Tensor) -> torch.Tensor:
    _req(x, "x", torch.bfloat16)
    y = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    check(_lib.load().ssd_cast_bf16_f32(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "cast_bf16_f32")
    return y


def conv1_first_fwd_bf16(x_nchw: torch.Tensor, w_rows: torch.Tensor, bias: Optional[torch.Tensor], relu: bool = True) -> torch.Tensor:
    """conv1_1 in the bf16-tensor mode: f32 NCHW image -> bf16 NHWC (N,H,W,64); operands rounded to bf16, f32 accumulate."""
    _req(x_nchw, "x"); _req(w_rows, "w_rows")
    n, c, h, w = x_nchw.shape
    if c != 3 or w_rows.numel() != 64 * 32:
        raise ValueError("conv1_first_fwd_bf16 expects 3 input channels and (64, 32) filter rows")
    if bias is not None:
        _req(bias, "bias")
    y = torch.empty((n, h, w, 64), device=x_nchw.device, dtype=torch.bfloat16)
    check(_lib.load().ssd_conv1_first_fwd_bf16(x_nchw.data_ptr(), w_rows.data_ptr(), _ptr(bias), y.data_ptr(), n, h, w, int(relu), _stream()),
          "conv1_first_fwd_bf16")
    return y


def conv1_first_wgrad_bf16(x_nchw: torch.Tensor, dy: torch.Tensor, want_bias: bool = True):
    """conv1_1 weight / bias gradient from the f32 NCHW image and the bf16 NHWC dy -> (dw rows (64,32,1,1), dbias or None), f32."""
    _req(x_nchw, "x"); _req(dy, "dy", torch.bfloat16)
    n, c, h, w = x_nchw.shape
    if c != 3 or tuple(dy.shape) != (n, h, w, 64):
        raise ValueError("conv1_first_wgrad_bf16: x (N,3,H,W) and dy (N,H,W,64)")
    lib = _lib.load()
    ws = workspace(lib.ssd_conv1_first_wgrad_workspace(n, h, w), x_nchw.device, "first_wgrad")
    dw = torch.empty((64, 32, 1, 1), device=x_nchw.device, dtype=torch.float32)
    db = torch.empty((64,), device=x_nchw.device, dtype=torch.float32) if want_bias else None
    check(lib.ssd_conv1_first_wgrad_bf16(x_nchw.data_ptr(), dy.data_ptr(), dw.data_ptr(), _ptr(db), n, h, w, ws.data_ptr(), ws.numel(), _stream()),
          "conv1_first_wgrad_bf16")
    return dw, db


def conv3x3_wgrad_bf16t(x: torch.Tensor, dy: torch.Tensor, g: ConvGeom, ldy: int, want_bias: bool = True,
                        dw_out: Optional[torch.Tensor] = None, db_out: Optional[torch.Tensor] = None):
    """Weight gradient of a 3x3 / s1 / p1 layer from bf16 x (N,H,W,Ci) and bf16 dy (N,H,W,ldy) -> (dw (Co,Ci,3,3), dbias) f32."""
    _req(x, "x", torch.bfloat16); _req(dy, "dy", torch.bfloat16)
    if tuple(x.shape) != (g.N, g.H, g.W, g.Ci) or dy.numel() != g.N * g.Ho * g.Wo * ldy or ldy < g.Co:
        raise ValueError("conv3x3_wgrad_bf16t: shapes do not match the geometry")
    lib = _lib.load()
    ws = workspace(lib.ssd_conv2d_wgrad_workspace(C.byref(g)), x.device)
    dw = _grad_dst(dw_out, (g.Co, g.Ci, 3, 3), x.device, "dw_out")
    db = _grad_dst(db_out, (g.Co,), x.device, "db_out") if want_bias else None
    check(lib.ssd_conv3x3_wgrad_bf16t(x.data_ptr(), dy.data_ptr(), ldy, dw.data_ptr(), _ptr(db), C.byref(g), ws.data_ptr(), ws.numel(), _stream()),
          "conv3x3_wgrad_bf16t")
    return dw, db


def heads_gather_bf16(dloc: torch.Tensor, dconf: torch.Tensor, ld: int, n: int, hw: int, a: int, prior_off: int) -> torch.Tensor:
    _req(dloc, "dloc"); _req(dconf, "dconf")
    p, ncls = dconf.shape[1], dconf.shape[2]
    if tuple(dloc.shape) != (n, p, 4) or dconf.shape[0] != n:
        raise ValueError("heads_gather shapes")
    packed = torch.empty((n * hw, ld), device=dloc.device, dtype=torch.bfloat16)
    check(_lib.load().ssd_heads_gather_bf16(dloc.data_ptr(), dconf.data_ptr(), packed.data_ptr(), ld, n, hw, a, prior_off, p, ncls, _stream()),
          "heads_gather_bf16")
    return packed


_ws_cache = {}


def workspace(nbytes: int, device, tag: str = "ws") -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream, tag); contents are dead between calls."""
    key = (str(device), _stream(), tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), device=device, dtype=torch.uint8)
        _ws_cache[key] = buf
    return buf


def conv2d_wgrad(x: torch.Tensor, dy: torch.Tensor, g: ConvGeom, ldy: int, want_bias: bool = True, bf16: bool = False,
                 dw_out: Optional[torch.Tensor] = None, db_out: Optional[torch.Tensor] = None):
    """-> (dw (Co,Ci,R,S) OIHW, dbias (Co,) or None); dw_out / db_out: write the gradients there (same element counts)."""
    _req(x, "x"); _req(dy, "dy")
    if tuple(x.shape) != (g.N, g.H, g.W, g.Ci):
        raise ValueError("x shape does not match geometry")
    if dy.numel() != g.N * g.Ho * g.Wo * ldy or ldy < g.Co:
        raise ValueError("dy size does not match geometry")
    lib = _lib.load()
    nbytes = lib.ssd_conv2d_wgrad_workspace(C.byref(g))
    ws = workspace(nbytes, x.device)
    dw = _grad_dst(dw_out, (g.Co, g.Ci, g.R, g.S), x.device, "dw_out")
    db = _grad_dst(db_out, (g.Co,), x.device, "db_out") if want_bias else None
    fn = lib.ssd_conv2d_wgrad_bf16 if bf16 else lib.ssd_conv2d_wgrad
    check(fn(x.data_ptr(), dy.data_ptr(), ldy, dw.data_ptr(), _ptr(db), C.byref(g), ws.data_ptr(), ws.numel(), _stream()),
          "conv2d_wgrad")
    return dw, db


def im2col_first(x_nchw: torch.Tensor) -> torch.Tensor:
    """(N,3,H,W) NCHW -> (N,H,W,32) rows of the 27 taps (k = (r*3+s)*3 + c) + 5 zero columns."""
    _req(x_nchw, "x")
    n, c, h, w = x_nchw.shape
    if c != 3:
        raise ValueError("im2col_first expects 3 input channels")
    out = torch.empty((n, h, w, 32), device=x_nchw.device, dtype=torch.float32)
    check(_lib.load().ssd_im2col_first(x_nchw.data_ptr(), out.data_ptr(), n, h, w, _stream()), "im2col_first")
    return out


def conv1_first_fwd(x_nchw: torch.Tensor, w_rows: torch.Tensor, bias: Optional[torch.Tensor], relu: bool = True, want_col: bool = False):
    """conv1_1 (3 -> 64 channels, 3x3, pad 1) + bias + ReLU from the NCHW batch in one kernel -> (y (N,H,W,64) NHWC, col or None);
    w_rows = `first_weight_rows(weight)`; want_col: also the (N,H,W,32) rows of `im2col_first` (the weight gradient's operand)."""
    _req(x_nchw, "x"); _req(w_rows, "w_rows")
    n, c, h, w = x_nchw.shape
    if c != 3 or w_rows.numel() != 64 * 32:
        raise ValueError("conv1_first_fwd expects 3 input channels and (64, 32) filter rows")
    if bias is not None:
        _req(bias, "bias")
        if bias.numel() != 64:
            raise ValueError("conv1_first_fwd: 64 biases")
    y = torch.empty((n, h, w, 64), device=x_nchw.device, dtype=torch.float32)
    col = torch.empty((n, h, w, 32), device=x_nchw.device, dtype=torch.float32) if want_col else None
    check(_lib.load().ssd_conv1_first_fwd(x_nchw.data_ptr(), w_rows.data_ptr(), _ptr(bias), y.data_ptr(), _ptr(col), n, h, w, int(relu), _stream()),
          "conv1_first_fwd")
    return y, col


def has_experimental() -> bool:
    """the library was built with SSD_EXPERIMENTAL=1 (the kernels that measured no better than the shipped ones)"""
    return bool(_lib.load().ssd_has_experimental())


def conv1_first_wino_fwd(x_nchw: torch.Tensor, w_rows: torch.Tensor, bias: Optional[torch.Tensor], want_bits: bool = True):
    """conv1_1 + bias + ReLU left as the F(4x4) input planes of the 64 -> 64 convolution behind it -> (planes (36, tiles, 64), ReLU bit
    words (tiles, 16) int64 or None): what `conv1_first_fwd` + that layer's input transform would have produced, bit for bit, without the
    activation tensor in between."""
    _req(x_nchw, "x"); _req(w_rows, "w_rows")
    n, c, h, w = x_nchw.shape
    if c != 3 or w_rows.numel() != 64 * 32:
        raise ValueError("conv1_first_wino_fwd expects 3 input channels and (64, 32) filter rows")
    if bias is not None:
        _req(bias, "bias")
        if bias.numel() != 64:
            raise ValueError("conv1_first_wino_fwd: 64 biases")
    tiles = n * ((h + 3) // 4) * ((w + 3) // 4)
    planes = torch.empty((36, tiles, 64), device=x_nchw.device, dtype=torch.float32)
    bits = torch.empty((tiles, 16), device=x_nchw.device, dtype=torch.int64) if want_bits else None
    check(_lib.load().ssd_conv1_first_wino_fwd(x_nchw.data_ptr(), w_rows.data_ptr(), _ptr(bias), planes.data_ptr(), _ptr(bits), n, h, w, _stream()),
          "conv1_first_wino_fwd")
    return planes, bits


def conv1_first_wgrad(x_nchw: torch.Tensor, dy: torch.Tensor, want_bias: bool = True):
    """Weight / bias gradient of conv1_1 from the NCHW input: -> (dw (64,32,1,1) rows for `first_weight_grad`, dbias (64,) or None)."""
    _req(x_nchw, "x"); _req(dy, "dy")
    n, c, h, w = x_nchw.shape
    if c != 3 or tuple(dy.shape) != (n, h, w, 64):
        raise ValueError("conv1_first_wgrad: x (N,3,H,W) and dy (N,H,W,64)")
    lib = _lib.load()
    ws = workspace(lib.ssd_conv1_first_wgrad_workspace(n, h, w), x_nchw.device, "first_wgrad")
    dw = torch.empty((64, 32, 1, 1), device=x_nchw.device, dtype=torch.float32)
    db = torch.empty((64,), device=x_nchw.device, dtype=torch.float32) if want_bias else None
    check(lib.ssd_conv1_first_wgrad(x_nchw.data_ptr(), dy.data_ptr(), dw.data_ptr(), _ptr(db), n, h, w, ws.data_ptr(), ws.numel(), _stream()),
          "conv1_first_wgrad")
    return dw, db


def im2col_nchw3(x_nchw: torch.Tensor, k: int, stride: int, pad: int, kpad: Optional[int] = None) -> torch.Tensor:
    """(N,3,H,W) -> (N,Ho,Wo,Kpad) rows with k = (r*k+s)*3 + c, zero-padded to a multiple of 32 (ResNet-34 stem)."""
    _req(x_nchw, "x")
    if x_nchw.dim() != 4 or x_nchw.shape[1] != 3:
        raise ValueError("im2col_nchw3 expects (N,3,H,W)")
    n, _, h, w = x_nchw.shape
    ho, wo = conv_out_hw(h, w, k, stride, pad, 1)
    kpad = pad32(k * k * 3) if kpad is None else kpad
    out = torch.empty((n, ho, wo, kpad), device=x_nchw.device, dtype=torch.float32)
    check(_lib.load().ssd_im2col_nchw3(x_nchw.data_ptr(), out.data_ptr(), n, h, w, k, k, stride, pad, ho, wo, kpad, _stream()),
          "im2col_nchw3")
    return out


def stem_weight_rows(w_oihw: torch.Tensor, kpad: Optional[int] = None) -> torch.Tensor:
    """(Co,3,R,S) -> (Co,1,Kpad) rows in im2col_nchw3's column order (the 1x1 OHWI layout)."""
    co, ci, r, s_ = w_oihw.shape
    kpad = pad32(r * s_ * ci) if kpad is None else kpad
    rows = torch.zeros((co, 1, kpad), device=w_oihw.device, dtype=torch.float32)
    rows[:, 0, :r * s_ * ci] = w_oihw.detach().permute(0, 2, 3, 1).reshape(co, -1)
    return rows


def channel_affine(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, relu: bool = False,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y[..., c] = x[..., c] * scale[c] + shift[c] over NHWC rows (eval-mode BatchNorm after a ReLU)."""
    _req(x, "x"); _req(scale, "scale"); _req(shift, "shift")
    c = x.shape[-1]
    if scale.numel() != c or shift.numel() != c or c % 4 != 0:
        raise ValueError("scale / shift must have one entry per channel, C % 4 == 0")
    out = torch.empty_like(x) if out is None else out
    check(_lib.load().ssd_channel_affine(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr(), x.numel() // c, c,
                                         int(relu), _stream()), "channel_affine")
    return out


def first_weight_rows(w_oihw: torch.Tensor) -> torch.Tensor:
    """(Co,3,3,3) OIHW -> (Co,1,32): rows ordered like im2col_first's columns, zero padded."""
    co = w_oihw.shape[0]
    rows = w_oihw.detach().permute(0, 2, 3, 1).reshape(co, 27)              # [co][(r,s),c]
    return torch.nn.functional.pad(rows, (0, 5)).reshape(co, 1, 32).contiguous()


def first_weight_grad(dw_rows: torch.Tensor) -> torch.Tensor:
    """inverse of first_weight_rows for the gradient: (Co,32,1,1) -> (Co,3,3,3) OIHW"""
    co = dw_rows.shape[0]
    return dw_rows.reshape(co, 32)[:, :27].reshape(co, 3, 3, 3).permute(0, 3, 1, 2).contiguous()


# ---- pooling / norm ----------------------------------------------------------------------
def pool_out(h: int, k: int, stride: int, pad: int, ceil_mode: bool) -> int:
    num = h + 2 * pad - k
    o = (-(-num // stride) if ceil_mode else num // stride) + 1
    if ceil_mode and (o - 1) * stride >= h + pad:      # torch: last window must start inside the input or left pad
        o -= 1
    return o


def maxpool_fwd(x: torch.Tensor, k: int, stride: int, pad: int, ceil_mode: bool, want_argmax: bool = True):
    """NHWC max pool -> (y, argmax codes or None); a bf16 x takes the bf16-tensor kernels (csrc/elementwise_bf16.hip)."""
    n, h, w, c = x.shape
    ho, wo = pool_out(h, k, stride, pad, ceil_mode), pool_out(w, k, stride, pad, ceil_mode)
    if x.dtype == torch.bfloat16:
        _req(x, "x", torch.bfloat16)
        y = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.bfloat16)
        am = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.uint8) if want_argmax else None
        check(_lib.load().ssd_maxpool_fwd_bf16(x.data_ptr(), y.data_ptr(), _ptr(am), n, h, w, c, k, stride, pad, ho, wo, _stream()),
              "maxpool_fwd_bf16")
        return y, am
    _req(x, "x")
    y = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.float32)
    am = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.uint8) if want_argmax else None
    check(_lib.load().ssd_maxpool_fwd(x.data_ptr(), y.data_ptr(), _ptr(am), n, h, w, c, k, stride, pad, ho, wo, _stream()),
          "maxpool_fwd")
    return y, am


def maxpool_bwd(dy: torch.Tensor, argmax: torch.Tensor, in_shape, k: int, stride: int, pad: int,
                dx: Optional[torch.Tensor] = None, relu_mask: Optional[torch.Tensor] = None, accumulate: bool = False,
                y_gate: Optional[torch.Tensor] = None):
    """y_gate: the pooled OUTPUT; with it (and no accumulate / relu_mask) the ReLU mask of the pool's input is applied as
    y > 0 per window, which reads a quarter of the bytes of relu_mask = x."""
    n, h, w, c = in_shape
    _, ho, wo, c2 = dy.shape
    if dy.dtype == torch.bfloat16:
        _req(dy, "dy", torch.bfloat16); _req(argmax, "argmax", torch.uint8)
        if c2 != c or tuple(argmax.shape) != tuple(dy.shape):
            raise ValueError("maxpool_bwd shapes")
        if y_gate is not None and (accumulate or relu_mask is not None):
            raise ValueError("y_gate replaces relu_mask and cannot accumulate")
        for t, nm in ((y_gate, "y_gate"), (relu_mask, "relu_mask")):
            if t is not None:
                _req(t, nm, torch.bfloat16)
        if y_gate is not None and tuple(y_gate.shape) != tuple(dy.shape):
            raise ValueError("y_gate must be the pooled output")
        if dx is None:
            if accumulate:
                raise ValueError("accumulate needs an existing dx")
            dx = torch.empty((n, h, w, c), device=dy.device, dtype=torch.bfloat16)
        _req(dx, "dx", torch.bfloat16)
        if dx.numel() != n * h * w * c or (relu_mask is not None and relu_mask.numel() != dx.numel()):
            raise ValueError("dx / relu_mask size")
        check(_lib.load().ssd_maxpool_bwd_bf16(dy.data_ptr(), argmax.data_ptr(), dx.data_ptr(), _ptr(relu_mask), _ptr(y_gate), int(accumulate),
                                               n, h, w, c, k, stride, pad, ho, wo, _stream()), "maxpool_bwd_bf16")
        return dx
    _req(dy, "dy"); _req(argmax, "argmax", torch.uint8)
    if c2 != c or tuple(argmax.shape) != tuple(dy.shape):
        raise ValueError("maxpool_bwd shapes")
    if y_gate is not None:
        if accumulate or relu_mask is not None:
            raise ValueError("y_gate replaces relu_mask and cannot accumulate")
        _req(y_gate, "y_gate")
        if tuple(y_gate.shape) != tuple(dy.shape):
            raise ValueError("y_gate must be the pooled output")
        dx = torch.empty((n, h, w, c), device=dy.device, dtype=torch.float32) if dx is None else dx
        _req(dx, "dx")
        if dx.numel() != n * h * w * c:
            raise ValueError("dx size")
        check(_lib.load().ssd_maxpool_bwd_gated(dy.data_ptr(), argmax.data_ptr(), y_gate.data_ptr(), dx.data_ptr(), n, h, w, c, k, stride,
                                                pad, ho, wo, _stream()), "maxpool_bwd_gated")
        return dx
    if dx is None:
        if accumulate:
            raise ValueError("accumulate needs an existing dx")
        dx = torch.empty((n, h, w, c), device=dy.device, dtype=torch.float32)
    _req(dx, "dx")
    if dx.numel() != n * h * w * c:
        raise ValueError("dx size")
    if relu_mask is not None:
        _req(relu_mask, "relu_mask")
        if relu_mask.numel() != dx.numel():
            raise ValueError("relu_mask size")
    check(_lib.load().ssd_maxpool_bwd(dy.data_ptr(), argmax.data_ptr(), dx.data_ptr(), _ptr(relu_mask), int(accumulate),
                                      n, h, w, c, k, stride, pad, ho, wo, _stream()), "maxpool_bwd")
    return dx


def l2norm_fwd(x: torch.Tensor, gamma: torch.Tensor) -> torch.Tensor:
    _req(x, "x", x.dtype if x.dtype == torch.bfloat16 else torch.float32); _req(gamma, "gamma")
    c = x.shape[-1]
    if gamma.numel() != c:
        raise ValueError("gamma length")
    if x.dtype == torch.bfloat16:
        y = torch.empty_like(x)
        check(_lib.load().ssd_l2norm_fwd_bf16(x.data_ptr(), gamma.data_ptr(), y.data_ptr(), x.numel() // c, c, _stream()), "l2norm_fwd_bf16")
        return y
    y = torch.empty_like(x)
    check(_lib.load().ssd_l2norm_fwd(x.data_ptr(), gamma.data_ptr(), y.data_ptr(), x.numel() // c, c, _stream()), "l2norm_fwd")
    return y


def l2norm_bwd(x: torch.Tensor, gamma: torch.Tensor, dy: torch.Tensor, dx: Optional[torch.Tensor] = None,
               dg_out: Optional[torch.Tensor] = None):
    c = x.shape[-1]
    m = x.numel() // c
    if x.dtype == torch.bfloat16:
        _req(x, "x", torch.bfloat16); _req(gamma, "gamma"); _req(dy, "dy", torch.bfloat16)
        if dy.numel() != x.numel():
            raise ValueError("dy size")
        lib = _lib.load()
        ws = workspace(lib.ssd_l2norm_bwd_bf16_workspace(m, c), x.device)
        dx = torch.empty_like(x) if dx is None else _req(dx, "dx", torch.bfloat16)
        dg = _grad_dst(dg_out, (c,), x.device, "dg_out")
        check(lib.ssd_l2norm_bwd_bf16(x.data_ptr(), gamma.data_ptr(), dy.data_ptr(), dx.data_ptr(), dg.data_ptr(), m, c, ws.data_ptr(),
                                      ws.numel(), _stream()), "l2norm_bwd_bf16")
        return dx, dg
    _req(x, "x"); _req(gamma, "gamma"); _req(dy, "dy")
    if dy.numel() != x.numel():
        raise ValueError("dy size")
    lib = _lib.load()
    ws = workspace(lib.ssd_l2norm_bwd_workspace(m, c), x.device)
    dx = torch.empty_like(x) if dx is None else _req(dx, "dx")
    dg = _grad_dst(dg_out, (c,), x.device, "dg_out")
    check(lib.ssd_l2norm_bwd(x.data_ptr(), gamma.data_ptr(), dy.data_ptr(), dx.data_ptr(), dg.data_ptr(), m, c,
                             ws.data_ptr(), ws.numel(), _stream()), "l2norm_bwd")
    return dx, dg


# ---- heads ---------------------------------------------------------------------------------
def heads_scatter(packed: torch.Tensor, ld: int, loc: torch.Tensor, conf: torch.Tensor, n: int, hw: int, a: int,
                  prior_off: int) -> None:
    _req(packed, "packed"); _req(loc, "loc"); _req(conf, "conf")
    p, ncls = conf.shape[1], conf.shape[2]
    if packed.numel() != n * hw * ld or tuple(loc.shape) != (n, p, 4) or conf.shape[0] != n:
        raise ValueError("heads_scatter shapes")
    check(_lib.load().ssd_heads_scatter(packed.data_ptr(), ld, loc.data_ptr(), conf.data_ptr(), n, hw, a, prior_off, p, ncls,
                                        _stream()), "heads_scatter")


def heads_gather(dloc: torch.Tensor, dconf: torch.Tensor, ld: int, n: int, hw: int, a: int, prior_off: int) -> torch.Tensor:
    _req(dloc, "dloc"); _req(dconf, "dconf")
    p, ncls = dconf.shape[1], dconf.shape[2]
    if tuple(dloc.shape) != (n, p, 4) or dconf.shape[0] != n:
        raise ValueError("heads_gather shapes")
    packed = torch.empty((n * hw, ld), device=dloc.device, dtype=torch.float32)
    check(_lib.load().ssd_heads_gather(dloc.data_ptr(), dconf.data_ptr(), packed.data_ptr(), ld, n, hw, a, prior_off, p, ncls,
                                       _stream()), "heads_gather")
    return packed


# ---- loss / decode ----------------------------------------------------------------------------
def multibox_loss(loc, conf, gt_boxes, gt_classes, img_start, priors_cxcywh, priors_xyxy, iou_threshold=0.5,
                  neg_pos_ratio=3, norm_mode=0, want_grads=True):
    """-> dict(losses (3,), obj (bs,P) i32, cls (bs,P) i32, dloc, dconf)."""
    _req(loc, "loc"); _req(conf, "conf"); _req(gt_boxes, "gt_boxes"); _req(gt_classes, "gt_classes")
    _req(img_start, "img_start", torch.int32); _req(priors_cxcywh, "priors"); _req(priors_xyxy, "priors_xyxy")
    bs, p, ncls = conf.shape
    n_gt = gt_boxes.shape[0]
    if tuple(loc.shape) != (bs, p, 4) or tuple(gt_boxes.shape) != (n_gt, 4) or gt_classes.numel() != n_gt or \
            img_start.numel() != bs + 1 or tuple(priors_cxcywh.shape) != (p, 4) or tuple(priors_xyxy.shape) != (p, 4):
        raise ValueError("multibox_loss shapes")
    lib = _lib.load()
    ws = workspace(lib.ssd_multibox_loss_workspace(bs, p, n_gt), loc.device, "loss")
    dev = loc.device
    losses = torch.empty((3,), device=dev, dtype=torch.float32)
    obj = torch.empty((bs, p), device=dev, dtype=torch.int32)
    cls = torch.empty((bs, p), device=dev, dtype=torch.int32)
    dloc = torch.empty_like(loc) if want_grads else None
    dconf = torch.empty_like(conf) if want_grads else None
    check(lib.ssd_multibox_loss(loc.data_ptr(), conf.data_ptr(), gt_boxes.data_ptr(), gt_classes.data_ptr(), img_start.data_ptr(),
                                bs, n_gt, priors_cxcywh.data_ptr(), priors_xyxy.data_ptr(), p, ncls, float(iou_threshold),
                                int(neg_pos_ratio), int(norm_mode), losses.data_ptr(), obj.data_ptr(), cls.data_ptr(),
                                _ptr(dloc), _ptr(dconf), ws.data_ptr(), ws.numel(), _stream()), "multibox_loss")
    return dict(losses=losses, obj=obj, cls=cls, dloc=dloc, dconf=dconf)


def decode_nms(l_, c_, priors_cxcywh, img_w, img_h, top_k=200, min_score=0.2, iou_threshold=0.45):
    """-> (boxes (top_k,4), classes (top_k,) i64, probs (top_k,), prior_ids (top_k,) i32, count (1,) i32), all device."""
    _req(l_, "l_"); _req(c_, "c_"); _req(priors_cxcywh, "priors")
    p, ncls = c_.shape
    if tuple(l_.shape) != (p, 4) or tuple(priors_cxcywh.shape) != (p, 4):
        raise ValueError("decode_nms shapes")
    lib = _lib.load()
    ws = workspace(lib.ssd_decode_nms_workspace(p, ncls), l_.device, "nms")
    dev = l_.device
    boxes = torch.zeros((top_k, 4), device=dev, dtype=torch.float32)
    classes = torch.zeros((top_k,), device=dev, dtype=torch.int64)
    probs = torch.zeros((top_k,), device=dev, dtype=torch.float32)
    ids = torch.zeros((top_k,), device=dev, dtype=torch.int32)
    count = torch.zeros((1,), device=dev, dtype=torch.int32)
    check(lib.ssd_decode_nms(l_.data_ptr(), c_.data_ptr(), priors_cxcywh.data_ptr(), p, ncls, float(min_score),
                             float(iou_threshold), int(top_k), float(img_w), float(img_h), boxes.data_ptr(), classes.data_ptr(),
                             probs.data_ptr(), ids.data_ptr(), count.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "decode_nms")
    return boxes, classes, probs, ids, count


def decode_nms_batch(l, c, priors_cxcywh, img_wh, top_k=200, min_score=0.2, iou_threshold=0.45):
    """l (B,P,4), c (B,P,C), img_wh (B,2) device floats -> (boxes (B,top_k,4), classes (B,top_k) i64, probs (B,top_k),
    prior_ids (B,top_k) i32, count (B,) i32), all on the device, one launch set, no host sync."""
    _req(l, "l"); _req(c, "c"); _req(priors_cxcywh, "priors"); _req(img_wh, "img_wh")
    b, p, ncls = c.shape
    if tuple(l.shape) != (b, p, 4) or tuple(priors_cxcywh.shape) != (p, 4) or tuple(img_wh.shape) != (b, 2):
        raise ValueError("decode_nms_batch shapes")
    lib = _lib.load()
    ws = workspace(lib.ssd_decode_nms_batch_workspace(b, p, ncls), l.device, "nms")
    dev = l.device
    boxes = torch.zeros((b, top_k, 4), device=dev, dtype=torch.float32)
    classes = torch.zeros((b, top_k), device=dev, dtype=torch.int64)
    probs = torch.zeros((b, top_k), device=dev, dtype=torch.float32)
    ids = torch.zeros((b, top_k), device=dev, dtype=torch.int32)
    count = torch.zeros((b,), device=dev, dtype=torch.int32)
    check(lib.ssd_decode_nms_batch(l.data_ptr(), c.data_ptr(), priors_cxcywh.data_ptr(), img_wh.data_ptr(), b, p, ncls,
                                   float(min_score), float(iou_threshold), int(top_k), boxes.data_ptr(), classes.data_ptr(),
                                   probs.data_ptr(), ids.data_ptr(), count.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
          "decode_nms_batch")
    return boxes, classes, probs, ids, count


def sgd_momentum_(param, grad, buf, lr, momentum, weight_decay, grad_scale=None, first_step=False):
    _req(param, "param"); _req(grad, "grad"); _req(buf, "buf")
    if grad.numel() != param.numel() or buf.numel() != param.numel():
        raise ValueError("sgd sizes")
    if grad_scale is not None:
        _req(grad_scale, "grad_scale")
    check(_lib.load().ssd_sgd_momentum(param.data_ptr(), grad.data_ptr(), buf.data_ptr(), param.numel(), float(lr), float(momentum),
                                       float(weight_decay), _ptr(grad_scale), int(first_step), _stream()), "sgd_momentum")


def map_eval(det_boxes, det_classes, det_scores, det_start, gt_boxes, gt_classes, gt_start, recall_levels, n_classes=20):
    """Concatenated detections / ground truth (see include/ssd_gfx950.h ssd_map_eval) -> (table (n_classes, n_levels)
    float64, tp (D,) uint8, counts (2, n_classes) int32).  recall_levels: host sequence of floats."""
    import numpy as np
    dev = det_start.device
    D, G, B = int(det_boxes.shape[0]), int(gt_boxes.shape[0]), int(det_start.numel()) - 1
    _req(det_boxes, "det_boxes"); _req(det_scores, "det_scores"); _req(gt_boxes, "gt_boxes")
    _req(det_classes, "det_classes", torch.int32); _req(gt_classes, "gt_classes", torch.int32)
    _req(det_start, "det_start", torch.int32); _req(gt_start, "gt_start", torch.int32)
    if det_classes.numel() != D or det_scores.numel() != D or gt_classes.numel() != G or gt_start.numel() != B + 1 or B < 1:
        raise ValueError("map_eval: inconsistent array lengths")
    lv = np.ascontiguousarray(np.asarray(recall_levels, np.float64))
    tp = torch.zeros(max(D, 1), device=dev, dtype=torch.uint8)
    table = torch.empty((n_classes, lv.size), device=dev, dtype=torch.float64)
    counts = torch.empty((2, n_classes), device=dev, dtype=torch.int32)
    lib = _lib.load()
    ws = workspace(lib.ssd_map_eval_workspace(D, G), dev, "map")
    check(lib.ssd_map_eval(_ptr(det_boxes), _ptr(det_classes), _ptr(det_scores), det_start.data_ptr(), D, _ptr(gt_boxes),
                           _ptr(gt_classes), gt_start.data_ptr(), G, B, n_classes, lv.ctypes.data, int(lv.size), tp.data_ptr(),
                           table.data_ptr(), counts.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "map_eval")
    return table, tp[:D], counts


def preprocess_u8(arena: torch.Tensor, descs, out_hw=(300, 300), mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225),
                  filler=(123, 116, 103)) -> torch.Tensor:
    """arena: uint8 device tensor holding the HWC RGB images; descs: ctypes array of _lib.ImageDesc (host).
    -> (B,3,out_h,out_w) float32: expand / crop / flip geometry, Pillow-exact bilinear resize, /255, (x-mean)/std."""
    import numpy as np
    _req(arena, "arena", torch.uint8)
    B = len(descs)
    oh, ow = out_hw
    for d in descs:
        if d.src_offset < 0 or d.src_offset + d.src_h * d.src_w * 3 > arena.numel():
            raise ValueError("image descriptor points outside the arena")
    lib = _lib.load()
    nbytes = lib.ssd_preprocess_workspace(C.byref(descs), B, oh, ow)
    if nbytes == 0:
        raise ValueError("preprocess_u8: bad geometry (window outside the canvas, empty image, or a down-scale factor above 31)")
    ws = workspace(nbytes, arena.device, "pre")
    raw = np.frombuffer(bytes(descs), dtype=np.uint8)          # bytes() copies: safe to hand to torch
    descs_dev = torch.from_numpy(raw.copy()).to(arena.device, non_blocking=False)
    out = torch.empty((B, 3, oh, ow), device=arena.device, dtype=torch.float32)
    m = (C.c_float * 3)(*mean); s_ = (C.c_float * 3)(*std); f = (C.c_uint8 * 3)(*filler)
    check(lib.ssd_preprocess_u8(arena.data_ptr(), descs_dev.data_ptr(), C.byref(descs), B, oh, ow, C.addressof(m), C.addressof(s_),
                                C.addressof(f), out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "preprocess_u8")
    return out


def clock_probe(device) -> torch.Tensor:
    """Enqueue a clock probe on the current stream; returns the (16, 2) int64 tensor it fills (counter, 100 MHz ticks per XCC)."""
    out = torch.zeros((16, 2), device=device, dtype=torch.int64)
    check(_lib.load().ssd_clock_probe(out.data_ptr(), _stream()), "clock_probe")
    return out


def shader_mhz(probe_a: torch.Tensor, probe_b: torch.Tensor) -> float:
    """Average shader clock between two probes (mean over the XCCs both probes landed on)."""
    a, b = probe_a.cpu(), probe_b.cpu()
    ok = (a[:, 1] > 0) & (b[:, 1] > a[:, 1])
    if not bool(ok.any()):
        return float("nan")
    return float((100.0 * (b[ok, 0] - a[ok, 0]).double() / (b[ok, 1] - a[ok, 1]).double()).mean())


def photometric_u8(arena: torch.Tensor, descs, photo) -> None:
    """In place on the arena's source images: photo = ctypes array of _lib.PhotoDesc (one per image, same order as descs)."""
    import numpy as np
    _req(arena, "arena", torch.uint8)
    B = len(descs)
    if len(photo) != B:
        raise ValueError("one photo descriptor per image")
    for d in descs:
        if d.src_offset < 0 or d.src_offset + d.src_h * d.src_w * 3 > arena.numel():
            raise ValueError("image descriptor points outside the arena")
    lib = _lib.load()
    ws = workspace(lib.ssd_photometric_workspace(B), arena.device, "photo")
    d_dev = torch.from_numpy(np.frombuffer(bytes(descs), dtype=np.uint8).copy()).to(arena.device)
    p_dev = torch.from_numpy(np.frombuffer(bytes(photo), dtype=np.uint8).copy()).to(arena.device)
    check(lib.ssd_photometric_u8(arena.data_ptr(), d_dev.data_ptr(), C.byref(descs), p_dev.data_ptr(), C.byref(photo), B, ws.data_ptr(),
                                 ws.numel(), _stream()), "photometric_u8")


# ---- Winograd F(mo x mo, 3x3), mo = 2 or 4 --------------------------------------------------------
def wino_x3(mo: int, K: int) -> bool:
    """True if the F(4x4) plane GEMMs of reduction length K multiply three bf16 limbs per f32 operand (csrc/gemm_x3.hip); the layer's
    transformed filter is then a bf16 limb tensor (`wino_filter_alloc`)."""
    return bool(_lib.load().ssd_wino_uses_x3(mo, K))


def wino_filter_alloc(mo: int, rows: int, K: int, device) -> torch.Tensor:
    """Destination of a transformed filter with `rows` output rows and reduction length K: (P, rows, K) f32, or -- where `wino_x3` --
    the limb planes (P, K/16, 3, pad128(rows), 16) bf16, zero-filled (the padding rows are never written)."""
    P = (mo + 2) ** 2
    if wino_x3(mo, K):
        return torch.zeros((P, K // 16, 3, (rows + 127) // 128 * 128, 16), device=device, dtype=torch.bfloat16)
    return torch.empty((P, rows, K), device=device, dtype=torch.float32)


def _wino_filter(u: torch.Tensor, name: str, rows: int, K: Optional[int]) -> int:
    """Check a transformed filter against its geometry and the library's current GEMM form; returns K."""
    mo = _wino_mo(u)
    if u.dtype == torch.bfloat16:
        _req(u, name, torch.bfloat16)
        k = u.shape[1] * 16
        ok = u.dim() == 5 and tuple(u.shape[2:]) == (3, (rows + 127) // 128 * 128, 16) and (K is None or k == K)
    else:
        _req(u, name)
        k = u.shape[2] if u.dim() == 3 else -1
        ok = u.dim() == 3 and u.shape[1] == rows and (K is None or k == K)
    if not ok:
        raise ValueError(f"{name}: shape does not match the geometry")
    if (u.dtype == torch.bfloat16) != wino_x3(mo, k):
        raise ValueError(f"{name}: transformed under another ssd_tune_set_wino_x3 setting than the one in force")
    return k


def wino_weights(w_oihw: torch.Tensor, co_pad: Optional[int] = None, want_bwd: bool = True, mo: int = 2):
    """(Co,Ci,3,3) -> U_fwd (P,Co,Ci) and, if asked, U_bwd (P,Ci,co_pad), P = (mo+2)^2, for conv2d_fwd_wino / conv2d_dgrad_wino
    (either may be a limb tensor instead: `wino_filter_alloc`)."""
    _req(w_oihw, "weight")
    co, ci, r, s = w_oihw.shape
    if (r, s) != (3, 3) or mo not in (2, 4):
        raise ValueError("Winograd needs 3x3 filters and mo in (2, 4)")
    co_pad = pad32(co) if co_pad is None else co_pad
    uf = wino_filter_alloc(mo, co, ci, w_oihw.device)
    ub = wino_filter_alloc(mo, ci, co_pad, w_oihw.device) if want_bwd else None
    check(_lib.load().ssd_wino_weights(w_oihw.data_ptr(), uf.data_ptr(), _ptr(ub), co, ci, co_pad, mo, _stream()), "wino_weights")
    return uf, ub


def _wino_mo(u: torch.Tensor) -> int:
    if u.shape[0] not in (16, 36):
        raise ValueError("transformed filter must have 16 or 36 planes")
    return 2 if u.shape[0] == 16 else 4


# ---- 1x1 / stride-1 convolutions on the three-limb GEMM kernels (csrc/gemm_x3.hip; fc7, seq8.0) ---------------------------------
def x3_filter_alloc(rows: int, K: int, device) -> torch.Tensor:
    """Limb planes of a [rows][K] filter matrix: (K/16, 3, pad128(rows), 16) bf16, zero-filled (padding rows are never written)."""
    return torch.zeros((K // 16, 3, (rows + 127) // 128 * 128, 16), device=device, dtype=torch.bfloat16)


def conv1x1_weights_x3(w_oihw: torch.Tensor, co_pad: int):
    """(Co,Ci,1,1) -> (limbs of w [Co][Ci], limbs of w^T [Ci][co_pad]) for conv1x1_fwd_x3 / conv1x1_dgrad_x3 (per-layer form of the kind-4
    weight job)."""
    _req(w_oihw, "weight")
    co, ci = int(w_oihw.shape[0]), int(w_oihw.shape[1])
    if tuple(w_oihw.shape[2:]) != (1, 1) or ci % 32 != 0 or co_pad % 32 != 0 or co_pad < co:
        raise ValueError("conv1x1_weights_x3: a 1x1 filter with Ci and co_pad multiples of 32")
    lib = _lib.load()
    wf = x3_filter_alloc(co, ci, w_oihw.device)
    check(lib.ssd_gemm_x3_split_weights(w_oihw.data_ptr(), wf.data_ptr(), co, ci, 1, _stream()), "conv1x1_weights_x3")
    wt = torch.zeros((ci, co_pad), device=w_oihw.device, dtype=torch.float32)
    wt[:, :co] = w_oihw.reshape(co, ci).t()
    wb = x3_filter_alloc(ci, co_pad, w_oihw.device)
    check(lib.ssd_gemm_x3_split_weights(wt.data_ptr(), wb.data_ptr(), ci, co_pad, 1, _stream()), "conv1x1_weights_x3")
    return wf, wb


def _x3_filter(w3: torch.Tensor, name: str, rows: int, K: int) -> None:
    _req(w3, name, torch.bfloat16)
    if tuple(w3.shape) != (K // 16, 3, (rows + 127) // 128 * 128, 16):
        raise ValueError(f"{name}: limb planes do not match the geometry")


def conv1x1_fwd_x3(x: torch.Tensor, w3: torch.Tensor, bias: Optional[torch.Tensor], g: ConvGeom, relu: bool) -> torch.Tensor:
    _req(x, "x")
    if tuple(x.shape) != (g.N, g.H, g.W, g.Ci):
        raise ValueError("conv1x1_fwd_x3: x shape does not match geometry")
    _x3_filter(w3, "w3", g.Co, g.Ci)
    if bias is not None:
        _req(bias, "bias")
    y = torch.empty((g.N, g.H, g.W, g.Co), device=x.device, dtype=torch.float32)
    check(_lib.load().ssd_conv1x1_fwd_x3(x.data_ptr(), w3.data_ptr(), _ptr(bias), y.data_ptr(), g.Co, C.byref(g), int(relu), _stream()), "conv1x1_fwd_x3")
    return y


def conv1x1_dgrad_x3(dy: torch.Tensor, w3t: torch.Tensor, g: ConvGeom, dx: Optional[torch.Tensor] = None, relu_mask: Optional[torch.Tensor] = None,
                     accumulate: bool = False) -> torch.Tensor:
    _req(dy, "dy")
    ldy = dy.shape[-1]
    if dy.numel() != g.N * g.H * g.W * ldy:
        raise ValueError("conv1x1_dgrad_x3: dy size does not match geometry")
    _x3_filter(w3t, "w3t", g.Ci, ldy)
    if dx is None:
        if accumulate:
            raise ValueError("accumulate needs an existing dx")
        dx = torch.empty((g.N, g.H, g.W, g.Ci), device=dy.device, dtype=torch.float32)
    _req(dx, "dx")
    if relu_mask is not None:
        _req(relu_mask, "relu_mask")
    check(_lib.load().ssd_conv1x1_dgrad_x3(dy.data_ptr(), ldy, w3t.data_ptr(), dx.data_ptr(), _ptr(relu_mask), int(accumulate), C.byref(g), _stream()),
          "conv1x1_dgrad_x3")
    return dx


def conv1x1_wgrad_x3(x: torch.Tensor, dy: torch.Tensor, g: ConvGeom, ldy: int, want_bias: bool = True, dw_out: Optional[torch.Tensor] = None,
                     db_out: Optional[torch.Tensor] = None):
    _req(x, "x"); _req(dy, "dy")
    if tuple(x.shape) != (g.N, g.H, g.W, g.Ci) or dy.numel() != g.N * g.H * g.W * ldy or ldy < g.Co:
        raise ValueError("conv1x1_wgrad_x3: shapes do not match geometry")
    lib = _lib.load()
    nbytes = lib.ssd_conv1x1_wgrad_x3_workspace(C.byref(g), ldy)
    if nbytes == 0:
        raise ValueError("conv1x1_wgrad_x3: not a 1x1 / stride 1 geometry")
    ws = workspace(nbytes, x.device)
    dw = _grad_dst(dw_out, (g.Co, g.Ci, 1, 1), x.device, "dw_out")
    db = _grad_dst(db_out, (g.Co,), x.device, "db_out") if want_bias else None
    check(lib.ssd_conv1x1_wgrad_x3(x.data_ptr(), dy.data_ptr(), ldy, dw.data_ptr(), _ptr(db), C.byref(g), ws.data_ptr(), ws.numel(), _stream()),
          "conv1x1_wgrad_x3")
    return dw, db


def wino_planes_shape(g: ConvGeom):
    return (36, wino_tiles(g), g.Ci)


def wino_planes(g: ConvGeom, device) -> torch.Tensor:
    """Buffer for the F(4x4) transformed input of a convolution, [36][tiles][Ci] (include/ssd_gfx950.h ssd_conv3x3_wino_fwd_keep)."""
    return torch.empty((36, wino_tiles(g), g.Ci), device=device, dtype=torch.float32)


def wino_uses_full(g: ConvGeom, direction: int) -> bool:
    """True if the library runs this geometry's F(4x4) forward (0) / data gradient from dy (1) as one kernel from the activation."""
    return bool(_lib.load().ssd_conv3x3_wino_uses_full(C.byref(g), direction))


def wino_relu_bits(g: ConvGeom, device) -> torch.Tensor:
    """Buffer for the ReLU mask of a convolution's INPUT as bits: one int64 word per (tile, channel quad) (include/ssd_gfx950.h
    ssd_conv3x3_wino_fwd_keep_bits)."""
    return torch.empty((wino_tiles(g), g.Ci // 4), device=device, dtype=torch.int64)


def conv2d_fwd_wino(x: torch.Tensor, u_fwd: torch.Tensor, bias: Optional[torch.Tensor], g: ConvGeom, relu: bool,
                    ld: Optional[int] = None, keep_planes: bool = False, want_bits: bool = False):
    """keep_planes: also return the transformed input (F(4x4) only) for `conv2d_wgrad_wino(..., planes=)` -> (y, planes);
    want_bits (with keep_planes): also the bit mask x > 0 for `conv2d_dgrad_wino(..., bits=)` -> (y, planes, bits)."""
    _req(x, "x")
    mo = _wino_mo(u_fwd)
    _wino_filter(u_fwd, "u_fwd", g.Co, g.Ci)
    if tuple(x.shape) != (g.N, g.H, g.W, g.Ci):
        raise ValueError("conv2d_fwd_wino: shapes do not match the geometry")
    if bias is not None:
        _req(bias, "bias")
    ld = g.Co if ld is None else ld
    out = torch.empty((g.N, g.H, g.W, ld), device=x.device, dtype=torch.float32) if ld == g.Co else \
        torch.zeros((g.N, g.H, g.W, ld), device=x.device, dtype=torch.float32)
    lib = _lib.load()
    nbytes = lib.ssd_conv3x3_wino_workspace(C.byref(g), 0, mo)
    if nbytes == 0:
        raise ValueError("conv2d_fwd_wino: not a 3x3 / stride 1 / pad 1 geometry")
    ws = workspace(nbytes, x.device, "wino")
    if keep_planes:
        if mo != 4:
            raise ValueError("conv2d_fwd_wino: keep_planes needs F(4x4,3x3) filters")
        planes = wino_planes(g, x.device)
        bits = wino_relu_bits(g, x.device) if want_bits else None
        check(lib.ssd_conv3x3_wino_fwd_keep_bits(x.data_ptr(), u_fwd.data_ptr(), _ptr(bias), out.data_ptr(), ld, C.byref(g), int(relu),
                                                 planes.data_ptr(), _ptr(bits), ws.data_ptr(), ws.numel(), _stream()), "conv2d_fwd_wino")
        return (out, planes, bits) if want_bits else (out, planes)
    if want_bits:
        raise ValueError("conv2d_fwd_wino: want_bits needs keep_planes")
    check(lib.ssd_conv3x3_wino_fwd(x.data_ptr(), u_fwd.data_ptr(), _ptr(bias), out.data_ptr(), ld, C.byref(g), int(relu), mo, ws.data_ptr(),
                                   ws.numel(), _stream()), "conv2d_fwd_wino")
    return out


def conv2d_fwd_wino_pool(x: torch.Tensor, u_fwd: torch.Tensor, bias: Optional[torch.Tensor], g: ConvGeom, ceil_mode: bool,
                         want_argmax: bool = True, keep_planes: bool = False, want_bits: bool = False):
    """conv3x3 -> ReLU -> max pool 2x2 / stride 2 in one pass (F(4x4,3x3) filters): (pooled y, argmax or None), the pair
    `conv2d_fwd_wino(relu=True)` + `maxpool_fwd(2, 2, 0)` returns, without the full-resolution activation in between."""
    _req(x, "x")
    if _wino_mo(u_fwd) != 4:
        raise ValueError("conv2d_fwd_wino_pool: needs F(4x4,3x3) filters")
    _wino_filter(u_fwd, "u_fwd", g.Co, g.Ci)
    if tuple(x.shape) != (g.N, g.H, g.W, g.Ci) or g.Co % 4 != 0:
        raise ValueError("conv2d_fwd_wino_pool: shapes do not match the geometry")
    if bias is not None:
        _req(bias, "bias")
    ho, wo = pool_out(g.H, 2, 2, 0, ceil_mode), pool_out(g.W, 2, 2, 0, ceil_mode)
    y = torch.empty((g.N, ho, wo, g.Co), device=x.device, dtype=torch.float32)
    am = torch.empty((g.N, ho, wo, g.Co), device=x.device, dtype=torch.uint8) if want_argmax else None
    lib = _lib.load()
    nbytes = lib.ssd_conv3x3_wino_workspace(C.byref(g), 0, 4)
    if nbytes == 0:
        raise ValueError("conv2d_fwd_wino_pool: not a 3x3 / stride 1 / pad 1 geometry")
    ws = workspace(nbytes, x.device, "wino")
    planes = wino_planes(g, x.device) if keep_planes else None
    bits = wino_relu_bits(g, x.device) if want_bits else None
    check(lib.ssd_conv3x3_wino_fwd_pool_bits(x.data_ptr(), u_fwd.data_ptr(), _ptr(bias), y.data_ptr(), _ptr(am), C.byref(g), int(ceil_mode),
                                             _ptr(planes), _ptr(bits), ws.data_ptr(), ws.numel(), _stream()), "conv2d_fwd_wino_pool")
    if want_bits:
        return y, am, planes, bits
    return (y, am, planes) if keep_planes else (y, am)


def conv2d_fwd_wino_from_planes(planes: torch.Tensor, u_fwd: torch.Tensor, bias: Optional[torch.Tensor], g: ConvGeom, relu: bool = True,
                                pool_ceil: Optional[bool] = None, want_argmax: bool = True):
    """Forward of an F(4x4) layer whose input planes exist already (`conv1_first_wino_fwd`): plane GEMMs + output transform.
    pool_ceil None -> y (N,H,W,Co); True / False -> (pooled y, argmax or None) of the fused conv -> ReLU -> 2x2 / stride-2 pool."""
    _req(planes, "planes")
    if _wino_mo(u_fwd) != 4 or tuple(planes.shape) != tuple(wino_planes_shape(g)) or g.Co % 4 != 0:
        raise ValueError("conv2d_fwd_wino_from_planes: planes / filters do not match the geometry")
    _wino_filter(u_fwd, "u_fwd", g.Co, g.Ci)
    if bias is not None:
        _req(bias, "bias")
    lib = _lib.load()
    nbytes = lib.ssd_conv3x3_wino_workspace(C.byref(g), 0, 4)
    if nbytes == 0:
        raise ValueError("conv2d_fwd_wino_from_planes: not a 3x3 / stride 1 / pad 1 geometry")
    ws = workspace(nbytes, planes.device, "wino")
    if pool_ceil is None:
        y = torch.empty((g.N, g.H, g.W, g.Co), device=planes.device, dtype=torch.float32)
        check(lib.ssd_conv3x3_wino_fwd_from_planes(planes.data_ptr(), u_fwd.data_ptr(), _ptr(bias), y.data_ptr(), g.Co, None, None, C.byref(g), int(relu),
                                                   0, ws.data_ptr(), ws.numel(), _stream()), "conv2d_fwd_wino_from_planes")
        return y
    ho, wo = pool_out(g.H, 2, 2, 0, pool_ceil), pool_out(g.W, 2, 2, 0, pool_ceil)
    y = torch.empty((g.N, ho, wo, g.Co), device=planes.device, dtype=torch.float32)
    am = torch.empty((g.N, ho, wo, g.Co), device=planes.device, dtype=torch.uint8) if want_argmax else None
    check(lib.ssd_conv3x3_wino_fwd_from_planes(planes.data_ptr(), u_fwd.data_ptr(), _ptr(bias), None, 0, y.data_ptr(), _ptr(am), C.byref(g), 1,
                                               int(pool_ceil), ws.data_ptr(), ws.numel(), _stream()), "conv2d_fwd_wino_from_planes")
    return y, am


def conv2d_dgrad_wino(dy: Optional[torch.Tensor], u_bwd: torch.Tensor, g: ConvGeom, dx: Optional[torch.Tensor] = None,
                      relu_mask: Optional[torch.Tensor] = None, accumulate: bool = False,
                      planes: Optional[torch.Tensor] = None, bits: Optional[torch.Tensor] = None) -> torch.Tensor:
    """planes: B^T dy B as `conv2d_wgrad_wino(..., dgrad_planes=True)` left it ((36, tiles, Co_pad), F(4x4)); dy is then not read.
    bits (with planes): the ReLU mask as the forward's input transform left it (`conv2d_fwd_wino(..., want_bits=True)`), applied
    instead of relu_mask."""
    mo = _wino_mo(u_bwd)
    co_pad = _wino_filter(u_bwd, "u_bwd", g.Ci, None)
    dev = u_bwd.device
    if planes is not None:
        _req(planes, "planes")
        if mo != 4 or tuple(planes.shape) != (36, wino_planes_shape(g)[1], co_pad):
            raise ValueError("conv2d_dgrad_wino: planes do not match the geometry")
    else:
        _req(dy, "dy")
        if dy.numel() != g.N * g.H * g.W * co_pad:
            raise ValueError("conv2d_dgrad_wino: shapes do not match the geometry")
    if dx is None:
        if accumulate:
            raise ValueError("accumulate needs an existing dx")
        dx = torch.empty((g.N, g.H, g.W, g.Ci), device=dev, dtype=torch.float32)
    _req(dx, "dx")
    if relu_mask is not None:
        _req(relu_mask, "relu_mask")
    lib = _lib.load()
    ws = workspace(lib.ssd_conv3x3_wino_workspace(C.byref(g), 1, mo), dev, "wino")
    if bits is not None:
        _req(bits, "bits", torch.int64)
        if mo != 4 or tuple(bits.shape) != (wino_planes_shape(g)[1], g.Ci // 4) or g.Ci % 4 != 0:
            raise ValueError("conv2d_dgrad_wino: bits need F(4x4) filters and one word per (tile, channel quad)")
        if planes is None:
            check(lib.ssd_conv3x3_wino_dgrad_bits(dy.data_ptr(), co_pad, u_bwd.data_ptr(), co_pad, dx.data_ptr(), bits.data_ptr(), int(accumulate),
                                                  C.byref(g), ws.data_ptr(), ws.numel(), _stream()), "conv2d_dgrad_wino")
            return dx
        check(lib.ssd_conv3x3_wino_dgrad_planes_bits(planes.data_ptr(), u_bwd.data_ptr(), co_pad, dx.data_ptr(), bits.data_ptr(), int(accumulate),
                                                     C.byref(g), ws.data_ptr(), ws.numel(), _stream()), "conv2d_dgrad_wino")
        return dx
    if planes is not None:
        check(lib.ssd_conv3x3_wino_dgrad_planes(planes.data_ptr(), u_bwd.data_ptr(), co_pad, dx.data_ptr(), _ptr(relu_mask), int(accumulate),
                                                C.byref(g), ws.data_ptr(), ws.numel(), _stream()), "conv2d_dgrad_wino")
        return dx
    check(lib.ssd_conv3x3_wino_dgrad(dy.data_ptr(), co_pad, u_bwd.data_ptr(), co_pad, dx.data_ptr(), _ptr(relu_mask), int(accumulate),
                                     C.byref(g), mo, ws.data_ptr(), ws.numel(), _stream()), "conv2d_dgrad_wino")
    return dx


def conv2d_wgrad_wino(x: Optional[torch.Tensor], dy: torch.Tensor, g: ConvGeom, ldy: int, want_bias: bool = True, mo: int = 2,
                      planes: Optional[torch.Tensor] = None, dgrad_planes: bool = False,
                      dw_out: Optional[torch.Tensor] = None, db_out: Optional[torch.Tensor] = None):
    """Winograd F(mo x mo, 3x3) weight gradient, mo = 2 or 4 -> (dw (Co,Ci,3,3) OIHW, dbias (Co,) or None).
    planes: the transformed input the forward kept (`conv2d_fwd_wino(..., keep_planes=True)`); x is then not read.
    dgrad_planes (needs planes): the pass over dy also forms this layer's dgrad input planes -> (dw, dbias, planes_dy (36, tiles, ldy))
    for `conv2d_dgrad_wino(None, ..., planes=planes_dy)`."""
    pooled = isinstance(dy, PooledGrad)
    if pooled and planes is None:
        raise ValueError("conv2d_wgrad_wino: a pooled gradient needs the kept forward planes")
    if not pooled:
        _req(dy, "dy")
    if planes is not None:
        _req(planes, "planes")
        if mo != 4 or tuple(planes.shape) != tuple(wino_planes_shape(g)):
            raise ValueError("conv2d_wgrad_wino: planes do not match the geometry")
    else:
        _req(x, "x")
        if tuple(x.shape) != (g.N, g.H, g.W, g.Ci):
            raise ValueError("conv2d_wgrad_wino: shapes do not match the geometry")
    if not pooled and dy.numel() != g.N * g.H * g.W * ldy:
        raise ValueError("conv2d_wgrad_wino: shapes do not match the geometry")
    lib = _lib.load()
    nbytes = lib.ssd_conv3x3_wino_wgrad_workspace(C.byref(g), ldy, mo)
    if nbytes == 0:
        raise ValueError("conv2d_wgrad_wino: not a 3x3 / stride 1 / pad 1 geometry")
    dev = planes.device if pooled else dy.device
    ws = workspace(nbytes, dev, "wino")
    dw = _grad_dst(dw_out, (g.Co, g.Ci, 3, 3), dev, "dw_out")
    db = _grad_dst(db_out, (g.Co,), dev, "db_out") if want_bias else None
    if dgrad_planes and planes is None:
        raise ValueError("conv2d_wgrad_wino: dgrad_planes needs the kept forward planes")
    if planes is not None:
        pd = torch.empty((36, planes.shape[1], ldy), device=dev, dtype=torch.float32) if dgrad_planes else None
        if pooled:
            check(lib.ssd_conv3x3_wino_wgrad_planes_pooled(planes.data_ptr(), *_pooled_args(dy, g, ldy), ldy, dw.data_ptr(), _ptr(db), C.byref(g),
                                                           _ptr(pd), ws.data_ptr(), ws.numel(), _stream()), "conv2d_wgrad_wino_pooled")
            return (dw, db, pd) if dgrad_planes else (dw, db)
        check(lib.ssd_conv3x3_wino_wgrad_planes(planes.data_ptr(), dy.data_ptr(), ldy, dw.data_ptr(), _ptr(db), C.byref(g), _ptr(pd),
                                                ws.data_ptr(), ws.numel(), _stream()), "conv2d_wgrad_wino")
        return (dw, db, pd) if dgrad_planes else (dw, db)
    check(lib.ssd_conv3x3_wino_wgrad(x.data_ptr(), dy.data_ptr(), ldy, dw.data_ptr(), _ptr(db), C.byref(g), mo, ws.data_ptr(), ws.numel(),
                                     _stream()), "conv2d_wgrad_wino")
    return dw, db


def wino_dy_transform(dy: torch.Tensor, g: ConvGeom, ldy: int, dgrad_planes: bool = True, want_bias: bool = True):
    """First half of the F(4x4) weight gradient on kept planes: one pass over dy -> (wgrad_planes (36, tiles, ldy),
    dgrad_planes (36, tiles, ldy) or None, bias_partial or None).  The data gradient only needs dgrad_planes."""
    pooled = isinstance(dy, PooledGrad)
    dev = dy.dpool.device if pooled else _req(dy, "dy").device
    tiles = wino_planes_shape(g)[1]
    if (not pooled and dy.numel() != g.N * g.H * g.W * ldy) or ldy % 4 != 0 or ldy < g.Co:
        raise ValueError("wino_dy_transform: shapes do not match the geometry")
    lib = _lib.load()
    Y = torch.empty((36, tiles, ldy), device=dev, dtype=torch.float32)
    Vd = torch.empty((36, tiles, ldy), device=dev, dtype=torch.float32) if dgrad_planes else None
    part = torch.empty((lib.ssd_wino4_bias_partial_floats(C.byref(g), ldy),), device=dev, dtype=torch.float32) if want_bias else None
    if pooled:
        check(lib.ssd_wino4_dy_transform_pooled(*_pooled_args(dy, g, ldy), ldy, C.byref(g), Y.data_ptr(), _ptr(Vd), _ptr(part), _stream()),
              "wino_dy_transform_pooled")
    else:
        check(lib.ssd_wino4_dy_transform(dy.data_ptr(), ldy, C.byref(g), Y.data_ptr(), _ptr(Vd), _ptr(part), _stream()), "wino_dy_transform")
    return Y, Vd, part


def wino_adj_weights(w_oihw: torch.Tensor, co_pad: int) -> torch.Tensor:
    """Filter planes of the adjoint-form data gradient (`wino_dgrad_adj_gemm`): the forward transform G g G^T laid out (36, Ci, co_pad)
    -- limb planes where `wino_x3(4, co_pad)`."""
    _req(w_oihw, "w")
    co, ci = int(w_oihw.shape[0]), int(w_oihw.shape[1])
    u = wino_filter_alloc(4, ci, co_pad, w_oihw.device)
    check(_lib.load().ssd_wino_weights_adj(w_oihw.data_ptr(), u.data_ptr(), co, ci, co_pad, _stream()), "wino_weights_adj")
    return u


def wino_dgrad_adj_gemm(y_planes: torch.Tensor, u_adj: torch.Tensor, g: ConvGeom, ldy: int) -> torch.Tensor:
    """The 36 plane GEMMs of the adjoint-form data gradient: (36, tiles, ldy) planes A dy A^T (`wino_dy_transform`'s first result, the
    weight gradient's operand) x the adjoint filter planes -> md (36, tiles, Ci)."""
    _req(y_planes, "y_planes")
    tiles = wino_planes_shape(g)[1]
    if tuple(y_planes.shape) != (36, tiles, ldy) or g.Ci % 4 != 0 or ldy % 32 != 0:
        raise ValueError("wino_dgrad_adj_gemm: planes do not match the geometry")
    _wino_filter(u_adj, "u_adj", g.Ci, ldy)
    md = torch.empty((36, tiles, g.Ci), device=y_planes.device, dtype=torch.float32)
    check(_lib.load().ssd_conv3x3_wino_dgrad_adj_gemm(y_planes.data_ptr(), ldy, u_adj.data_ptr(), md.data_ptr(), C.byref(g), _stream()),
          "wino_dgrad_adj_gemm")
    return md


def _adj_mask_args(g: ConvGeom, tiles: int, relu_mask, bits):
    if bits is not None:
        _req(bits, "bits", torch.int64)
        if tuple(bits.shape) != (tiles, g.Ci // 4):
            raise ValueError("adjoint output: one mask word per (tile, channel quad)")
    if relu_mask is not None:
        _req(relu_mask, "relu_mask")
        if relu_mask.numel() != g.N * g.H * g.W * g.Ci:
            raise ValueError("adjoint output: relu_mask must have dx's shape")


def wino_adj_output(md: torch.Tensor, g: ConvGeom, dx: Optional[torch.Tensor] = None, relu_mask: Optional[torch.Tensor] = None,
                    bits: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """dx (N,H,W,Ci) [+=] the overlap-added 6x6 patches B md B^T, then the ReLU mask (bit words of the forward's input transform, or an
    f32 tensor > 0)."""
    _req(md, "md")
    tiles = wino_planes_shape(g)[1]
    if tuple(md.shape) != (36, tiles, g.Ci):
        raise ValueError("wino_adj_output: planes do not match the geometry")
    _adj_mask_args(g, tiles, relu_mask, bits)
    if dx is None:
        if accumulate:
            raise ValueError("accumulate needs an existing dx")
        dx = torch.empty((g.N, g.H, g.W, g.Ci), device=md.device, dtype=torch.float32)
    _req(dx, "dx")
    if dx.numel() != g.N * g.H * g.W * g.Ci:
        raise ValueError("wino_adj_output: dx shape")
    check(_lib.load().ssd_wino4_adj_output(md.data_ptr(), dx.data_ptr(), _ptr(relu_mask) if bits is None else None, _ptr(bits), int(accumulate),
                                           C.byref(g), _stream()), "wino_adj_output")
    return dx


def wino_adj_output_to_planes(md: torch.Tensor, g: ConvGeom, g_below: ConvGeom, relu_mask: Optional[torch.Tensor] = None,
                              bits: Optional[torch.Tensor] = None, want_bias: bool = True):
    """The chained form: the masked dx block of layer `g` is the dy block of the layer below it -> (planes A dy A^T (36, tiles, g.Ci) of
    `g_below`, its bias partial sums or None), without the gradient tensor in between."""
    _req(md, "md")
    tiles = wino_planes_shape(g)[1]
    if tuple(md.shape) != (36, tiles, g.Ci) or g_below.Co != g.Ci or (g_below.N, g_below.H, g_below.W) != (g.N, g.H, g.W):
        raise ValueError("wino_adj_output_to_planes: the two layers do not chain")
    _adj_mask_args(g, tiles, relu_mask, bits)
    lib = _lib.load()
    Y = torch.empty((36, tiles, g.Ci), device=md.device, dtype=torch.float32)
    part = torch.empty((lib.ssd_wino4_bias_partial_floats(C.byref(g_below), g.Ci),), device=md.device, dtype=torch.float32) if want_bias else None
    check(lib.ssd_wino4_adj_output_to_planes(md.data_ptr(), _ptr(relu_mask) if bits is None else None, _ptr(bits), C.byref(g), C.byref(g_below),
                                             Y.data_ptr(), _ptr(part), _stream()), "wino_adj_output_to_planes")
    return Y, part


class AdjPlanesGrad:
    """A data gradient that exists only as the product planes `md` of the adjoint-form GEMMs of layer `g` (plus the ReLU mask of the
    tensor it is the gradient of): the layer below turns it into its own dy planes (`wino_adj_output_to_planes`) without the tensor
    being written; `materialize()` is the stand-alone output transform for any other reader."""

    def __init__(self, md: torch.Tensor, g: ConvGeom, relu_mask: Optional[torch.Tensor], bits: Optional[torch.Tensor]):
        self.md, self.g, self.relu_mask, self.bits = md, g, relu_mask, bits

    @property
    def shape(self):
        return (self.g.N, self.g.H, self.g.W, self.g.Ci)

    def materialize(self) -> torch.Tensor:
        return wino_adj_output(self.md, self.g, None, self.relu_mask, self.bits, False)


class PooledGrad:
    """A gradient that exists only as the gradient of the 2x2 / stride-2 max pool behind it: (dpool (N,Hp,Wp,C), argmax codes, pooled forward
    output = the ReLU gate).  `wino_dy_transform` / `conv2d_wgrad_wino` take it in place of dy; `materialize()` is the scatter to memory."""

    def __init__(self, dpool: torch.Tensor, argmax: torch.Tensor, y_pooled: torch.Tensor, in_shape):
        _req(dpool, "dpool"); _req(y_pooled, "y_pooled"); _req(argmax, "argmax", torch.uint8)
        if tuple(dpool.shape) != tuple(y_pooled.shape) or tuple(argmax.shape) != tuple(dpool.shape) or dpool.shape[3] != in_shape[3]:
            raise ValueError("PooledGrad: dpool, argmax and the pooled output must have one shape, and the channels of the pool's input")
        self.dpool, self.argmax, self.y_pooled, self.in_shape = dpool, argmax, y_pooled, tuple(in_shape)

    def materialize(self, dx: Optional[torch.Tensor] = None) -> torch.Tensor:
        return maxpool_bwd(self.dpool, self.argmax, self.in_shape, 2, 2, 0, dx, y_gate=self.y_pooled)


def _pooled_args(dy: "PooledGrad", g: ConvGeom, ldy: int):
    if dy.in_shape != (g.N, g.H, g.W, g.Co) or ldy != g.Co:
        raise ValueError("pooled gradient does not match the geometry")
    return dy.dpool.data_ptr(), dy.argmax.data_ptr(), dy.y_pooled.data_ptr(), dy.dpool.shape[1], dy.dpool.shape[2]


def wino_wgrad_gemm(wgrad_planes: torch.Tensor, x_planes: torch.Tensor, bias_partial: Optional[torch.Tensor], g: ConvGeom, ldy: int,
                    dw_out: Optional[torch.Tensor] = None, db_out: Optional[torch.Tensor] = None):
    """Second half: the TN GEMMs over the tiles, the inverse transform and the bias gradient -> (dw (Co,Ci,3,3), dbias or None).
    Runs on the CURRENT stream, which may differ from the one `wino_dy_transform` ran on (the caller orders them with an event)."""
    _req(wgrad_planes, "wgrad_planes"); _req(x_planes, "x_planes")
    tiles = wino_planes_shape(g)[1]
    if tuple(wgrad_planes.shape) != (36, tiles, ldy) or tuple(x_planes.shape) != (36, tiles, g.Ci):
        raise ValueError("wino_wgrad_gemm: planes do not match the geometry")
    lib = _lib.load()
    dev = x_planes.device
    ws = workspace(lib.ssd_wino4_wgrad_gemm_workspace(C.byref(g), ldy), dev, "wino_wgrad")
    dw = _grad_dst(dw_out, (g.Co, g.Ci, 3, 3), dev, "dw_out")
    db = _grad_dst(db_out, (g.Co,), dev, "db_out") if bias_partial is not None else None
    check(lib.ssd_wino4_wgrad_gemm(wgrad_planes.data_ptr(), x_planes.data_ptr(), ldy, _ptr(bias_partial), dw.data_ptr(), _ptr(db), C.byref(g),
                                   ws.data_ptr(), ws.numel(), _stream()), "wino_wgrad_gemm")
    return dw, db


class WeightTable:
    """Job table of `ssd_weights_prepare`: every filter transform / re-layout of a training step in one launch.  `jobs` is a list of
    dicts(kind, w0, w1 or None, co0, co, ci, taps, co_pad, out_fwd, out_bwd or None) holding tensors; the table keeps them alive and is
    valid while their storage does not move."""

    def __init__(self, jobs, device):
        import numpy as np
        lib = _lib.load()
        arr = (_lib.WeightJob * len(jobs))()
        starts, total = [], 0
        self.keep = []
        for a, j in zip(arr, jobs):
            for t in (j["w0"], j.get("w1"), j.get("out_fwd"), j.get("out_bwd")):
                if t is not None:
                    _req(t, "weight job tensor", t.dtype if (j["kind"] in (0, 3, 4) and t.dtype == torch.bfloat16) else torch.float32)
                    self.keep.append(t)
            a.w0 = j["w0"].data_ptr()
            a.w1 = j["w1"].data_ptr() if j.get("w1") is not None else j["w0"].data_ptr()
            a.out_fwd = _ptr(j.get("out_fwd"))
            a.out_bwd = _ptr(j.get("out_bwd"))
            a.co0, a.co, a.ci, a.taps, a.co_pad, a.kind = j["co0"], j["co"], j["ci"], j["taps"], j["co_pad"], j["kind"]
            a.pad1 = j.get("pad1", 0)
            if j["kind"] == 0:                 # Winograd filters: which outputs are limb tensors (wino_filter_alloc); bit 2: out_bwd in the adjoint form
                a.pad0 = sum(bit for bit, key in ((1, "out_fwd"), (2, "out_bwd")) if j.get(key) is not None and j[key].dtype == torch.bfloat16)
                if j.get("adj"):
                    a.pad0 |= 4
            nb = lib.ssd_weight_job_blocks(C.byref(a))
            if nb <= 0:
                raise ValueError("bad weight job")
            starts.append(total)
            total += nb
        self.njobs, self.total_blocks = len(jobs), total
        self.jobs_dev = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(device)
        self.starts_dev = torch.tensor(starts, dtype=torch.int32).to(device)

    def run(self) -> None:
        check(_lib.load().ssd_weights_prepare(self.jobs_dev.data_ptr(), self.starts_dev.data_ptr(), self.njobs, self.total_blocks, _stream()),
              "weights_prepare")
