"""GPU parity tests of the individual gfx950 kernels, through the C ABI
(objectdetection_ssd_amd.ops -> libssd_gfx950.so).  Floating-point kernels are
compared with plain torch-CPU f32 ops of the same mathematical op; tolerance:
|a-b| <= 1e-4 * max(1, max|ref|) (north_star: loc/conf/loss within 1e-4 fp32).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _close(got, ref, tol=1e-4, what=""):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(1.0, float(ref.abs().max()))
    err = float((got - ref).abs().max())
    assert err <= tol * scale, f"{what}: max err {err:.3e} > {tol * scale:.3e} (scale {scale:.3g})"


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


CONV_CASES = [
    # n, h, w, ci, co, k, stride, pad, dil      (what it exercises)
    (2, 19, 19, 64, 64, 3, 1, 1, 1),       # 64x64 tile, M=722 ragged
    (1, 38, 38, 128, 256, 3, 1, 1, 1),     # 64x64 tile
    (4, 64, 64, 64, 512, 3, 1, 1, 1),      # 128x128 tile (>=512 blocks)
    (2, 256, 256, 32, 64, 3, 1, 1, 1),     # 256x64 tile
    (1, 19, 19, 512, 1024, 3, 1, 4, 4),    # fc6: dilation 4
    (2, 19, 19, 1024, 256, 1, 1, 0, 1),    # 1x1
    (2, 19, 19, 256, 512, 3, 2, 1, 1),     # stride 2 pad 1 -> 10x10
    (3, 10, 10, 128, 256, 3, 2, 1, 1),     # stride 2 -> 5x5
    (2, 5, 5, 128, 256, 3, 1, 0, 1),       # pad 0 -> 3x3
    (2, 3, 3, 128, 256, 3, 1, 0, 1),       # -> 1x1
    (2, 38, 38, 512, 100, 3, 1, 1, 1),     # head: Co=100 (not a tile multiple)
    (2, 19, 19, 1024, 150, 3, 1, 1, 1),    # head: Co=150
]


def _conv_data(case, seed=0):
    n, h, w, ci, co, k, s, p, d = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5
    b = torch.randn(co, generator=g) * 0.1
    return x, wt, b


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("relu", [True, False])
def test_conv2d_fwd(case, relu):
    from objectdetection_ssd_amd import ops
    n, h, w, ci, co, k, s, p, d = case
    dev = _dev()
    x, wt, b = _conv_data(case)
    ref = F.conv2d(x, wt, b, stride=s, padding=p, dilation=d)
    if relu:
        ref = F.relu(ref)
    g = ops.make_geom(n, h, w, ci, co, k, s, p, d)
    ld = ops.pad32(co)
    wf = ops.weight_ohwi(wt.to(dev), ld)
    y = ops.conv2d_fwd(_nhwc(x).to(dev), wf, b.to(dev), g, relu, ld=ld)
    torch.cuda.synchronize()
    assert y.shape == (n, g.Ho, g.Wo, ld)
    _close(y[..., :co], _nhwc(ref), what=f"conv fwd {case}")
    if ld != co:
        assert float(y[..., co:].abs().max()) == 0.0        # pad columns untouched


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_dgrad_and_wgrad(case):
    from objectdetection_ssd_amd import ops
    n, h, w, ci, co, k, s, p, d = case
    dev = _dev()
    x, wt, b = _conv_data(case, seed=1)
    x.requires_grad_(True); wt.requires_grad_(True); b.requires_grad_(True)
    y = F.conv2d(x, wt, b, stride=s, padding=p, dilation=d)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(2))
    y.backward(dy)
    g = ops.make_geom(n, h, w, ci, co, k, s, p, d)
    ld = ops.pad32(co)
    dy_p = torch.zeros(n, g.Ho, g.Wo, ld)
    dy_p[..., :co] = _nhwc(dy)
    dy_d = dy_p.to(dev)
    wb = ops.weight_ihwo(wt.detach().to(dev), ld)
    x_d = _nhwc(x.detach()).to(dev)
    # plain dgrad
    dx = ops.conv2d_dgrad(dy_d, wb, g)
    _close(dx, _nhwc(x.grad), what=f"dgrad {case}")
    # accumulate + relu mask
    prev = torch.randn(dx.shape, generator=torch.Generator().manual_seed(3)).to(dev)
    mask = torch.randn(dx.shape, generator=torch.Generator().manual_seed(4)).clamp_min(0).to(dev)
    dx2 = ops.conv2d_dgrad(dy_d, wb, g, dx=prev.clone(), relu_mask=mask, accumulate=True)
    ref2 = (_nhwc(x.grad) + prev.cpu()) * (mask.cpu() > 0)
    _close(dx2, ref2, what=f"dgrad acc+mask {case}")
    # wgrad + bias grad (deterministic: two runs are bitwise equal)
    dw, db = ops.conv2d_wgrad(x_d, dy_d, g, ld, True)
    dw_b, db_b = ops.conv2d_wgrad(x_d, dy_d, g, ld, True)
    torch.cuda.synchronize()
    _close(dw, wt.grad, tol=2e-4, what=f"wgrad {case}")
    _close(db, b.grad, tol=2e-4, what=f"bias grad {case}")
    assert torch.equal(dw, dw_b) and torch.equal(db, db_b)


POOLS = [(2, 2, 0, False, 300), (2, 2, 0, False, 75), (2, 2, 0, True, 75), (2, 2, 0, False, 38), (3, 1, 1, True, 19)]


@pytest.mark.parametrize("k,s,p,ceil,hw", POOLS)
def test_maxpool_fwd_bwd(k, s, p, ceil, hw):
    from objectdetection_ssd_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(6)
    n, c = 2, 64
    x = F.relu(torch.randn(n, c, hw, hw, generator=g)).requires_grad_(True)   # post-ReLU: many tied zeros
    y = F.max_pool2d(x, k, s, p, ceil_mode=ceil)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    yd, am = ops.maxpool_fwd(_nhwc(x.detach()).to(dev), k, s, p, ceil)
    assert yd.shape == _nhwc(y).shape
    assert torch.equal(yd.cpu(), _nhwc(y.detach()))                       # max is exact
    xd = _nhwc(x.detach()).to(dev)
    # with the ReLU mask the routing of tied zeros cannot matter
    dx = ops.maxpool_bwd(_nhwc(dy).to(dev), am, tuple(xd.shape), k, s, p, relu_mask=xd)
    ref = _nhwc(x.grad * (x.detach() > 0))
    _close(dx, ref, tol=1e-6, what="maxpool bwd + mask")
    dxg = ops.maxpool_bwd(_nhwc(dy).to(dev), am, tuple(xd.shape), k, s, p, y_gate=yd)       # same mask through the pooled output
    assert torch.equal(dxg, dx)
    with pytest.raises(ValueError):
        ops.maxpool_bwd(_nhwc(dy).to(dev), am, tuple(xd.shape), k, s, p, relu_mask=xd, y_gate=yd)
    prev = torch.randn(ref.shape, generator=g)
    dx2 = ops.maxpool_bwd(_nhwc(dy).to(dev), am, tuple(xd.shape), k, s, p, dx=prev.to(dev), relu_mask=xd, accumulate=True)
    _close(dx2, (_nhwc(x.grad) + prev) * (_nhwc(x.detach()) > 0), tol=1e-6, what="maxpool bwd acc")
    # without mask and without ties: exact torch routing
    x2 = torch.randn(n, c, hw, hw, generator=g).requires_grad_(True)
    y2 = F.max_pool2d(x2, k, s, p, ceil_mode=ceil)
    y2.backward(dy)
    _, am2 = ops.maxpool_fwd(_nhwc(x2.detach()).to(dev), k, s, p, ceil)
    dx3 = ops.maxpool_bwd(_nhwc(dy).to(dev), am2, tuple(xd.shape), k, s, p)
    _close(dx3, _nhwc(x2.grad), tol=1e-6, what="maxpool bwd plain")


def test_l2norm_fwd_bwd():
    from objectdetection_ssd_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(7)
    x = F.relu(torch.randn(2, 512, 38, 38, generator=g)).requires_grad_(True)
    gamma = (20 + torch.randn(1, 512, 1, 1, generator=g)).requires_grad_(True)
    y = x / x.pow(2).sum(dim=1, keepdim=True).sqrt() * gamma                # Model.py:207-209
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xd = _nhwc(x.detach()).to(dev)
    gd = gamma.detach().reshape(-1).to(dev)
    yd = ops.l2norm_fwd(xd, gd)
    _close(yd, _nhwc(y), tol=1e-5, what="l2norm fwd")
    dx, dg = ops.l2norm_bwd(xd, gd, _nhwc(dy).to(dev))
    _close(dx, _nhwc(x.grad), tol=1e-4, what="l2norm dx")
    _close(dg, gamma.grad.reshape(-1), tol=2e-4, what="l2norm dgamma")


def test_heads_scatter_gather_roundtrip():
    from objectdetection_ssd_amd import ops
    dev = _dev()
    n, hw, a, P = 2, 25, 6, 8732
    ld = ops.pad32(a * 25)
    g = torch.Generator().manual_seed(8)
    packed = torch.randn(n * hw, ld, generator=g).to(dev)
    loc = torch.zeros(n, P, 4, device=dev)
    conf = torch.zeros(n, P, 21, device=dev)
    off = 8542
    ops.heads_scatter(packed, ld, loc, conf, n, hw, a, off)
    pc = packed.cpu().view(n, hw, ld)
    ref_loc = pc[:, :, :4 * a].reshape(n, hw * a, 4)                        # Model.py:212 permute+view
    ref_conf = pc[:, :, 4 * a:25 * a].reshape(n, hw * a, 21)
    assert torch.equal(loc.cpu()[:, off:off + hw * a], ref_loc)
    assert torch.equal(conf.cpu()[:, off:off + hw * a], ref_conf)
    assert float(loc.cpu()[:, :off].abs().max()) == 0.0
    back = ops.heads_gather(loc, conf, ld, n, hw, a, off)
    exp = packed.clone()
    exp[:, 25 * a:] = 0
    assert torch.equal(back, exp)


def test_sgd_momentum_matches_torch():
    from objectdetection_ssd_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    p0 = torch.randn(100003, generator=g)
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([p_ref], lr=1e-2, momentum=0.9, weight_decay=5e-4)
    p = p0.clone().to(dev)
    buf = torch.zeros_like(p)
    for step in range(3):
        gr = torch.randn(100003, generator=g)
        p_ref.grad = gr.clone()
        opt.step()
        ops.sgd_momentum_(p, gr.to(dev), buf, 1e-2, 0.9, 5e-4, first_step=(step == 0))
    _close(p, p_ref.detach(), tol=1e-6, what="sgd")


def test_conv_first_via_im2col():
    """conv1_1 = im2col (K 27->32) + 1x1 MFMA conv; wgrad through the same buffer"""
    from objectdetection_ssd_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(15)
    n, h, w = 2, 60, 77
    x = torch.randn(n, 3, h, w, generator=g)
    wt = (torch.randn(64, 3, 3, 3, generator=g) * 0.27).requires_grad_(True)
    b = (torch.randn(64, generator=g) * 0.1).requires_grad_(True)
    y = F.relu(F.conv2d(x, wt, b, padding=1))
    dy = torch.randn(y.shape, generator=g) * (y > 0)
    y.backward(dy)
    col = ops.im2col_first(x.to(dev))
    ref_col = F.unfold(x, 3, padding=1).view(n, 3, 9, h, w).permute(0, 3, 4, 2, 1).reshape(n, h, w, 27)   # [(r,s)][c]
    assert torch.equal(col[..., :27].cpu(), ref_col) and float(col[..., 27:].abs().max()) == 0.0
    geom = ops.make_geom(n, h, w, 32, 64, 1, 1, 0, 1)
    yd = ops.conv2d_fwd(col, ops.first_weight_rows(wt.detach().to(dev)), b.detach().to(dev), geom, True)
    _close(yd, _nhwc(y), what="conv1_1 via im2col")
    dw, db = ops.conv2d_wgrad(col, _nhwc(dy).to(dev), geom, 64, True)
    _close(ops.first_weight_grad(dw), wt.grad, tol=2e-4, what="conv1_1 wgrad via im2col")
    _close(db, b.grad, tol=2e-4, what="conv1_1 bias grad via im2col")


VARIANT_CASES = [(2, 19, 19, 64, 64, 3, 1, 1, 1), (2, 38, 38, 128, 256, 3, 1, 1, 1), (2, 19, 19, 256, 100, 3, 1, 1, 1),
                 (2, 10, 10, 128, 256, 3, 2, 1, 1), (3, 5, 5, 128, 256, 3, 1, 0, 1), (2, 19, 19, 512, 160, 1, 1, 0, 1),
                 (1, 37, 53, 64, 96, 3, 1, 1, 1), (1, 75, 75, 64, 64, 3, 1, 1, 1)]


@pytest.mark.parametrize("case", VARIANT_CASES)
def test_every_kernel_variant(case):
    """every instantiated tile / stage variant of the igemm and wgrad kernels (the dispatcher's choice is a
    speed matter only: results must not depend on it)"""
    from objectdetection_ssd_amd import _lib, ops
    lib = _lib.load()
    n, h, w, ci, co, k, s, p, d = case
    dev = _dev()
    x, wt, b = _conv_data(case, seed=21)
    x.requires_grad_(True); wt.requires_grad_(True); b.requires_grad_(True)
    y = F.conv2d(x, wt, b, stride=s, padding=p, dilation=d)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(22))
    y.backward(dy)
    g = ops.make_geom(n, h, w, ci, co, k, s, p, d)
    ld = ops.pad32(co)
    x_d = _nhwc(x.detach()).to(dev)
    dy_p = torch.zeros(n, g.Ho, g.Wo, ld)
    dy_p[..., :co] = _nhwc(dy)
    dy_d = dy_p.to(dev)
    wf = ops.weight_ohwi(wt.detach().to(dev), ld)
    wb = ops.weight_ihwo(wt.detach().to(dev), ld)
    try:
        for tile in range(4):
            for nbuf in (1, 2):
                assert lib.ssd_tune_set_igemm(tile, nbuf) == 0
                yd = ops.conv2d_fwd(x_d, wf, b.detach().to(dev), g, False, ld=ld)
                _close(yd[..., :co], _nhwc(y), what=f"fwd tile {tile} nbuf {nbuf} {case}")
                dx = ops.conv2d_dgrad(dy_d, wb, g)
                _close(dx, _nhwc(x.grad), what=f"dgrad tile {tile} nbuf {nbuf} {case}")
        lib.ssd_tune_set_igemm(-1, -1)
        for bt, nbuf in ((3, 1), (64, 1), (64, 2), (128, 1), (128, 2)):
            for bpc in (-1, 1, 40):
                lib.ssd_tune_set_wgrad(bt, nbuf, bpc)
                dw, db = ops.conv2d_wgrad(x_d, dy_d, g, ld, True)
                _close(dw, wt.grad, tol=2e-4, what=f"wgrad bt {bt} nbuf {nbuf} bpc {bpc} {case}")
                _close(db, b.grad, tol=2e-4, what=f"bias bt {bt} nbuf {nbuf} bpc {bpc} {case}")
        if (k, s, p, d) == (3, 1, 1, 1):                  # fused nine-tap kernel: every patch shape (4x8, 1x38, 2x19)
            for shape in (0, 1, 2):
                lib.ssd_tune_set_wgrad(3, 1, -1)
                assert lib.ssd_tune_set_wgrad_patch(shape) == 0
                dw, db = ops.conv2d_wgrad(x_d, dy_d, g, ld, True)
                _close(dw, wt.grad, tol=2e-4, what=f"fused wgrad patch shape {shape} {case}")
                _close(db, b.grad, tol=2e-4, what=f"fused bias patch shape {shape} {case}")
    finally:
        lib.ssd_tune_set_igemm(-1, -1)
        lib.ssd_tune_set_wgrad(-1, -1, -1)
        lib.ssd_tune_set_wgrad_patch(-1)


@pytest.mark.parametrize("case", [CONV_CASES[i] for i in (0, 1, 2, 4, 5, 6, 8, 10, 11)])
def test_conv_bf16_operand_variant(case):
    """bf16-operand kernels (configs[2]): against an f32 torch conv of the SAME bf16-rounded operands the result must
    agree to f32 accuracy (bf16 x bf16 products are exact in f32); against the unrounded conv it is bf16-accurate."""
    from objectdetection_ssd_amd import _lib, ops
    lib = _lib.load()
    n, h, w, ci, co, k, s, p, d = case
    dev = _dev()
    x, wt, b = _conv_data(case, seed=31)
    xr, wr = x.bfloat16().float().requires_grad_(True), wt.bfloat16().float().requires_grad_(True)
    y = F.conv2d(xr, wr, b, stride=s, padding=p, dilation=d)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(32)).bfloat16().float()
    y.backward(dy)
    g = ops.make_geom(n, h, w, ci, co, k, s, p, d)
    ld = ops.pad32(co)
    wf, wb = ops.weight_ohwi(wt.to(dev), ld), ops.weight_ihwo(wt.to(dev), ld)
    dy_p = torch.zeros(n, g.Ho, g.Wo, ld)
    dy_p[..., :co] = _nhwc(dy)
    try:
        for tile in (-1, 0, 1, 2, 3):
            assert lib.ssd_tune_set_igemm_bf16(tile) == 0
            yd = ops.conv2d_fwd(_nhwc(x).to(dev), wf, b.to(dev), g, False, ld=ld, bf16=True)
            _close(yd[..., :co], _nhwc(y), tol=2e-5, what=f"bf16 fwd tile {tile} {case}")
            dx = ops.conv2d_dgrad(dy_p.to(dev), wb, g, bf16=True)
            _close(dx, _nhwc(xr.grad), tol=2e-5, what=f"bf16 dgrad tile {tile} {case}")
        if (k, s, p, d) == (3, 1, 1, 1) and ci % 32 == 0:       # halo-tile kernel on plane 0 of the split weights
            wf3, wb3 = ops.weight_split3(wf), ops.weight_split3(wb)
            for halo in (1, 2):
                assert lib.ssd_tune_set_halo(halo) == 0
                yh = torch.zeros_like(yd)
                yh = ops.conv2d_fwd(_nhwc(x).to(dev), wf, b.to(dev), g, False, ld=ld, out=yh, bf16=True, w3=wf3)
                _close(yh[..., :co], _nhwc(y), tol=2e-5, what=f"bf16 halo fwd {halo} {case}")
                dxh = ops.conv2d_dgrad(dy_p.to(dev), wb, g, bf16=True, w3=wb3)
                _close(dxh, _nhwc(xr.grad), tol=2e-5, what=f"bf16 halo dgrad {halo} {case}")
                mask = (torch.rand(dxh.shape, generator=torch.Generator().manual_seed(5)) > .5).float().to(dev)
                dxa = ops.conv2d_dgrad(dy_p.to(dev), wb, g, dx=dxh.clone(), relu_mask=mask, accumulate=True, bf16=True, w3=wb3)
                _close(dxa, 2 * dxh * mask, tol=2e-5, what=f"bf16 halo dgrad accumulate+mask {halo} {case}")
    finally:
        lib.ssd_tune_set_igemm_bf16(-1)
        lib.ssd_tune_set_halo(-1)
    # weight gradient: the bf16 patch kernel where it applies (stride 1 and: 3x3 with padding = dilation 1 or 4, or 1x1), f32 kernels elsewhere
    dw, db = ops.conv2d_wgrad(_nhwc(x).to(dev), dy_p.to(dev), g, ld, True, bf16=True)
    fused = s == 1 and ((k == 3 and p == d and d in (1, 4)) or (k == 1 and p == 0))
    ref_dw = wr.grad if fused else None
    if ref_dw is None:                                    # f32 path: unrounded x
        x2 = x.clone().requires_grad_(False)
        w2 = wt.clone().requires_grad_(True)
        F.conv2d(x2, w2, b, stride=s, padding=p, dilation=d).backward(dy)
        ref_dw = w2.grad
    _close(dw, ref_dw, tol=1e-4, what=f"bf16-mode wgrad {case}")
    _close(db, dy.sum(dim=(0, 2, 3)), tol=1e-4, what=f"bf16-mode bias grad {case}")
    full = F.conv2d(x, wt, b, stride=s, padding=p, dilation=d)
    rel = float((yd[..., :co].cpu() - _nhwc(full)).norm() / _nhwc(full).norm())
    assert 1e-4 < rel < 2e-2, rel             # it really is bf16 arithmetic, and no worse than bf16


@pytest.mark.parametrize("case", [CONV_CASES[i] for i in (0, 1, 2, 4, 5, 6, 8, 10, 11)])
def test_conv_f32_from_three_bf16_limbs(case):
    """"f32x3" kernels: operands split exactly into three bf16 limbs, six limb products per block.  Judged against an f64
    convolution: the error must be at f32 level -- no worse than 2x the exact-f32 MFMA kernel's own error (+1e-7)."""
    from objectdetection_ssd_amd import _lib, ops
    lib = _lib.load()
    n, h, w, ci, co, k, s, p, d = case
    dev = _dev()
    x, wt, b = _conv_data(case, seed=41)
    x64 = x.double().requires_grad_(True)
    y64 = F.conv2d(x64, wt.double(), b.double(), stride=s, padding=p, dilation=d)
    dy = torch.randn(y64.shape, generator=torch.Generator().manual_seed(42))
    y64.backward(dy.double())
    g = ops.make_geom(n, h, w, ci, co, k, s, p, d)
    ld = ops.pad32(co)
    wf, wb = ops.weight_ohwi(wt.to(dev), ld), ops.weight_ihwo(wt.to(dev), ld)
    wf3, wb3 = ops.weight_split3(wf), ops.weight_split3(wb)
    assert torch.equal(wf3.float().sum(0), wf)                     # hi + mid + lo == w exactly
    dy_p = torch.zeros(n, g.Ho, g.Wo, ld)
    dy_p[..., :co] = _nhwc(dy)
    xd, dyd = _nhwc(x).to(dev), dy_p.to(dev)
    lib.ssd_tune_set_igemm_splitk(1)                  # baseline: the plain f32 kernel, one accumulation chain over all of K
    y_f32 = ops.conv2d_fwd(xd, wf, b.to(dev), g, False, ld=ld)[..., :co].cpu().double()
    dx_f32 = ops.conv2d_dgrad(dyd, wb, g).cpu().double()
    lib.ssd_tune_set_igemm_splitk(-1)
    ref_y, ref_dx = _nhwc(y64.detach()), _nhwc(x64.grad)
    e_y32 = float((y_f32 - ref_y).norm() / ref_y.norm())
    e_dx32 = float((dx_f32 - ref_dx).norm() / ref_dx.norm())
    try:
        # (tile, halo): generic x3 kernel in every tile; for the 3x3/s1 layers also the halo-tile kernel in both patch shapes
        combos = [(-1, 0), (1, 0), (2, 0), (3, 0), (-1, -1)] + ([(-1, 1), (-1, 2)] if (k, s, p, d) == (3, 1, 1, 1) else [])
        for tile, halo in combos:
            assert lib.ssd_tune_set_igemm_x3(tile) == 0 and lib.ssd_tune_set_halo(halo) == 0
            y3 = ops.conv2d_fwd_x3(xd, wf3, b.to(dev), g, False, ld=ld)[..., :co].cpu().double()
            dx3 = ops.conv2d_dgrad_x3(dyd, wb3, g).cpu().double()
            e_y3 = float((y3 - ref_y).norm() / ref_y.norm())
            e_dx3 = float((dx3 - ref_dx).norm() / ref_dx.norm())
            assert e_y3 <= 2 * e_y32 + 1e-7, (tile, halo, e_y3, e_y32)
            assert e_dx3 <= 2 * e_dx32 + 1e-7, (tile, halo, e_dx3, e_dx32)
            _close(y3, ref_y, tol=1e-5, what=f"x3 fwd {case}")
            _close(dx3, ref_dx, tol=1e-5, what=f"x3 dgrad {case}")
    finally:
        lib.ssd_tune_set_igemm_x3(-1)
        lib.ssd_tune_set_halo(-1)
    print(f"{case}: fwd rel err f32 {e_y32:.2e} x3 {e_y3:.2e}; dgrad f32 {e_dx32:.2e} x3 {e_dx3:.2e}")


# ---- SSD_resnet34 pieces (Model.py:12-126) ---------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 224, 224, 7, 2, 3), (1, 37, 53, 7, 2, 3), (2, 30, 30, 3, 1, 1), (1, 16, 20, 5, 3, 0)])
def test_stem_im2col_and_conv(shape):
    """3-channel NCHW im2col + the 1x1 MFMA convolution == F.conv2d (the ResNet-34 stem is 7x7/s2/p3)."""
    from objectdetection_ssd_amd import ops
    n, h, w, k, s, p = shape
    dev = _dev()
    g0 = torch.Generator().manual_seed(3)
    x = torch.randn(n, 3, h, w, generator=g0)
    wt = torch.randn(64, 3, k, k, generator=g0) * (2.0 / (3 * k * k)) ** 0.5
    b = torch.randn(64, generator=g0) * 0.1
    col = ops.im2col_nchw3(x.to(dev), k, s, p)
    ref_col = F.unfold(x, k, padding=p, stride=s)                         # (n, 3*k*k, L) with row index c*k*k + t
    L = ref_col.shape[2]
    ref_col = ref_col.view(n, 3, k * k, L).permute(0, 3, 2, 1).reshape(n, L, k * k * 3)
    got = col.view(n, L, -1).cpu()
    assert torch.equal(got[..., :k * k * 3], ref_col) and float(got[..., k * k * 3:].abs().max()) == 0.0
    g = ops.make_geom(n, col.shape[1], col.shape[2], col.shape[3], 64, 1, 1, 0, 1)
    y = ops.conv2d_fwd(col, ops.stem_weight_rows(wt.to(dev)), b.to(dev), g, True)
    _close(y, _nhwc(F.relu(F.conv2d(x, wt, b, stride=s, padding=p))), what=f"stem {shape}")


@pytest.mark.gpu
@pytest.mark.parametrize("case", [CONV_CASES[i] for i in (0, 1, 5, 6)])
@pytest.mark.parametrize("bf16", [False, True])
def test_conv2d_fwd_accumulate_residual(case, bf16):
    """relu(conv(x) + bias + identity) accumulated in place into the identity buffer (BasicBlock tail)."""
    from objectdetection_ssd_amd import ops
    n, h, w, ci, co, k, s, p, d = case
    dev = _dev()
    x, wt, b = _conv_data(case, seed=51)
    if bf16:
        x, wt = x.bfloat16().float(), wt.bfloat16().float()
    y0 = F.conv2d(x, wt, b, stride=s, padding=p, dilation=d)
    idt = torch.randn(y0.shape, generator=torch.Generator().manual_seed(52))
    g = ops.make_geom(n, h, w, ci, co, k, s, p, d)
    buf = _nhwc(idt).to(dev)
    out = ops.conv2d_fwd(_nhwc(x).to(dev), ops.weight_ohwi(wt.to(dev)), b.to(dev), g, True, out=buf, bf16=bf16, accumulate=True)
    assert out.data_ptr() == buf.data_ptr()
    _close(out, _nhwc(F.relu(y0 + idt)), tol=2e-5 if bf16 else 1e-4, what=f"fwd accumulate {case}")


@pytest.mark.gpu
def test_channel_affine():
    from objectdetection_ssd_amd import ops
    dev = _dev()
    g0 = torch.Generator().manual_seed(9)
    x = torch.randn(3, 7, 7, 256, generator=g0)
    sc, sh = torch.rand(256, generator=g0) + .5, torch.randn(256, generator=g0)
    y = ops.channel_affine(x.to(dev), sc.to(dev), sh.to(dev))
    assert torch.equal(y.cpu(), x * sc + sh) or float((y.cpu() - (x * sc + sh)).abs().max()) < 1e-6
    xin = x.to(dev)
    y2 = ops.channel_affine(xin, sc.to(dev), sh.to(dev), relu=True, out=xin)          # in place
    assert float((y2.cpu() - F.relu(x * sc + sh)).abs().max()) < 1e-6
    with pytest.raises(ValueError):
        ops.channel_affine(torch.zeros(2, 6, device=dev), torch.zeros(6, device=dev), torch.zeros(6, device=dev))


@pytest.mark.parametrize("case", [CONV_CASES[i] for i in (0, 4, 5, 6, 8, 9, 10)])
@pytest.mark.parametrize("ksplit", [1, 2, 3, 7])
def test_conv_split_k(case, ksplit):
    """Split-K path of the forward / dgrad igemm (small grids, deep K): forced slice counts incl. ones that do not divide
    the K steps; bias + ReLU + padded leading dimension (fwd), accumulate + mask + stride 2 (dgrad); bitwise reproducible."""
    from objectdetection_ssd_amd import _lib, ops
    lib = _lib.load()
    n, h, w, ci, co, k, s, p, d = case
    dev = _dev()
    x, wt, b = _conv_data(case, seed=61)
    x.requires_grad_(True)
    y = F.conv2d(x, wt, b, stride=s, padding=p, dilation=d)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(62))
    y.backward(dy)
    g = ops.make_geom(n, h, w, ci, co, k, s, p, d)
    ld = ops.pad32(co)
    wf, wb = ops.weight_ohwi(wt.to(dev), ld), ops.weight_ihwo(wt.to(dev), ld)
    dy_p = torch.zeros(n, g.Ho, g.Wo, ld)
    dy_p[..., :co] = _nhwc(dy)
    try:
        assert lib.ssd_tune_set_igemm_splitk(ksplit) == 0
        if ksplit > 1:
            assert lib.ssd_conv2d_igemm_workspace(g, 0) > 0 and lib.ssd_conv2d_igemm_workspace(g, 1) > 0
        else:
            assert lib.ssd_conv2d_igemm_workspace(g, 0) == 0
        yd = ops.conv2d_fwd(_nhwc(x.detach()).to(dev), wf, b.to(dev), g, True, ld=ld)
        yd2 = ops.conv2d_fwd(_nhwc(x.detach()).to(dev), wf, b.to(dev), g, True, ld=ld)
        _close(yd[..., :co], _nhwc(F.relu(y.detach())), what=f"split-K {ksplit} fwd {case}")
        assert torch.equal(yd, yd2)
        if ld != co:
            assert float(yd[..., co:].abs().max()) == 0.0
        prev = torch.randn(n, h, w, ci, generator=torch.Generator().manual_seed(63)).to(dev)
        mask = torch.randn(n, h, w, ci, generator=torch.Generator().manual_seed(64)).clamp_min(0).to(dev)
        dx = ops.conv2d_dgrad(dy_p.to(dev), wb, g, dx=prev.clone(), relu_mask=mask, accumulate=True)
        _close(dx, (_nhwc(x.grad) + prev.cpu()) * (mask.cpu() > 0), what=f"split-K {ksplit} dgrad {case}")
    finally:
        lib.ssd_tune_set_igemm_splitk(-1)


FULL_SIZE_LAYERS = [  # BASELINE configs[1] geometries at batch 32: (H, Ci, Co, k, stride, pad, dil)
    (300, 64, 64, 3, 1, 1, 1),       # conv1_2
    (75, 256, 256, 3, 1, 1, 1),      # conv3_2
    (38, 512, 512, 3, 1, 1, 1),      # conv4_2
    (19, 512, 1024, 3, 1, 4, 4),     # fc6 (dilated)
    (19, 1024, 150, 3, 1, 1, 1),     # head c_7 (split-K forward)
    (19, 256, 512, 3, 2, 1, 1),      # seq8.2 (stride 2)
    (3, 128, 256, 3, 1, 0, 1),       # seq11.2
]


@pytest.mark.parametrize("layer", FULL_SIZE_LAYERS)
def test_conv_adjoint_identities_at_full_batch(layer):
    """Size-independent properties at the benchmark's batch 32 (no CPU reference needed): the three convolution kernels are
    adjoints of one another -- <conv(x; W), dy> == <x, dgrad(dy; W)> == <W, wgrad(x, dy)> (bias off), and forward is linear
    in x.  Dots are accumulated in float64; 1e-4 relative."""
    from objectdetection_ssd_amd import ops
    h, ci, co, k, s, p, d = layer
    dev = _dev()
    g0 = torch.Generator(device=dev).manual_seed(h * 7 + co)
    x1 = torch.randn(32, h, h, ci, device=dev, generator=g0)
    x2 = torch.randn(32, h, h, ci, device=dev, generator=g0)
    w = torch.randn(co, ci, k, k, device=dev, generator=g0) * (2.0 / (ci * k * k)) ** 0.5
    g = ops.make_geom(32, h, h, ci, co, k, s, p, d)
    ld = ops.pad32(co)
    wf, wb = ops.weight_ohwi(w, ld), ops.weight_ihwo(w, ld)
    dy = torch.zeros(32, g.Ho, g.Wo, ld, device=dev)
    dy[..., :co] = torch.randn(32, g.Ho, g.Wo, co, device=dev, generator=g0)
    y1 = ops.conv2d_fwd(x1, wf, None, g, False, ld=ld)
    y2 = ops.conv2d_fwd(x2, wf, None, g, False, ld=ld)
    y12 = ops.conv2d_fwd(x1 + x2, wf, None, g, False, ld=ld)
    _close(y12, y1 + y2, tol=1e-4, what=f"linearity {layer}")
    dx = ops.conv2d_dgrad(dy, wb, g)
    dw, db = ops.conv2d_wgrad(x1, dy, g, ld, True)

    def dot(a, b):
        return float((a.double() * b.double()).sum())
    lhs = dot(y1, dy)
    for name, rhs in (("dgrad", dot(x1, dx)), ("wgrad", dot(w, dw))):
        assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs)), (name, layer, lhs, rhs)
    assert abs(float(db.double().sum()) - float(dy.double().sum())) <= 1e-6 * float(dy.abs().double().sum())       # bias grad = column sums


WINO_BENCH_LAYERS = [  # the Winograd layers of BASELINE configs[1] as bench.py runs them at batch 32: (H, Ci, Co, pool ceil_mode or None)
    (300, 64, 64, False),     # conv1_2 + pool1 (fused output transform); 2 x 1.66 GB planes
    (150, 64, 128, None),     # conv2_1
    (150, 128, 128, False),   # conv2_2 + pool2
    (75, 256, 256, None),     # conv3_2
    (75, 256, 256, True),     # conv3_3 + pool3 (ceil_mode, 75 -> 38)
    (38, 512, 512, None),     # conv4_2
    (19, 512, 512, None),     # conv5_2
    (38, 512, 100, None),     # head c_4 (Co = 100 -> ld 128)
    (19, 1024, 150, None),    # head c_7
]


@pytest.mark.parametrize("layer", WINO_BENCH_LAYERS)
def test_winograd_entry_points_at_bench_batch(layer):
    """The code path bench.py times, at the size it times it (batch 32): `ssd_conv3x3_wino_fwd_keep` / `_fwd_pool` (kept planes),
    `_wgrad_planes` with `dgrad_planes_out` (the shared dy pass), `_dgrad_planes` -- held against the DIRECT exact-f32 kernels
    on the same operands (1e-4 of the scale, the bar of every f32 kernel here) and against each other through the adjoint
    identities <conv(x;W), dy> == <x, dgrad(dy;W)> == <W, wgrad(x,dy)> in float64 dots.  Split-K plans, plane sizes and the
    32-bit buffer offsets of these kernels depend on the batch, so small-batch tests do not cover them."""
    from objectdetection_ssd_amd import ops
    h, ci, co, pool = layer
    n = 32
    dev = _dev()
    gen = torch.Generator(device=dev).manual_seed(1000 + h + co)
    x = torch.randn(n, h, h, ci, device=dev, generator=gen)
    w = torch.randn(co, ci, 3, 3, device=dev, generator=gen) * (2.0 / (ci * 9)) ** 0.5
    b = torch.randn(co, device=dev, generator=gen) * 0.1
    g = ops.make_geom(n, h, h, ci, co, 3, 1, 1, 1)
    ld = ops.pad32(co)
    uf, ub = ops.wino_weights(w, ld, mo=4)
    wf, wb = ops.weight_ohwi(w, ld), ops.weight_ihwo(w, ld)
    dy = torch.zeros(n, h, h, ld, device=dev)
    dy[..., :co] = torch.randn(n, h, h, co, device=dev, generator=gen)

    def dot(a, c):
        return float((a.double() * c.double()).sum())

    # ---- forward, planes kept --------------------------------------------------------------------------------------
    y_d = ops.conv2d_fwd(x, wf, b, g, False, ld=ld)                       # direct exact-f32 MFMA kernel
    y_w, planes = ops.conv2d_fwd_wino(x, uf, b, g, False, ld=ld, keep_planes=True)
    assert tuple(planes.shape) == ops.wino_planes_shape(g)
    _close(y_w[..., :co], y_d[..., :co], what=f"wino fwd_keep vs direct {layer}")
    if ld != co:
        assert float(y_w[..., co:].abs().max()) == 0.0
    lhs = dot(y_w[..., :co] - b, dy[..., :co])                            # <conv(x;W), dy>, bias taken out
    if pool is not None:
        yp, am, planes_p = ops.conv2d_fwd_wino_pool(x, uf, b, g, pool, keep_planes=True)
        assert torch.equal(planes_p, planes)                              # the same transformed input either way
        y_r, planes_r = ops.conv2d_fwd_wino(x, uf, b, g, True, keep_planes=True)
        yp_ref, am_ref = ops.maxpool_fwd(y_r, 2, 2, 0, pool)
        assert torch.equal(yp, yp_ref) and torch.equal(am, am_ref)        # fused == two kernels, bit for bit
        yp_d, _ = ops.maxpool_fwd(torch.relu(y_d), 2, 2, 0, pool)
        _close(yp, yp_d, what=f"wino fwd_pool vs direct conv + relu + pool {layer}")
        del yp, am, planes_p, y_r, planes_r, yp_ref, am_ref, yp_d
    del y_d, y_w
    # ---- weight gradient on the kept planes + the data gradient's planes from the same pass over dy --------------------
    dw, db, dyp = ops.conv2d_wgrad_wino(None, dy, g, ld, True, mo=4, planes=planes, dgrad_planes=True)
    dw_d, db_d = ops.conv2d_wgrad(x, dy, g, ld, True)
    _close(dw, dw_d, what=f"wino wgrad_planes vs direct {layer}")
    _close(db, db_d, what=f"wino bias grad vs direct {layer}")
    dw2, db2 = ops.conv2d_wgrad_wino(x, dy, g, ld, True, mo=4)            # transforming x again gives the same bits
    assert torch.equal(dw2, dw) and torch.equal(db2, db)
    # every dw element is an f32 sum over N*H*W (2.9 M at conv1_2) products; its error bar is 1e-4 of max|dw| like above, so the dot
    # with W can be off by that times ||W||_2 (independent errors) -- |lhs| itself is a random-walk sum and no yardstick for it
    rhs_w = dot(w, dw)
    assert abs(lhs - rhs_w) <= 1e-4 * float(w.norm()) * float(dw.abs().max()), ("wgrad adjoint", layer, lhs, rhs_w)
    assert abs(float(db.double().sum()) - float(dy.double().sum())) <= 1e-5 * float(dy.abs().double().sum())
    del planes, dw2, db2, dw_d, db_d
    # ---- data gradient from those planes ------------------------------------------------------------------------------------
    dx_w = ops.conv2d_dgrad_wino(None, ub, g, planes=dyp)
    dx_d = ops.conv2d_dgrad(dy, wb, g)
    _close(dx_w, dx_d, what=f"wino dgrad_planes vs direct {layer}")
    assert torch.equal(ops.conv2d_dgrad_wino(dy, ub, g), dx_w)            # own dy transform == shared pass
    rhs_x = dot(x, dx_w)
    assert abs(lhs - rhs_x) <= 1e-4 * max(1.0, abs(lhs)), ("dgrad adjoint", layer, lhs, rhs_x)
    # accumulate + ReLU mask epilogue of the planes entry (what the backward of a two-consumer tensor uses)
    prev = torch.randn(n, h, h, ci, device=dev, generator=gen)
    mask = torch.randn(n, h, h, ci, device=dev, generator=gen).clamp_min(0)
    dx_a = ops.conv2d_dgrad_wino(None, ub, g, dx=prev.clone(), relu_mask=mask, accumulate=True, planes=dyp)
    _close(dx_a, (dx_d + prev) * (mask > 0), what=f"wino dgrad_planes accumulate + mask {layer}")
    # the ReLU mask as bits: the forward's input transform leaves x > 0 per (tile, channel quad) word; the dgrad epilogue applying
    # those bits must equal, bit for bit, the one reading x itself as the float mask
    xm = x * (torch.rand(x.shape, device=dev, generator=gen) > 0.5)        # a post-ReLU-like input: half of it exact zeros
    res = ops.conv2d_fwd_wino_pool(xm, uf, b, g, pool, keep_planes=True, want_bits=True) if pool is not None else \
        ops.conv2d_fwd_wino(xm, uf, b, g, True, ld=ld, keep_planes=True, want_bits=True)
    bits = res[-1]
    assert bits.dtype == torch.int64 and tuple(bits.shape) == (ops.wino_planes_shape(g)[1], ci // 4)
    tiles_side = (h + 3) // 4
    pth, ptw = min(5, tiles_side - 1), min(7, tiles_side - 1)              # a tile of image 1 (the last one on small maps: cut by the edge)
    t_probe = tiles_side * tiles_side + pth * tiles_side + ptw
    word = int(bits[t_probe, 3].item()) & ((1 << 64) - 1)
    for a_, b_, e_ in ((0, 0, 0), (1, 2, 3), (3, 3, 1), (2, 0, 2)):
        want = bool(xm[1, 4 * pth + a_, 4 * ptw + b_, 12 + e_] > 0) if (4 * pth + a_ < h and 4 * ptw + b_ < h) else False
        assert bool((word >> ((a_ * 4 + b_) * 4 + e_)) & 1) == want, (layer, a_, b_, e_)
    dx_m = ops.conv2d_dgrad_wino(None, ub, g, relu_mask=xm, planes=dyp)
    dx_b = ops.conv2d_dgrad_wino(None, ub, g, planes=dyp, bits=bits)
    assert torch.equal(dx_b, dx_m), f"bit mask != float mask {layer}"
    dx_ba = ops.conv2d_dgrad_wino(None, ub, g, dx=prev.clone(), accumulate=True, planes=dyp, bits=bits)
    assert torch.equal(dx_ba, ops.conv2d_dgrad_wino(None, ub, g, dx=prev.clone(), relu_mask=xm, accumulate=True, planes=dyp))


@pytest.mark.parametrize("layer", [(300, 64, 64, 3, 1, 1, 1), (38, 512, 512, 3, 1, 1, 1), (19, 512, 1024, 3, 1, 4, 4)])
def test_bf16_operand_kernels_at_bench_batch(layer):
    """BASELINE configs[2] at its per-GPU batch (32): conv1_2, conv4_2 and fc6 through the bf16-operand kernels the mode uses
    (halo-tile / generic igemm forward and dgrad, fused bf16 weight gradient or the f32 one) against an f32 torch-CPU
    convolution of the SAME bf16-rounded operands (bf16 x bf16 products are exact in f32, so only the f32 summation order
    differs): 2e-5 of the scale for forward / dgrad, 1e-4 for the weight gradient."""
    from objectdetection_ssd_amd import ops
    h, ci, co, k, s, p, d = layer
    n = 32
    dev = _dev()
    torch.set_num_threads(16)
    gen = torch.Generator().manual_seed(77 + h)
    x = torch.randn(n, ci, h, h, generator=gen)
    wt = torch.randn(co, ci, k, k, generator=gen) * (2.0 / (ci * k * k)) ** 0.5
    b = torch.randn(co, generator=gen) * 0.1
    xr, wr = x.bfloat16().float().requires_grad_(True), wt.bfloat16().float().requires_grad_(True)
    y = F.conv2d(xr, wr, b, stride=s, padding=p, dilation=d)
    dy = torch.randn(y.shape, generator=gen)
    dyr = dy.bfloat16().float()
    y.backward(dyr)
    g = ops.make_geom(n, h, h, ci, co, k, s, p, d)
    wd = wt.to(dev)
    wf, wb = ops.weight_ohwi(wd, co), ops.weight_ihwo(wd, co)
    wf3, wb3 = ops.weight_split3(wf), ops.weight_split3(wb)
    xd, dyd = _nhwc(x).to(dev), _nhwc(dy).to(dev)
    yd = ops.conv2d_fwd(xd, wf, b.to(dev), g, False, bf16=True, w3=wf3)
    _close(yd, _nhwc(y), tol=2e-5, what=f"bf16 fwd @32 {layer}")
    dx = ops.conv2d_dgrad(dyd, wb, g, bf16=True, w3=wb3)
    _close(dx, _nhwc(xr.grad), tol=2e-5, what=f"bf16 dgrad @32 {layer}")
    dw, db = ops.conv2d_wgrad(xd, dyd, g, co, True, bf16=True)
    if s == 1 and ((k == 3 and p == d and d in (1, 4)) or (k == 1 and p == 0)):
        ref_dw = wr.grad                                                   # bf16 patch kernel (incl. fc6 / the 1x1 layers): both operands rounded
    else:
        w2 = wt.clone().requires_grad_(True)                               # f32 kernel: unrounded x and dy
        F.conv2d(x, w2, None, stride=s, padding=p, dilation=d).backward(dy)
        ref_dw = w2.grad
    _close(dw, ref_dw, tol=1e-4, what=f"bf16-mode wgrad @32 {layer}")
    _close(db, dy.sum(dim=(0, 2, 3)), tol=1e-4, what=f"bf16-mode bias grad @32 {layer}")


FUSED_CASES = [  # n, h, w, ci, co: K = 64 / 128 / 256, ragged tile groups (tiles % 32 != 0), partial channel blocks (co % 64 != 0), cut tiles
    (1, 19, 19, 64, 64), (2, 30, 26, 128, 100), (1, 75, 75, 256, 32), (3, 10, 7, 64, 32), (1, 38, 38, 128, 256), (2, 13, 21, 64, 132),
]


@pytest.mark.parametrize("case", FUSED_CASES)
def test_winograd_fused_gemm_output_kernel_equals_two_kernel_form(case):
    """csrc/wino_fused.hip (36 plane GEMMs + output transform in one kernel, accumulators of all planes in registers) against the
    batched-GEMM + output-transform kernels it replaces, and against an f64 convolution: forward with bias / ReLU, forward fused with
    the 2x2 pool in both rounding modes (pooled values and argmax codes), dgrad from planes with float mask, bit mask, accumulate.
    Same sums in another order: 2e-5 of the scale between the two forms, 1e-4 against f64."""
    from objectdetection_ssd_amd import _lib, ops
    lib = _lib.load()
    n, h, w, ci, co = case
    dev = _dev()
    full = (n, h, w, ci, co, 3, 1, 1, 1)
    x, wt, b = _conv_data(full, seed=101)
    x = F.relu(x)                                                          # post-ReLU input: the bit mask has something to say
    x64 = x.double().requires_grad_(True)
    y64 = F.conv2d(x64, wt.double(), b.double(), padding=1)
    dy = torch.randn(y64.shape, generator=torch.Generator().manual_seed(102))
    y64.backward(dy.double())
    g = ops.make_geom(*full)
    ld = ops.pad32(co)
    uf, ub = ops.wino_weights(wt.to(dev), ld, mo=4)
    xd = _nhwc(x).to(dev)
    dy_p = torch.zeros(n, h, w, ld)
    dy_p[..., :co] = _nhwc(dy)
    dy_p = dy_p.to(dev)
    prev = torch.randn(n, h, w, ci, generator=torch.Generator().manual_seed(103)).to(dev)
    out = {}
    try:
        for mode in (0, 1, 2):                                             # two-kernel form, fused GEMMs + output, the whole convolution in one kernel
            if mode == 2 and not lib.ssd_has_experimental():               # wino4_full_kernel ships only in SSD_EXPERIMENTAL builds
                assert lib.ssd_tune_set_wino_full(1) != 0
                continue
            _lib.check(lib.ssd_tune_set_wino_fused(min(mode, 1)), "tune")
            _lib.check(lib.ssd_tune_set_wino_full(1 if mode == 2 else 0), "tune")
            if mode == 2 and not (ops.wino_uses_full(g, 0) or ops.wino_uses_full(g, 1)):
                continue
            r = {}
            r["y"], planes, bits = ops.conv2d_fwd_wino(xd, uf, b.to(dev), g, False, ld=ld, keep_planes=True, want_bits=True)
            r["y_relu"] = ops.conv2d_fwd_wino(xd, uf, b.to(dev), g, True, ld=ld)
            if co % 4 == 0:
                for ceil in (False, True):
                    r[f"pool{ceil}"], r[f"am{ceil}"] = ops.conv2d_fwd_wino_pool(xd, uf, b.to(dev), g, ceil)
            _, _, dyp = ops.conv2d_wgrad_wino(None, dy_p, g, ld, True, mo=4, planes=planes, dgrad_planes=True)
            r["dx"] = ops.conv2d_dgrad_wino(None, ub, g, planes=dyp)
            r["dx_mask"] = ops.conv2d_dgrad_wino(None, ub, g, relu_mask=xd, planes=dyp)
            r["dx_bits"] = ops.conv2d_dgrad_wino(None, ub, g, planes=dyp, bits=bits)
            r["dx_acc"] = ops.conv2d_dgrad_wino(None, ub, g, dx=prev.clone(), relu_mask=xd, accumulate=True, planes=dyp)
            r["dx_own"] = ops.conv2d_dgrad_wino(dy_p, ub, g)
            r["dx_own_bits"] = ops.conv2d_dgrad_wino(dy_p, ub, g, bits=bits)
            r["planes"], r["bits"] = planes, bits
            out[mode] = r
    finally:
        _lib.check(lib.ssd_tune_set_wino_fused(-1), "tune")
        _lib.check(lib.ssd_tune_set_wino_full(-1), "tune")
    a = out[0]
    for mode in out:
        if mode == 0:
            continue
        for k in a:
            if k.startswith("am") or k in ("planes", "bits"):
                continue
            _close(out[mode][k], a[k], tol=2e-5, what=f"mode {mode} vs two-kernel: {k} {case}")
        assert torch.equal(out[mode]["planes"], a["planes"]) and torch.equal(out[mode]["bits"], a["bits"]), f"kept planes / bits differ, mode {mode}"
    for r in out.values():
        _close(r["y"][..., :co], _nhwc(y64.detach()), what=f"fwd vs f64 {case}")
        _close(r["y_relu"][..., :co], _nhwc(F.relu(y64.detach())), what=f"fwd relu vs f64 {case}")
        if ld != co:
            assert float(r["y"][..., co:].abs().max()) == 0.0
        _close(r["dx"], _nhwc(x64.grad), what=f"dgrad vs f64 {case}")
        assert torch.equal(r["dx_bits"], r["dx_mask"]) and torch.equal(r["dx_own_bits"] * 1.0, r["dx_own"] * (xd > 0))
        _close(r["dx_own"], r["dx"], tol=2e-5, what=f"dgrad from dy vs from planes {case}")
        _close(r["dx_mask"], _nhwc(x64.grad) * (xd.cpu() > 0), what=f"dgrad mask vs f64 {case}")
        _close(r["dx_acc"], (_nhwc(x64.grad) + prev.cpu()) * (xd.cpu() > 0), what=f"dgrad accumulate + mask vs f64 {case}")
        if co % 4 == 0:
            for ceil in (False, True):
                p64 = F.max_pool2d(F.relu(y64.detach()), 2, 2, 0, ceil_mode=ceil)
                _close(r[f"pool{ceil}"], _nhwc(p64), what=f"fused pool vs f64 {case}")
                # the argmax code must point at a window element that carries the pooled value
                yfull = r["y_relu"][..., :co]
                code = r[f"am{ceil}"].long()
                oh = torch.arange(code.shape[1], device=dev).view(1, -1, 1, 1) * 2 + (code >> 1)
                ow = torch.arange(code.shape[2], device=dev).view(1, 1, -1, 1) * 2 + (code & 1)
                assert int(oh.max()) < h and int(ow.max()) < w
                picked = yfull[torch.arange(n, device=dev).view(-1, 1, 1, 1), oh, ow, torch.arange(co, device=dev).view(1, 1, 1, -1)]
                assert torch.equal(picked, r[f"pool{ceil}"]), f"argmax code does not select the maximum {case} ceil={ceil} mode"


WINO_CASES = [(2, 19, 19, 64, 64), (1, 38, 38, 128, 256), (2, 75, 75, 32, 64), (1, 19, 19, 512, 100), (3, 10, 7, 64, 32), (1, 5, 5, 32, 32)]


@pytest.mark.parametrize("case", WINO_CASES)
def test_winograd_f4x4_3x3_forward_and_dgrad(case):
    """F(4x4,3x3): 36 multiplies per 4x4 tile.  Its f32 error is larger than F(2x2)'s (coefficients up to 8 and 1/24) -- judged
    against an f64 convolution it has to stay within 1e-4 of the output scale, the bar of every f32 kernel here."""
    from objectdetection_ssd_amd import ops
    n, h, w, ci, co = case
    dev = _dev()
    full = (n, h, w, ci, co, 3, 1, 1, 1)
    x, wt, b = _conv_data(full, seed=81)
    x64 = x.double().requires_grad_(True)
    y64 = F.conv2d(x64, wt.double(), b.double(), padding=1)
    dy = torch.randn(y64.shape, generator=torch.Generator().manual_seed(82))
    y64.backward(dy.double())
    g = ops.make_geom(*full)
    ld = ops.pad32(co)
    uf, ub = ops.wino_weights(wt.to(dev), ld, mo=4)
    assert uf.shape[0] == 36
    yd = ops.conv2d_fwd_wino(_nhwc(x).to(dev), uf, b.to(dev), g, False, ld=ld)
    _close(yd[..., :co], _nhwc(y64.detach()), what=f"winograd F(4x4) fwd {case}")
    dy_p = torch.zeros(n, h, w, ld)
    dy_p[..., :co] = _nhwc(dy)
    dx = ops.conv2d_dgrad_wino(dy_p.to(dev), ub, g)
    _close(dx, _nhwc(x64.grad), what=f"winograd F(4x4) dgrad {case}")
    e = float((yd[..., :co].cpu().double() - _nhwc(y64.detach())).abs().max() / max(1.0, float(y64.detach().abs().max())))
    # weight gradient in the F(4x4) domain against f64
    w64 = wt.double().requires_grad_(True); b64 = b.double().requires_grad_(True)
    F.conv2d(x.double(), w64, b64, padding=1).backward(dy.double())
    dw, db = ops.conv2d_wgrad_wino(_nhwc(x).to(dev), dy_p.to(dev), g, ld, True, mo=4)
    _close(dw, w64.grad, tol=2e-4, what=f"winograd F(4x4) wgrad {case}")
    _close(db, b64.grad, tol=2e-4, what=f"winograd F(4x4) bias grad {case}")
    ew = float((dw.cpu().double() - w64.grad).abs().max() / max(1.0, float(w64.grad.abs().max())))
    print(f"F(4x4,3x3) {case}: fwd max err / scale {e:.2e}, wgrad {ew:.2e}")
    # the forward can leave its transformed input behind; the weight gradient on those planes is the same computation
    yk, planes = ops.conv2d_fwd_wino(_nhwc(x).to(dev), uf, b.to(dev), g, False, ld=ld, keep_planes=True)
    assert torch.equal(yk, yd) and tuple(planes.shape) == ops.wino_planes_shape(g)
    dw_p, db_p = ops.conv2d_wgrad_wino(None, dy_p.to(dev), g, ld, True, mo=4, planes=planes)
    assert torch.equal(dw_p, dw) and torch.equal(db_p, db)
    # ... and the same pass over dy can leave the data gradient's input planes: dgrad on them == dgrad on dy
    dw_q, db_q, dy_planes = ops.conv2d_wgrad_wino(None, dy_p.to(dev), g, ld, True, mo=4, planes=planes, dgrad_planes=True)
    assert torch.equal(dw_q, dw) and torch.equal(db_q, db)
    assert torch.equal(ops.conv2d_dgrad_wino(None, ub, g, planes=dy_planes), dx)
    # and the transposed-plane formulation (NT GEMM) of the same gradient agrees
    from objectdetection_ssd_amd import _lib
    _lib.check(_lib.load().ssd_tune_set_wino_wgrad_tn(0))
    try:
        dw_t, db_t = ops.conv2d_wgrad_wino(_nhwc(x).to(dev), dy_p.to(dev), g, ld, True, mo=4)
    finally:
        _lib.check(_lib.load().ssd_tune_set_wino_wgrad_tn(1))
    _close(dw_t, w64.grad, tol=2e-4, what=f"winograd F(4x4) wgrad, transposed planes {case}")
    _close(dw_t, dw, tol=2e-5, what=f"winograd F(4x4) wgrad TN vs NT {case}")


@pytest.mark.parametrize("case", [(2, 300, 300), (1, 37, 70), (3, 8, 8), (1, 129, 66)])
def test_conv1_1_written_as_winograd_planes(case):
    """csrc/conv_first.hip conv_first_wino_kernel (Model.py:135 features[0:4]): conv1_1 + ReLU left as the F(4x4) input planes + ReLU bit
    words of the 64 -> 64 layer behind it must be, bit for bit, what conv1_1's own kernel followed by that layer's input transform writes
    -- on the full 300 x 300 map, on maps that are no multiple of 4 or of the 64-column workgroup strip, and on a map smaller than a strip;
    and the layer's forward from those planes (plain and with the fused 2x2 pool) equals its forward from the activation."""
    from objectdetection_ssd_amd import ops
    if not ops.has_experimental():
        pytest.skip("conv_first_wino_kernel is compiled only with SSD_EXPERIMENTAL=1 (bit-identical, measured slower than the two kernels it replaces)")
    n, h, w = case
    dev = _dev()
    gen = torch.Generator().manual_seed(7 + h)
    x = torch.randn(n, 3, h, w, generator=gen).to(dev)
    w1 = (torch.randn(64, 3, 3, 3, generator=gen) / 5.0).to(dev)
    b1 = torch.randn(64, generator=gen).to(dev)
    w2 = (torch.randn(64, 64, 3, 3, generator=gen) / 24.0).to(dev)
    b2 = torch.randn(64, generator=gen).to(dev)
    rows = ops.first_weight_rows(w1)
    y1, _ = ops.conv1_first_fwd(x, rows, b1, True)
    g = ops.make_geom(n, h, w, 64, 64, 3, 1, 1, 1)
    uf, _ = ops.wino_weights(w2, 64, want_bwd=False, mo=4)
    y2, planes_ref, bits_ref = ops.conv2d_fwd_wino(y1, uf, b2, g, True, keep_planes=True, want_bits=True)
    planes, bits = ops.conv1_first_wino_fwd(x, rows, b1, want_bits=True)
    assert planes.shape == planes_ref.shape and bits.shape == bits_ref.shape
    assert torch.equal(bits, bits_ref), "ReLU bit words differ"
    assert torch.equal(planes, planes_ref), "input planes differ"
    assert torch.equal(ops.conv2d_fwd_wino_from_planes(planes, uf, b2, g, True), y2)
    for ceil in (False, True):
        yp_ref, am_ref = ops.conv2d_fwd_wino_pool(y1, uf, b2, g, ceil)
        yp, am = ops.conv2d_fwd_wino_from_planes(planes, uf, b2, g, True, pool_ceil=ceil)
        assert torch.equal(yp, yp_ref) and torch.equal(am, am_ref)
    p2, none = ops.conv1_first_wino_fwd(x, rows, b1, want_bits=False)
    assert none is None and torch.equal(p2, planes)


def _relu_bits(mask_nhwc: torch.Tensor, th: int, tw: int) -> torch.Tensor:
    """(N,H,W,C) -> (N*th*tw, C/4) int64 words, bit (a*4+b)*4+e = mask[n, 4th+a, 4tw+b, 4c4+e] > 0: what wino4_input_kernel leaves"""
    n, h, w, c = mask_nhwc.shape
    pad = torch.zeros(n, 4 * th, 4 * tw, c, dtype=torch.bool)
    pad[:, :h, :w] = mask_nhwc > 0
    t = pad.view(n, th, 4, tw, 4, c // 4, 4).permute(0, 1, 3, 5, 2, 4, 6).reshape(n * th * tw, c // 4, 64).to(torch.int64)
    sh = torch.arange(64, dtype=torch.int64)
    return (t << sh).sum(-1)                      # bit 63 wraps into the sign: the same 64-bit pattern


ADJ_CASES = [(2, 19, 19, 64, 256), (1, 38, 38, 128, 256), (2, 10, 13, 32, 512), (1, 75, 75, 64, 256), (3, 5, 9, 32, 288)]


@pytest.mark.parametrize("case", ADJ_CASES)
def test_adjoint_winograd_data_gradient(case):
    """The adjoint-form data gradient (csrc/winograd.hip wino4_adj_out_kernel; autograd of nn.Conv2d, reference Model.py:135-143): the
    planes A dy A^T of the weight gradient x the forward filter transform, 6x6 patches overlap-added by a gather over the neighbouring
    tiles.  Against an f64 `conv2d_input` within the f32 bar and no further from it than the rotated-filter form; += and ReLU masks (f32
    tensor and bit words) as the direct kernel applies them; and the CHAINED form -- the masked block taken as the dy of the layer below,
    written as that layer's planes + bias partial sums -- equals the dy transform of the stored tensor."""
    from objectdetection_ssd_amd import _lib, ops
    n, h, w, ci, co = case
    dev = _dev()
    full = (n, h, w, ci, co, 3, 1, 1, 1)
    x, wt, b = _conv_data(full, seed=181)
    g = ops.make_geom(*full)
    gen = torch.Generator().manual_seed(182)
    dy = torch.randn(n, co, h, w, generator=gen) * torch.exp(torch.randn(n, co, h, w, generator=gen))
    ref = torch.nn.grad.conv2d_input((n, ci, h, w), wt.double(), dy.double(), padding=1)
    dyd = _nhwc(dy).contiguous().to(dev)
    uf, ub = ops.wino_weights(wt.to(dev), co, mo=4)
    uadj = ops.wino_adj_weights(wt.to(dev), co)
    assert uadj.shape == ub.shape and uadj.dtype == ub.dtype
    Y, none, _ = ops.wino_dy_transform(dyd, g, co, False, False)
    assert none is None
    md = ops.wino_dgrad_adj_gemm(Y, uadj, g, co)
    dx = ops.wino_adj_output(md, g)
    dx_rot = ops.conv2d_dgrad_wino(dyd, ub, g)
    _close(dx, _nhwc(ref), what=f"adjoint dgrad {case}")
    ea = float((dx.cpu().double() - _nhwc(ref)).norm() / ref.norm())
    er = float((dx_rot.cpu().double() - _nhwc(ref)).norm() / ref.norm())
    print(f"adjoint dgrad {case}: rel L2 vs f64 {ea:.2e} (rotated-filter form {er:.2e})")
    assert ea <= 1.5 * er + 1e-7, (ea, er)
    # ReLU mask as an f32 tensor and as bit words, and accumulation into an existing gradient
    act = torch.randn(n, h, w, ci, generator=gen)
    base = torch.randn(n, h, w, ci, generator=gen)
    want = torch.where(act > 0, base + dx.cpu(), torch.zeros(()))
    got = ops.wino_adj_output(md, g, base.clone().to(dev), relu_mask=act.to(dev), accumulate=True)
    assert torch.equal(got.cpu(), want)
    th, tw = (h + 3) // 4, (w + 3) // 4
    bits = _relu_bits(act, th, tw).to(dev)
    got_b = ops.wino_adj_output(md, g, base.clone().to(dev), bits=bits, accumulate=True)
    assert torch.equal(got_b.cpu(), want)
    # chained: the masked block is the dy of a layer below with ci output channels
    if ci % 32 == 0:
        g_below = ops.make_geom(n, h, w, 32, ci, 3, 1, 1, 1)
        dmask = torch.where(act > 0, dx.cpu(), torch.zeros(())).contiguous().to(dev)
        Y_ref, _, part_ref = ops.wino_dy_transform(dmask, g_below, ci, False, True)
        Y2, part2 = ops.wino_adj_output_to_planes(md, g, g_below, bits=bits, want_bias=True)
        Y3, part3 = ops.wino_adj_output_to_planes(md, g, g_below, relu_mask=act.to(dev), want_bias=False)
        assert part3 is None and torch.equal(Y2, Y3)
        _close(Y2, Y_ref, tol=1e-6, what=f"chained dy planes {case}")
        lib = _lib.load()
        nb = part2.numel() // ci
        s2, sr = part2.view(nb, ci).double().sum(0), dmask.double().sum((0, 1, 2))
        # (the partial buffer has rows beyond the launched blocks: compare through the weight-gradient path instead when it is ragged)
        kept = torch.randn(36, Y2.shape[1], 32, device=dev)
        dw2, db2 = ops.wino_wgrad_gemm(Y2, kept, part2, g_below, ci)
        dwr, dbr = ops.wino_wgrad_gemm(Y_ref, kept, part_ref, g_below, ci)
        _close(dw2, dwr, tol=1e-6, what=f"chained wgrad {case}")
        _close(db2, dbr, tol=1e-6, what=f"chained bias gradient {case}")
        assert float((db2.cpu().double() - sr.cpu()).abs().max()) <= 1e-4 * max(1.0, float(sr.abs().max()))


@pytest.mark.parametrize("case", [(2, 22, 22, 32, 8, False), (1, 75, 75, 32, 12, True), (2, 9, 13, 64, 4, False), (1, 15, 11, 32, 8, True),
                                  (1, 150, 150, 64, 64, False)])
def test_winograd_forward_fused_with_maxpool(case):
    """conv3x3 -> ReLU -> MaxPool2d(2,2[,ceil]) in one pass == the F(4x4) convolution followed by the pool kernel, bit for bit
    (values and argmax codes), on even, odd and non-multiple-of-4 maps in both rounding modes; and the pooled result is the
    f64 convolution's within the f32 bar."""
    from objectdetection_ssd_amd import ops
    n, h, w, ci, co, ceil = case
    dev = _dev()
    full = (n, h, w, ci, co, 3, 1, 1, 1)
    x, wt, b = _conv_data(full, seed=91)
    g = ops.make_geom(*full)
    uf, _ = ops.wino_weights(wt.to(dev), co, want_bwd=False, mo=4)
    xd = _nhwc(x).to(dev)
    y_full = ops.conv2d_fwd_wino(xd, uf, b.to(dev), g, True)
    y_ref, am_ref = ops.maxpool_fwd(y_full, 2, 2, 0, ceil)
    yp, am = ops.conv2d_fwd_wino_pool(xd, uf, b.to(dev), g, ceil)
    assert yp.shape == y_ref.shape and am.shape == am_ref.shape
    assert torch.equal(yp, y_ref), f"pooled values differ {case}"
    assert torch.equal(am, am_ref), f"argmax codes differ {case}"
    yp2, am2 = ops.conv2d_fwd_wino_pool(xd, uf, b.to(dev), g, ceil, want_argmax=False)
    assert am2 is None and torch.equal(yp2, yp)
    yp3, am3, planes = ops.conv2d_fwd_wino_pool(xd, uf, b.to(dev), g, ceil, keep_planes=True)
    _, planes_ref = ops.conv2d_fwd_wino(xd, uf, b.to(dev), g, True, keep_planes=True)
    assert torch.equal(yp3, yp) and torch.equal(am3, am) and torch.equal(planes, planes_ref)
    y64 = F.max_pool2d(F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1)), 2, 2, 0, ceil_mode=ceil)
    _close(yp, _nhwc(y64), what=f"fused conv+pool {case}")


@pytest.mark.parametrize("case", WINO_CASES)
def test_winograd_f2x2_3x3_forward_and_dgrad(case):
    """Winograd F(2x2,3x3) == the direct 3x3 / stride 1 / pad 1 convolution (odd maps, padded channel counts, fused bias + ReLU,
    dgrad with accumulate + mask); the transforms cost a few ulps: 1e-4 of the output scale like every f32 kernel here."""
    from objectdetection_ssd_amd import ops
    n, h, w, ci, co = case
    dev = _dev()
    full = (n, h, w, ci, co, 3, 1, 1, 1)
    x, wt, b = _conv_data(full, seed=71)
    x.requires_grad_(True)
    y = F.conv2d(x, wt, b, padding=1)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(72))
    y.backward(dy)
    g = ops.make_geom(*full)
    ld = ops.pad32(co)
    uf, ub = ops.wino_weights(wt.to(dev), ld)
    yd = ops.conv2d_fwd_wino(_nhwc(x.detach()).to(dev), uf, b.to(dev), g, True, ld=ld)
    _close(yd[..., :co], _nhwc(F.relu(y.detach())), what=f"winograd fwd {case}")
    if ld != co:
        assert float(yd[..., co:].abs().max()) == 0.0
    dy_p = torch.zeros(n, h, w, ld)
    dy_p[..., :co] = _nhwc(dy)
    dx = ops.conv2d_dgrad_wino(dy_p.to(dev), ub, g)
    _close(dx, _nhwc(x.grad), what=f"winograd dgrad {case}")
    prev = torch.randn(n, h, w, ci, generator=torch.Generator().manual_seed(73)).to(dev)
    mask = torch.randn(n, h, w, ci, generator=torch.Generator().manual_seed(74)).clamp_min(0).to(dev)
    dx2 = ops.conv2d_dgrad_wino(dy_p.to(dev), ub, g, dx=prev.clone(), relu_mask=mask, accumulate=True)
    _close(dx2, (_nhwc(x.grad) + prev.cpu()) * (mask.cpu() > 0), what=f"winograd dgrad acc+mask {case}")
    # weight gradient in the Winograd domain (adjoint transforms), bias gradient by column sums; reproducible
    wt2 = wt.clone().requires_grad_(True); b2 = b.clone().requires_grad_(True)
    F.conv2d(x.detach(), wt2, b2, padding=1).backward(dy)
    dw, db = ops.conv2d_wgrad_wino(_nhwc(x.detach()).to(dev), dy_p.to(dev), g, ld, True)
    dw_b, db_b = ops.conv2d_wgrad_wino(_nhwc(x.detach()).to(dev), dy_p.to(dev), g, ld, True)
    _close(dw, wt2.grad, tol=2e-4, what=f"winograd wgrad {case}")
    _close(db, b2.grad, tol=2e-4, what=f"winograd bias grad {case}")
    assert torch.equal(dw, dw_b) and torch.equal(db, db_b)


def test_weight_job_table_equals_the_per_layer_transforms_bit_for_bit():
    """ssd_weights_prepare (one launch for every filter transform / re-layout of a training step) against the per-layer entry points it
    replaces: Winograd F(4x4) filters forward + rotated/transposed backward, OHWI / IHWO copies with padded output channels (a head's
    loc and conf filters come as two OIHW pieces; a 1x1 layer; fc6's dilated 3x3), conv1_1's im2col rows.  Pure data movement and the
    same G g G^T arithmetic: bit-exact."""
    from objectdetection_ssd_amd import ops
    dev = _dev()
    g_ = torch.Generator().manual_seed(21)
    rnd = lambda *s: torch.randn(*s, generator=g_).to(dev)
    w_wino, w_bb, w_cl = rnd(128, 64, 3, 3), rnd(16, 256, 3, 3), rnd(84, 256, 3, 3)
    w_hb, w_hc = rnd(24, 64, 3, 3), rnd(126, 64, 3, 3)
    w_1x1, w_fc6, w_first = rnd(96, 160, 1, 1), rnd(64, 32, 3, 3), rnd(64, 3, 3, 3)
    e = lambda *s: torch.full(s, float("nan"), device=dev)
    head = torch.cat([w_bb, w_cl])
    headw = torch.cat([w_hb, w_hc])                                   # 150 filters -> co_pad 160
    jobs = [
        dict(kind=0, w0=w_wino, co0=128, co=128, ci=64, taps=9, co_pad=128, out_fwd=e(36, 128, 64), out_bwd=e(36, 64, 128)),
        dict(kind=0, w0=w_hb, w1=w_hc, co0=24, co=150, ci=64, taps=9, co_pad=160, out_fwd=e(36, 150, 64), out_bwd=e(36, 64, 160)),
        dict(kind=1, w0=w_bb, w1=w_cl, co0=16, co=100, ci=256, taps=9, co_pad=128, out_fwd=e(128, 9, 256), out_bwd=e(256, 9, 128)),
        dict(kind=1, w0=w_1x1, co0=96, co=96, ci=160, taps=1, co_pad=96, out_fwd=e(96, 1, 160), out_bwd=e(160, 1, 96)),
        dict(kind=1, w0=w_fc6, co0=64, co=64, ci=32, taps=9, co_pad=64, out_fwd=e(64, 9, 32), out_bwd=None),
        dict(kind=2, w0=w_first, co0=64, co=64, ci=3, taps=9, co_pad=64, out_fwd=e(64, 1, 32)),
    ]
    # limb-plane outputs (reduction length >= 256 under the default ssd_tune_set_wino_x3): forward only, backward only, both
    w_xa, w_xb, w_xc = rnd(160, 256, 3, 3), rnd(256, 64, 3, 3), rnd(288, 256, 3, 3)
    for wq, cp in ((w_xa, 160), (w_xb, 256), (w_xc, 288)):
        co_, ci_ = wq.shape[0], wq.shape[1]
        jobs.append(dict(kind=0, w0=wq, co0=co_, co=co_, ci=ci_, taps=9, co_pad=cp, out_fwd=ops.wino_filter_alloc(4, co_, ci_, dev),
                         out_bwd=ops.wino_filter_alloc(4, ci_, cp, dev)))
    on = ops.wino_x3(4, 256)                       # (SSD_WINO_X3=0 in the environment: every filter stays f32 and the limb checks below are skipped)
    assert [jb["out_fwd"].dtype == torch.bfloat16 for jb in jobs[6:]] == [on, False, on]
    assert [jb["out_bwd"].dtype == torch.bfloat16 for jb in jobs[6:]] == [False, on, on]
    table = ops.WeightTable(jobs, dev)
    assert table.njobs == 9 and table.total_blocks > 9
    table.run()
    torch.cuda.synchronize()
    for jb, cp in zip(jobs[6:], (160, 256, 288)):
        uf, ub = ops.wino_weights(jb["w0"], co_pad=cp, mo=4)
        assert uf.dtype == jb["out_fwd"].dtype and ub.dtype == jb["out_bwd"].dtype
        assert torch.equal(jb["out_fwd"].view(torch.int16) if uf.dtype == torch.bfloat16 else jb["out_fwd"],
                           uf.view(torch.int16) if uf.dtype == torch.bfloat16 else uf)
        assert torch.equal(jb["out_bwd"].view(torch.int16) if ub.dtype == torch.bfloat16 else jb["out_bwd"],
                           ub.view(torch.int16) if ub.dtype == torch.bfloat16 else ub)
    if on:   # the limbs of a filter plane add up to the f32 transform: hi + mid + lo == U exactly
        lib = __import__("objectdetection_ssd_amd._lib", fromlist=["load"]).load()
        try:
            assert lib.ssd_tune_set_wino_x3(0) == 0
            uf32, _ = ops.wino_weights(w_xc, co_pad=288, mo=4)
        finally:
            assert lib.ssd_tune_set_wino_x3(-1) == 0
        lim = jobs[8]["out_fwd"].float().sum(2)[:, :, :288, :].permute(0, 2, 1, 3).reshape(36, 288, 256)
        assert torch.equal(lim, uf32)
    uf, ub = ops.wino_weights(w_wino, co_pad=128, mo=4)
    assert torch.equal(jobs[0]["out_fwd"], uf) and torch.equal(jobs[0]["out_bwd"], ub)
    uf, ub = ops.wino_weights(headw, co_pad=160, mo=4)
    assert torch.equal(jobs[1]["out_fwd"], uf) and torch.equal(jobs[1]["out_bwd"], ub)
    assert torch.equal(jobs[2]["out_fwd"], ops.weight_ohwi(head, 128)) and torch.equal(jobs[2]["out_bwd"], ops.weight_ihwo(head, 128))
    assert float(jobs[2]["out_fwd"][100:].abs().max()) == 0.0
    assert torch.equal(jobs[3]["out_fwd"], ops.weight_ohwi(w_1x1, 96)) and torch.equal(jobs[3]["out_bwd"], ops.weight_ihwo(w_1x1, 96))
    assert torch.equal(jobs[4]["out_fwd"], ops.weight_ohwi(w_fc6, 64))
    assert torch.equal(jobs[5]["out_fwd"], ops.first_weight_rows(w_first))
    # the table reads the live storage: a second run after an in-place update follows it
    w_wino.mul_(0.5)
    table.run()
    assert torch.equal(jobs[0]["out_fwd"], ops.wino_weights(w_wino, co_pad=128, mo=4)[0])
    with pytest.raises(ValueError):
        ops.WeightTable([dict(kind=7, w0=w_wino, co0=128, co=128, ci=64, taps=9, co_pad=128, out_fwd=e(4))], dev)


@pytest.mark.parametrize("case", [(2, 22, 22, 32, 64, False), (1, 75, 75, 32, 32, True), (2, 9, 13, 64, 64, False), (1, 15, 11, 32, 96, True),
                                  (32, 300, 300, 64, 64, False), (32, 75, 75, 256, 256, True)])
def test_winograd_dy_pass_from_the_pooled_gradient_equals_scatter_then_transform(case):
    """The backward of conv -> ReLU -> MaxPool2d(2, 2[, ceil]) (Model.py:135-137) without ever writing the pool's input gradient: the dy
    pass of the Winograd weight / data gradient reads (pooled gradient, argmax codes, pooled output as the ReLU gate) and forms dy on the
    fly.  Selection and gating move no bits, and the transform arithmetic is the same code: every output -- both plane sets, dw, dbias --
    must EQUAL the two-step form (ssd_maxpool_bwd_gated to memory, then the dy pass over it), odd sizes and ceil-mode edges included."""
    from objectdetection_ssd_amd import ops
    dev = _dev()
    n, h, w, ci, co, ceil = case
    g_ = torch.Generator().manual_seed(h * 131 + w)
    act = torch.relu(torch.randn(n, h, w, co, generator=g_)).to(dev)           # ~half the windows gate to zero somewhere
    y, am = ops.maxpool_fwd(act, 2, 2, 0, ceil)
    assert float((y == 0).float().mean()) > 0.01
    dpool = torch.randn(y.shape, generator=g_).to(dev)
    g = ops.make_geom(n, h, w, ci, co, 3, 1, 1, 1)
    dy_full = ops.maxpool_bwd(dpool, am, (n, h, w, co), 2, 2, 0, y_gate=y)
    lazy = ops.PooledGrad(dpool, am, y, (n, h, w, co))
    assert torch.equal(lazy.materialize(), dy_full)
    for want_planes in (True, False):
        Y0, V0, _ = ops.wino_dy_transform(dy_full, g, co, want_planes, True)
        Y1, V1, _ = ops.wino_dy_transform(lazy, g, co, want_planes, True)
        assert torch.equal(Y0, Y1), "wgrad planes"
        if want_planes:
            assert torch.equal(V0, V1), "dgrad planes"
            del V0, V1
        del Y0, Y1
    kept = torch.randn(ops.wino_planes_shape(g), generator=g_).to(dev) if n < 32 else torch.randn(ops.wino_planes_shape(g), device=dev)
    dw0, db0, p0 = ops.conv2d_wgrad_wino(None, dy_full, g, co, True, mo=4, planes=kept, dgrad_planes=True)
    dw1, db1, p1 = ops.conv2d_wgrad_wino(None, lazy, g, co, True, mo=4, planes=kept, dgrad_planes=True)
    assert torch.equal(dw0, dw1) and torch.equal(db0, db1) and torch.equal(p0, p1)
    # the bias gradient is the plain sum of the scattered gradient
    _close(db1, dy_full.double().sum((0, 1, 2)).float(), 1e-4, "dbias")
    with pytest.raises(ValueError):
        ops.wino_dy_transform(ops.PooledGrad(dpool, am, y, (n, h, w, co)), ops.make_geom(n, h + 4, w, ci, co, 3, 1, 1, 1), co)


NT_CASES = [  # n, h, w, ci, co: full 128-wide tiles, ragged rows (tiles % 128 != 0), fewer than 128 rows, column tiles cut (co % 128 != 0), K = 32
    (2, 38, 38, 256, 256), (1, 19, 19, 512, 128), (1, 10, 10, 256, 128), (2, 30, 26, 288, 100), (1, 75, 75, 128, 384), (3, 13, 21, 32, 160),
]


@pytest.mark.parametrize("case", NT_CASES)
def test_winograd_plane_gemm_128_tile_dma_kernel_equals_generic_kernel(case):
    """csrc/gemm_nt.hip (128 x 128 tile, LDS-DMA double buffer, XOR-swizzled unpadded rows, XCD-grouped block order) against the generic
    64 x 64 implicit-GEMM kernel on the plane GEMMs of the Winograd forward and data gradient.  Both walk K in the same order with the
    same MFMA, so the outputs must be EQUAL; both stay within 1e-4 of an f64 convolution.  Forced on (mode 1) it also takes shapes that
    the automatic rule leaves to the generic kernel: cut column tiles, fewer rows than a tile, K = 32."""
    from objectdetection_ssd_amd import _lib, ops
    lib = _lib.load()
    if not lib.ssd_has_experimental():
        assert lib.ssd_tune_set_gemm_nt(1) != 0 and lib.ssd_tune_set_gemm_nt(-1) == 0        # refused, not silently ignored
        pytest.skip("csrc/gemm_nt.hip is compiled only with SSD_EXPERIMENTAL=1 (off by default: it measured no better than the 64 x 64 kernel)")
    n, h, w, ci, co = case
    dev = _dev()
    full = (n, h, w, ci, co, 3, 1, 1, 1)
    x, wt, b = _conv_data(full, seed=201)
    x64 = x.double().requires_grad_(True)
    y64 = F.conv2d(x64, wt.double(), b.double(), padding=1)
    dy = torch.randn(y64.shape, generator=torch.Generator().manual_seed(202))
    y64.backward(dy.double())
    g = ops.make_geom(*full)
    ld = ops.pad32(co)
    uf, ub = ops.wino_weights(wt.to(dev), ld, mo=4)
    xd = _nhwc(x).to(dev)
    dy_p = torch.zeros(n, h, w, ld)
    dy_p[..., :co] = _nhwc(dy)
    dy_p = dy_p.to(dev)
    out = {}
    try:
        _lib.check(lib.ssd_tune_set_wino_fused(0), "tune")                 # K <= 128 would otherwise take the one-kernel form
        for mode in (0, 1):
            _lib.check(lib.ssd_tune_set_gemm_nt(mode), "tune")
            out[mode] = (ops.conv2d_fwd_wino(xd, uf, b.to(dev), g, True, ld=ld), ops.conv2d_dgrad_wino(dy_p, ub, g))
    finally:
        _lib.check(lib.ssd_tune_set_wino_fused(-1), "tune")
        _lib.check(lib.ssd_tune_set_gemm_nt(-1), "tune")
    assert torch.equal(out[0][0], out[1][0]), "forward differs between the two GEMM kernels"
    assert torch.equal(out[0][1], out[1][1]), "data gradient differs between the two GEMM kernels"
    _close(out[1][0][..., :co], _nhwc(F.relu(y64.detach())), what=f"fwd vs f64 {case}")
    _close(out[1][1], _nhwc(x64.grad), what=f"dgrad vs f64 {case}")


X3_GEMM_CASES = [  # M, K, N, planes: full tiles; ragged rows and a cut column tile; fewer rows than a tile, N just over one tile; deep K
    (384, 256, 256, 3), (1000, 512, 100, 2), (129, 256, 130, 1), (64, 1024, 512, 2), (3200, 512, 512, 1),
]


@pytest.mark.parametrize("mfma", [32, 16, 256])
@pytest.mark.parametrize("case", X3_GEMM_CASES)
def test_plane_gemm_from_three_bf16_limbs_is_as_exact_as_the_f32_mfma(case, mfma):
    """csrc/gemm_x3.hip: out[b] = a[b] * w[b]^T with every f32 operand split exactly into three bf16 limbs and six limb products per block
    on the bf16 MFMA (f32 accumulate).  Against an f64 product it must be no worse than the f32-MFMA kernel it replaces (the dropped
    limb products are <= 2^-24 relative), on operands with a wide dynamic range, and rows / columns outside M x N must stay untouched."""
    from objectdetection_ssd_amd import _lib
    lib = _lib.load()
    M, K, N, P = case
    dev = _dev()
    gen = torch.Generator().manual_seed(M + K + N)
    a = torch.randn(P, M, K, generator=gen) * torch.exp(2.0 * torch.randn(P, M, K, generator=gen))
    w = torch.randn(P, N, K, generator=gen) * torch.exp(2.0 * torch.randn(P, N, K, generator=gen)) / K ** 0.5
    a[0, 0, :8] = 0.0
    w[0, 0, 3] = 0.0
    ad, wd = a.to(dev), w.to(dev)
    st = torch.cuda.current_stream().cuda_stream
    w3 = torch.zeros(lib.ssd_gemm_x3_weights_bytes(N, K, P), dtype=torch.uint8, device=dev)
    assert w3.numel() == P * K * ((N + 127) // 128 * 128) * 6
    _lib.check(lib.ssd_gemm_x3_split_weights(wd.data_ptr(), w3.data_ptr(), N, K, P, st), "split")
    # the three limbs add up to the weight exactly
    lim = w3.view(torch.bfloat16).view(P, K // 16, 3, -1, 16).float()
    back = lim.sum(2)[:, :, :N, :].permute(0, 2, 1, 3).reshape(P, N, K)
    assert torch.equal(back.cpu(), w), "hi + mid + lo != w"
    assert float(lim[:, :, :, N:, :].abs().max()) == 0.0 if lim.shape[3] > N else True
    guard = 7.0
    o3 = torch.full((P, M, N), guard, device=dev)
    o32 = torch.full((P, M, N), guard, device=dev)
    # 16: two limb products per v_mfma_f32_16x16x32_bf16 (limbs concatenated along K); 256: the 256 x 256 ping-pong kernel forced on every shape
    # it takes (csrc/gemm_x3v2.hip) -- held bit-identical to the 128 x 128 kernel: same products in the same order per accumulator
    if mfma != 32 and not lib.ssd_has_experimental():
        pytest.skip("the 16x16x32 form and csrc/gemm_x3v2.hip are compiled only with SSD_EXPERIMENTAL=1 (off by default: no faster in the step)")
    _lib.check(lib.ssd_tune_set_x3_mfma(16 if mfma == 16 else 32), "tune")
    _lib.check(lib.ssd_tune_set_x3_big(2 if mfma == 256 else 0), "tune")
    try:
        _lib.check(lib.ssd_gemm_planes_x3(ad.data_ptr(), w3.data_ptr(), o3.data_ptr(), M, K, N, N, P, st), "x3")
        if mfma == 256:
            o_small = torch.full((P, M, N), guard, device=dev)
            _lib.check(lib.ssd_tune_set_x3_big(0), "tune")
            _lib.check(lib.ssd_gemm_planes_x3(ad.data_ptr(), w3.data_ptr(), o_small.data_ptr(), M, K, N, N, P, st), "x3")
            assert torch.equal(o3, o_small), "256 x 256 ping-pong kernel differs from the 128 x 128 kernel"
    finally:
        _lib.check(lib.ssd_tune_set_x3_mfma(32), "tune")
        _lib.check(lib.ssd_tune_set_x3_big(0), "tune")
    _lib.check(lib.ssd_gemm_planes_f32(ad.data_ptr(), wd.data_ptr(), o32.data_ptr(), M, K, N, N, P, st), "f32")
    ref = torch.bmm(a.double(), w.double().transpose(1, 2))
    scale = torch.bmm(a.double().abs(), w.double().abs().transpose(1, 2))            # sum_k |a||w|: what an f32 sum's error is relative to
    e3 = float(((o3.cpu().double() - ref).abs() / scale.clamp_min(1e-300)).max())
    e32 = float(((o32.cpu().double() - ref).abs() / scale.clamp_min(1e-300)).max())
    r3 = float((o3.cpu().double() - ref).norm() / ref.norm())
    r32 = float((o32.cpu().double() - ref).norm() / ref.norm())
    # measured: 2.4e-7 ... 3.7e-7 against 2.6e-7 ... 4.1e-7 of the f32 MFMA (which rounds once per product block, the limb form six times)
    assert r3 <= 1e-6 and r3 <= 1.5 * r32 + 1e-9, (r3, r32)
    assert e3 <= 2.0 * e32 + 1e-9 and e3 <= 2e-6, (e3, e32)


def test_limb_gemm_non_finite_operands():
    """csrc/gemm_x3.hip header, "Non-finite and out-of-range operands": an activation row holding +inf, or a finite value above the
    largest bf16 (3.3895e38), makes every output of that ROW NaN in the limb form (inf - inf in the split) where the f32 MFMA returns
    +-inf / a finite overflow; NaN stays NaN; all other rows are bit-identical to the run without the special values -- the limb form
    never turns a non-finite result into a finite one and never spreads it."""
    from objectdetection_ssd_amd import _lib
    lib = _lib.load()
    M, K, N, P = 256, 256, 128, 2
    dev = _dev()
    gen = torch.Generator().manual_seed(77)
    a = torch.randn(P, M, K, generator=gen)
    w = torch.randn(P, N, K, generator=gen) / K ** 0.5
    st = torch.cuda.current_stream().cuda_stream
    wd = w.to(dev)
    w3 = torch.zeros(lib.ssd_gemm_x3_weights_bytes(N, K, P), dtype=torch.uint8, device=dev)
    _lib.check(lib.ssd_gemm_x3_split_weights(wd.data_ptr(), w3.data_ptr(), N, K, P, st), "split")

    def run(act):
        ad = act.to(dev)
        o3 = torch.zeros((P, M, N), device=dev)
        o32 = torch.zeros((P, M, N), device=dev)
        _lib.check(lib.ssd_gemm_planes_x3(ad.data_ptr(), w3.data_ptr(), o3.data_ptr(), M, K, N, N, P, st), "x3")
        _lib.check(lib.ssd_gemm_planes_f32(ad.data_ptr(), wd.data_ptr(), o32.data_ptr(), M, K, N, N, P, st), "f32")
        return o3.cpu(), o32.cpu()

    clean3, clean32 = run(a)
    assert torch.isfinite(clean3).all() and torch.isfinite(clean32).all()
    b = a.clone()
    b[0, 5, 17] = float("inf")
    b[0, 40, 3] = 3.40e38                 # finite in f32, above the bf16 range
    b[1, 100, 200] = float("nan")
    b[1, 7, 9] = -float("inf")
    hot3, hot32 = run(b)
    rows = {(0, 5), (0, 40), (1, 100), (1, 7)}
    for p_ in range(P):
        for m in range(M):
            if (p_, m) in rows:
                assert not torch.isfinite(hot3[p_, m]).any(), (p_, m)          # limb form: the whole row is NaN
                assert torch.isnan(hot3[p_, m]).all(), (p_, m)
                assert not torch.isfinite(hot32[p_, m]).any() or (p_, m) == (0, 40)   # f32 MFMA: inf (or NaN); the 3.4e38 row may stay finite
    keep = torch.ones(P, M, dtype=torch.bool)
    for p_, m in rows:
        keep[p_, m] = False
    assert torch.equal(hot3[keep], clean3[keep]) and torch.equal(hot32[keep], clean32[keep])
    assert torch.isinf(hot32[0, 5]).all() and torch.isinf(hot32[1, 7]).all() and torch.isnan(hot32[1, 100]).all()


X31_CASES = [  # n, h, w, ci, co: fc7's shape class at a small batch, ragged pixel count, Co not a multiple of 128, deep reduction
    (2, 19, 19, 1024, 1024), (3, 13, 7, 256, 160), (1, 38, 38, 512, 128), (2, 10, 10, 2048, 256),
]


@pytest.mark.parametrize("case", X31_CASES)
def test_conv1x1_on_the_three_limb_gemms_forward_dgrad_wgrad(case):
    """1x1 / stride-1 convolutions through csrc/gemm_x3.hip (fc7, seq8.0 in the f32 step): forward with bias + ReLU, data gradient with
    accumulate + ReLU mask, weight and bias gradient (split-K TN GEMM + slice sum + column sums) against an f64 convolution: within
    FIXED 1e-6 (forward, data gradient) / 2e-6 (weight gradient) relative L2 (measured 4.7e-7 at K = 1024, 6.4e-7 at K = 2048, where
    the exact-f32 MFMA kernels read 1.6e-7: 6 x K/16 accumulating MFMAs per output instead of K/2 exact FMA steps -- both are f32
    summation-order noise, reported side by side in the assertion message); filter limbs from the per-layer split and from the kind-4 weight job must be the same bits."""
    from objectdetection_ssd_amd import ops
    n, h, w, ci, co = case
    dev = _dev()
    full = (n, h, w, ci, co, 1, 1, 0, 1)
    x, wt, b = _conv_data(full, seed=311)
    x = torch.relu(x)
    x64 = x.double().requires_grad_(True)
    w64 = wt.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True)
    y64 = F.relu(F.conv2d(x64, w64, b64))
    dy = torch.randn(y64.shape, generator=torch.Generator().manual_seed(312)) * (y64.detach() > 0)
    y64.backward(dy.double())
    g = ops.make_geom(*full)
    xd, dyd, wd, bd = _nhwc(x).to(dev), _nhwc(dy).to(dev), wt.to(dev), b.to(dev)
    w3f, w3b = ops.conv1x1_weights_x3(wd, co)
    wf_j, wb_j = ops.x3_filter_alloc(co, ci, dev), ops.x3_filter_alloc(ci, co, dev)
    ops.WeightTable([dict(kind=4, w0=wd, co0=co, co=co, ci=ci, taps=1, co_pad=co, out_fwd=wf_j, out_bwd=wb_j)], dev).run()
    assert torch.equal(wf_j.view(torch.int16), w3f.view(torch.int16)) and torch.equal(wb_j.view(torch.int16), w3b.view(torch.int16))
    rel = lambda a, r: float((a.double().cpu() - r).norm() / r.norm())
    # forward
    y3 = ops.conv1x1_fwd_x3(xd, w3f, bd, g, True)
    y32 = ops.conv2d_fwd(xd, ops.weight_ohwi(wd), bd, g, True)
    ref = _nhwc(y64.detach())
    assert rel(y3, ref) <= 1e-6, (rel(y3, ref), rel(y32, ref))
    assert torch.equal(y3 > 0, y32 > 0) or float(((y3 > 0) != (y32 > 0)).float().mean()) < 1e-5
    # data gradient: += into an existing tensor, masked by the layer input's ReLU
    prev = torch.randn(n, h, w, ci, generator=torch.Generator().manual_seed(313)).to(dev)
    dx3 = ops.conv1x1_dgrad_x3(dyd, w3b, g, dx=prev.clone(), relu_mask=xd, accumulate=True)
    dx32 = ops.conv2d_dgrad(dyd, ops.weight_ihwo(wd), g, prev.clone(), xd, True)
    refdx = (prev.cpu().double() + _nhwc(x64.grad)) * (_nhwc(x) > 0)
    assert rel(dx3, refdx) <= 1e-6, (rel(dx3, refdx), rel(dx32, refdx))
    dx3n = ops.conv1x1_dgrad_x3(dyd, w3b, g)
    assert rel(dx3n, _nhwc(x64.grad)) <= 1e-6
    # weight and bias gradient
    dw3, db3 = ops.conv1x1_wgrad_x3(xd, dyd, g, co, True)
    dw32, db32 = ops.conv2d_wgrad(xd, dyd, g, co, True)
    assert rel(dw3, w64.grad) <= 2e-6, (rel(dw3, w64.grad), rel(dw32, w64.grad))
    assert rel(db3, b64.grad) <= 2e-6, rel(db3, b64.grad)
    dwa, dba = ops.conv1x1_wgrad_x3(xd, dyd, g, co, True)
    assert torch.equal(dwa, dw3) and torch.equal(dba, db3)               # fixed slice order: reproducible


DILATED_CASES = [  # n, h, w, ci, co, dilation: fc6's shape class, odd maps, lattices of unequal size, a map smaller than the dilation
    (2, 19, 19, 64, 96, 4), (1, 10, 13, 32, 64, 2), (2, 7, 5, 32, 32, 3), (1, 3, 3, 32, 32, 4), (1, 23, 17, 32, 40, 4), (32, 19, 19, 512, 1024, 4),
]


@pytest.mark.parametrize("case", DILATED_CASES)
def test_winograd_on_the_sub_lattices_of_a_dilated_conv(case):
    """fc6 is Conv2d(512, 1024, 3, padding=4, dilation=4) (Model.py:149): d x d independent 3x3 / pad-1 convolutions on the sub-lattices
    x[d*i + a][d*j + b].  The F(4x4) transforms walk those lattices (csrc/winograd.hip, struct Lat); GEMMs and filters are the plain
    ones.  Forward (+ bias, ReLU, kept planes), data gradient (from dy, from the planes of the shared dy pass, with mask / accumulate)
    and weight + bias gradient against an f64 dilated convolution, 1e-4 of the scale; the split entry points equal the combined one."""
    from objectdetection_ssd_amd import ops
    n, h, w, ci, co, d = case
    dev = _dev()
    full = (n, h, w, ci, co, 3, 1, d, d)
    big = n >= 32
    if big:
        gen = torch.Generator(device=dev).manual_seed(7)
        x = torch.randn(n, ci, h, w, device=dev, generator=gen)
        wt = torch.randn(co, ci, 3, 3, device=dev, generator=gen) * (2.0 / (ci * 9)) ** 0.5
        b = torch.randn(co, device=dev, generator=gen) * 0.1
        dy = torch.randn(n, co, h, w, device=dev, generator=gen)
        # reference on the GPU through torch's own (MIOpen / rocBLAS) f64 convolution would not be ours either; use f32 CPU autograd in f64 on a slice
        xs, dys = x[:2].cpu(), dy[:2].cpu()
    else:
        x, wt, b = _conv_data(full, seed=301)
        dy = torch.randn(n, co, h, w, generator=torch.Generator().manual_seed(302))
        xs, dys = x, dy
    x64 = xs.double().requires_grad_(True)
    w64 = wt.cpu().double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, b.cpu().double(), padding=d, dilation=d)
    y64.backward(dys.double())
    g = ops.make_geom(*full)
    ld = ops.pad32(co)
    assert ops.wino_tiles(g) == n * sum((((h - a + d - 1) // d) + 3) // 4 for a in range(min(d, h))) * sum((((w - a + d - 1) // d) + 3) // 4 for a in range(min(d, w)))
    uf, ub = ops.wino_weights(wt.to(dev), ld, mo=4)
    xd = _nhwc(x).to(dev)
    dy_p = torch.zeros(n, h, w, ld, device=dev)
    dy_p[..., :co] = _nhwc(dy).to(dev)
    y, planes = ops.conv2d_fwd_wino(xd, uf, b.to(dev), g, True, ld=ld, keep_planes=True)
    k = xs.shape[0]
    _close(y[:k, ..., :co], _nhwc(F.relu(y64.detach())), what=f"dilated fwd {case}")
    if ld != co:
        assert float(y[..., co:].abs().max()) == 0.0
    y2 = ops.conv2d_fwd_wino(xd, uf, b.to(dev), g, False, ld=ld)
    _close(y2[:k, ..., :co], _nhwc(y64.detach()), what=f"dilated fwd no relu {case}")
    dx = ops.conv2d_dgrad_wino(dy_p, ub, g)
    _close(dx[:k], _nhwc(x64.grad), what=f"dilated dgrad {case}")
    dw, db, dyp = ops.conv2d_wgrad_wino(None, dy_p, g, ld, True, mo=4, planes=planes, dgrad_planes=True)
    dx2 = ops.conv2d_dgrad_wino(None, ub, g, planes=dyp)
    _close(dx2, dx, tol=2e-5, what=f"dilated dgrad from the dy pass planes {case}")
    prev = torch.randn(n, h, w, ci, device=dev)
    dx3 = ops.conv2d_dgrad_wino(None, ub, g, dx=prev.clone(), relu_mask=xd, accumulate=True, planes=dyp)
    assert torch.equal(dx3, torch.where(xd > 0, dx2 + prev, torch.zeros_like(dx2)))
    if not big:
        _close(dw, w64.grad, what=f"dilated wgrad {case}")
        _close(db, dy.double().sum((0, 2, 3)), what=f"dilated bias grad {case}")
    else:
        # adjoint identity at the bench size: <dw, w> = <dy, conv(x, w)> (no bias), in f64 dots of the f32 results
        y_nb = ops.conv2d_fwd_wino(xd, uf, None, g, False, ld=ld)
        lhs = float((dw.double() * wt.double()).sum())
        rhs = float((y_nb.double() * dy_p.double()).sum())
        assert abs(lhs - rhs) <= 1e-4 * max(abs(rhs), float(wt.double().norm() * dw.double().abs().max())), (lhs, rhs)
        _close(db, dy.double().sum((0, 2, 3)), what=f"dilated bias grad {case}")
    # the two-call form of the weight gradient gives the same bits
    Y, Vd, part = ops.wino_dy_transform(dy_p, g, ld, True, True)
    dw2, db2 = ops.wino_wgrad_gemm(Y, planes, part, g, ld)
    assert torch.equal(dw2, dw) and torch.equal(db2, db) and torch.equal(Vd, dyp)
    # what the dilated form must refuse: the fused pool, F(2x2)
    with pytest.raises((ValueError, RuntimeError)):
        ops.conv2d_fwd_wino_pool(xd, uf, b.to(dev), g, False)


@pytest.mark.parametrize("shape", [(2, 300, 300), (1, 37, 70), (3, 5, 5), (1, 64, 64), (2, 4, 129)])
def test_conv1_1_forward_in_one_kernel(shape):
    """csrc/conv_first.hip: Conv2d(3, 64, 3, padding=1) + ReLU (Model.py:135, features[0:2]) straight from the NCHW batch -- halo tile in
    LDS, K = 27 on the MFMA -- against an f64 convolution (1e-4 of the scale), against the im2col + 1x1-convolution form it replaces
    (2e-5: same sum, another order), and its optional [pixel][32] rows against `im2col_first` bit for bit; tiles cut by the map's edges,
    maps smaller than a tile, a NaN in the input (must reach exactly the outputs whose window holds it)."""
    from objectdetection_ssd_amd import ops
    n, h, w = shape
    dev = _dev()
    g_ = torch.Generator().manual_seed(h * 7 + w)
    x = torch.randn(n, 3, h, w, generator=g_)
    wt = torch.randn(64, 3, 3, 3, generator=g_) * (2.0 / 27) ** 0.5
    b = torch.randn(64, generator=g_) * 0.1
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    rows = ops.first_weight_rows(wt.to(dev))
    xd = x.to(dev)
    for relu in (True, False):
        y, col = ops.conv1_first_fwd(xd, rows, b.to(dev), relu, want_col=True)
        _close(y, _nhwc(F.relu(ref) if relu else ref), what=f"conv1_1 fused relu={relu} {shape}")
        assert torch.equal(col, ops.im2col_first(xd))
        y2, none = ops.conv1_first_fwd(xd, rows, b.to(dev), relu)
        assert none is None and torch.equal(y2, y)
    g = ops.make_geom(n, h, w, 32, 64, 1, 1, 0, 1)
    old = ops.conv2d_fwd(ops.im2col_first(xd), rows, b.to(dev), g, True)
    _close(y2 * 0 + ops.conv1_first_fwd(xd, rows, b.to(dev), True)[0], old, tol=2e-5, what=f"fused vs im2col form {shape}")
    ynb, _ = ops.conv1_first_fwd(xd, rows, None, False)
    _close(ynb, _nhwc(ref - b.double().view(1, -1, 1, 1)), what="no bias")
    if h >= 5 and w >= 5:
        xn = xd.clone()
        xn[0, 1, 2, 3] = float("nan")
        yn, _ = ops.conv1_first_fwd(xn, rows, b.to(dev), False)
        bad = torch.isnan(yn[0]).any(dim=2)
        want = torch.zeros(h, w, dtype=torch.bool, device=dev)
        want[1:4, 2:5] = True
        assert torch.equal(bad, want) and not torch.isnan(yn[1:]).any()


@pytest.mark.parametrize("shape", [(2, 300, 300), (1, 37, 70), (3, 5, 5), (2, 4, 129), (32, 300, 300)])
def test_conv1_1_weight_gradient_from_the_nchw_input(shape):
    """csrc/conv_first.hip: dw / db of Conv2d(3, 64, 3, padding=1) (autograd of Model.py:135 features[0]) with the reduction over the
    pixels on the MFMA, the [pixel][27] operand read from a halo image of x in LDS (never in memory), the bias gradient as a column of
    ones.  Against f64 autograd (1e-4 of the scale; at batch 32 against the im2col + generic weight-gradient form, 2e-5) and bit-stable
    from run to run (persistent workgroups, partial sums added in index order)."""
    from objectdetection_ssd_amd import ops
    n, h, w = shape
    dev = _dev()
    gen = torch.Generator(device=dev).manual_seed(h + w)
    x = torch.randn(n, 3, h, w, device=dev, generator=gen)
    dy = torch.randn(n, h, w, 64, device=dev, generator=gen)
    dw, db = ops.conv1_first_wgrad(x, dy, True)
    dw2, db2 = ops.conv1_first_wgrad(x, dy, True)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    assert float(dw.reshape(64, 32)[:, 27:].abs().max()) == 0.0
    g = ops.make_geom(n, h, w, 32, 64, 1, 1, 0, 1)
    dw_old, db_old = ops.conv2d_wgrad(ops.im2col_first(x), dy, g, 64, True)
    _close(dw, dw_old, tol=2e-5, what=f"dw vs im2col form {shape}")
    _close(db, db_old, tol=2e-5, what=f"db vs im2col form {shape}")
    if n <= 3:
        wt = torch.zeros(64, 3, 3, 3, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(x.cpu().double(), wt, torch.zeros(64, dtype=torch.float64), padding=1)
        y.backward(_nchw(dy.cpu().double()))
        _close(ops.first_weight_grad(dw), wt.grad, what=f"dw vs f64 {shape}")
        _close(db, dy.cpu().double().sum((0, 1, 2)), what=f"db vs f64 {shape}")
    assert ops.conv1_first_wgrad(x, dy, False)[1] is None


STRIDE2_CASES = [  # n, h, w, ci, co, k, pad: seq8.2 / seq9.2 shapes, even and odd maps, 1x1 / stride 2, pad 0, more images than a tile holds
    (2, 19, 19, 64, 96, 3, 1), (3, 10, 10, 32, 64, 3, 1), (1, 8, 13, 32, 32, 3, 1), (2, 7, 6, 32, 40, 1, 0), (2, 9, 9, 32, 32, 3, 0),
    (32, 19, 19, 256, 512, 3, 1),
]


@pytest.mark.parametrize("case", STRIDE2_CASES)
def test_stride2_data_gradient_grouped_by_pixel_parity(case):
    """dx of a stride-2 convolution (seq8.2 / seq9.2, Model.py:163-164): a dx pixel of parity (a, b) only receives the filter taps of one
    parity -- 1, 2, 2 or 4 of the 9 -- while the plain implicit-GEMM kernel walks all nine and multiplies zeros for the rest.  The
    PARITY instantiation groups the rows by class so that a block skips those taps.  Skipped products are exact zeros and the remaining
    ones are added in the same order, so the result must EQUAL the plain kernel's (mask and accumulate included), and match autograd."""
    from objectdetection_ssd_amd import _lib, ops
    lib = _lib.load()
    n, h, w, ci, co, k, pad = case
    dev = _dev()
    gen = torch.Generator(device=dev).manual_seed(h * 31 + w)
    g = ops.make_geom(n, h, w, ci, co, k, 2, pad, 1)
    ld = ops.pad32(co)
    wt = torch.randn(co, ci, k, k, device=dev, generator=gen) * (2.0 / (ci * k * k)) ** 0.5
    dy = torch.zeros(n, g.Ho, g.Wo, ld, device=dev)
    dy[..., :co] = torch.randn(n, g.Ho, g.Wo, co, device=dev, generator=gen)
    wb = ops.weight_ihwo(wt, ld)
    prev = torch.randn(n, h, w, ci, device=dev, generator=gen)
    mask = torch.randn(n, h, w, ci, device=dev, generator=gen).clamp_min(0)
    out = {}
    try:
        _lib.check(lib.ssd_tune_set_igemm_splitk(1), "tune")               # the plain kernel without K slices: one sum order for both
        for mode in (0, 1):
            _lib.check(lib.ssd_tune_set_dgrad_parity(mode), "tune")
            out[mode] = (ops.conv2d_dgrad(dy, wb, g), ops.conv2d_dgrad(dy, wb, g, dx=prev.clone(), relu_mask=mask, accumulate=True))
    finally:
        _lib.check(lib.ssd_tune_set_dgrad_parity(1), "tune")
        _lib.check(lib.ssd_tune_set_igemm_splitk(-1), "tune")
    assert torch.equal(out[0][0], out[1][0]), "plain vs parity-grouped data gradient"
    assert torch.equal(out[0][1], out[1][1]), "with accumulate + mask"
    assert torch.equal(out[1][1], torch.where(mask > 0, out[1][0] + prev, torch.zeros_like(prev)))
    k_ = min(n, 2)
    x64 = torch.zeros(k_, ci, h, w, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x64, wt.cpu().double(), None, stride=2, padding=pad)
    y.backward(_nchw(dy[:k_, ..., :co].cpu().double()))
    _close(out[1][0][:k_], _nhwc(x64.grad), what=f"stride-2 dgrad vs f64 {case}")


# ---- bf16 TENSORS (round 3): csrc/conv_bf16.hip ------------------------------------------------------------------------------
def _bf16_ulps(got, ref_f32):
    """distance of a bf16 result from the f32 reference in units of the reference's bf16 spacing (2^-8 relative)"""
    g, r = got.detach().cpu().float(), ref_f32.detach().cpu().float()
    return (g - r).abs() / (r.abs().clamp_min(1e-3) * 2.0 ** -8)


HALO_BF16_CASES = [
    # mode, bn, N, H, W, K (in channels), n_out, w_rows
    (0, 128, 2, 11, 45, 128, 128, 128),      # 8x32 patches, ragged in both directions
    (0, 128, 1, 17, 33, 64, 160, 150),       # second N tile half empty, zero weight rows (a head: 150 -> pad 160)
    (0, 64, 2, 19, 40, 64, 64, 64),          # K = 64, 64 channels: the persistent kernel (filter resident in LDS, two halo buffers)
    (0, 64, 4, 70, 260, 64, 64, 64),         # ... 324 patches for 256 workgroups: the patch loop runs twice for some
    (0, 64, 1, 20, 70, 64, 48, 48),          # ... fewer filter rows than the tile
    (0, 64, 1, 33, 35, 128, 64, 64),         # ... two chunks: the halo is re-loaded between them
    (1, 128, 2, 19, 23, 192, 256, 256),      # 16x16 patches, three chunks
    (1, 64, 1, 16, 16, 64, 64, 64),
    (2, 128, 3, 13, 11, 128, 128, 128),      # flat: tiles span images
    (2, 128, 2, 38, 38, 64, 128, 100),       # flat at a real map size, head-like rows
    (2, 64, 2, 19, 19, 128, 64, 64),
    (-1, -1, 1, 75, 75, 64, 128, 128),       # automatic choice: flat with the 7-piece halo buffers (two of them + the weight ring = 160 KB)
    (-1, -1, 2, 75, 75, 256, 256, 256),      # ... four chunks, tiles spanning the two images
    (2, 128, 1, 9, 94, 64, 128, 128),        # ... the widest map the flat form takes
    (-1, -1, 1, 75, 75, 64, 64, 64),         # 64 output channels at 75 x 75: 16x16 patches (the 64-channel flat form has the 6-piece buffers)
    (1, 128, 1, 75, 75, 64, 128, 128),       # 16x16 patches forced at 75 x 75 (what rounds 1-3 ran)
    (-1, -1, 1, 10, 10, 64, 128, 128),       # automatic: flat
]


@pytest.mark.parametrize("mfma", [16, 32])
@pytest.mark.parametrize("case", HALO_BF16_CASES)
def test_conv3x3_on_bf16_tensors_forward_and_data_gradient(case, mfma):
    """3x3 / s1 / p1 convolution with activations, gradients and weights in bf16 (f32 accumulate), every position space and both
    N tiles: forward (bias + ReLU, bf16 out; f32 out as the heads take it) and data gradient (mirrored taps, += an existing dx,
    ReLU mask from the bf16 activation) against torch-CPU f32 convolutions of the SAME bf16 operands.  f32 outputs within 2e-5
    of the scale; bf16 outputs are the f32 result rounded once: within one bf16 spacing of the reference everywhere (a sum that
    lands near a rounding boundary may go either way) and exactly the rounded reference in >= 99 % of the elements."""
    from objectdetection_ssd_amd import _lib, ops
    mode, bn, n, h, w, k, n_out, rows = case
    dev = _dev()
    g = torch.Generator().manual_seed(hash(case) % 10000)
    x = torch.randn(n, h, w, k, generator=g).bfloat16()
    wt = (torch.randn(rows, 9, k, generator=g) * (2.0 / (9 * k)) ** 0.5).bfloat16()
    bias = torch.zeros(n_out)
    bias[:rows] = torch.randn(rows, generator=g)
    _lib.check(_lib.load().ssd_tune_set_conv_bf16(mode, bn), "tune")
    _lib.check(_lib.load().ssd_tune_set_conv_bf16_mfma(mfma), "tune")          # both MFMA shapes (32x32x16 is the default)
    try:
        w_nchw = torch.zeros(n_out, k, 3, 3)
        w_nchw[:rows] = wt.float().reshape(rows, 3, 3, k).permute(0, 3, 1, 2)
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), w_nchw, bias, padding=1)
        y32 = ops.conv3x3_bf16(x.to(dev), wt.to(dev), bias.to(dev), n_out, relu=False, out_f32=True)
        _close(_nchw(y32), ref, 2e-5, "forward, f32 out")
        y = ops.conv3x3_bf16(x.to(dev), wt.to(dev), bias.to(dev), n_out, relu=True)
        assert y.dtype == torch.bfloat16
        u = _bf16_ulps(_nchw(y), ref.relu())
        assert float(u.max()) <= 1.5, float(u.max())          # (the spacing doubles at a power of two: 1 spacing of the larger side)
        assert float((_nchw(y).cpu() == ref.relu().bfloat16()).float().mean()) >= 0.99
        # wider output rows: columns beyond n_out stay untouched
        ld = n_out + 32
        canvas = torch.full((n, h, w, ld), 7.0, dtype=torch.bfloat16, device=dev)
        ops.conv3x3_bf16(x.to(dev), wt.to(dev), bias.to(dev), n_out, relu=True, out=canvas, ldo=ld)
        assert torch.equal(canvas[..., :n_out], y) and bool((canvas[..., n_out:] == 7.0).all())
        # data gradient of a convolution with `rows` input channels... here: dy has k channels, dx has n_out; weights [n_out rows][9][k]
        dy = x
        prev = torch.randn(n, h, w, n_out, generator=g).bfloat16()
        act = torch.randn(n, h, w, n_out, generator=g).relu().bfloat16()
        w_t = torch.zeros(k, n_out, 3, 3)
        w_t[:, :rows] = wt.float().reshape(rows, 3, 3, k).permute(3, 0, 1, 2)            # conv_transpose2d weight: (in = k, out = n_out, r, s)
        dref = F.conv_transpose2d(dy.float().permute(0, 3, 1, 2), w_t, padding=1)
        dref = (dref + prev.float().permute(0, 3, 1, 2)) * (act.float().permute(0, 3, 1, 2) > 0)
        dx = prev.clone().to(dev)
        ops.conv3x3_bf16(dy.to(dev), wt.to(dev), None, n_out, relu=False, flip=True, out=dx, relu_mask=act.to(dev), accumulate=True)
        u = _bf16_ulps(_nchw(dx), dref)
        assert float(u.max()) <= 1.5, float(u.max())          # (the spacing doubles at a power of two: 1 spacing of the larger side)
        assert float((_nchw(dx).cpu() == dref.bfloat16()).float().mean()) >= 0.99
        assert bool((dx[act.to(dev) <= 0] == 0).all())
    finally:
        _lib.check(_lib.load().ssd_tune_set_conv_bf16(-1, -1), "tune")
        _lib.check(_lib.load().ssd_tune_set_conv_bf16_mfma(32), "tune")


@pytest.mark.parametrize("k,s,p,ceil,hw", POOLS)
def test_maxpool_on_bf16_tensors(k, s, p, ceil, hw):
    """bf16 pools: values and arg-max codes equal the f32 kernel's on the same (bf16-representable) input, the three backward forms
    (plain + ReLU mask + accumulate, gated by the pooled output, the 2x2 scatter) equal the f32 kernel's result rounded once."""
    from objectdetection_ssd_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(k * 100 + hw)
    x = torch.randn(2, hw, hw + 3, 16, generator=g).relu().bfloat16()
    y32, am32 = ops.maxpool_fwd(x.float().to(dev), k, s, p, ceil)
    y16, am16 = ops.maxpool_fwd(x.to(dev), k, s, p, ceil)
    assert y16.dtype == torch.bfloat16 and torch.equal(y16.float(), y32) and torch.equal(am16, am32)
    dy = torch.randn(y32.shape, generator=g).bfloat16()
    prev = torch.randn(x.shape, generator=g).bfloat16()
    ref = ops.maxpool_bwd(dy.float().to(dev), am32, tuple(x.shape), k, s, p, prev.float().to(dev).clone(), x.float().to(dev), True)
    got = ops.maxpool_bwd(dy.to(dev), am16, tuple(x.shape), k, s, p, prev.to(dev).clone(), x.to(dev), True)
    assert got.dtype == torch.bfloat16 and torch.equal(got, ref.bfloat16())
    ref = ops.maxpool_bwd(dy.float().to(dev), am32, tuple(x.shape), k, s, p, y_gate=y32)
    got = ops.maxpool_bwd(dy.to(dev), am16, tuple(x.shape), k, s, p, y_gate=y16)
    assert torch.equal(got, ref.bfloat16())


def test_l2norm_gather_casts_on_bf16_tensors():
    from objectdetection_ssd_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 7, 5, 512, generator=g).bfloat16()
    gamma = (20 + torch.randn(512, generator=g)).to(dev)
    y = ops.l2norm_fwd(x.to(dev), gamma)
    ref = ops.l2norm_fwd(x.float().to(dev), gamma)
    assert y.dtype == torch.bfloat16 and float((y.float() - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max())
    dy = torch.randn(x.shape, generator=g).bfloat16()
    dx, dg = ops.l2norm_bwd(x.to(dev), gamma, dy.to(dev))
    rdx, rdg = ops.l2norm_bwd(x.float().to(dev), gamma, dy.float().to(dev))
    assert dx.dtype == torch.bfloat16 and float((dx.float() - rdx).abs().max()) <= 2.0 ** -8 * float(rdx.abs().max())
    _close(dg, rdg, 1e-5, "dgamma")
    # gather: bf16 packed rows = the f32 kernel's, rounded; pad columns zero
    n, hw, a, P = 2, 9, 4, 60
    dloc, dconf = torch.randn(n, P, 4, generator=g).to(dev), torch.randn(n, P, 21, generator=g).to(dev)
    p16 = ops.heads_gather_bf16(dloc, dconf, 128, n, hw, a, 12)
    p32 = ops.heads_gather(dloc, dconf, 128, n, hw, a, 12)
    assert torch.equal(p16, p32.bfloat16()) and bool((p16[:, 100:] == 0).all())
    t = torch.randn(64, 24, generator=g).to(dev)
    assert torch.equal(ops.cast_bf16(t), t.bfloat16()) and torch.equal(ops.cast_f32(t.bfloat16()), t.bfloat16().float())


def test_conv1_1_and_nine_tap_weight_gradient_on_bf16_tensors():
    """conv1_1 with bf16-rounded operands and bf16 NHWC output + its weight gradient from a bf16 dy; the fused nine-tap weight gradient
    fed bf16 x / dy equals the same kernel fed the f32 copies of those values (identical products, identical order)."""
    from objectdetection_ssd_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 3, 21, 70, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g) * 0.2
    b = torch.randn(64, generator=g)
    rows = ops.first_weight_rows(w.to(dev))
    y = ops.conv1_first_fwd_bf16(x.to(dev), rows, b.to(dev), True)
    ref = F.conv2d(x.bfloat16().float(), w.bfloat16().float(), b, padding=1).relu()
    u = _bf16_ulps(_nchw(y), ref)
    assert y.dtype == torch.bfloat16 and float(u.max()) <= 1.5 and float((_nchw(y).cpu() == ref.bfloat16()).float().mean()) >= 0.99
    dy = torch.randn(2, 21, 70, 64, generator=g).bfloat16()
    dw16, db16 = ops.conv1_first_wgrad_bf16(x.to(dev), dy.to(dev), True)
    dw32, db32 = ops.conv1_first_wgrad(x.to(dev), dy.float().to(dev), True)
    assert torch.equal(dw16, dw32) and torch.equal(db16, db32)
    for (n, h, w_, ci, co, ldy) in ((2, 19, 23, 64, 128, 128), (1, 38, 38, 128, 100, 128), (3, 9, 12, 256, 64, 64)):
        geom = ops.make_geom(n, h, w_, ci, co, 3, 1, 1, 1)
        xa = torch.randn(n, h, w_, ci, generator=g).bfloat16()
        da = torch.randn(n, h, w_, ldy, generator=g).bfloat16()
        da[..., co:] = 0
        dwa, dba = ops.conv3x3_wgrad_bf16t(xa.to(dev), da.to(dev), geom, ldy, True)
        dwb, dbb = ops.conv2d_wgrad(xa.float().to(dev), da.float().to(dev), geom, ldy, True, bf16=True)
        assert torch.equal(dwa, dwb) and torch.equal(dba, dbb)


# ---- MultiBox loss: the three-launch form (wide per-prior kernel, one workgroup per image, wide gradients) against the four-launch form of rounds 1-3 --------------------
# (Losses.py:119-199.)  obj / cls / the selected hard negatives must be IDENTICAL; the sums differ by their order of addition only.
LOSS_FORM_CASES = [
    # bs, P, C, boxes per image (min, max), what
    (32, 8732, 21, 1, 8, "SSD300 at batch 32"),
    (3, 8732, 21, 9, 40, "more boxes than one pass of the per-box arg-max takes"),
    (2, 24564, 21, 1, 6, "SSD512 prior count"),
    (2, 1000, 64, 1, 3, "64 classes, a last chunk of 40 priors"),
    (1, 36000, 64, 2, 2, "the largest image the entry takes: 144 KB of values in LDS"),
    (4, 777, 5, 1, 2, "ragged everything"),
]


def _loss_inputs(bs, P, C, nmin, nmax, seed):
    g = torch.Generator().manual_seed(seed)
    cxcy = torch.rand(P, 2, generator=g)
    wh = torch.rand(P, 2, generator=g) * 0.5 + 0.03
    pri = torch.cat([cxcy, wh], 1)
    pri_xyxy = torch.cat([cxcy - wh / 2, cxcy + wh / 2], 1)
    boxes, classes, start = [], [], [0]
    for i in range(bs):
        n = int(torch.randint(nmin, nmax + 1, (1,), generator=g))
        c = torch.rand(n, 2, generator=g) * 0.6 + 0.2
        s = torch.rand(n, 2, generator=g) * 0.4 + 0.05
        b = torch.cat([c - s / 2, c + s / 2], 1)
        if n >= 2:
            b[1] = b[0]                                   # two identical boxes: ties in both arg-max directions
        if i == 0:
            b[0] = pri_xyxy[5]                            # a box that IS a prior (IoU exactly 1)
        boxes.append(b)
        classes.append(torch.randint(0, C - 1, (n,), generator=g).float())
        start.append(start[-1] + n)
    loc = torch.randn(bs, P, 4, generator=g)
    conf = torch.randn(bs, P, C, generator=g) * 2.0
    n_eq = conf[:, 1::7].shape[1]
    conf[:, ::7][:, :n_eq] = conf[:, 1::7]                             # equal rows -> equal CE values: the tie rule of the selection
    return loc, conf, torch.cat(boxes), torch.cat(classes), torch.tensor(start, dtype=torch.int32), pri, pri_xyxy


@pytest.mark.parametrize("norm_mode", [0, 1])
@pytest.mark.parametrize("case", LOSS_FORM_CASES, ids=[c[-1] for c in LOSS_FORM_CASES])
def test_multibox_loss_three_launch_form_equals_the_four_launch_form(case, norm_mode):
    from objectdetection_ssd_amd import _lib, ops
    bs, P, C, nmin, nmax, _ = case
    dev = _dev()
    args = [t.to(dev) for t in _loss_inputs(bs, P, C, nmin, nmax, seed=bs * 1000 + C)]
    lib = _lib.load()
    outs = {}
    try:
        for form in (0, 1):
            _lib.check(lib.ssd_tune_set_loss_form(form), "tune")
            o = ops.multibox_loss(*args, iou_threshold=0.5, neg_pos_ratio=3, norm_mode=norm_mode)
            torch.cuda.synchronize()
            outs[form] = {k: v.cpu() for k, v in o.items()}
    finally:
        _lib.check(lib.ssd_tune_set_loss_form(1), "tune")
    a, b = outs[0], outs[1]
    assert torch.equal(a["obj"], b["obj"]) and torch.equal(a["cls"], b["cls"])
    assert float(a["losses"][2]) == float(b["losses"][2]) and float(a["losses"][2]) >= bs       # n_pos: every box forces one prior
    # the same priors carry a conf gradient (positives + selected negatives), the loc gradient is a sign pattern: equal up to the 1/n_pos scale
    assert torch.equal(a["dconf"] != 0, b["dconf"] != 0)
    assert torch.equal(a["dloc"], b["dloc"])
    _close(b["dconf"], a["dconf"], 1e-6, "dconf")
    for q in (0, 1):
        ra, rb = float(a["losses"][q]), float(b["losses"][q])
        assert abs(ra - rb) <= 2e-6 * max(1.0, abs(ra)), (q, ra, rb)
    # forward-only call (no gradient buffers) gives the same losses
    o = ops.multibox_loss(*args, iou_threshold=0.5, neg_pos_ratio=3, norm_mode=norm_mode, want_grads=False)
    assert torch.equal(o["losses"].cpu(), b["losses"]) and torch.equal(o["cls"].cpu(), b["cls"])


def test_weight_job_table_bf16_copies_are_the_rounded_permutes():
    """Job kind 3 of ssd_weights_prepare (the bf16-tensor mode's filter copies, Model.py:135-143 / 176-184): OHWI [co][taps][ci] and IHWO
    [ci][taps][pad1] bf16 copies of OIHW f32 masters, written from one read of 32 x 32-channel bricks.  Pure rounding + data movement:
    bit-exact against torch's own permute + round-to-nearest-even cast, with a head's two filter pieces, ragged channel counts and 1 tap."""
    from objectdetection_ssd_amd import ops
    dev = _dev()
    g_ = torch.Generator().manual_seed(33)
    rnd = lambda *s: torch.randn(*s, generator=g_).to(dev)
    cases = [
        ((rnd(512, 256, 3, 3),), 512, 512),                       # a trunk layer
        ((rnd(16, 512, 3, 3), rnd(84, 512, 3, 3)), 100, 128),      # the c_4 head: two pieces, 100 filters, K of the data gradient 128
        ((rnd(70, 80, 3, 3),), 70, 128),                          # ragged bricks both ways
        ((rnd(96, 160, 1, 1),), 96, 128),                         # one tap
    ]
    jobs = []
    for pieces, co_all, pad1 in cases:
        ci, taps = pieces[0].shape[1], pieces[0].shape[2] * pieces[0].shape[3]
        jobs.append(dict(kind=3, w0=pieces[0], w1=pieces[1] if len(pieces) > 1 else None, co0=pieces[0].shape[0], co=co_all, ci=ci, taps=taps,
                         co_pad=co_all, pad1=pad1,
                         out_fwd=torch.full((co_all, taps, ci), float("nan"), device=dev, dtype=torch.bfloat16),
                         out_bwd=torch.full((ci, taps, pad1), float("nan"), device=dev, dtype=torch.bfloat16)))
    jobs.append(dict(jobs[0], out_fwd=torch.full((512, 9, 256), float("nan"), device=dev, dtype=torch.bfloat16), out_bwd=None))   # forward copy only
    table = ops.WeightTable(jobs, dev)
    table.run()
    torch.cuda.synchronize()
    for jb, (pieces, co_all, pad1) in zip(jobs, cases + [cases[0]]):
        w = torch.cat(pieces)
        co, ci, r, s_ = w.shape
        ref_f = w.permute(0, 2, 3, 1).reshape(co, r * s_, ci).to(torch.bfloat16)
        assert torch.equal(jb["out_fwd"].view(torch.int16), ref_f.view(torch.int16))
        if jb["out_bwd"] is not None:
            ref_b = torch.zeros(ci, r * s_, pad1, device=dev, dtype=torch.bfloat16)
            ref_b[:, :, :co] = w.permute(1, 2, 3, 0).reshape(ci, r * s_, co).to(torch.bfloat16)
            assert torch.equal(jb["out_bwd"].view(torch.int16), ref_b.view(torch.int16))
