"""GPU parity of the assembled hot path against the oracle and the reference-generated
golden vectors: matching (bit-exact), MultiBox loss + gradients, NMS decode
(bit-exact keep sets), SSD_300 forward / train step, and size-independent
properties at BASELINE.json's full batch size.
"""
import os

import numpy as np
import pytest
import torch

import ssd_oracle as O
from helpers import nms_case, split_case, synth_gt

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(DEV) if dtype is None else t.to(DEV, dtype)


# ---------------------------------------------------------------------------------------------
# matching + loss
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ci", range(18))
def test_ssd_loss_vs_golden_and_oracle(gold_dir, ci):
    from objectdetection_ssd_amd import Losses
    z = np.load(os.path.join(gold_dir, "match_loss.npz"))
    boxes, classes, loc, conf, p = split_case(z, ci)
    lt = _t(loc).requires_grad_(True)
    ct = _t(conf).requires_grad_(True)
    l_loc, l_conf = Losses.ssd((lt, ct), [_t(c) for c in classes], [_t(b) for b in boxes])
    (l_loc + l_conf).backward()
    torch.cuda.synchronize()
    cls = Losses.last_match["cls"].cpu().numpy()
    obj = Losses.last_match["obj"].cpu().numpy()
    # integer outputs: bit-exact against the reference
    assert np.array_equal(cls.astype(np.int8), z[p + "cls"])
    pos = cls != 20
    allb = np.concatenate(boxes)
    assert np.array_equal(O.xyxy_to_xywh(allb)[obj][pos], z[p + "gt_pos"])          # matched GT of every positive
    ora = O.multibox_loss(loc, conf, boxes, classes)
    assert np.array_equal(obj[pos], ora["obj"][pos])
    # floating point: 1e-4 (north_star)
    assert abs(l_loc.item() - float(z[p + "loc_loss"])) <= 1e-4 * max(1, abs(float(z[p + "loc_loss"])))
    assert abs(l_conf.item() - float(z[p + "conf_loss"])) <= 1e-4 * max(1, abs(float(z[p + "conf_loss"])))
    dloc = lt.grad.cpu().numpy()
    dconf = ct.grad.cpu().numpy()
    np.testing.assert_allclose(dloc, ora["dloc"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(dloc[pos], z[p + "dloc_pos"], rtol=1e-5, atol=1e-8)
    touched = np.nonzero(np.abs(dconf.reshape(-1, 21)).sum(1) > 0)[0]
    assert np.array_equal(touched, z[p + "dconf_touched"])                          # same hard-negative set
    np.testing.assert_allclose(dconf, ora["dconf"], rtol=1e-3, atol=1e-6)
    rows = z[p + "dconf_rows"]
    np.testing.assert_allclose(dconf.reshape(-1, 21)[rows], z[p + "dconf_vals"], rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("ci", range(3))
def test_ssd_loss_degenerate_ground_truth_vs_reference(gold_dir, ci):
    """Zero-area / zero-height ground truth (tests/golden/degenerate.npz, from the reference): forced match only, loc loss
    inf like the reference's, finite gradients, identical classes and hard-negative set."""
    from objectdetection_ssd_amd import Losses
    z = np.load(os.path.join(gold_dir, "degenerate.npz"))
    boxes, classes, loc, conf, p = split_case(z, ci)
    lt = _t(loc).requires_grad_(True)
    ct = _t(conf).requires_grad_(True)
    l_loc, l_conf = Losses.ssd((lt, ct), [_t(c) for c in classes], [_t(b) for b in boxes])
    (l_loc + l_conf).backward()
    cls = Losses.last_match["cls"].cpu().numpy()
    assert np.array_equal(cls.astype(np.int8), z[p + "cls"])
    assert np.isinf(l_loc.item()) and l_loc.item() > 0 and np.isinf(z[p + "loc_loss"])
    assert abs(l_conf.item() - float(z[p + "conf_loss"])) <= 1e-4 * max(1, abs(float(z[p + "conf_loss"])))
    dloc, dconf = lt.grad.cpu().numpy(), ct.grad.cpu().numpy()
    assert np.isfinite(dloc).all() and np.isfinite(dconf).all()
    np.testing.assert_allclose(dloc[cls != 20], z[p + "dloc_pos"], rtol=1e-5, atol=1e-8)
    touched = np.nonzero(np.abs(dconf.reshape(-1, 21)).sum(1) > 0)[0]
    assert np.array_equal(touched, z[p + "dconf_touched"])
    np.testing.assert_allclose(np.abs(dconf).astype(np.float64).sum(), z[p + "dconf_abs_sum"], rtol=1e-4)


def test_ssd_loss_is_deterministic_and_rejects_bad_input():
    from objectdetection_ssd_amd import Losses
    rng = np.random.default_rng(11)
    boxes, classes = synth_gt(rng, 4)
    loc = _t(rng.standard_normal((4, 8732, 4), dtype=np.float32))
    conf = _t(rng.standard_normal((4, 8732, 21), dtype=np.float32))
    a = Losses.ssd((loc, conf), [_t(c) for c in classes], [_t(b) for b in boxes])
    b = Losses.ssd((loc, conf), [_t(c) for c in classes], [_t(b) for b in boxes])
    assert a[0].item() == b[0].item() and a[1].item() == b[1].item()
    with pytest.raises(ValueError):
        Losses.ssd((loc, conf), [_t(c) for c in classes[:3]] + [torch.zeros(0, device=DEV)],
                   [_t(b) for b in boxes[:3]] + [torch.zeros(0, 4, device=DEV)])
    with pytest.raises(RuntimeError):
        Losses.ssd((loc.cpu(), conf.cpu()), [torch.from_numpy(c) for c in classes], [torch.from_numpy(b) for b in boxes])


def test_loss_norm_mode_1_is_unnormalised_sum():
    """data-parallel form: sums / gradients before the division by n_pos."""
    from objectdetection_ssd_amd import Losses
    rng = np.random.default_rng(12)
    boxes, classes = synth_gt(rng, 3)
    loc = _t(rng.standard_normal((3, 8732, 4), dtype=np.float32))
    conf = _t(rng.standard_normal((3, 8732, 21), dtype=np.float32))
    cl = [_t(c) for c in classes]
    bx = [_t(b) for b in boxes]
    l0 = loc.clone().requires_grad_(True); c0 = conf.clone().requires_grad_(True)
    a = Losses.ssd((l0, c0), cl, bx)
    (a[0] + a[1]).backward()
    n_pos = float(Losses.last_match["n_pos"].item())
    l1 = loc.clone().requires_grad_(True); c1 = conf.clone().requires_grad_(True)
    b = Losses.ssd((l1, c1), cl, bx, norm_mode=1)
    (b[0] + b[1]).backward()
    assert abs(b[0].item() / n_pos - a[0].item()) <= 1e-5 * max(1, a[0].item())
    assert abs(b[1].item() / n_pos - a[1].item()) <= 1e-5 * max(1, a[1].item())
    torch.testing.assert_close(l1.grad / n_pos, l0.grad, rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(c1.grad / n_pos, c0.grad, rtol=1e-5, atol=1e-9)


# ---------------------------------------------------------------------------------------------
# decode + NMS
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ni", range(6))
def test_inference_vs_golden(gold_dir, ni):
    from objectdetection_ssd_amd import Losses
    z = np.load(os.path.join(gold_dir, "nms.npz"))
    l_, c_, top_k, p = nms_case(z, ni)
    w, h = [int(v) for v in z["img_wh"]]
    out = Losses.inference(_t(l_), _t(c_), (w, h), top_k=top_k, toDraw=False)
    ob, oc, op_, oi = O.decode_nms(l_, c_, w, h, top_k=top_k)
    if z[p + "boxes"].shape[0] == 0:
        assert out == ([], [], [])
        return
    boxes, classes, probs = [t.cpu().numpy() for t in out]
    ids = Losses.inference.last_prior_ids.cpu().numpy()
    assert boxes.shape == z[p + "boxes"].shape
    # keep set on ORIGINAL prior ids, bit-exact (SURVEY section 7 "hard parts")
    assert np.array_equal(ids, oi)
    assert np.array_equal(classes, z[p + "classes"])
    np.testing.assert_allclose(probs, z[p + "probs"], rtol=1e-5)
    np.testing.assert_allclose(boxes, z[p + "boxes"], rtol=1e-5, atol=1e-3)


def test_inference_worst_case_all_candidates():
    """every prior is a candidate of one class with identical boxes -> one survivor per class"""
    from objectdetection_ssd_amd import Losses
    l_ = torch.zeros(8732, 4, device=DEV)
    c_ = torch.full((8732, 21), -10.0, device=DEV)
    c_[:, 3] = 10.0
    boxes, classes, probs = Losses.inference(l_, c_, (300, 300), toDraw=False)
    ob, oc, op_, oi = O.decode_nms(l_.cpu().numpy(), c_.cpu().numpy(), 300, 300)
    assert np.array_equal(Losses.inference.last_prior_ids.cpu().numpy(), oi)
    assert np.array_equal(classes.cpu().numpy(), oc)
    assert boxes.shape[0] == ob.shape[0] <= 200


# ---------------------------------------------------------------------------------------------
# network
# ---------------------------------------------------------------------------------------------
def _load_params(net, params):
    named = dict(net.named_parameters())
    with torch.no_grad():
        for k, v in params.items():
            named[k].copy_(v)


@pytest.fixture(scope="module")
def golden_net(gold_dir):
    from objectdetection_ssd_amd import Model
    z = np.load(os.path.join(gold_dir, "network.npz"))
    params = O.ssd300_random_params(int(z["param_seed"]))
    net = Model.SSD_300()
    _load_params(net, params)
    net = net.to(DEV)
    return net, params, z


def test_ssd300_forward_vs_reference_golden(golden_net):
    net, params, z = golden_net
    bs = int(z["bs"])
    x = np.random.default_rng(int(z["x_seed"])).standard_normal((bs, 3, 300, 300), dtype=np.float32)
    net.eval()
    with torch.no_grad():
        loc, conf = net(_t(x))
    assert loc.shape == (bs, 8732, 4) and conf.shape == (bs, 8732, 21)
    idx = z["prior_idx"]
    loc_c, conf_c = loc.cpu().numpy(), conf.cpu().numpy()
    for got, ref, nm in ((loc_c[:, idx], z["loc_s"], "loc"), (conf_c[:, idx], z["conf_s"], "conf")):
        err = np.abs(got - ref).max()
        assert err <= 1e-4 * max(1.0, np.abs(ref).max()), f"{nm}: {err}"
    assert abs(np.abs(loc_c).astype(np.float64).sum() - float(z["loc_abs"])) <= 1e-4 * float(z["loc_abs"])
    assert abs(np.abs(conf_c).astype(np.float64).sum() - float(z["conf_abs"])) <= 1e-4 * float(z["conf_abs"])
    # full tensors against the oracle network
    with torch.no_grad():
        lo, co = O.ssd300_forward(torch.from_numpy(x), params)
    assert np.abs(loc_c - lo.numpy()).max() <= 1e-4 * max(1.0, float(lo.abs().max()))
    assert np.abs(conf_c - co.numpy()).max() <= 1e-4 * max(1.0, float(co.abs().max()))


@pytest.mark.parametrize("engine", ["wino", "direct"])
def test_ssd300_train_step_vs_reference_golden(golden_net, engine):
    """forward + ssd loss + backward through every kernel: losses (= loss delta vs CPU, 1e-4) and all 71 gradients against the
    reference's own f32 CPU step.  Gradient bars are PER TENSOR: 2 x the distance measured on an MI355X for this engine
    (tests/golden/grad_bars.json, written by tools/grad_bars.py) -- the reference's f32 backward is itself 1e-4 .. 2e-3 from an
    f64 evaluation on the backbone (ReLU / max-pool decisions flip on last-bit differences), so the distance grows towards the
    input, and a blanket bar wide enough for conv1 would hide a regression on conv4 / conv5 / the heads."""
    import grad_measure as M
    net, params, z = golden_net
    table = M.load_bars()
    x, cl, bx = M.golden_case(z)
    M.set_engine(net, engine)
    try:
        _, _, l1, l2, grads = M.train_step(net, x, cl, bx)
    finally:
        M.set_engine(net, "wino")
    assert abs(l1 - float(z["loc_loss"])) <= 1e-4 * max(1, float(z["loc_loss"]))       # loss delta vs CPU
    assert abs(l2 - float(z["conf_loss"])) <= 1e-4 * max(1, float(z["conf_loss"]))
    names = [str(n) for n in z["grad_names"]]
    assert len(names) == 71 and set(names) == set(table["gold_" + engine])
    bad = []
    for k, ref_l2 in zip(names, z["grad_l2"]):
        assert k in grads, k
        got = float(grads[k].double().norm())
        lim = M.bar(table, "gold_" + engine, k, M.FLOOR_NORM)
        if abs(got - ref_l2) > lim * max(ref_l2, 1e-6):
            bad.append((k, abs(got - ref_l2) / max(ref_l2, 1e-6), lim))
    assert not bad, bad
    for k in ("model.features.0.weight", "model.features.21.bias", "c_11_cl.weight", "seq10.2.weight",
              "rescaling_conv_4_3", "c_4_bb.bias"):
        ref = z["g_" + k].astype(np.float64)
        got = grads[k].cpu().numpy().astype(np.float64)
        rel = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-12)
        lim = M.bar(table, "gold_elem_" + engine, k)
        assert rel <= lim, (k, rel, lim)
    # dead VGG classifier receives nothing
    assert "model.classifier.0.weight" not in grads


@pytest.mark.parametrize("engine", ["wino", "direct"])
def test_train_step_gradients_vs_f64_oracle(golden_net, engine):
    """Every gradient of the train step against an f64 CPU evaluation of the oracle network, on ground truth with large boxes
    so that all six scales (incl. the 3x3 and 1x1 maps) carry positives.  Per-tensor bars = 2 x the measured distance
    (tests/golden/grad_bars.json): ~4e-5 on conv5 / fc / aux / heads, ~2e-4 at conv3, a few 1e-3 at conv1 (f32 activations that
    differ by an ulp flip a few ReLU / max-pool decisions out of ~1e8; the reference's own f32 CPU backward sits at 1.8e-3 on
    conv1_1 -- `cpu32_f64` in the same table)."""
    import grad_measure as M
    net, params, _ = golden_net
    table = M.load_bars()
    lo64, co64, a1, a2, g64 = M.f64_oracle_grads(params)
    M.set_engine(net, engine)
    try:
        loc, conf, l1, l2, grads = M.gpu_f64_case(net)
    finally:
        M.set_engine(net, "wino")
    assert abs(l1 - a1) <= 1e-4 * max(1, a1) and abs(l2 - a2) <= 1e-4 * max(1, a2)
    assert float((loc.cpu().double() - lo64).abs().max()) <= 1e-4 * max(1, float(lo64.abs().max()))
    assert float((conf.cpu().double() - co64).abs().max()) <= 1e-4 * max(1, float(co64.abs().max()))
    assert set(g64) == set(table["f64_" + engine]) and len(g64) == 71
    bad = []
    for k, ref in g64.items():
        assert float(ref.norm()) > 0, f"{k}: scale not exercised"
        rel = M.rel_l2(grads[k], ref)
        lim = M.bar(table, "f64_" + engine, k)
        if rel > lim:
            bad.append((k, rel, lim))
    assert not bad, bad
    # the table itself must describe an f32-accurate backward: conv1 .. conv3 (maps of 300 .. 75 pixels, 1e7 .. 1e8 ReLU / pool
    # decisions per image) within 1e-2, everything from conv4 on within 1e-3
    for k, v in table["f64_" + engine].items():
        early = any(k.startswith(f"model.features.{i}.") for i in (0, 2, 5, 7, 10, 12, 14))
        assert v <= (1e-2 if early else 1e-3), (k, v)


@pytest.mark.parametrize("engine", ["direct", "wino"])
def test_train_step_gradients_vs_decision_pinned_f64_oracle(golden_net, engine):
    """The gradient check that does not depend on a table measured with the code under test.  The f64 oracle is run FOLLOWING the
    discrete decisions of the HIP forward (its ReLU masks, its max-pool arg-max codes, its hard negatives -- exported from what the
    engine saves for its own backward): both sides then evaluate the same fixed piecewise-linear function, no ReLU or pool flips on
    a last-bit difference, and what remains is arithmetic error alone.  Every one of the 71 gradient tensors must be within ONE
    fixed bar of the f64 result: 1e-5 relative L2 on the exact-f32 direct engine, 5e-5 on the Winograd engine (F(4x4,3x3)
    re-associates the sums with coefficients up to 8 and 1/24) -- measured 1.6e-6 / 7.7e-6 at worst, i.e. the 1e-3 .. 4e-3 that
    conv1 / conv2 show against an UNPINNED f64 run (tests/golden/grad_bars.json) is flip noise, not arithmetic."""
    import grad_measure as M
    net, params, _ = golden_net
    x, boxes, classes = M.f64_case()
    xd, cl, bx = M._t(x), [M._t(c) for c in classes], [M._t(b) for b in boxes]
    M.set_engine(net, engine)
    try:
        decisions, neg = M.gpu_decisions(net, xd, cl, bx)
        _, _, l1, l2, grads = M.train_step(net, xd, cl, bx)
    finally:
        M.set_engine(net, "wino")
    assert any(m is None for m in decisions["relu"].values()) == (engine == "wino")       # Winograd fuses three conv -> ReLU -> pool layers
    a1, a2, g64 = M.f64_pinned_grads(params, decisions, neg)
    assert abs(l1 - a1) <= 1e-4 * max(1, a1) and abs(l2 - a2) <= 1e-4 * max(1, a2)
    assert len(g64) == 71
    rows = sorted(((M.rel_l2(grads[k], g64[k]), k) for k in g64), reverse=True)
    print(f"decision-pinned f64 distance [{engine}]: worst " + ", ".join(f"{k} {v:.2e}" for v, k in rows[:6]) + f"; median {rows[35][0]:.2e}")
    bad = [(k, v) for v, k in rows if v > M.PINNED_BAR[engine]]
    assert not bad, bad


def test_train_step_at_bench_batch_winograd_vs_direct_engine():
    """bench.py's own step at bench.py's size: batch 32 of its synthetic input through the default engine (Winograd F(4x4) with
    kept planes, fused pools, the shared dy pass) against the direct exact-f32 MFMA engine: loc / conf / losses within 1e-4 of
    their scale, every one of the 71 gradients within its measured per-tensor bar (`wd32` of tests/golden/grad_bars.json, x 2),
    per-prior classes identical."""
    import grad_measure as M
    from objectdetection_ssd_amd import Losses, Model
    table = M.load_bars()
    z = np.load(os.path.join(M.ROOT, "tests", "golden", "network.npz"))
    net = Model.SSD_300()
    _load_params(net, O.ssd300_random_params(int(z["param_seed"])))
    net = net.to(DEV)
    x, cl, bx = M.bench_batch()
    res = {}
    for eng in M.ENGINES:
        M.set_engine(net, eng)
        assert net.winograd is (eng == "wino")
        res[eng] = M.train_step(net, x, cl, bx) + (Losses.last_match["cls"].clone(),)
    (la, ca, a1, a2, ga, ma), (lb, cb, b1, b2, gb, mb) = res["wino"], res["direct"]
    assert torch.isfinite(la).all() and torch.isfinite(ca).all()
    assert float((la - lb).abs().max()) <= 1e-4 * max(1.0, float(lb.abs().max()))
    assert float((ca - cb).abs().max()) <= 1e-4 * max(1.0, float(cb.abs().max()))
    assert abs(a1 - b1) <= 1e-4 * max(1.0, abs(b1)) and abs(a2 - b2) <= 1e-4 * max(1.0, abs(b2))
    assert torch.equal(ma, mb)
    assert set(ga) == set(gb) == set(table["wd32"]) and len(ga) == 71
    bad = [(n, M.rel_l2(ga[n], gb[n]), M.bar(table, "wd32", n)) for n in ga if M.rel_l2(ga[n], gb[n]) > M.bar(table, "wd32", n)]
    assert not bad, bad


def test_full_batch_properties():
    """BASELINE config 2 size (bs=32): finite outputs, batch independence (image i of a batch ==
    the same image alone), per-image matching independence, loss reproducible."""
    from objectdetection_ssd_amd import Losses, Model
    torch.manual_seed(0)
    net = Model.SSD_300().to(DEV).eval()
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(32, 3, 300, 300, generator=g).to(DEV)
    with torch.no_grad():
        loc, conf = net(x)
        loc1, conf1 = net(x[5:6].contiguous())
    assert torch.isfinite(loc).all() and torch.isfinite(conf).all()
    assert (loc[5:6] - loc1).abs().max().item() <= 1e-4 * max(1.0, loc1.abs().max().item())
    assert (conf[5:6] - conf1).abs().max().item() <= 1e-4 * max(1.0, conf1.abs().max().item())
    boxes, classes = synth_gt(np.random.default_rng(99), 32)
    bx = [_t(b) for b in boxes]; cl = [_t(c) for c in classes]
    l = Losses.ssd((loc, conf), cl, bx)
    cls_all = Losses.last_match["cls"].clone()
    Losses.ssd((loc[7:8].contiguous(), conf[7:8].contiguous()), cl[7:8], bx[7:8])
    assert torch.equal(Losses.last_match["cls"][0], cls_all[7])
    assert torch.isfinite(l[0]) and torch.isfinite(l[1])


def _sgd_groups(named):
    """train.py:44-55: biases at 2x lr, everything else at lr; momentum .9, weight decay 5e-4"""
    named = list(named)
    biases = [p for n, p in named if p.requires_grad and n.endswith(".bias")]
    others = [p for n, p in named if p.requires_grad and not n.endswith(".bias")]
    return biases, others


@pytest.mark.parametrize("engine", ["direct", "wino"])
def test_training_loop_as_train_function_vs_cpu_oracle(gold_dir, engine):
    """The caller's loop (train_function.py:59-95: zero_grad, cnn(inputs), ssd(...), loss1+loss2, .item(),
    backward, optimizer.step with train.py's SGD groups) for three iterations on the GPU path against the same
    loop on the CPU oracle: the loss trajectory must agree step by step.  Bars: the first step (same weights) 1e-4 for both
    engines.  After an update every last-bit difference of a gradient moves the weights, and a handful of ReLU / max-pool decisions
    out of ~1e8 flip in the next forward -- each flip a discrete change of one path.  Measured (loc / conf): direct engine
    (plain f32 fma chains, other summation order than the CPU's) 1.4e-5 / 7e-6 at step 2, 1.2e-5 / 4.9e-5 at step 3, bar 1e-4;
    Winograd engine (re-associated sums) 1.4e-5 / 3.4e-5 at step 2, 3.4e-4 / 8e-5 at step 3, bar 1e-3.  (The gradients themselves
    are held per tensor against f64 in test_train_step_gradients_vs_f64_oracle.)"""
    from objectdetection_ssd_amd import Losses, Model
    lr = 1e-4                                   # train.py:53
    bs = 2
    x = np.random.default_rng(31).standard_normal((bs, 3, 300, 300), dtype=np.float32)
    boxes, classes = synth_gt(np.random.default_rng(32), bs)
    params = O.ssd300_random_params(5)
    # CPU oracle loop
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    b_cpu, o_cpu = _sgd_groups(P.items())
    opt_cpu = torch.optim.SGD([{"params": b_cpu, "lr": 2 * lr}, {"params": o_cpu}], lr=lr, momentum=0.9, weight_decay=5e-4)
    ref = []
    for _ in range(3):
        opt_cpu.zero_grad()
        loc, conf = O.ssd300_forward(torch.from_numpy(x), P)
        l1, l2 = O.multibox_loss_torch(loc, conf, [torch.from_numpy(b) for b in boxes], [torch.from_numpy(c) for c in classes])
        (l1 + l2).backward()
        opt_cpu.step()
        ref.append((l1.item(), l2.item()))
    # GPU path, written like the reference's caller
    cnn = Model.SSD_300()
    _load_params(cnn, params)
    cnn = cnn.to(DEV)
    cnn.winograd = engine == "wino"
    biases, not_biases = _sgd_groups(cnn.named_parameters())
    assert len(biases) == 38
    optimizer = torch.optim.SGD(params=[{"params": biases, "lr": 2 * lr}, {"params": not_biases}], lr=lr, momentum=0.9, weight_decay=5e-4)
    inputs = _t(x)
    cl = [_t(c) for c in classes]
    bx = [_t(b) for b in boxes]
    cnn.train()
    got = []
    for _ in range(3):
        optimizer.zero_grad()
        with torch.set_grad_enabled(True):
            outputs = cnn(inputs)
            loss1, loss2 = Losses.ssd(outputs, cl, bx)
            loss = loss1 + loss2
            got.append((loss1.item(), loss2.item()))
            loss.backward()
            optimizer.step()
    for it, ((a1, a2), (b1, b2)) in enumerate(zip(got, ref)):
        tol = 1e-4 if (it == 0 or engine == "direct") else 1e-3
        assert abs(a1 - b1) <= tol * max(1, abs(b1)), (it, got, ref)
        assert abs(a2 - b2) <= tol * max(1, abs(b2)), (it, got, ref)
    assert len(not_biases) == 39          # 77 named parameters - 38 biases (incl. the dead VGG classifier)
    assert got[2][0] + got[2][1] < got[0][0] + got[0][1]            # it trains
    # eval / no_grad path gives the same outputs as the autograd path
    cnn.eval()
    with torch.no_grad():
        l_e, c_e = cnn(inputs)
    cnn.train()
    l_t, c_t = cnn(inputs)
    assert torch.equal(l_e, l_t.detach()) and torch.equal(c_e, c_t.detach())


def test_flat_sgd_data_parallel_step_equals_torch_sgd():
    """ddp.FlatSGDDataParallel (un-normalised sums, flat buffers, fused SGD * 1/n_pos) == reference normalisation +
    torch.optim.SGD with train.py's groups, for three steps at world size 1."""
    from objectdetection_ssd_amd import Losses, Model
    from objectdetection_ssd_amd.ddp import FlatSGDDataParallel
    lr, bs = 1e-4, 2
    x = _t(np.random.default_rng(41).standard_normal((bs, 3, 300, 300), dtype=np.float32))
    boxes, classes = synth_gt(np.random.default_rng(42), bs)
    cl = [_t(c) for c in classes]
    bx = [_t(b) for b in boxes]
    params = O.ssd300_random_params(6)
    a = Model.SSD_300(); _load_params(a, params); a = a.to(DEV).train()
    b = Model.SSD_300(); _load_params(b, params); b = b.to(DEV).train()
    biases, others = _sgd_groups(a.named_parameters())
    opt = torch.optim.SGD([{"params": biases, "lr": 2 * lr}, {"params": others}], lr=lr, momentum=0.9, weight_decay=5e-4)
    dp = FlatSGDDataParallel(b, lr=lr, momentum=0.9, weight_decay=5e-4)
    for _ in range(3):
        opt.zero_grad()
        l1, l2 = Losses.ssd(a(x), cl, bx)
        (l1 + l2).backward()
        opt.step()
        dp.zero_grad()
        m1, m2, n_pos = Losses.ssd(b(x), cl, bx, norm_mode=1, with_n_pos=True)
        (m1 + m2).backward()
        dp.reduce_and_step(n_pos)
    na, nb = dict(a.named_parameters()), dict(b.named_parameters())
    for k in a._engine.names:
        ref = na[k].detach()
        err = float((nb[k].detach() - ref).abs().max())
        assert err <= 2e-5 * max(1.0, float(ref.abs().max())), (k, err)


def test_fused_sgd_optimizer_is_bitwise_torch_sgd():
    """(f)-1: optim.SGD built exactly like train.py:44-55 -- parameters, momentum buffers and state_dict after several steps are
    bit-identical to torch.optim.SGD's, including a parameter that gets its first gradient late and a checkpoint hand-over in
    both directions (train_function.py:27,116)."""
    from objectdetection_ssd_amd.optim import SGD
    g = torch.Generator().manual_seed(77)
    shapes = [(64, 3, 3, 3), (64,), (150, 256, 3, 3), (150,), (1, 512, 1, 1), (7,), (33, 5)]
    lr = 1e-4

    def make():
        ps = [torch.nn.Parameter(torch.randn(s, generator=torch.Generator().manual_seed(i)).to(DEV)) for i, s in enumerate(shapes)]
        biases = [p for p in ps if p.dim() == 1]
        others = [p for p in ps if p.dim() != 1]
        return ps, [{"params": biases, "lr": 2 * lr}, {"params": others}]

    pa, ga = make()
    pb, gb = make()
    ref = torch.optim.SGD(params=ga, lr=lr, momentum=0.9, weight_decay=5e-4)
    opt = SGD(params=gb, lr=lr, momentum=0.9, weight_decay=5e-4)
    assert [sorted(k for k in grp if k != "params") for grp in opt.state_dict()["param_groups"]] == \
           [sorted(k for k in grp if k != "params") for grp in ref.state_dict()["param_groups"]]

    def run(o_ref, o_new, p_ref, p_new, steps, skip):
        for it in range(steps):
            o_ref.zero_grad(); o_new.zero_grad()
            for i, (a, b) in enumerate(zip(p_ref, p_new)):
                if i == 5 and it < skip:
                    continue                              # no gradient yet: torch leaves this parameter untouched
                gr = (torch.randn(a.shape, generator=g) * 10).to(DEV)
                a.grad = gr.clone(); b.grad = gr.clone()
            o_ref.step(); o_new.step()
            for i, (a, b) in enumerate(zip(p_ref, p_new)):
                assert torch.equal(a.detach(), b.detach()), (it, i)

    run(ref, opt, pa, pb, 4, skip=2)
    sa, sb = ref.state_dict(), opt.state_dict()
    assert sa["state"].keys() == sb["state"].keys()
    for k in sa["state"]:
        assert torch.equal(sa["state"][k]["momentum_buffer"], sb["state"][k]["momentum_buffer"]), k
    # hand the checkpoints over crosswise and keep going
    pc, gc_ = make(); pd, gd = make()
    with torch.no_grad():
        for dst, src in zip(pc, pa): dst.copy_(src)
        for dst, src in zip(pd, pa): dst.copy_(src)
    ref2 = torch.optim.SGD(params=gc_, lr=lr, momentum=0.9, weight_decay=5e-4)
    opt2 = SGD(params=gd, lr=lr, momentum=0.9, weight_decay=5e-4)
    ref2.load_state_dict(sb)                             # ours -> torch
    opt2.load_state_dict(sa)                             # torch -> ours
    for grp in opt2.param_groups:
        grp["lr"] = grp["lr"] * 3                         # train_function.py:29-30 rewrites lr after a resume
    for grp in ref2.param_groups:
        grp["lr"] = grp["lr"] * 3
    run(ref2, opt2, pc, pd, 3, skip=0)
    # plain SGD (momentum 0, no decay) keeps no state, like torch
    pe, ge = make(); pf, gf = make()
    r3, o3 = torch.optim.SGD(ge, lr=1e-2), SGD(gf, lr=1e-2)
    run(r3, o3, pe, pf, 2, skip=0)
    assert all("momentum_buffer" not in st or st["momentum_buffer"] is None for st in o3.state.values())
    with pytest.raises(ValueError):
        SGD(gf, lr=1e-2, nesterov=True, momentum=0.9)


def test_fused_sgd_optimizer_in_the_callers_loop():
    """train.py's optimizer construction with optim.SGD in place of torch.optim.SGD: after three iterations of the caller's loop
    the two models hold bit-identical weights (the path itself is deterministic), and the re-laid weight caches follow the
    in-place update."""
    from objectdetection_ssd_amd import Losses, Model
    from objectdetection_ssd_amd.optim import SGD
    lr, bs = 1e-4, 2
    x = _t(np.random.default_rng(51).standard_normal((bs, 3, 300, 300), dtype=np.float32))
    boxes, classes = synth_gt(np.random.default_rng(52), bs)
    cl = [_t(c) for c in classes]
    bx = [_t(b) for b in boxes]
    params = O.ssd300_random_params(7)
    a = Model.SSD_300(); _load_params(a, params); a = a.to(DEV).train()
    b = Model.SSD_300(); _load_params(b, params); b = b.to(DEV).train()
    ba, oa = _sgd_groups(a.named_parameters())
    bb, ob = _sgd_groups(b.named_parameters())
    opt_a = torch.optim.SGD(params=[{"params": ba, "lr": 2 * lr}, {"params": oa}], lr=lr, momentum=0.9, weight_decay=5e-4)
    opt_b = SGD(params=[{"params": bb, "lr": 2 * lr}, {"params": ob}], lr=lr, momentum=0.9, weight_decay=5e-4)
    losses = []
    for _ in range(3):
        step = []
        for net, opt in ((a, opt_a), (b, opt_b)):
            opt.zero_grad()
            l1, l2 = Losses.ssd(net(x), cl, bx)
            (l1 + l2).backward()
            opt.step()
            step.append((l1.item(), l2.item()))
        losses.append(step)
        assert step[0] == step[1], losses
    assert losses[2][1] != losses[0][1]                   # the forward sees the updated weights
    na, nb = dict(a.named_parameters()), dict(b.named_parameters())
    for k in a._engine.names:
        assert torch.equal(na[k].detach(), nb[k].detach()), k
    sd = b.state_dict()                                   # checkpoint layout is unchanged by the flat storage
    assert set(sd.keys()) == set(a.state_dict().keys())


def test_overlapped_gradient_exchange_equals_the_single_allreduce_path():
    """ddp.FlatSGDDataParallel(overlap=True): the engine reports every gradient the moment it exists, slices of the flat buffer
    are handed to the collective as they fill up; at world size 1 the bookkeeping runs without the collectives and the weights
    after three steps must be bit-identical to the plain path."""
    from objectdetection_ssd_amd import Losses, Model
    from objectdetection_ssd_amd.ddp import FlatSGDDataParallel
    lr, bs = 1e-4, 2
    x = _t(np.random.default_rng(61).standard_normal((bs, 3, 300, 300), dtype=np.float32))
    boxes, classes = synth_gt(np.random.default_rng(62), bs)
    cl = [_t(c) for c in classes]
    bx = [_t(b) for b in boxes]
    params = O.ssd300_random_params(8)
    a = Model.SSD_300(); _load_params(a, params); a = a.to(DEV).train()
    b = Model.SSD_300(); _load_params(b, params); b = b.to(DEV).train()
    da = FlatSGDDataParallel(a, lr=lr, momentum=0.9, weight_decay=5e-4)
    db = FlatSGDDataParallel(b, lr=lr, momentum=0.9, weight_decay=5e-4, overlap=True, bucket_bytes=8 << 20)
    assert len(db._bucket_rng) >= 8 and sum(db._need) == len(db.w_names)
    for _ in range(3):
        for net, tr in ((a, da), (b, db)):
            tr.zero_grad()
            l1, l2, n_pos = Losses.ssd(net(x), cl, bx, norm_mode=1, with_n_pos=True)
            (l1 + l2).backward()
            tr.reduce_and_step(n_pos)
    assert torch.equal(da.flat_param, db.flat_param)
    assert torch.equal(da.flat_grad[:da.n + 1], db.flat_grad[:db.n + 1])


def test_inference_batch_equals_single_image_calls(gold_dir):
    """(f)-4: the batched decode gives, image by image, exactly what `inference` gives (incl. an empty image)"""
    from objectdetection_ssd_amd import Losses
    z = np.load(os.path.join(gold_dir, "nms.npz"))
    ls, cs = [], []
    for ni in (0, 3, 5, 2):                      # case 5 has no detections
        l_, c_, _, _ = nms_case(z, ni)
        ls.append(l_); cs.append(c_)
    sizes = [(500, 375), (300, 300), (640, 480), (123, 77)]
    L, C = _t(np.stack(ls)), _t(np.stack(cs))
    outs = Losses.inference_batch(L, C, sizes, top_k=100)
    assert len(outs) == 4
    for i, (sz, o) in enumerate(zip(sizes, outs)):
        single = Losses.inference(L[i], C[i], sz, top_k=100, toDraw=False)
        if single == ([], [], []):
            assert o == ([], [], [])
            continue
        for a, b in zip(o, single):
            assert torch.equal(a, b)
        assert torch.equal(Losses.inference_batch.last_prior_ids[i], Losses.inference.last_prior_ids)


def test_ssd512_build_defined_extension_vs_oracle():
    """SSD512 does not exist in the reference; this checks the same kernels at the larger
    geometry (64x64 ... 1x1 maps, 24564 priors) against the oracle's own restatement: forward, loss, matching
    (bit-exact), backward, and decode."""
    from objectdetection_ssd_amd import Losses, Model
    params = O.ssd300_random_params(8, variant=512)
    net = Model.SSD_512()
    assert set(net._engine.names) == set(params)
    _load_params(net, params)
    net = net.to(DEV).train()
    bs = 1
    x = np.random.default_rng(61).standard_normal((bs, 3, 512, 512), dtype=np.float32)
    boxes = [np.array([[.05, .05, .95, .95], [.1, .3, .475, .675], [.40, .40, .47, .48], [.6, .2, .8, .45]], np.float32)]
    classes = [np.array([1., 5., 12., 7.], np.float32)]
    pri = O.create_priors_ssd512()
    assert np.array_equal(Losses._priors_on(torch.device(DEV), 24564)[0].cpu().numpy(), pri)
    P64 = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    lo, co = O.ssd300_forward(torch.from_numpy(x), P64, variant=512)
    assert lo.shape == (bs, 24564, 4) and co.shape == (bs, 24564, 21)
    ref = O.multibox_loss(lo.detach().numpy(), co.detach().numpy(), boxes, classes, pri_cxcywh=pri, want_grads=False)
    a1, a2 = O.multibox_loss_torch(lo, co, [torch.from_numpy(b) for b in boxes], [torch.from_numpy(c) for c in classes], pri_cxcywh=pri)
    (a1 + a2).backward()
    loc, conf = net(_t(x))
    l1, l2 = Losses.ssd((loc, conf), [_t(c) for c in classes], [_t(b) for b in boxes])
    (l1 + l2).backward()
    assert float((loc.detach().cpu() - lo.detach()).abs().max()) <= 1e-4 * max(1, float(lo.abs().max()))
    assert float((conf.detach().cpu() - co.detach()).abs().max()) <= 1e-4 * max(1, float(co.abs().max()))
    assert np.array_equal(Losses.last_match["cls"].cpu().numpy(), ref["cls"])
    assert abs(l1.item() - a1.item()) <= 1e-4 * max(1, a1.item()) and abs(l2.item() - a2.item()) <= 1e-4 * max(1, a2.item())
    named = dict(net.named_parameters())
    for k, p64 in P64.items():
        g = p64.grad
        if float(g.norm()) == 0:
            continue
        rel = float((named[k].grad.cpu() - g).norm() / g.norm())
        assert rel <= (2e-2 if k.startswith("model.features") else 5e-3), (k, rel)


def test_decode_at_ssd512_prior_count():
    """decode + NMS with 24564 priors (the NMS kernel then works in 32-row chunks: 64-row suppression words for that
    many columns do not fit the LDS); inputs are drawn until every decision has a margin, as for the golden cases"""
    from objectdetection_ssd_amd import Losses
    pri = O.create_priors_ssd512()
    seed = 700
    while True:
        r = np.random.default_rng(seed)
        seed += 1
        l_ = r.standard_normal((24564, 4), dtype=np.float32) * np.float32(0.5)
        c_ = r.standard_normal((24564, 21), dtype=np.float32) * np.float32(1.0)
        pr = torch.softmax(torch.from_numpy(c_).double(), 1).numpy()
        if np.min(np.abs(pr[:, :20] - 0.2)) < 2e-6:
            continue
        bx = O.xywh_to_xyxy(O.decode_offsets(l_, pri))
        ok = True
        for c in range(20):
            idx = np.nonzero(pr[:, c] >= 0.2)[0]
            if idx.size < 2:
                continue
            if np.min(np.diff(np.sort(pr[idx, c]))) < 1e-7 or np.min(np.abs(O.iou_matrix(bx[idx], bx[idx]).astype(np.float64) - 0.45)) < 2e-6:
                ok = False
                break
        if ok:
            break
    out = Losses.inference(_t(l_), _t(c_), (512, 512), top_k=200, toDraw=False)
    ob, oc, op_, oi = O.decode_nms(l_, c_, 512, 512, top_k=200, pri_cxcywh=pri)
    assert ob.shape[0] > 0
    assert np.array_equal(Losses.inference.last_prior_ids.cpu().numpy(), oi)
    assert np.array_equal(out[1].cpu().numpy(), oc)
    np.testing.assert_allclose(out[0].cpu().numpy(), ob, rtol=1e-5, atol=1e-3)


def test_bf16_conv_mode_config3(golden_net):
    """BASELINE configs[2] ("bf16 convs") against the ORACLE's bf16-operand train step (`ssd300_forward(operand_round="bf16")`:
    every convolution on bf16-rounded operands with f32 accumulation, forward and backward, f32 weight gradients where the HIP
    mode keeps them), followed LAYER BY LAYER.  Two bf16 computations that differ only in f32 summation order cannot stay equal
    through 20 layers: a last-bit difference flips the bf16 rounding of ~1e-6/4e-3 of the activations, each flip is a 4e-3
    relative error of one operand of the next layer, and that feeds more flips -- measured (tests/golden/grad_bars.json,
    `bf16_act`): conv1_1 3e-8, conv1_2 1e-6, conv2_1 3e-5, conv2_2 9e-5, conv3 2e-4 .. 1e-3, saturating at the mode's own
    rounding noise (~1e-2 = distance of the bf16 oracle to the f32 oracle) from conv4 on.  So: the first layers, where no flip
    has happened yet, must agree to f32 accuracy; every activation, loc / conf and every gradient must stay within 2 x its
    measured distance and within the mode's own noise; losses within 1e-3; per-prior classes bit-exact.
    Round 3: the mode stores the VGG trunk's activations and gradients in bf16 (`_Engine.bf16_tensors`); the oracle has the same rounding
    points (`ssd300_forward(operand_round="bf16", store_round=True)`: every trunk tensor and its gradient rounded once, at the store)."""
    import grad_measure as M
    from objectdetection_ssd_amd import Losses
    net, params, z = golden_net
    assert net._engine.bf16_tensors
    table = M.load_bars()
    acts, e_loc, e_conf = M.layerwise_forward_distance(net, params, "bf16")
    assert set(acts) == set(table["bf16_act"]) and len(acts) == 24
    # before the flips compound.  bf16-tensor mode (round 3): the compared tensors are themselves rounded to bf16 at the store, so a sum
    # that lands next to a rounding boundary differs by one bf16 spacing (2^-8 relative) in that element -- conv1_1 (K = 27) is bit-equal
    # to the oracle, conv1_2 has a few such elements in 1e5 (1.9e-5 relative L2 measured), conv2_1 7.8e-5
    assert acts["a1_1"] <= 1e-6 and acts["a1_2"] <= 1e-4 and acts["a2_1"] <= 3e-4
    bad = [(k, v, M.bar(table, "bf16_act", k)) for k, v in acts.items() if v > M.bar(table, "bf16_act", k, 1e-6)]
    assert not bad, bad
    noise = table["bf16_mode_noise"]
    assert e_loc <= min(2 * table["bf16_oracle_out"]["loc"], 1.5 * noise["loc"])
    assert e_conf <= min(2 * table["bf16_oracle_out"]["conf"], 1.5 * noise["conf"])
    lo, co, a1, a2, gref = M.f64_oracle_grads(params, operand_round="bf16", dtype=torch.float32, store_round=net._engine.bf16_tensors)
    lo32, co32, *_ = M.f64_oracle_grads(params, dtype=torch.float32)
    x, boxes, classes = M.f64_case()
    ref_match = O.multibox_loss(lo.numpy(), co.numpy(), boxes, classes, want_grads=False)
    try:
        M.set_engine(net, "wino", "bf16")
        assert net.conv_dtype == "bf16"
        loc, conf, l1, l2, grads = M.gpu_f64_case(net)
        cls = Losses.last_match["cls"].cpu().numpy()
    finally:
        M.set_engine(net, "wino", "f32")
    assert abs(l1 - a1) <= 1e-3 * max(1.0, abs(a1)) and abs(l2 - a2) <= 1e-3 * max(1.0, abs(a2))
    assert np.array_equal(cls, ref_match["cls"])
    assert float((loc.cpu() - lo32).abs().max()) > 1e-3 * float(lo32.abs().max())            # bf16 arithmetic, not the f32 path
    assert set(gref) == set(table["bf16_oracle"])
    bad = [(k, M.rel_l2(grads[k], gref[k]), M.bar(table, "bf16_oracle", k)) for k in gref
           if M.rel_l2(grads[k], gref[k]) > M.bar(table, "bf16_oracle", k)]
    assert not bad, bad
    # heads see one layer of backward: close; the distance grows towards the input and stays a perturbation everywhere
    assert max(table["bf16_oracle"][k] for k in gref if k.startswith("c_")) <= 2e-2
    assert max(table["bf16_oracle"].values()) <= 0.3


def test_bf16_conv_mode_at_bench_batch():
    """configs[2] at its per-GPU batch: one bf16-operand train step on bench.py's batch of 32 -- finite, the matching identical to
    the f32 step's (it depends on ground truth and priors only), losses within 2e-2 of the f32 step's, gradient direction kept
    (cosine > 0.99 over all parameters)."""
    import grad_measure as M
    from objectdetection_ssd_amd import Losses, Model
    torch.manual_seed(0)
    net = Model.SSD_300().to(DEV)
    x, cl, bx = M.bench_batch()
    out = {}
    for mode in ("f32", "bf16", "bf16_f32_tensors"):
        M.set_engine(net, "wino", mode[:4].rstrip("_"))
        net._engine.bf16_tensors = mode != "bf16_f32_tensors"        # the round-2 form (f32 tensors, operands rounded into LDS) stays selectable
        try:
            loc, conf, l1, l2, g = M.train_step(net, x, cl, bx)
        finally:
            net._engine.bf16_tensors = True
        assert torch.isfinite(loc).all() and torch.isfinite(conf).all()
        out[mode] = (l1, l2, Losses.last_match["cls"].clone(), torch.cat([g[n].flatten() for n in sorted(g)]))
    M.set_engine(net, "wino", "f32")
    a = out["f32"]
    for mode in ("bf16", "bf16_f32_tensors"):
        b = out[mode]
        assert abs(a[0] - b[0]) <= 2e-2 * a[0] and abs(a[1] - b[1]) <= 2e-2 * a[1]
        assert torch.equal(a[2], b[2])
        assert torch.isfinite(b[3]).all()
        cos = float(torch.dot(a[3], b[3]) / (a[3].norm() * b[3].norm()))
        assert cos > 0.99, (mode, cos)


def test_f32x3_mode_is_f32_accurate(golden_net):
    """conv_dtype = "f32x3" (three exact bf16 limbs per operand, six bf16 MFMAs per product block, f32 accumulate):
    the full train step must meet the SAME bars as the exact-f32 path -- golden loc/conf/losses within 1e-4, gradients
    against the golden f32 reference within 5e-3 relative L2."""
    from objectdetection_ssd_amd import Losses
    net, params, z = golden_net
    bs = int(z["bs"])
    x = _t(np.random.default_rng(int(z["x_seed"])).standard_normal((bs, 3, 300, 300), dtype=np.float32))
    boxes, classes = synth_gt(np.random.default_rng(int(z["gt_seed"])), bs)
    net.train()
    try:
        net.conv_dtype = "f32x3"
        net.zero_grad()
        loc, conf = net(x)
        l1, l2 = Losses.ssd((loc, conf), [_t(c) for c in classes], [_t(b) for b in boxes])
        (l1 + l2).backward()
    finally:
        net.conv_dtype = "f32"
    idx = z["prior_idx"]
    for got, ref in ((loc.detach().cpu().numpy()[:, idx], z["loc_s"]), (conf.detach().cpu().numpy()[:, idx], z["conf_s"])):
        assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    assert abs(l1.item() - float(z["loc_loss"])) <= 1e-4 * max(1, float(z["loc_loss"]))
    assert abs(l2.item() - float(z["conf_loss"])) <= 1e-4 * max(1, float(z["conf_loss"]))
    named = dict(net.named_parameters())
    for k in ("model.features.0.weight", "model.features.21.bias", "c_11_cl.weight", "seq10.2.weight", "rescaling_conv_4_3", "c_4_bb.bias"):
        ref = z["g_" + k].astype(np.float64)
        got = named[k].grad.cpu().numpy().astype(np.float64)
        assert np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-12) <= 5e-3, k


# ---- A16 / configs[4]: SSD_resnet34 (Model.py:12-126), eval-mode forward ---------------------------------------------------
def _resnet34_with_state(seed):
    from objectdetection_ssd_amd import Model
    net = Model.SSD_resnet34(20)
    state = O.ssd_resnet34_random_state(seed)
    full = dict(state)
    for alias, trunk in O.ssd_resnet34_aliases().items():
        for k in state:
            if k.startswith(trunk):
                full[alias + k[len(trunk):]] = state[k]
    net.load_state_dict(full)                       # strict: the reference's checkpoint layout
    return net.to("cuda:0").eval(), state


@pytest.mark.gpu
def test_resnet34_forward_vs_reference(gold_dir):
    """Against the reference's own eval-mode outputs (tests/golden/resnet34.npz).  BatchNorm is folded into the
    convolution weights here, so the comparison is at f32 rounding level, not bit level: 1e-4 of the output scale."""
    z = np.load(os.path.join(gold_dir, "resnet34.npz"))
    net, _ = _resnet34_with_state(int(z["state_seed"]))
    x = np.random.default_rng(int(z["x_seed"])).standard_normal((2, 3, 224, 224), dtype=np.float32)
    loc, conf = net(torch.from_numpy(x).cuda())
    assert tuple(loc.shape) == (2, 63, 4) and tuple(conf.shape) == (2, 63, 21)
    for got, ref in ((loc, z["loc"]), (conf, z["conf"])):
        err = float(np.abs(got.cpu().numpy() - ref).max())
        assert err <= 1e-4 * max(1.0, float(np.abs(ref).max())), err
    # parameter updates invalidate the folded weights
    with torch.no_grad():
        net.conv2d_02_c1.bias.add_(1.0)
    _, conf2 = net(torch.from_numpy(x).cuda())
    np.testing.assert_allclose(conf2[:, 60:].cpu().numpy(), z["conf"][:, 60:] + 1.0, rtol=0, atol=2e-3)
    np.testing.assert_allclose(conf2[:, :60].cpu().numpy(), conf[:, :60].cpu().numpy(), rtol=0, atol=0)


@pytest.mark.gpu
def test_resnet34_forward_vs_oracle_other_batch_and_bf16():
    """A different batch size / seed against the CPU oracle; bf16 operand mode stays within bf16 accuracy."""
    net, state = _resnet34_with_state(7)
    x = torch.randn(5, 3, 224, 224, generator=torch.Generator().manual_seed(8))
    with torch.no_grad():
        rl, rc = O.ssd_resnet34_forward(x, state)
    loc, conf = net(x.cuda())
    for got, ref in ((loc, rl), (conf, rc)):
        assert float((got.cpu() - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    net.conv_dtype = "bf16"
    lb, cb = net(x.cuda())
    rel = float((cb.cpu() - rc).norm() / rc.norm())
    assert 1e-5 < rel < 3e-2, rel


# ---- (f)-4: mAP evaluator (Util.py:783-885) ---------------------------------------------------------------------------------
def _map_case(z, ci):
    p = f"c{ci}_"

    def split(a, cnt):
        o = np.cumsum(np.r_[0, cnt])
        return [a[o[i]:o[i + 1]] for i in range(len(cnt))]
    dc, gc = z[p + "det_count"], z[p + "gt_count"]
    return (split(z[p + "det_boxes"], dc), split(z[p + "det_classes"], dc), split(z[p + "det_scores"], dc),
            split(z[p + "gt_boxes"], gc), split(z[p + "gt_classes"], gc)), z[p + "ap"]


@pytest.mark.parametrize("ci", range(4))
def test_get_map_vs_reference(gold_dir, ci):
    """Bit-exact against the reference's own get_map output (tests/golden/map.npz)."""
    from objectdetection_ssd_amd import Util
    z = np.load(os.path.join(gold_dir, "map.npz"))
    args, ref = _map_case(z, ci)
    det_b, det_c, det_s, gt_b, gt_c = args
    aps = Util.get_map([_t(b) for b in det_b], [torch.from_numpy(c) for c in det_c], [_t(s) for s in det_s],
                       [torch.from_numpy(b) for b in gt_b], [torch.from_numpy(c.astype(np.float32)) for c in gt_c])
    assert sorted(aps) == list(range(20)) and isinstance(aps[0], np.float64)
    assert np.array_equal(np.asarray([aps[c] for c in range(20)]), ref)


def test_get_map_large_with_ties_vs_oracle():
    """600 images x up to 200 detections with repeated scores (tie rule: lower flat index first), images without
    detections or ground truth, classes outside 0..19 ignored: TP flags, precision table and APs equal the oracle's."""
    from objectdetection_ssd_amd import ops
    rng = np.random.default_rng(99)
    n_img = 600
    gt_b, gt_c = synth_gt(rng, n_img)
    gt_c = [c.astype(np.int64) for c in gt_c]
    det_b, det_c, det_s = [], [], []
    for i in range(n_img):
        n = int(rng.integers(0, 201)) if i % 7 else 0
        k = rng.integers(0, len(gt_b[i]), n)
        b = gt_b[i][k] + rng.normal(0, .04, (n, 4)).astype(np.float32)
        b = np.stack([np.minimum(b[:, 0], b[:, 2]), np.minimum(b[:, 1], b[:, 3]),
                      np.maximum(b[:, 0], b[:, 2]) + np.float32(.01), np.maximum(b[:, 1], b[:, 3]) + np.float32(.01)], 1).astype(np.float32)
        c = np.where(rng.uniform(size=n) < .8, gt_c[i][k], rng.integers(0, 22, n)).astype(np.int64)   # some classes 20, 21
        det_b.append(b); det_c.append(c)
        det_s.append((rng.integers(1, 50, n) / np.float32(50)).astype(np.float32))                      # many ties
    gt_b[5] = np.zeros((0, 4), np.float32); gt_c[5] = np.zeros(0, np.int64)
    aps, tp_ref, table_ref = O.get_map(det_b, det_c, det_s, gt_b, gt_c, return_details=True)

    def flat(parts, dtype):
        start = torch.tensor(np.cumsum([0] + [len(p) for p in parts]), dtype=torch.int32, device=DEV)
        return torch.from_numpy(np.concatenate(parts)).to(DEV, dtype).contiguous(), start
    db, dstart = flat(det_b, torch.float32)
    dc, _ = flat(det_c, torch.int32)
    ds, _ = flat(det_s, torch.float32)
    gb, gstart = flat(gt_b, torch.float32)
    gc, _ = flat(gt_c, torch.int32)
    table, tp, counts = ops.map_eval(db, dc, ds, dstart, gb, gc, gstart, O.ap_recall_thresholds(), 20)
    assert np.array_equal(tp.cpu().numpy(), tp_ref)
    assert np.array_equal(table.cpu().numpy(), table_ref)
    dcat, gcat = np.concatenate(det_c), np.concatenate(gt_c)
    assert counts[0].tolist() == [int((dcat == c).sum()) for c in range(20)]
    assert counts[1].tolist() == [int((gcat == c).sum()) for c in range(20)]
    assert 0.05 < float(np.mean([aps[c] for c in range(20)])) < 0.95


# ---- (f)-3: input pipeline (Dataset.py:10-13,24-39; Util.py:610-749) ---------------------------------------------------------
def test_preprocess_batch_equals_pillow_resize_and_normalize():
    """Ragged batch of VOC-like sizes (down-scale, up-scale, unchanged axis): the device batch equals
    Pillow resize -> /255 -> (x-mean)/std bit for bit."""
    from PIL import Image
    from objectdetection_ssd_amd import Dataset
    rng = np.random.default_rng(21)
    shapes = [(375, 500), (500, 333), (120, 90), (300, 300), (300, 451), (900, 1300), (37, 53), (301, 299), (500, 500)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    out = Dataset.preprocess_batch(imgs)
    assert tuple(out.shape) == (len(imgs), 3, 300, 300) and out.dtype == torch.float32 and out.is_cuda
    mean = torch.tensor(Dataset.MEAN).view(3, 1, 1)
    std = torch.tensor(Dataset.STD).view(3, 1, 1)
    for i, a in enumerate(imgs):
        u8 = np.asarray(Image.fromarray(a).resize((300, 300), Image.BILINEAR))
        ref = torch.from_numpy(u8.copy()).permute(2, 0, 1).contiguous().float().div(255).sub_(mean).div_(std)
        assert torch.equal(out[i].cpu(), ref), (shapes[i], float((out[i].cpu() - ref).abs().max()))
        assert np.array_equal(O.preprocess_image(a), ref.numpy())
    small = Dataset.preprocess_batch(imgs[:3], size=(64, 48))
    assert np.array_equal(small[1].cpu().numpy(), O.preprocess_image(imgs[1], 64, 48))


def test_preprocess_batch_with_reference_geometry(gold_dir):
    """expand / crop / flip plans drawn like the reference (tests/golden/augment.npz seeds), pixels on the device:
    equal to the oracle's composition + Pillow-exact resize of the same plan."""
    import random
    from objectdetection_ssd_amd import Dataset
    z = np.load(os.path.join(gold_dir, "augment.npz"))
    imgs, plans = [], []
    for ci in range(int(z["n_cases"])):
        p = f"c{ci}_"
        img = z[p + "img"]
        random.seed(int(z[p + "seed"]))
        plan, _, _ = Dataset.plan_transform(img.shape[1], img.shape[0], torch.from_numpy(z[p + "boxes"]), torch.from_numpy(z[p + "labels"]),
                                            photometric=False)
        imgs.append(img); plans.append(plan)
    out = Dataset.preprocess_batch(imgs, plans).cpu().numpy()
    for i, (img, plan) in enumerate(zip(imgs, plans)):
        ref = O.preprocess_image(img, 300, 300, canvas=plan.canvas, crop=plan.crop, flip=plan.flip)
        assert np.array_equal(out[i], ref), i
    with pytest.raises(ValueError):
        Dataset.preprocess_batch([imgs[0]], [plans[1]])


def test_graphed_inference_forward_replays_bitwise():
    """HIP-graph capture of the inference forward: replay equals the eager forward bit for bit, follows new inputs, and the
    library calls are capture-safe (enqueue only)."""
    from objectdetection_ssd_amd import Model
    torch.manual_seed(3)
    net = Model.SSD_300().to(DEV).eval()
    x1 = torch.randn(2, 3, 300, 300, device=DEV)
    x2 = torch.randn(2, 3, 300, 300, device=DEV)
    with torch.no_grad():
        l1, c1 = net(x1)
        l2, c2 = net(x2)
    g = net.graphed_forward(x1)
    a, b = g(x1)
    assert torch.equal(a, l1) and torch.equal(b, c1)
    a, b = g(x2)
    assert torch.equal(a, l2) and torch.equal(b, c2)
    a, b = g(x1)
    assert torch.equal(a, l1) and torch.equal(b, c1)
    with pytest.raises(ValueError):
        g(torch.zeros(1, 3, 300, 300, device=DEV))


# ---- size-independent properties at the benchmark's batch (no CPU reference involved) --------------------------------------
def test_loss_is_additive_over_images_at_full_batch():
    """bs = 32: the un-normalised sums (norm_mode=1) of the batch equal the sums of its images taken alone, the positive
    counts add up, per-image classes are unchanged -- matching and mining never look across images (Losses.py:152-167)."""
    from objectdetection_ssd_amd import Losses
    rng = np.random.default_rng(321)
    boxes, classes = synth_gt(rng, 32)
    loc = _t(rng.standard_normal((32, 8732, 4), dtype=np.float32))
    conf = _t(rng.standard_normal((32, 8732, 21), dtype=np.float32) * np.float32(2))
    bx, cl = [_t(b) for b in boxes], [_t(c) for c in classes]
    l_all, c_all = Losses.ssd((loc, conf), cl, bx, norm_mode=1)
    n_all = float(Losses.last_match["n_pos"].item())
    cls_all = Losses.last_match["cls"].clone()
    s_loc = s_conf = n_sum = 0.0
    for i in range(32):
        li, ci = Losses.ssd((loc[i:i + 1].contiguous(), conf[i:i + 1].contiguous()), cl[i:i + 1], bx[i:i + 1], norm_mode=1)
        s_loc += float(li.item()); s_conf += float(ci.item()); n_sum += float(Losses.last_match["n_pos"].item())
        assert torch.equal(Losses.last_match["cls"][0], cls_all[i])
    assert n_sum == n_all
    assert abs(s_loc - l_all.item()) <= 1e-5 * abs(l_all.item()) and abs(s_conf - c_all.item()) <= 1e-5 * abs(c_all.item())


def test_padded_decode_is_the_list_form_without_the_sync():
    """`inference_batch_padded` (device tensors + counts, no host synchronisation) row by row equals `inference_batch`; rows past the
    count are zero; a device tensor of sizes is used as it is.  Two calls give identical results although the candidate keys land
    in a different order every time (slots come from atomics): the ranks are computed from the unique keys."""
    from objectdetection_ssd_amd import Losses
    g = torch.Generator().manual_seed(77)
    l = torch.randn(5, 8732, 4, generator=g).to(DEV)
    c = (3 * torch.randn(5, 8732, 21, generator=g)).to(DEV)
    c[3, :, :20] = -20.0                                                # an image without detections
    sizes = torch.tensor([[500., 375.], [300., 300.], [640., 480.], [100., 100.], [375., 500.]], device=DEV)
    b, k, p, ids, cnt = Losses.inference_batch_padded(l, c, sizes, top_k=150)
    b2, k2, p2, ids2, cnt2 = Losses.inference_batch_padded(l, c, sizes, top_k=150)
    assert torch.equal(b, b2) and torch.equal(k, k2) and torch.equal(p, p2) and torch.equal(ids, ids2) and torch.equal(cnt, cnt2)
    outs = Losses.inference_batch(l, c, sizes.cpu(), top_k=150)
    assert cnt.tolist()[3] == 0 and outs[3] == ([], [], [])
    for i, n in enumerate(cnt.tolist()):
        if n:
            assert torch.equal(b[i, :n], outs[i][0]) and torch.equal(k[i, :n], outs[i][1]) and torch.equal(p[i, :n], outs[i][2])
        assert bool((b[i, n:] == 0).all()) and bool((p[i, n:] == 0).all())


def test_decode_output_invariants_at_full_batch():
    """32 images x 8732 priors through the batched decode: every kept score >= min_score, classes 0..19 in class-major order,
    scores descending inside a class, at most top_k rows, and no two kept boxes of one class overlap by >= the NMS threshold
    (IoU recomputed by the oracle's bit-exact formula on the returned pixel boxes is only approximate, hence the 1e-3 slack)."""
    from objectdetection_ssd_amd import Losses
    g = torch.Generator().manual_seed(77)
    l = (torch.randn(32, 8732, 4, generator=g) * 0.5).to(DEV)
    c = (torch.randn(32, 8732, 21, generator=g) * 3).to(DEV)
    outs = Losses.inference_batch(l, c, [(500, 375)] * 32, top_k=200, min_score=0.2, iou_threshold=0.45)
    assert len(outs) == 32
    seen = 0
    for boxes, classes, probs in outs:
        if isinstance(boxes, list):
            continue
        seen += 1
        b, k, p = boxes.cpu().numpy(), classes.cpu().numpy(), probs.cpu().numpy()
        assert len(k) <= 200 and (p >= 0.2).all() and ((k >= 0) & (k <= 19)).all()
        capped = len(k) == 200
        if not capped:
            assert (np.diff(k) >= 0).all()                                   # class-major
        for cc in np.unique(k):
            idx = np.nonzero(k == cc)[0]
            if not capped:
                assert (np.diff(p[idx]) <= 0).all()                          # descending inside the class
            if idx.size > 1:
                frac = b[idx] / np.float32([500, 375, 500, 375])
                iou = O.iou_matrix(frac, frac)
                np.fill_diagonal(iou, 0)
                assert float(iou.max()) < 0.45 + 1e-3
    assert seen >= 30


def test_dataset_class_end_to_end_on_image_files(tmp_path):
    """`Dataset.MultiImageMultiBBoxDataset` + `collate_fn` with the reference's constructor arguments (Dataset.py:7-53) on
    real image files: isTest=True items give the Pillow-resized, normalised batch and boxes divided by (w,h,w,h); difficult
    objects are dropped unless keep_difficult; isTest=False items carry a drawn geometry whose pixels equal the oracle's."""
    import random
    from PIL import Image
    from objectdetection_ssd_amd import Dataset
    rng = np.random.default_rng(5)
    paths, sizes = [], [(120, 90), (64, 200), (300, 300)]
    for i, (h, w) in enumerate(sizes):
        pth = str(tmp_path / f"im{i}.png")
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(pth)
        paths.append(pth)
    bboxes = [[[10., 20., 60., 80.], [5., 5., 30., 40.]], [[0., 0., 100., 50.]], [[30., 40., 200., 250.], [100., 100., 280., 290.]]]
    labels = [["dog", "cat"], ["person"], ["car", "bus"]]
    difficult = [[0, 1], [0], [0, 0]]
    ds = Dataset.MultiImageMultiBBoxDataset(paths, bboxes, labels, difficult, [7, 8, 9], isTest=True)
    items = [ds[i] for i in range(3)]
    raw, classes, boxes, idx = Dataset.collate_fn(items)
    assert isinstance(raw, Dataset.RawBatch) and not raw.arena.is_cuda and raw.shape[0] == raw.size(0) == 3
    x = raw.to(DEV)                                                               # train_function.py:61
    assert idx == [7, 8, 9] and tuple(x.shape) == (3, 3, 300, 300) and x.is_cuda
    assert classes[0].tolist() == [float(Dataset.label_to_class["dog"])]          # the difficult cat is dropped
    mean = torch.tensor(Dataset.MEAN).view(3, 1, 1)
    std = torch.tensor(Dataset.STD).view(3, 1, 1)
    for i, (h, w) in enumerate(sizes):
        u8 = np.asarray(Image.open(paths[i]).convert("RGB").resize((300, 300), Image.BILINEAR))
        ref = torch.from_numpy(u8.copy()).permute(2, 0, 1).contiguous().float().div(255).sub_(mean).div_(std)
        assert torch.equal(x[i].cpu(), ref)
        keep = [b for b, d in zip(bboxes[i], difficult[i]) if d == 0]
        assert torch.equal(boxes[i], torch.tensor(keep) / torch.tensor([w, h, w, h], dtype=torch.float32))
    kd = Dataset.MultiImageMultiBBoxDataset(paths, bboxes, labels, difficult, [7, 8, 9], isTest=True, keep_difficult=True)
    assert kd[0][1].numel() == 2
    tr = Dataset.MultiImageMultiBBoxDataset(paths, bboxes, labels, difficult, [7, 8, 9], isTest=False)
    random.seed(11)
    items = [tr[i] for i in range(3)]
    x2, _, boxes2, _ = Dataset.collate_fn(items)
    x2 = x2.to(DEV)
    for i, it in enumerate(items):
        plan = it[0].plan
        ref = O.preprocess_image(O.photometric_apply(it[0].pixels, plan.photo), 300, 300, canvas=plan.canvas, crop=plan.crop, flip=plan.flip)
        assert np.array_equal(x2[i].cpu().numpy(), ref)
        assert boxes2[i].shape[1] == 4 and boxes2[i].shape[0] == it[1].shape[0]


def test_dataloader_with_the_references_arguments_and_worker_processes(tmp_path):
    """train.py:29,40: `DataLoader(ds, batch_size=.., shuffle=True, num_workers=2, collate_fn=collate_fn)` AFTER the model is on
    the GPU (train.py:43 comes later in the file, but train_model iterates the loaders with HIP long initialised).  `collate_fn`
    runs in the forked workers and must stay host-only; `inputs.to(device)` in the main process renders the batch.  Every
    batch equals the single-process result for the same plans."""
    from PIL import Image
    from objectdetection_ssd_amd import Dataset
    torch.zeros(1, device=DEV)                                                    # HIP is up in the parent before the fork
    rng = np.random.default_rng(9)
    paths, n = [], 6
    for i in range(n):
        h, w = int(rng.integers(40, 120)), int(rng.integers(40, 120))
        pth = str(tmp_path / f"w{i}.png")
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(pth)
        paths.append(pth)
    bboxes = [[[2., 3., 30., 35.]] for _ in range(n)]
    labels = [["dog"] for _ in range(n)]
    ds = Dataset.MultiImageMultiBBoxDataset(paths, bboxes, labels, [[0]] * n, list(range(n)))
    dl = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=True, num_workers=2, collate_fn=Dataset.collate_fn)
    seen = []
    for inputs, classes, boxes, indices in dl:
        assert isinstance(inputs, Dataset.RawBatch) and not inputs.arena.is_cuda
        plans = inputs.plans
        x = inputs.to(DEV)
        assert x.is_cuda and tuple(x.shape) == (2, 3, 300, 300) and bool(torch.isfinite(x).all())
        again = Dataset.preprocess_batch([np.asarray(Image.open(paths[i]).convert("RGB")) for i in indices], plans)
        assert torch.equal(x, again)
        assert all(b.shape[1] == 4 for b in boxes) and len(classes) == 2
        seen += indices
    assert sorted(seen) == list(range(n))


def test_photometric_kernels_equal_pillow_arithmetic():
    """photometric_distort on the device: every op alone at the extremes of its factor range and random 1-4 op sequences in
    random order, on ragged images, bit-equal to the oracle (which tests/test_oracle_golden.py pins on Pillow itself)."""
    from objectdetection_ssd_amd import Dataset
    rng = np.random.default_rng(31)
    imgs, plans, refs = [], [], []
    singles = [(0, .5), (0, 1.5), (0, 1.0), (1, .5), (1, 1.5), (2, .5), (2, 1.5), (2, .999), (3, -18 / 255.), (3, 18 / 255.), (3, 0.0), (3, .031)]
    for i in range(40):
        h, w = int(rng.integers(7, 90)), int(rng.integers(7, 90))
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if i % 6 == 0:
            a = (a // 16 + 120).astype(np.uint8)                     # low contrast
        if i % 9 == 0:
            a[..., 1] = a[..., 0]; a[..., 2] = a[..., 0]              # grey: s == 0 branch of the hue path
        if i < len(singles):
            ops_ = (singles[i],)
        else:
            kinds = rng.permutation(4)[: int(rng.integers(1, 5))]
            ops_ = tuple((int(k), float(rng.uniform(-18 / 255., 18 / 255.)) if k == 3 else float(rng.uniform(.5, 1.5))) for k in kinds)
        plan = Dataset.identity_plan(h, w)
        plan.photo = ops_
        imgs.append(a); plans.append(plan)
        refs.append(O.preprocess_image(O.photometric_apply(a, ops_), 64, 64))
    out = Dataset.preprocess_batch(imgs, plans, size=(64, 64)).cpu().numpy()
    for i in range(len(imgs)):
        assert np.array_equal(out[i], refs[i]), (i, plans[i].photo)


def test_weight_cache_is_refreshed_in_training_and_invalidated_on_request():
    """The re-laid weight copies are keyed on (data_ptr, _version), which a write through `.data` does not change.  A training
    forward therefore never trusts the cache; an inference forward does until `invalidate_weight_cache()` (or any autograd-visible
    in-place update) -- documented on the method."""
    from objectdetection_ssd_amd import Model
    torch.manual_seed(3)
    net = Model.SSD_300().to(DEV)
    x = torch.randn(1, 3, 300, 300, device=DEV)
    p = net.model.features[2].weight                                    # conv1_2 (a Winograd layer: transformed filters are cached)
    q = net.conv_fc7.weight                                             # a direct-kernel layer
    net.train()
    loc0, _ = net(x)
    for t in (p, q):
        ptr, ver = t.data_ptr(), t._version
        t.data.mul_(0.5)
        assert (t.data_ptr(), t._version) == (ptr, ver)                 # invisible to the cache key
        loc1, _ = net(x)
        assert float((loc1 - loc0).abs().max()) > 1e-3, "training forward used stale weight layouts"
        loc0 = loc1
    net.eval()
    with torch.no_grad():
        e0, _ = net(x)
        p.data.mul_(2.0)
        net.invalidate_weight_cache()
        e1, _ = net(x)
        assert float((e1 - e0).abs().max()) > 1e-3
        q.mul_(2.0)                                                     # autograd-visible in-place op: picked up by itself
        e2, _ = net(x)
        assert float((e2 - e1).abs().max()) > 1e-3
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        sd["model.features.2.weight"] = sd["model.features.2.weight"] * 0.25
        sd["conv_4_3.2.weight"] = sd["model.features.2.weight"]
        net.load_state_dict(sd)
        e3, _ = net(x)
        assert float((e3 - e2).abs().max()) > 1e-3


def test_ops_refuse_tensors_of_another_device_than_the_current_one():
    from objectdetection_ssd_amd import ops
    if torch.cuda.device_count() < 2:
        # one-GPU box: the guard compares with the current device index; fake a mismatch through the hook it uses
        saved = ops._current_device
        ops._current_device = lambda: 1
        try:
            with pytest.raises(RuntimeError, match="current device"):
                ops.l2norm_fwd(torch.zeros(4, 512, device=DEV), torch.ones(512, device=DEV))
        finally:
            ops._current_device = saved
    else:
        with pytest.raises(RuntimeError, match="current device"):
            ops.l2norm_fwd(torch.zeros(4, 512, device="cuda:1"), torch.ones(512, device="cuda:1"))


def test_data_parallel_optimizer_checkpoint_roundtrip_with_the_real_kernel(tmp_path):
    """FlatSGDDataParallel as the caller's optimizer (train_function.py:27-30,76,95,116; train.py:57): param_groups in train.py's
    order, a StepLR on top, state_dict -> torch.save -> fresh model + optimizer -> load_state_dict, lr forced back; the step
    after the restore equals the uninterrupted run BIT FOR BIT (momentum restored into the flat buffer)."""
    from objectdetection_ssd_amd import Losses, Model
    from objectdetection_ssd_amd.ddp import FlatSGDDataParallel
    lr = 1e-3
    x = _t(np.random.default_rng(71).standard_normal((2, 3, 300, 300), dtype=np.float32))
    boxes, classes = synth_gt(np.random.default_rng(72), 2)
    cl = [_t(c) for c in classes]; bx = [_t(b) for b in boxes]
    params = O.ssd300_random_params(9)

    def make():
        n = Model.SSD_300(); _load_params(n, params); n = n.to(DEV).train()
        return n, FlatSGDDataParallel(n, lr=lr, momentum=0.9, weight_decay=5e-4)

    def step(n, o):
        o.zero_grad()
        l1, l2, n_pos = Losses.ssd(n(x), cl, bx, norm_mode=1, with_n_pos=True)
        (l1 + l2).backward()
        o.reduce_gradients(n_pos)
        o.step()

    a, oa = make()
    assert isinstance(oa, torch.optim.Optimizer) and [g["lr"] for g in oa.param_groups] == [2 * lr, lr]
    assert len(oa.param_groups[0]["params"]) == 35 and len(oa.param_groups[1]["params"]) == 36      # biases | weights + the L2-norm scale
    sched = torch.optim.lr_scheduler.StepLR(oa, step_size=1, gamma=0.5)
    for _ in range(2):
        step(a, oa)
        sched.step()
    assert abs(oa.param_groups[1]["lr"] - lr / 4) < 1e-12
    torch.save({"cnn_state_dict": a.state_dict(), "optimizer_state_dict": oa.state_dict()}, tmp_path / "ck.pt")
    b, ob = make()
    ck = torch.load(tmp_path / "ck.pt", weights_only=True)
    b.load_state_dict(ck["cnn_state_dict"])
    ob.load_state_dict(ck["optimizer_state_dict"])
    assert torch.equal(ob.flat_param, oa.flat_param) and torch.equal(ob.flat_mom, oa.flat_mom)
    for o in (oa, ob):
        for g in o.param_groups:
            g["lr"] = lr                                                # train_function.py:29-30
    step(a, oa)
    step(b, ob)
    assert torch.equal(oa.flat_param, ob.flat_param) and torch.equal(oa.flat_mom, ob.flat_mom)
    c, oc = make()                                                      # control: without the momentum the runs differ
    c.load_state_dict(ck["cnn_state_dict"])
    step(c, oc)
    assert not torch.equal(oc.flat_param, oa.flat_param)


def _dp_rank(rank, world, port, q):
    """one rank of the two-rank rehearsal below: the REAL model and engine on cuda:0, torch.distributed over gloo"""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from objectdetection_ssd_amd import Losses, Model
        from objectdetection_ssd_amd.ddp import FlatSGDDataParallel
        torch.cuda.set_device(0)
        net = Model.SSD_300()
        _load_params(net, O.ssd300_random_params(11 + rank))            # ranks start different: the broadcast must fix that
        net = net.to(DEV).train()
        dp = FlatSGDDataParallel(net, lr=1e-3, momentum=0.9, weight_decay=5e-4, overlap=(rank >= 0 and os.environ.get("DP_OVERLAP") == "1"))
        dp.broadcast_parameters(0)
        if os.environ.get("DP_SIDE_DELAY") == "1":
            # ~20 ms of nothing on the engine's side stream in front of the tail group's deferred weight gradients: a collective
            # that did not wait for them would read the flat buffer long before they are written
            net._engine._test_side_delay = lambda: torch.cuda._sleep(40_000_000)
        x = np.random.default_rng(81).standard_normal((4, 3, 300, 300), dtype=np.float32)
        boxes, classes = synth_gt(np.random.default_rng(82), 4)
        sl = slice(2 * rank, 2 * rank + 2)
        for _ in range(2):
            dp.zero_grad()
            l1, l2, n_pos = Losses.ssd(net(_t(x[sl])), [_t(c) for c in classes[sl]], [_t(b) for b in boxes[sl]], norm_mode=1, with_n_pos=True)
            (l1 + l2).backward()
            dp.reduce_and_step(n_pos)
        torch.cuda.synchronize()
        q.put((rank, dp.flat_param.cpu().numpy(), float(dp.flat_grad[dp.n].item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_two_rank_data_parallel_step_of_the_real_engine_equals_the_global_batch_step(overlap):
    """N > 1 with the real model: two processes (both on cuda:0, gloo between them -- the one-GPU box's rehearsal of the RCCL
    run), each with half of a global batch of 4, two steps of forward + un-normalised loss + backward + ONE all-reduce of the
    flat buffer (or its overlapped slices) + fused SGD.  Both ranks must end with identical parameters, equal to a single
    process stepping on the whole batch with the reference's normalisation (train.py's SGD groups), and n_pos must be global."""
    import socket
    import torch.multiprocessing as mp
    from objectdetection_ssd_amd import Losses, Model
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    os.environ["DP_OVERLAP"] = "1" if overlap else "0"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        out = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    finally:
        for p in procs:
            p.join(timeout=120)
    assert all(p.exitcode == 0 for p in procs)
    assert np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]
    # single process, global batch, reference normalisation + torch.optim.SGD with train.py's groups
    ref = Model.SSD_300(); _load_params(ref, O.ssd300_random_params(11)); ref = ref.to(DEV).train()
    biases, others = _sgd_groups(ref.named_parameters())
    opt = torch.optim.SGD([{"params": biases, "lr": 2e-3}, {"params": others}], lr=1e-3, momentum=0.9, weight_decay=5e-4)
    x = np.random.default_rng(81).standard_normal((4, 3, 300, 300), dtype=np.float32)
    boxes, classes = synth_gt(np.random.default_rng(82), 4)
    for _ in range(2):
        opt.zero_grad()
        l1, l2 = Losses.ssd(ref(_t(x)), [_t(c) for c in classes], [_t(b) for b in boxes])
        (l1 + l2).backward()
        opt.step()
    assert out[0][2] == float(Losses.last_match["n_pos"].item())        # the extra slot carried the GLOBAL positive count
    from objectdetection_ssd_amd.ddp import FlatSGDDataParallel
    probe = FlatSGDDataParallel(ref, lr=1e-3)                           # lays the reference weights out in the same flat order
    got, want = torch.from_numpy(out[0][1]), probe.flat_param.cpu()
    err = float((got - want).abs().max())
    assert err <= 2e-5 * max(1.0, float(want.abs().max())), err


def test_overlapped_exchange_waits_for_gradients_written_on_the_side_stream():
    """ADVICE (round 2): with the tail group on a second stream, a 16 MB bucket mixes gradients of the caller's stream (conv_fc7)
    with gradients the side stream writes later (c_8, seq8).  The listener must hear of the latter only once the caller's stream has
    waited for them.  Two ranks, overlapped exchange, the side stream held up by 20 ms in front of its deferred weight gradients:
    parameters after two steps are bit-identical to the single all-reduce run (two-rank sums commute)."""
    import socket
    import torch.multiprocessing as mp
    res = {}
    for mode, env in (("plain", {"DP_OVERLAP": "0", "DP_SIDE_DELAY": "0"}), ("overlap_delayed", {"DP_OVERLAP": "1", "DP_SIDE_DELAY": "1"})):
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
        os.environ.update(env)
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_dp_rank, args=(r, 2, port, q)) for r in range(2)]
        for p in procs:
            p.start()
        try:
            out = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
        finally:
            for p in procs:
                p.join(timeout=120)
        assert all(p.exitcode == 0 for p in procs)
        assert np.array_equal(out[0][1], out[1][1])
        res[mode] = out[0][1]
    os.environ["DP_SIDE_DELAY"] = "0"
    assert np.array_equal(res["plain"], res["overlap_delayed"])


@pytest.mark.parametrize("form", ["adjoint", "rotated + weight-gradient stream"])
def test_second_stream_schedule_is_bitwise_the_single_stream_step(form):
    """(form: the default engine, whose >= 256-channel data gradients take the adjoint Winograd form and share the weight gradient's
    planes -- there is no third stream then -- and the round-3 engine with the rotated-filter form and the weight-gradient stream.)
    The engine runs the tiny-map group (c_8, seq9 ... c_11: ~60 latency-bound launches per direction) on a second HIP stream beside
    the c_4 / c_7 head convolutions, forward and backward, and the Winograd weight-gradient GEMMs (which nothing in the backward pass
    waits for) on a third beside the data-gradient chain.  Same kernels, same operands, only the stream differs: outputs, losses and
    every gradient must be bit-identical to the single-stream schedule, three steps in a row (events order every cross-stream use)."""
    import grad_measure as M
    from objectdetection_ssd_amd import Model
    torch.manual_seed(11)
    net = Model.SSD_300().to(DEV)
    net._engine.adjoint_dgrad = form == "adjoint"
    x, cl, bx = M.bench_batch(bs=4, seed=77)
    res = {}
    for overlap in (True, False, True):
        net._engine.overlap_tail = overlap
        net._engine.overlap_wgrad = overlap and form != "adjoint"      # ... and the Winograd weight-gradient GEMMs on a third one
        net._engine.batch_weights = overlap                  # ... and all filter transforms of the step in one launch vs layer by layer
        steps = [M.train_step(net, x, cl, bx) for _ in range(3)]
        for a in steps[1:]:
            assert torch.equal(a[0], steps[0][0]) and torch.equal(a[1], steps[0][1])
            assert all(torch.equal(a[4][k], steps[0][4][k]) for k in steps[0][4])
        res.setdefault(overlap, steps[0])
    net._engine.overlap_tail = net._engine.batch_weights = True
    net._engine.overlap_wgrad = False
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3]
    assert set(a[4]) == set(b[4]) and len(a[4]) == 71
    for k in a[4]:
        assert torch.equal(a[4][k], b[4][k]), k
    net.eval()
    with torch.no_grad():
        net._engine.overlap_tail = False
        l0, c0 = net(x)
        net._engine.overlap_tail = True
        l1, c1 = net(x)
    assert torch.equal(l0, l1) and torch.equal(c0, c1)


def test_pool_gradient_formed_inside_the_dy_pass_is_bitwise_the_scattered_one():
    """Behind conv1_2 / conv2_2 / conv3_3 (fused conv -> ReLU -> 2x2 pool, Model.py:135-137) the engine never writes the pool's input
    gradient: the layer's Winograd dy pass reads the pooled gradient, the argmax codes and the pooled output.  Same values, same
    arithmetic: every output and gradient of a train step equals the engine with the scatter kernel, bit for bit; a layer whose
    weights need no gradient falls back to the scatter and still gives the same data gradients."""
    import grad_measure as M
    from objectdetection_ssd_amd import Model
    torch.manual_seed(12)
    net = Model.SSD_300().to(DEV)
    x, cl, bx = M.bench_batch(bs=2, seed=78)
    res = {}
    for lazy in (True, False):
        net._engine.lazy_pool_grad = lazy
        res[lazy] = M.train_step(net, x, cl, bx)
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and len(a[4]) == 71
    for k in a[4]:
        assert torch.equal(a[4][k], b[4][k]), k
    frozen = ["model.features.2.weight", "model.features.2.bias"]                 # conv1_2 frozen: its dy must be materialised after all
    P = dict(net.named_parameters())
    for nme in frozen:
        P[nme].requires_grad_(False)
    net._engine.lazy_pool_grad = True
    c = M.train_step(net, x, cl, bx)
    for nme in frozen:
        P[nme].requires_grad_(True)
    for k in c[4]:
        assert torch.equal(c[4][k], a[4][k]), k
    assert not any(k in c[4] for k in frozen)


def test_ssd512_at_its_per_gpu_batch_vs_oracle_and_direct_engine():
    """BASELINE configs[3] (build-defined SSD512, 16 images per GPU) at that batch: the forward + loss of the default engine against the
    oracle's restatement on the CPU (loc / conf / both losses 1e-4, per-prior classes bit-exact; parity UNPINNED by the reference,
    which has no SSD512) and one full train step against the direct exact-f32 engine (outputs 1e-4, every gradient within 5e-3
    relative L2 -- no per-tensor table exists for this geometry)."""
    import grad_measure as M
    from objectdetection_ssd_amd import Losses, Model
    bs = 16
    params = O.ssd300_random_params(8, variant=512)
    net = Model.SSD_512()
    _load_params(net, params)
    net = net.to(DEV)
    g = torch.Generator().manual_seed(512)
    x = torch.randn(bs, 3, 512, 512, generator=g)
    boxes, classes = synth_gt(np.random.default_rng(513), bs)
    pri = O.create_priors_ssd512()
    torch.set_num_threads(16)
    with torch.no_grad():
        lo, co = O.ssd300_forward(x, params, variant=512)
    ref = O.multibox_loss(lo.numpy(), co.numpy(), boxes, classes, pri_cxcywh=pri, want_grads=False)
    cl, bx = [_t(c) for c in classes], [_t(b) for b in boxes]
    res = {}
    for eng in M.ENGINES:
        M.set_engine(net, eng)
        res[eng] = M.train_step(net, x.to(DEV), cl, bx) + (Losses.last_match["cls"].cpu().numpy(),)
    M.set_engine(net, "wino")
    for eng in M.ENGINES:
        loc, conf, l1, l2, _, cls = res[eng]
        assert tuple(loc.shape) == (bs, 24564, 4)
        assert float((loc.cpu() - lo).abs().max()) <= 1e-4 * max(1, float(lo.abs().max())), eng
        assert float((conf.cpu() - co).abs().max()) <= 1e-4 * max(1, float(co.abs().max())), eng
        assert abs(l1 - float(ref["loc_loss"])) <= 1e-4 * max(1, float(ref["loc_loss"])), eng
        assert abs(l2 - float(ref["conf_loss"])) <= 1e-4 * max(1, float(ref["conf_loss"])), eng
        assert np.array_equal(cls, ref["cls"])
    ga, gb = res["wino"][4], res["direct"][4]
    assert set(ga) == set(gb) and len(ga) == 79
    bad = [(k, M.rel_l2(ga[k], gb[k])) for k in ga if M.rel_l2(ga[k], gb[k]) > 5e-3]
    assert not bad, bad


def test_resnet34_eval_forward_at_its_per_gpu_batch_vs_oracle():
    """BASELINE configs[4] first half (SSD_resnet34, 32 images per GPU, eval mode as section 8 A16 scopes it) against the oracle, 1e-4."""
    net, state = _resnet34_with_state(9)
    x = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(10))
    torch.set_num_threads(16)
    with torch.no_grad():
        rl, rc = O.ssd_resnet34_forward(x, state)
        loc, conf = net(x.to(DEV))
    assert tuple(loc.shape) == (32, 63, 4) and tuple(conf.shape) == (32, 63, 21)
    for got, ref in ((loc, rl), (conf, rc)):
        assert float((got.cpu() - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))



def test_overflowing_step_is_non_finite_in_both_gemm_forms():
    """Step-level pin of csrc/gemm_x3.hip's non-finite rule (round-3 review, weak 3): an input large enough to overflow the trunk's
    activations gives non-finite outputs and losses through the default engine (limb GEMMs: NaN) AND through SSD_WINO_X3=0 (f32 MFMA:
    inf / NaN) -- neither form reports a finite loss for a diverged step -- while an ordinary input is finite and within 1e-4 in both."""
    from objectdetection_ssd_amd import Losses, Model, _lib
    lib = _lib.load()
    net = Model.SSD_300()
    _load_params(net, O.ssd300_random_params(8))
    net = net.to(DEV).train()
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 3, 300, 300), dtype=np.float32)
    boxes, classes = synth_gt(np.random.default_rng(6), 1)
    cl, bx = [_t(c) for c in classes], [_t(b) for b in boxes]
    res = {}
    try:
        for mode in (1, 0):
            _lib.check(lib.ssd_tune_set_wino_x3(mode), "tune")
            net.invalidate_weight_cache()
            with torch.no_grad():
                fin = Losses.ssd(net(_t(x)), cl, bx)
                big = x.copy()
                big[0, :, 100:140, 100:140] = 3.0e37            # conv1_1 / conv1_2 stay finite, the 256+-channel Winograd layers overflow
                loc, conf = net(_t(big))
                hot = Losses.ssd((loc, conf), cl, bx)
            res[mode] = (float(fin[0]), float(fin[1]), bool(torch.isfinite(loc).all() and torch.isfinite(conf).all()),
                         float(hot[0]), float(hot[1]))
    finally:
        _lib.check(lib.ssd_tune_set_wino_x3(-1), "tune")
        net.invalidate_weight_cache()
    for mode in (1, 0):
        f0, f1, outs_finite, h0, h1 = res[mode]
        assert np.isfinite(f0) and np.isfinite(f1)
        assert not outs_finite, f"mode {mode}: overflowing activations left finite outputs"
        assert not (np.isfinite(h0) and np.isfinite(h1)), (mode, h0, h1)
    assert abs(res[1][0] - res[0][0]) <= 1e-4 * max(1, abs(res[0][0])) and abs(res[1][1] - res[0][1]) <= 1e-4 * max(1, abs(res[0][1]))


@pytest.mark.parametrize("conv_dtype,two_streams", [("f32", True), ("f32", False), ("bf16", True)])
def test_graphed_train_step_is_bitwise_the_eager_step(conv_dtype, two_streams):
    """ddp.GraphedTrainStep (the captured form of train_function.py:80-95's loop body): forward + MultiBox loss + backward + fused SGD
    replayed from ONE HIP graph must leave the weights, the momentum and the loss sums bit-identical to the eager step, step after
    step, on batches whose ground-truth counts differ from the captured one (the offsets are data, not shape)."""
    from objectdetection_ssd_amd import Losses, Model
    from objectdetection_ssd_amd.ddp import FlatSGDDataParallel, GraphedTrainStep
    lr, bs = 1e-3, 2
    params = O.ssd300_random_params(8)
    nets, trs = [], []
    for _ in range(2):
        n = Model.SSD_300(); _load_params(n, params); n = n.to(DEV).train()
        n.conv_dtype = conv_dtype
        nets.append(n)
        trs.append(FlatSGDDataParallel(n, lr=lr, momentum=0.9, weight_decay=5e-4))
    gstep = GraphedTrainStep(nets[1], trs[1], max_boxes_per_image=8, warmup=2, two_streams=two_streams)
    for it in range(6):
        x = _t(np.random.default_rng(100 + it).standard_normal((bs, 3, 300, 300), dtype=np.float32))
        boxes, classes = synth_gt(np.random.default_rng(200 + it), bs)
        cl, bx = [_t(c) for c in classes], [_t(b) for b in boxes]
        if it == 4:                                          # ground truth handed over on the host: the pinned staging path
            cl, bx = [torch.from_numpy(c) for c in classes], [torch.from_numpy(b) for b in boxes]
        trs[0].zero_grad()
        l1, l2, n_pos = Losses.ssd(nets[0](x), [_t(c) for c in classes], [_t(b) for b in boxes], norm_mode=1, with_n_pos=True)
        (l1 + l2).backward()
        trs[0].reduce_and_step(n_pos)
        g1, g2, gn = gstep(x, cl, bx)
        torch.cuda.synchronize()
        assert float(g1) == float(l1) and float(g2) == float(l2) and float(gn) == float(n_pos), (it, float(g1), float(l1))
        assert torch.equal(trs[0].flat_param, trs[1].flat_param), f"weights differ after step {it}"
        assert torch.equal(trs[0].flat_mom, trs[1].flat_mom), it
        assert trs[0].steps == trs[1].steps == it + 1
    assert gstep.graph is not None and gstep._calls == 6
    if gstep.kernel_nodes is not None:
        assert 100 <= gstep.kernel_nodes <= 400, gstep.kernel_nodes
    # a changed learning rate is a changed graph: captured again, still bit-identical
    for tr in trs:
        tr.lr = 5e-4
    x = _t(np.random.default_rng(300).standard_normal((bs, 3, 300, 300), dtype=np.float32))
    boxes, classes = synth_gt(np.random.default_rng(301), bs)
    cl, bx = [_t(c) for c in classes], [_t(b) for b in boxes]
    trs[0].zero_grad()
    l1, l2, n_pos = Losses.ssd(nets[0](x), cl, bx, norm_mode=1, with_n_pos=True)
    (l1 + l2).backward()
    trs[0].reduce_and_step(n_pos)
    gstep(x, cl, bx)
    torch.cuda.synchronize()
    assert torch.equal(trs[0].flat_param, trs[1].flat_param)
    # too many boxes for the static buffers is an error, not a silent truncation
    many = [np.tile(b, (9, 1)) for b in boxes]
    with pytest.raises(ValueError):
        gstep(x, [_t(np.tile(c, 9)) for c in classes], [_t(b) for b in many])


def test_bf16_k64_persistent_kernel_equals_the_general_kernel():
    """conv3x3_bf16_k64_kernel (inline-asm fragment reads, guarded at build time by asm_guard.py) against the general LDS-DMA kernel on
    conv1_2's shape class, forward and data gradient: the same products in the same k order -- bit-identical bf16 outputs."""
    from objectdetection_ssd_amd import _lib, ops
    lib = _lib.load()
    gen = torch.Generator().manual_seed(3)
    x = (torch.randn(2, 60, 70, 64, generator=gen)).to(DEV).to(torch.bfloat16)
    w = (torch.randn(64, 64, 3, 3, generator=gen) / 24.0)
    wf = w.permute(0, 2, 3, 1).reshape(64, 9, 64).contiguous().to(DEV).to(torch.bfloat16)
    wb = w.permute(1, 2, 3, 0).reshape(64, 9, 64).contiguous().to(DEV).to(torch.bfloat16)
    bias = torch.randn(64, generator=gen).to(DEV)
    outs = {}
    try:
        for k64 in (1, 0):
            _lib.check(lib.ssd_tune_set_conv_bf16_k64(k64), "tune")
            y = ops.conv3x3_bf16(x, wf, bias, 64, True)
            dx = ops.conv3x3_bf16(y, wb, None, 64, False, flip=True, relu_mask=x)
            outs[k64] = (y.clone(), dx.clone())
    finally:
        _lib.check(lib.ssd_tune_set_conv_bf16_k64(1), "tune")
    assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][1], outs[0][1])



def test_bf16_train_step_vs_decision_and_rounding_pinned_f64_oracle(golden_net):
    """The whole-step check of the bf16-tensor mode (BASELINE configs[2]) that does not depend on a table measured with the code under
    test (round-3 review, weak 1).  The f64 oracle with bf16 operand rounding is run FOLLOWING the HIP step's discrete choices: ReLU
    masks, max-pool arg-max codes, hard negatives -- and the bf16 (or f32) VALUE every activation and every activation gradient was
    stored as: each layer of the oracle is fed exactly what the HIP layer was fed.  Then
      (a) layer by layer, the oracle's own result must round to what the HIP kernel stored: within half a bf16 spacing (one
          round-to-nearest; conv4_3's gradient has two contributions and two roundings, each followed and checked on its own) -- a
          per-layer check of every forward and data-gradient kernel that no flip upstream can blur;
      (b) all 71 parameter gradients are within ONE fixed relative-L2 bar, 1e-4, of the f64 result (what is left is f32 summation order
          inside one layer); the losses within 1e-4."""
    import grad_measure as M
    net, params, _ = golden_net
    assert net._engine.bf16_tensors
    x, boxes, classes = M.f64_case()
    xd, cl, bx = M._t(x), [M._t(c) for c in classes], [M._t(b) for b in boxes]
    decisions, neg, pinned, (l1, l2), grads = M.gpu_pinned_step(net, xd, cl, bx, "bf16")
    assert {"a1_1", "a1_2", "p1", "a4_3", "n4_3", "p5"} <= pinned["bf16"] and "a6" in pinned["fwd"] and "a6" not in pinned["bf16"]
    assert set(pinned["bwd"]) >= set(pinned["fwd"]) - {"a1_1"} or "a1_1" in pinned["bwd"]
    a1, a2, g64 = M.f64_rounding_pinned_grads(params, decisions, neg, pinned)
    rep = pinned["report"]
    assert len(rep) >= 2 * len(pinned["fwd"]) - 2, sorted(rep)
    worst = sorted(((v, k) for k, v in rep.items()), reverse=True)
    print("rounding-pinned layer distances (spacings): " + ", ".join(f"{k} {v:.3f}" for v, k in worst[:8]))
    assert "a4_3:1" in pinned["bwd"] and "a4_3:1:bwd" in rep
    bad = [(k, v) for k, v in rep.items() if v > (0.51 if k.split(":")[0] in pinned["bf16"] else 1.0)]
    assert not bad, bad
    assert abs(l1 - a1) <= 1e-4 * max(1, a1) and abs(l2 - a2) <= 1e-4 * max(1, a2), (l1, a1, l2, a2)
    assert len(g64) == 71 and set(g64) == set(grads)
    rows = sorted(((M.rel_l2(grads[k], g64[k]), k) for k in g64), reverse=True)
    print("rounding-pinned bf16 gradient distance: worst " + ", ".join(f"{k} {v:.2e}" for v, k in rows[:6]) + f"; median {rows[35][0]:.2e}")
    bad = [(k, v) for v, k in rows if v > M.BF16_PINNED_BAR]
    assert not bad, bad



def test_first_layer_into_planes_is_bitwise_the_two_kernel_step():
    """`_Engine.first_wino` (round 4): conv1_1 writes conv1_2's Winograd input planes and ReLU bits itself and its activation tensor is never
    stored.  Same arithmetic, same order: outputs, losses and all 71 gradients of a train step equal the engine with the two separate
    kernels bit for bit (batch 3 of 300 x 300), and so does the inference forward."""
    import grad_measure as M
    from objectdetection_ssd_amd import Model, ops
    from objectdetection_ssd_amd.Model import _Elided
    if not ops.has_experimental():
        pytest.skip("conv_first_wino_kernel is compiled only with SSD_EXPERIMENTAL=1 (off by default: measured slower)")
    torch.manual_seed(14)
    net = Model.SSD_300().to(DEV)
    x, cl, bx = M.bench_batch(bs=3, seed=79)
    res = {}
    for on in (True, False):
        net._engine.first_wino = on
        res[on] = M.train_step(net, x, cl, bx)
        with torch.no_grad():
            _, _, saved = net._engine.forward(x, net._forward_params(), save=True)
        assert isinstance(saved["T"]["a1_1"], _Elided) == on
    net._engine.first_wino = False
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3]
    assert set(a[4]) == set(b[4]) and len(a[4]) == 71
    for k in a[4]:
        assert torch.equal(a[4][k], b[4][k]), k
    net.eval()
    with torch.no_grad():
        net._engine.first_wino = True
        l1, c1 = net(x)
        net._engine.first_wino = False
        l0, c0 = net(x)
    assert torch.equal(l0, l1) and torch.equal(c0, c1)


def test_rccl_calls_of_the_data_parallel_step_run_with_one_rank():
    """The N > 1 path has only ever had one GPU at a time to run on, so its collectives were rehearsed over gloo.  This runs the SAME calls on
    RCCL (backend "nccl" with device_id, as bench.py forms it) with a one-rank group and the N > 1 branches forced: parameter broadcast, the
    overlapped sliced all-reduce in f32 and in bf16 payload, barrier, the int32 MAX all-reduce of bench.py's spin-up.  One rank moves no
    bytes between GPUs; what it proves is that every call is one RCCL accepts (dtype, device, stream use) -- in a child process, because a
    process group is process-global state."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "nccl_single_rank.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "rccl single-rank path ok" in r.stdout, (r.returncode, r.stdout[-600:], r.stderr[-1200:])
    assert r.stdout.count("backend nccl") == 2
