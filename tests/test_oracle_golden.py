"""Pins oracle/ssd_oracle.py to vectors produced by the reference itself
(tests/golden, written by oracle/gen_golden.py in the build container)."""
import os

import numpy as np
import pytest
import torch

import ssd_oracle as O
from helpers import nms_case, split_case, synth_gt


def test_priors_bit_exact(gold_dir):
    z = np.load(os.path.join(gold_dir, "priors_ssd300.npz"))
    pri = O.create_priors_ssd300()
    assert pri.dtype == np.float32 and pri.shape == (8732, 4)
    assert np.array_equal(pri, z["cxcywh"])
    assert np.array_equal(O.xywh_to_xyxy(pri), z["xyxy"])
    # SURVEY 8(a) A6 probes
    assert np.allclose(pri[0], [.013158, .013158, .1, .1], atol=1e-6)
    assert np.allclose(pri[-1], [.5, .5, .6364, 1.0], atol=1e-4)
    offs = np.cumsum([0] + [g * g * a for g, a in zip((38, 19, 10, 5, 3, 1), O.ANCHORS_PER_CELL)])
    assert tuple(offs[:-1]) == O.SCALE_OFFSETS and offs[-1] == 8732


def test_boxmath_bit_exact(gold_dir):
    z = np.load(os.path.join(gold_dir, "boxmath.npz"))
    assert np.array_equal(O.iou_matrix(z["a"], z["b"]), z["iou"])
    assert np.array_equal(O.xyxy_to_xywh(z["a"]), z["a_xywh"])
    assert np.array_equal(O.xywh_to_xyxy(O.xyxy_to_xywh(z["a"])), z["a_back"])
    # exp/log are libm-dependent: tolerance, not bit equality
    np.testing.assert_allclose(O.decode_offsets(z["g"], z["pri"]), z["dec"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(O.encode_offsets(z["dec"], z["pri"]), z["enc"], rtol=1e-5, atol=2e-5)


def _n_cases(gold_dir):
    return int(np.load(os.path.join(gold_dir, "match_loss.npz"))["n_cases"])


@pytest.mark.parametrize("ci", range(18))
def test_match_and_loss_vs_reference(gold_dir, ci):
    z = np.load(os.path.join(gold_dir, "match_loss.npz"))
    assert int(z["n_cases"]) == 18
    boxes, classes, loc, conf, p = split_case(z, ci)
    out = O.multibox_loss(loc, conf, boxes, classes)
    # integer outputs: bit-exact
    assert np.array_equal(out["cls"].astype(np.int8), z[p + "cls"])
    pos = out["pos"]
    allb = np.concatenate(boxes)
    assert np.array_equal(O.xyxy_to_xywh(allb)[out["obj"]][pos], z[p + "gt_pos"])
    np.testing.assert_allclose(out["enc"], z[p + "enc_pos"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(out["loc_loss"], z[p + "loc_loss"], rtol=2e-6)
    np.testing.assert_allclose(out["conf_loss"], z[p + "conf_loss"], rtol=2e-6)
    np.testing.assert_allclose(out["dloc"][pos], z[p + "dloc_pos"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(np.abs(out["dloc"]).astype(np.float64).sum(), z[p + "dloc_abs_sum"], rtol=1e-6)
    rows = z[p + "dconf_rows"]
    np.testing.assert_allclose(out["dconf"].reshape(-1, 21)[rows], z[p + "dconf_vals"], rtol=1e-4, atol=1e-8)
    touched = np.nonzero(np.abs(out["dconf"].reshape(-1, 21)).sum(1) > 0)[0]
    assert np.array_equal(touched, z[p + "dconf_touched"])          # same hard-negative set
    np.testing.assert_allclose(np.abs(out["dconf"]).astype(np.float64).sum(), z[p + "dconf_abs_sum"], rtol=1e-5)


def test_match_rejects_empty_image(priors):
    with pytest.raises(ValueError):
        O.match_priors([np.zeros((0, 4), np.float32)], [np.zeros((0,), np.float32)], O.xywh_to_xyxy(priors))


@pytest.mark.parametrize("ni", range(6))
def test_nms_vs_reference(gold_dir, ni):
    z = np.load(os.path.join(gold_dir, "nms.npz"))
    assert int(z["n_cases"]) == 6
    l_, c_, top_k, p = nms_case(z, ni)
    w, h = [int(v) for v in z["img_wh"]]
    boxes, classes, probs, ids = O.decode_nms(l_, c_, w, h, top_k=top_k)
    assert boxes.shape == z[p + "boxes"].shape
    assert np.array_equal(classes, z[p + "classes"])
    np.testing.assert_allclose(probs, z[p + "probs"], rtol=1e-6)
    np.testing.assert_allclose(boxes, z[p + "boxes"], rtol=1e-5, atol=1e-3)
    if ni == 5:
        assert boxes.shape[0] == 0          # nothing reaches min_score: reference returns three empty lists


def test_network_forward_and_step_vs_reference(gold_dir):
    z = np.load(os.path.join(gold_dir, "network.npz"))
    assert int(z["ref_named_parameters"]) == 77
    params = {k: v.requires_grad_(True) for k, v in O.ssd300_random_params(int(z["param_seed"])).items()}
    assert len(params) == 71 and sum(v.numel() for v in params.values()) == 26285486
    bs = int(z["bs"])
    x = np.random.default_rng(int(z["x_seed"])).standard_normal((bs, 3, 300, 300), dtype=np.float32)
    boxes, classes = synth_gt(np.random.default_rng(int(z["gt_seed"])), bs)
    torch.set_num_threads(8)
    loc, conf = O.ssd300_forward(torch.from_numpy(x), params)
    assert loc.shape == (bs, 8732, 4) and conf.shape == (bs, 8732, 21)
    idx = z["prior_idx"]
    np.testing.assert_allclose(loc.detach().numpy()[:, idx], z["loc_s"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(conf.detach().numpy()[:, idx], z["conf_s"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(float(loc.detach().double().abs().sum()), z["loc_abs"], rtol=1e-5)
    np.testing.assert_allclose(float(conf.detach().double().abs().sum()), z["conf_abs"], rtol=1e-5)
    # numpy loss on the torch outputs == reference loss
    out = O.multibox_loss(loc.detach().numpy(), conf.detach().numpy(), boxes, classes, want_grads=False)
    np.testing.assert_allclose(out["loc_loss"], z["loc_loss"], rtol=1e-5)
    np.testing.assert_allclose(out["conf_loss"], z["conf_loss"], rtol=1e-5)
    # differentiable form + autograd == reference backward
    l1, l2 = O.multibox_loss_torch(loc, conf, [torch.from_numpy(b) for b in boxes],
                                   [torch.from_numpy(c) for c in classes])
    np.testing.assert_allclose(l1.item(), z["loc_loss"], rtol=1e-5)
    np.testing.assert_allclose(l2.item(), z["conf_loss"], rtol=1e-5)
    (l1 + l2).backward()
    names = [str(n) for n in z["grad_names"]]
    l2n = np.asarray([float(params[k].grad.double().norm()) for k in names])
    np.testing.assert_allclose(l2n, z["grad_l2"], rtol=2e-4)
    for k in ("model.features.0.weight", "model.features.21.bias", "c_11_cl.weight", "seq10.2.weight",
              "rescaling_conv_4_3", "c_4_bb.bias"):
        g = params[k].grad.numpy()
        ref = z["g_" + k]
        np.testing.assert_allclose(g, ref, rtol=1e-3, atol=1e-5 * max(1.0, float(np.abs(ref).max())))


def test_resnet34_variant_vs_reference(gold_dir):
    """SSD_resnet34 eval forward (Model.py:72-126) and the zoom/ratio anchors (Util.py:142-164)."""
    z = np.load(os.path.join(gold_dir, "resnet34.npz"))
    st = O.ssd_resnet34_random_state(int(z["state_seed"]))
    x = np.random.default_rng(int(z["x_seed"])).standard_normal((2, 3, 224, 224), dtype=np.float32)
    torch.set_num_threads(8)
    with torch.no_grad():
        loc, conf = O.ssd_resnet34_forward(torch.from_numpy(x), st)
    assert tuple(loc.shape) == (2, 63, 4) and tuple(conf.shape) == (2, 63, 21)
    np.testing.assert_allclose(loc.numpy(), z["loc"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(conf.numpy(), z["conf"], rtol=1e-4, atol=1e-4)
    # state_dict layout: our shape table + the seq1..seq5 aliases == the reference's keys, in shapes too
    ref = dict(zip(z["state_dict_keys"].tolist(), z["state_dict_shapes"].tolist()))
    shapes = O.ssd_resnet34_state_shapes()
    full = dict(shapes)
    for alias, trunk in O.ssd_resnet34_aliases().items():
        for k, v in shapes.items():
            if k.startswith(trunk):
                full[alias + k[len(trunk):]] = v
    assert set(full) == set(ref)
    for k, v in full.items():
        assert ",".join(str(d) for d in v) == ref[k], k
    anc = O.create_ancs_xywh_zoom_ratio()
    assert anc.shape == (189, 4)
    np.testing.assert_allclose(anc, z["ancs_zoom_ratio"], rtol=0, atol=1e-7)


def _map_case(z, ci):
    p = f"c{ci}_"
    def split(a, cnt):
        o = np.cumsum(np.r_[0, cnt])
        return [a[o[i]:o[i + 1]] for i in range(len(cnt))]
    dc, gc = z[p + "det_count"], z[p + "gt_count"]
    return (split(z[p + "det_boxes"], dc), split(z[p + "det_classes"], dc), split(z[p + "det_scores"], dc),
            split(z[p + "gt_boxes"], gc), split(z[p + "gt_classes"], gc)), z[p + "ap"]


@pytest.mark.parametrize("ci", range(4))
def test_map_vs_reference(gold_dir, ci):
    """get_map (Util.py:783-885): the APs are sums/ratios of small integers in float64 -> bit-exact."""
    z = np.load(os.path.join(gold_dir, "map.npz"))
    assert int(z["n_cases"]) == 4
    args, ref = _map_case(z, ci)
    aps = O.get_map(*args)
    assert np.array_equal(np.asarray([aps[c] for c in range(20)]), ref)
    assert np.array_equal(O.ap_recall_thresholds(), z["recall_levels"])


@pytest.mark.parametrize("shape", [(375, 500, 300, 300), (500, 333, 300, 300), (120, 90, 300, 300), (300, 300, 300, 300),
                                   (300, 451, 300, 300), (900, 1300, 300, 300), (37, 53, 64, 48), (301, 299, 300, 300)])
def test_resize_restatement_equals_pillow(shape):
    """The oracle's ImagingResample restatement against Pillow itself (the library Dataset.py:10 calls through
    torchvision's Resize): bit-exact on 8-bit RGB, down-scaling (antialiased), up-scaling and size-preserving axes."""
    from PIL import Image
    h, w, oh, ow = shape
    a = np.random.default_rng(h * 1000 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(a).resize((ow, oh), Image.BILINEAR))
    assert np.array_equal(O.resize_bilinear_u8(a, oh, ow), ref)


def test_preprocess_restatement_layout():
    a = np.random.default_rng(3).integers(0, 256, (40, 50, 3), dtype=np.uint8)
    out = O.preprocess_image(a, 30, 20)
    assert out.shape == (3, 30, 20) and out.dtype == np.float32
    u8 = O.resize_bilinear_u8(a, 30, 20)
    np.testing.assert_allclose(out[1], (u8[..., 1] / 255.0 - 0.456) / 0.224, rtol=0, atol=1e-5)
    assert tuple(O.mean_filler_u8()) == (123, 116, 103)


@pytest.mark.parametrize("ci", range(3))
def test_degenerate_ground_truth_vs_reference(gold_dir, ci):
    """Zero-area / zero-height boxes (tests/golden/degenerate.npz): matched through the forced match only, loc loss = inf
    (target log(0)), finite gradients, the same positives and hard negatives as the reference."""
    z = np.load(os.path.join(gold_dir, "degenerate.npz"))
    boxes, classes, loc, conf, p = split_case(z, ci)
    with np.errstate(all="ignore"):
        out = O.multibox_loss(loc, conf, boxes, classes)
    assert np.array_equal(out["cls"].astype(np.int8), z[p + "cls"])
    assert np.isinf(z[p + "loc_loss"]) and np.isinf(out["loc_loss"])
    np.testing.assert_allclose(out["conf_loss"], z[p + "conf_loss"], rtol=2e-6)
    np.testing.assert_allclose(out["dloc"][out["pos"]], z[p + "dloc_pos"], rtol=1e-6, atol=1e-9)
    assert np.isfinite(out["dloc"]).all()
    touched = np.nonzero(np.abs(out["dconf"].reshape(-1, 21)).sum(1) > 0)[0]
    assert np.array_equal(touched, z[p + "dconf_touched"])
    np.testing.assert_allclose(np.abs(out["dconf"]).astype(np.float64).sum(), z[p + "dconf_abs_sum"], rtol=1e-5)


def test_photometric_restatement_equals_pillow():
    """Brightness / contrast / saturation through ImageEnhance and the hue path through Pillow's HSV mode: the oracle's
    restatement is bit-equal on random images; the two colour-space conversions are checked on all 2^24 inputs."""
    from PIL import Image, ImageEnhance
    rng = np.random.default_rng(1)
    for trial in range(24):
        h, w = int(rng.integers(5, 80)), int(rng.integers(5, 80))
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if trial % 5 == 0:
            a = (a // 8 + 100).astype(np.uint8)
        f = float(rng.uniform(0.5, 1.5)); hf = float(rng.uniform(-18 / 255., 18 / 255.))
        im = Image.fromarray(a)
        refs = [np.asarray(ImageEnhance.Brightness(im).enhance(f)), np.asarray(ImageEnhance.Contrast(im).enhance(f)),
                np.asarray(ImageEnhance.Color(im).enhance(f))]
        hh, ss, vv = im.convert("HSV").split()
        nh = np.array(hh, dtype=np.uint8)
        with np.errstate(over="ignore"):
            nh += np.int32(hf * 255).astype(np.uint8)              # torchvision adjust_hue's wrap-around add
        refs.append(np.asarray(Image.merge("HSV", (Image.fromarray(nh, "L"), ss, vv)).convert("RGB")))
        for kind in range(4):
            assert np.array_equal(O.photometric_apply(a, [(kind, hf if kind == 3 else f)]), refs[kind]), (trial, kind)
    g = np.arange(256, dtype=np.uint8)
    for lo in range(0, 256, 64):                                       # all 2^24 triples, 2^22 at a time
        c0, c1, c2 = np.meshgrid(g[lo:lo + 64], g, g, indexing="ij")
        tri = np.stack([c0, c1, c2], -1).reshape(2048, 2048, 3)
        assert np.array_equal(O.rgb_to_hsv_u8(tri), np.asarray(Image.fromarray(tri, "RGB").convert("HSV")))
        assert np.array_equal(O.hsv_to_rgb_u8(tri), np.asarray(Image.fromarray(tri, "HSV").convert("RGB")))
