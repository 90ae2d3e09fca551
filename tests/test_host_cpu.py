"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header declares,
host-side mirrors agree with the oracle, the drop-in surface has the reference's names, and the
product path refuses to run without the GPU instead of falling back."""
import os
import re

import numpy as np
import pytest
import torch

import ssd_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from objectdetection_ssd_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "ssd_gfx950.h")).read()
    declared = set(re.findall(r"\b(ssd_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ssd_status"}
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ssd_abi_version() == 1      # entry points were added since round 1, none changed
    assert lib.ssd_status_string(-2) == b"workspace too small"


def test_workspace_queries_need_no_gpu():
    import ctypes as C
    from objectdetection_ssd_amd import _lib, ops
    lib = _lib.load()
    g = ops.make_geom(32, 300, 300, 64, 64, 3, 1, 1, 1)
    assert (g.Ho, g.Wo) == (300, 300)
    assert 0 < lib.ssd_conv2d_wgrad_workspace(C.byref(g)) < 1 << 30
    assert lib.ssd_multibox_loss_workspace(32, 8732, 90) > 32 * 8732 * 5
    assert lib.ssd_decode_nms_workspace(8732, 21) > 0
    assert ops.pool_out(75, 2, 2, 0, True) == 38 and ops.pool_out(75, 2, 2, 0, False) == 37
    assert ops.pool_out(19, 3, 1, 1, True) == 19 and ops.pool_out(300, 2, 2, 0, False) == 150


def test_host_priors_and_coders_match_oracle():
    from objectdetection_ssd_amd import Losses, Util
    pri = O.create_priors_ssd300()
    assert np.array_equal(Losses.ancs_xywh.numpy(), pri)
    assert np.array_equal(Losses.ancs_xyxy.numpy(), O.xywh_to_xyxy(pri))
    r = np.random.default_rng(0)
    g = r.standard_normal((50, 4)).astype(np.float32)
    p = pri[r.integers(0, 8732, 50)]
    np.testing.assert_allclose(Util.gcxgcy_to_cxcy(torch.from_numpy(g), torch.from_numpy(p)).numpy(),
                               O.decode_offsets(g, p), rtol=1e-6)
    assert Util.class_to_label[20] == "bg" and len(Util.class_to_label) == 21


def test_ssd300_module_surface_matches_reference_names(gold_dir):
    from objectdetection_ssd_amd import Model
    z = np.load(os.path.join(gold_dir, "network.npz"))
    net = Model.SSD_300()
    named = dict(net.named_parameters())
    assert len(named) == int(z["ref_named_parameters"]) == 77
    shapes = O.ssd300_param_shapes()
    assert set(net._engine.names) == set(shapes)
    for k, s in shapes.items():
        assert tuple(named[k].shape) == s, k
    biases = [n for n in named if n.endswith(".bias")]
    assert len(biases) == 38                                       # train.py:46-51 2x-lr group
    sd = net.state_dict()
    for alias in ("conv_4_3.0.weight", "seq5.1.weight", "seq7.0.weight", "seq7.2.bias", "model.classifier.0.weight"):
        assert alias in sd, alias
    assert torch.equal(sd["seq7.0.weight"], sd["conv_fc6.weight"])
    assert float(net.rescaling_conv_4_3.mean()) == 20.0
    # op list geometry: 8732 priors, offsets as SURVEY 8(a) A5
    from objectdetection_ssd_amd import ops
    hw = {"x": 300}
    offs, off = [], 0
    for op in net._engine.ops:
        if op["op"] == "conv_first":
            hw[op["y"]] = hw[op["x"]]
        elif op["op"] == "conv":
            hw[op["y"]] = ops.conv_out_hw(hw[op["x"]], hw[op["x"]], op["k"], op["s"], op["pad"], op["dil"])[0]
        elif op["op"] == "pool":
            hw[op["y"]] = ops.pool_out(hw[op["x"]], op["k"], op["s"], op["pad"], op["ceil"])
        elif op["op"] == "l2norm":
            hw[op["y"]] = hw[op["x"]]
        elif op["op"] == "head":
            offs.append(off)
            off += hw[op["x"]] ** 2 * op["a"]
    assert tuple(offs) == O.SCALE_OFFSETS and off == 8732


def test_product_path_refuses_cpu_tensors():
    from objectdetection_ssd_amd import Losses, Model, ops
    net = Model.SSD_300()
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 300, 300))
    with pytest.raises(RuntimeError):
        Losses.ssd((torch.zeros(1, 8732, 4), torch.zeros(1, 8732, 21)), [torch.zeros(1)], [torch.tensor([[0., 0., 1., 1.]])])
    with pytest.raises(RuntimeError):
        Losses.inference(torch.zeros(8732, 4), torch.zeros(8732, 21), (300, 300), toDraw=False)
    with pytest.raises(RuntimeError):
        ops.l2norm_fwd(torch.zeros(4, 512), torch.ones(512))
    with pytest.raises(ValueError):
        Losses.ssd((torch.zeros(1, 8732, 4), torch.zeros(1, 8732, 21)), [torch.zeros(0)], [torch.zeros(0, 4)])


def test_dropin_module_names(gold_dir):
    """Every name the reference's callers import from Model / Losses / Util / Dataset resolves after install_dropin():
    tests/golden/import_names.json is read off train.py, train_function.py and Dataset.py with `ast` by oracle/gen_golden.py
    (train.py:1-6, train_function.py:3-4, Dataset.py:1-6)."""
    import importlib
    import json
    import sys
    import objectdetection_ssd_amd as pkg
    names = json.load(open(os.path.join(gold_dir, "import_names.json")))
    assert set(names) == {"train.py", "train_function.py", "Dataset.py"}
    assert "all_multi_bboxes" in names["train.py"]["explicit"]["Util"] and "transform" in names["Dataset.py"]["explicit"]["Util"]
    saved = {k: sys.modules.get(k) for k in ("Model", "Losses", "Util", "Dataset")}
    try:
        for k in saved:
            sys.modules.pop(k, None)
        pkg.install_dropin()
        from Losses import ancs_xywh, ancs_xyxy, device, inference, ssd  # noqa: F401
        from Model import SSD_300, SSD_resnet34  # noqa: F401
        from Util import class_to_label, get_map, create_ancs_xywh_zoom_ratio  # noqa: F401
        from Dataset import MultiImageMultiBBoxDataset, collate_fn  # noqa: F401
        for fn, spec in names.items():
            for mod, wanted in spec["explicit"].items():
                m = importlib.import_module(mod)
                assert m.__name__.startswith("objectdetection_ssd_amd."), (fn, mod, m.__name__)
                for n in wanted:
                    assert hasattr(m, n), f"{fn}: from {mod} import {n}"
            ns = {}
            for mod in spec["star"]:
                exec(f"from {mod} import *", ns)
            for n in spec["star_used"]:
                assert n in ns, f"{fn}: `{n}` is used after star imports of {spec['star']}"
        import Util
        for lst in ("all_images", "all_multi_bboxes", "all_multi_labels", "all_difficulties"):       # train.py:12,22-25 index them by phase
            assert set(getattr(Util, lst)) >= {"train"}
        # hot path only: the reference's own Util / Dataset stay whatever `import Util` finds
        for k in saved:
            sys.modules.pop(k, None)
        pkg.install_dropin(dataset=False)
        assert sys.modules["Model"].__name__ == "objectdetection_ssd_amd.Model" and "Util" not in sys.modules and "Dataset" not in sys.modules
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_collate_fn_is_host_only_in_dataloader_workers(tmp_path):
    """train.py:29,40 hands `collate_fn` to `DataLoader(num_workers=2)`: it runs in forked workers and must never touch the
    GPU.  The batch comes back as a `Dataset.RawBatch` of host data; rendering it is `.to(device)` in the main process, and
    there is no CPU rendering to fall back to."""
    from PIL import Image
    from objectdetection_ssd_amd import Dataset, Util
    rng = np.random.default_rng(3)
    paths, n = [], 5
    for i in range(n):
        h, w = int(rng.integers(30, 90)), int(rng.integers(30, 90))
        pth = str(tmp_path / f"i{i}.png")
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(pth)
        paths.append(pth)
    ds = Dataset.MultiImageMultiBBoxDataset(paths, [[[1., 2., 20., 25.], [3., 3., 15., 28.]]] * n, [["dog", "cat"]] * n, [[0, 1]] * n,
                                            list(range(n)))
    dl = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=True, num_workers=2, collate_fn=Dataset.collate_fn)
    seen = []
    for inputs, classes, boxes, indices in dl:
        assert isinstance(inputs, Dataset.RawBatch) and inputs.arena.dtype == torch.uint8 and not inputs.arena.is_cuda
        bs = len(indices)
        assert inputs.shape == (bs, 3, 300, 300) and inputs.size(0) == bs               # train_function.py:64,98
        assert all(c.tolist() == [float(Util.label_to_class["dog"])] or c.numel() == 0 for c in classes)    # difficult cat dropped
        for p, idx in zip(inputs.plans, indices):
            with Image.open(paths[idx]) as im:
                assert (p.src_w, p.src_h) == im.size
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            inputs.to("cpu")
        seen += indices
    assert sorted(seen) == list(range(n))
    # Util.transform keeps the reference's signature (Dataset.py:33) and returns an image with `.size`
    img, b, l = Util.transform(Image.open(paths[0]), torch.tensor([[1., 2., 20., 25.]]), torch.tensor([3.]))
    assert isinstance(img, Dataset.RawImage) and img.size == img.plan.size and b.shape[1] == 4 and b.shape[0] == l.shape[0]


def test_state_dict_layout_equals_reference_checkpoint_layout(gold_dir, tmp_path):
    """(f)-2: `cnn.state_dict()` written by train_function.py:114-120 / read by :25-27 -- same 107 keys, same
    shapes, same order as the reference model's; save/load round trip keeps the aliases tied."""
    from objectdetection_ssd_amd import Model
    z = np.load(os.path.join(gold_dir, "network.npz"))
    ref_keys = [str(k) for k in z["state_dict_keys"]]
    ref_shapes = [tuple(int(d) for d in s.split(",")) if s else () for s in z["state_dict_shapes"]]
    net = Model.SSD_300()
    sd = net.state_dict()
    assert list(sd.keys()) == ref_keys and len(ref_keys) == 107
    assert [tuple(v.shape) for v in sd.values()] == ref_shapes
    assert [n for n, _ in net.named_parameters()] == [str(k) for k in z["named_parameter_keys"]]
    path = tmp_path / "ckpt.pt"
    torch.save({"epoch": 3, "cnn_state_dict": sd}, path)               # the reference's checkpoint dict shape
    other = Model.SSD_300()
    ck = torch.load(path, weights_only=True)
    other.load_state_dict(ck["cnn_state_dict"])                         # strict: every key present
    assert torch.equal(other.conv_4_3[0].weight, net.model.features[0].weight)
    assert other.conv_4_3[0].weight is other.model.features[0].weight   # aliases still share storage
    assert other.seq7[0].weight is other.conv_fc6.weight
    assert torch.equal(other.c_11_cl.bias, net.c_11_cl.bias)


def test_resnet34_variant_surface_and_checkpoint_layout(gold_dir):
    """A16 / configs[4]: `SSD_resnet34(n_classes)` keeps the reference's constructor, state_dict keys (order and shapes)
    and parameter names; it refuses train mode (unreproducible dropout) and CPU tensors (no fallback)."""
    from objectdetection_ssd_amd import Model, Util
    z = np.load(os.path.join(gold_dir, "resnet34.npz"))
    net = Model.SSD_resnet34(20)
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in z["state_dict_keys"]]
    assert [",".join(str(d) for d in v.shape) for v in sd.values()] == [str(s) for s in z["state_dict_shapes"]]
    assert [n for n, _ in net.named_parameters()] == [str(k) for k in z["named_parameter_keys"]]
    assert net.seq1[0] is net.resnet.conv1 and net.seq3[0] is net.resnet.layer2[0]
    assert float(net.conv2d_02_c4.bias[0]) == -2.0                        # Model.py:39
    with pytest.raises(RuntimeError, match="eval"):
        net(torch.zeros(1, 3, 224, 224))
    net.eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 3, 224, 224))
    with pytest.raises(ValueError):
        Model.SSD_resnet34(10)
    anc = Util.create_ancs_xywh_zoom_ratio()
    assert anc.dtype == torch.float32 and tuple(anc.shape) == (189, 4)
    np.testing.assert_allclose(anc.numpy(), z["ancs_zoom_ratio"], rtol=0, atol=1e-7)


def test_get_map_needs_the_gpu():
    from objectdetection_ssd_amd import Util
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Util.get_map([torch.zeros(1, 4)], [torch.zeros(1)], [torch.zeros(1)], [torch.zeros(1, 4)], [torch.zeros(1)])


def test_transform_geometry_equals_reference_functions(gold_dir):
    """(f)-3 host half: `Dataset.plan_transform` makes the reference's `random` draws in the reference's order and the
    same box arithmetic as Util.expand / random_crop / flip (Util.py:610-749): on every seeded case of
    tests/golden/augment.npz the composed 8-bit image (sha256), boxes and labels equal the reference's."""
    import hashlib
    import random
    from objectdetection_ssd_amd import Dataset
    z = np.load(os.path.join(gold_dir, "augment.npz"))
    seen = set()
    for ci in range(int(z["n_cases"])):
        p = f"c{ci}_"
        img = z[p + "img"]
        h, w = img.shape[:2]
        random.seed(int(z[p + "seed"]))
        plan, b, l = Dataset.plan_transform(w, h, torch.from_numpy(z[p + "boxes"]), torch.from_numpy(z[p + "labels"]),
                                            photometric=False)        # the fixture's streams hold the geometric draws only
        out = O.compose_input(img, canvas=plan.canvas, crop=plan.crop, flip=plan.flip)
        assert tuple(out.shape[:2]) == tuple(z[p + "out_shape"]) == (plan.size[1], plan.size[0])
        assert hashlib.sha256(out.tobytes()).hexdigest() == str(z[p + "out_sha256"])
        assert np.array_equal(b.numpy(), z[p + "out_boxes"]) and np.array_equal(l.numpy(), z[p + "out_labels"])
        assert plan.flip == bool(z[p + "flipped"])
        if p + "out_img" in z.files:
            assert np.array_equal(out, z[p + "out_img"])
        seen.add((plan.canvas[:2] != (h, w), plan.crop[2:] != plan.canvas[:2], plan.flip))
    assert len(seen) >= 6                                    # the cases cover the combinations
    # collate_fn keeps the reference's tuple layout for already-normalised items
    x, c, bb, idx = Dataset.collate_fn([(torch.zeros(3, 4, 4), torch.zeros(1), torch.zeros(1, 4), 7)] * 2)
    assert tuple(x.shape) == (2, 3, 4, 4) and idx == [7, 7]
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            Dataset.preprocess_batch([np.zeros((8, 8, 3), np.uint8)])


def test_bench_refuses_to_run_without_the_gpu():
    """bench.py measures the HIP path only: without a GPU it must stop with a clear message, never time a fallback."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout)
    assert '"metric"' not in r.stdout


def test_bench_gpus_n_starts_n_ranks_itself():
    """`python bench.py --gpus 2` (no launcher around it, as the driver runs it) must form a 2-rank process group by itself, and a
    launcher that started a different number of ranks than --gpus says must be refused -- the line can never time one GPU and
    label it N.  `--rendezvous-only` stops after the process group's first all-reduce, before any GPU is touched."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-device",
                        "--rendezvous-only"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout                                        # rank 0 only
    out = json.loads(line[0])
    assert out == {"rendezvous": True, "n_gpus": 2, "backend": "gloo", "ranks_counted": 2}
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rendezvous-only"], capture_output=True, text=True,
                         timeout=300, env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode != 0 and "WORLD_SIZE=4" in (bad.stderr + bad.stdout) and '"metric"' not in bad.stdout


def test_photometric_draws_equal_reference(gold_dir):
    """`Dataset.plan_photometric` == the order and factors Util.photometric_distort (Util.py:752-780) applies on the same
    seeded `random` stream (recorded from the reference's own function), and it leaves the stream at the same position."""
    import random
    from objectdetection_ssd_amd import Dataset
    z = np.load(os.path.join(gold_dir, "photometric_draws.npz"))
    kinds_seen = set()
    for ci in range(int(z["n_cases"])):
        random.seed(12000 + ci)
        ops_ = Dataset.plan_photometric()
        assert [k for k, _ in ops_] == z[f"c{ci}_kinds"].tolist()
        assert [f for _, f in ops_] == z[f"c{ci}_factors"].tolist()
        assert random.random() == float(z[f"c{ci}_next"])
        kinds_seen.update(k for k, _ in ops_)
        for k, f in ops_:
            assert (-18 / 255. <= f <= 18 / 255.) if k == 3 else (0.5 <= f <= 1.5)
    assert kinds_seen == {0, 1, 2, 3}
    # the full training plan draws photometric first, then expand / crop / flip (Util.py:586-606)
    random.seed(5)
    want = Dataset.plan_photometric()
    random.seed(5)
    plan, _, _ = Dataset.plan_transform(50, 40, torch.tensor([[5., 5., 30., 30.]]), torch.tensor([3.]))
    assert plan.photo == want


def test_fused_sgd_optimizer_interface():
    """optim.SGD keeps torch.optim.SGD's constructor, param-group keys and state_dict layout (train.py:53-55,
    train_function.py:27,116); options the reference never uses are refused, and there is no CPU step."""
    from objectdetection_ssd_amd.optim import SGD
    lr = 1e-4
    w, b = torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(4))
    groups = lambda: [{"params": [b], "lr": 2 * lr}, {"params": [w]}]
    ours = SGD(params=groups(), lr=lr, momentum=0.9, weight_decay=5e-4)
    ref = torch.optim.SGD(params=groups(), lr=lr, momentum=0.9, weight_decay=5e-4)
    assert isinstance(ours, torch.optim.Optimizer)
    a, r = ours.state_dict(), ref.state_dict()
    assert a["param_groups"] == r["param_groups"] and a["state"] == r["state"] == {}
    assert [g["lr"] for g in ours.param_groups] == [2 * lr, lr]
    sched = torch.optim.lr_scheduler.StepLR(ours, step_size=7, gamma=0.1)        # train.py:57 constructs one
    assert sched.get_last_lr() == [2 * lr, lr]
    for bad in (dict(nesterov=True, momentum=0.9), dict(dampening=0.1), dict(maximize=True), dict(lr=-1.0)):
        with pytest.raises(ValueError):
            SGD(groups(), **({"lr": lr} | bad))
    ours.zero_grad()
    w.grad, b.grad = torch.ones_like(w), torch.ones_like(b)
    with pytest.raises(ValueError, match="on the GPU"):
        ours.step()


def test_winograd_entry_points_validate_before_launching():
    """Error convention of the C ABI (include/ssd_gfx950.h: 0 or a negative ssd_status, never a throw): the Winograd entry points
    reject NULL / misaligned pointers, unsupported geometry and a short workspace before anything is enqueued -- so this runs
    without a GPU, with made-up pointer values that are never dereferenced."""
    import ctypes as C
    from objectdetection_ssd_amd import _lib, ops
    lib = _lib.load()
    OK_PTR, ODD_PTR = 0x10000, 0x10004
    g = ops.make_geom(2, 38, 38, 64, 64, 3, 1, 1, 1)
    gb = C.byref(g)
    ws_fwd = lib.ssd_conv3x3_wino_workspace(gb, 0, 4)
    assert ws_fwd == 2 * 36 * 2 * 10 * 10 * 64 * 4                     # V and M planes: 36 x tiles x channels floats each
    assert lib.ssd_conv3x3_wino_workspace(gb, 0, 3) == 0                # only F(2x2) and F(4x4) exist
    # forward fused with the pool
    f = lib.ssd_conv3x3_wino_fwd_pool
    assert f(None, OK_PTR, None, OK_PTR, None, gb, 0, None, OK_PTR, ws_fwd, None) == -3
    assert f(OK_PTR, OK_PTR, None, ODD_PTR, None, gb, 0, None, OK_PTR, ws_fwd, None) == -5
    assert f(OK_PTR, OK_PTR, None, OK_PTR, None, gb, 0, ODD_PTR, OK_PTR, ws_fwd, None) == -5
    assert f(OK_PTR, OK_PTR, None, OK_PTR, None, gb, 0, None, OK_PTR, ws_fwd - 1, None) == -2
    g5 = ops.make_geom(2, 38, 38, 64, 64, 5, 1, 2, 1)
    assert f(OK_PTR, OK_PTR, None, OK_PTR, None, C.byref(g5), 0, None, OK_PTR, ws_fwd, None) == -1
    g2 = ops.make_geom(2, 38, 38, 64, 62, 3, 1, 1, 1)                  # the pooled form needs whole channel quads
    assert f(OK_PTR, OK_PTR, None, OK_PTR, None, C.byref(g2), 0, None, OK_PTR, ws_fwd, None) == -1
    # forward that keeps its planes
    k = lib.ssd_conv3x3_wino_fwd_keep
    assert k(OK_PTR, OK_PTR, None, OK_PTR, 64, gb, 1, None, OK_PTR, ws_fwd, None) == -3
    assert k(OK_PTR, OK_PTR, None, OK_PTR, 60, gb, 1, OK_PTR, OK_PTR, ws_fwd, None) == -1
    assert k(OK_PTR, OK_PTR, None, OK_PTR, 64, gb, 1, OK_PTR, OK_PTR, 16, None) == -2
    # weight gradient on kept planes, data gradient on the planes its dy pass leaves
    ws_w = lib.ssd_conv3x3_wino_wgrad_workspace(gb, 64, 4)
    assert ws_w > 0
    w = lib.ssd_conv3x3_wino_wgrad_planes
    assert w(None, OK_PTR, 64, OK_PTR, None, gb, None, OK_PTR, ws_w, None) == -3
    assert w(OK_PTR, OK_PTR, 64, ODD_PTR, None, gb, None, OK_PTR, ws_w, None) == -5
    assert w(OK_PTR, OK_PTR, 62, OK_PTR, None, gb, None, OK_PTR, ws_w, None) == -1
    assert w(OK_PTR, OK_PTR, 64, OK_PTR, None, gb, None, OK_PTR, ws_w - 1, None) == -2
    d = lib.ssd_conv3x3_wino_dgrad_planes
    ws_d = lib.ssd_conv3x3_wino_workspace(gb, 1, 4)
    assert d(None, OK_PTR, 64, OK_PTR, None, 0, gb, OK_PTR, ws_d, None) == -3
    assert d(OK_PTR, OK_PTR, 48, OK_PTR, None, 0, gb, OK_PTR, ws_d, None) == -1      # Co_pad must be a multiple of 32 and >= Co
    assert d(OK_PTR, OK_PTR, 64, OK_PTR, ODD_PTR, 0, gb, OK_PTR, ws_d, None) == -5
    assert d(OK_PTR, OK_PTR, 64, OK_PTR, None, 0, gb, OK_PTR, 16, None) == -2
    assert lib.ssd_status_string(-5).startswith(b"pointer")


def test_round2_entry_points_validate_and_size_without_a_gpu():
    """Host-side rules of the entry points added in round 2, with made-up pointers that are never dereferenced: the tile count of a
    dilated layer (d x d sub-lattices tiled one by one) as the workspace query sees it, what the dilated form refuses (fused pool,
    F(2x2), pooled gradient source), the shape rules of the pooled-gradient source, conv1_1's one-kernel forms and the weight job table."""
    import ctypes as C
    from objectdetection_ssd_amd import _lib, ops
    lib = _lib.load()
    OK_PTR, ODD_PTR = 0x10000, 0x10004
    # fc6: 19x19, dilation 4 -> lattices of 5,5,5,4 rows -> 2+2+2+1 = 7 tile rows per dimension
    g6 = ops.make_geom(32, 19, 19, 512, 1024, 3, 1, 4, 4)
    assert ops.wino_tiles(g6) == 32 * 49
    for (h, w, d) in ((19, 19, 4), (10, 13, 2), (7, 5, 3), (3, 3, 4), (300, 300, 1), (75, 38, 1)):
        g = ops.make_geom(2, h, w, 32, 32, 3, 1, d, d)
        rows = sum(-(-len(range(a, h, d)) // 4) for a in range(d))
        cols = sum(-(-len(range(b, w, d)) // 4) for b in range(d))
        assert ops.wino_tiles(g) == 2 * rows * cols, (h, w, d)
        assert lib.ssd_conv3x3_wino_workspace(C.byref(g), 0, 4) == 2 * (36 * 2 * rows * cols * 32 * 4)
        if d > 1:
            assert lib.ssd_conv3x3_wino_workspace(C.byref(g), 0, 2) == 0                         # no F(2x2) on sub-lattices
    gb6 = C.byref(g6)
    ws6 = lib.ssd_conv3x3_wino_workspace(gb6, 0, 4)
    assert lib.ssd_conv3x3_wino_workspace(gb6, 0, 2) == 0                                       # no F(2x2) on sub-lattices
    assert lib.ssd_conv3x3_wino_fwd_pool(OK_PTR, OK_PTR, None, OK_PTR, None, gb6, 0, None, OK_PTR, ws6, None) == -1     # nor the fused pool
    assert lib.ssd_conv3x3_wino_uses_full(gb6, 0) == 0
    g5 = ops.make_geom(2, 19, 19, 64, 64, 3, 1, 5, 5)
    assert lib.ssd_conv3x3_wino_workspace(C.byref(g5), 0, 4) == 0                                # dilation 1 .. 4
    gpd = ops.make_geom(2, 19, 19, 64, 64, 3, 1, 4, 2)
    assert lib.ssd_conv3x3_wino_workspace(C.byref(gpd), 0, 4) == 0                               # padding must equal the dilation
    # pooled-gradient source: the 2x2 / stride-2 pool over exactly this layer's output, ldy = Co, plain geometry only
    g = ops.make_geom(2, 75, 75, 64, 64, 3, 1, 1, 1)
    gb = C.byref(g)
    f = lib.ssd_wino4_dy_transform_pooled
    assert f(OK_PTR, OK_PTR, OK_PTR, 38, 38, 64, gb, OK_PTR, None, None, None) in (0, -4)        # accepted (a launch without a GPU may fail)
    assert f(OK_PTR, OK_PTR, OK_PTR, 37, 37, 64, gb, None, None, None, None) == -3
    assert f(OK_PTR, OK_PTR, OK_PTR, 36, 38, 64, gb, OK_PTR, None, None, None) == -1             # not this map's pool
    assert f(OK_PTR, OK_PTR, OK_PTR, 39, 38, 64, gb, OK_PTR, None, None, None) == -1
    assert f(OK_PTR, OK_PTR, OK_PTR, 38, 38, 96, gb, OK_PTR, None, None, None) == -1             # ldy must be Co
    assert f(OK_PTR, 0x10002, OK_PTR, 38, 38, 64, gb, OK_PTR, None, None, None) == -5            # argmax words
    assert f(OK_PTR, OK_PTR, OK_PTR, 10, 10, 1024, gb6, OK_PTR, None, None, None) == -1          # no pooled source on a dilated layer
    # conv1_1 in one kernel
    assert lib.ssd_conv1_first_fwd(None, OK_PTR, None, OK_PTR, None, 2, 300, 300, 1, None) == -3
    assert lib.ssd_conv1_first_fwd(OK_PTR, OK_PTR, None, ODD_PTR, None, 2, 300, 300, 1, None) == -5
    assert lib.ssd_conv1_first_fwd(OK_PTR, OK_PTR, None, OK_PTR, None, 0, 300, 300, 1, None) == -1
    wsf = lib.ssd_conv1_first_wgrad_workspace(32, 300, 300)
    assert wsf >= 768 * 2048 * 4                                                                  # one [64][32] partial per workgroup
    assert lib.ssd_conv1_first_wgrad(OK_PTR, OK_PTR, OK_PTR, None, 32, 300, 300, OK_PTR, 16, None) == -2
    assert lib.ssd_conv1_first_wgrad(OK_PTR, ODD_PTR, OK_PTR, None, 32, 300, 300, OK_PTR, wsf, None) == -5
    # weight job table: block counts per job kind
    job = _lib.WeightJob()
    job.w0 = job.w1 = OK_PTR; job.out_fwd = OK_PTR; job.out_bwd = OK_PTR
    job.co0 = job.co = 128; job.ci = 64; job.taps = 9; job.co_pad = 128; job.kind = 0
    assert lib.ssd_weight_job_blocks(C.byref(job)) == (128 * 64 + 64 * 128 + 255) // 256
    job.kind = 1                                                                                  # one block per 32 x 32-channel brick
    assert lib.ssd_weight_job_blocks(C.byref(job)) == (128 // 32) * (64 // 32)
    job.kind = 3; job.pad1 = 192                                                                  # bf16 copies: the data gradient's K may exceed co_pad
    assert lib.ssd_weight_job_blocks(C.byref(job)) == (192 // 32) * (64 // 32)
    job.taps = 25
    assert lib.ssd_weight_job_blocks(C.byref(job)) == -1
    job.taps = 9
    job.kind = 2; job.co = job.co0 = 64; job.ci = 3
    assert lib.ssd_weight_job_blocks(C.byref(job)) == 64 * 32 // 256
    job.kind = 9
    assert lib.ssd_weight_job_blocks(C.byref(job)) == -1
    assert lib.ssd_weights_prepare(None, OK_PTR, 1, 1, None) == -3


def test_round3_limb_gemm_rules_and_sizes_without_a_gpu():
    """Host-side rules of the three-limb plane GEMMs (csrc/gemm_x3.hip) with made-up pointers that are never dereferenced: which reduction
    lengths take them, the environment / tune switch, the size of a limb filter tensor, what the bare GEMM entry points refuse, and the block
    count of a weight job whose outputs are limb planes (8 elements per thread)."""
    import ctypes as C
    import os
    import subprocess
    import sys
    from objectdetection_ssd_amd import _lib
    lib = _lib.load()
    OK_PTR, ODD_PTR = 0x10000, 0x10004
    try:
        assert lib.ssd_tune_set_wino_x3(1) == 0
        assert [lib.ssd_wino_uses_x3(4, k) for k in (64, 128, 256, 512, 1024)] == [0, 0, 1, 1, 1]
        assert lib.ssd_wino_uses_x3(2, 512) == 0                       # F(2x2) (a tuning aid) stays on the f32 MFMA
        assert lib.ssd_tune_set_wino_x3(0) == 0
        assert lib.ssd_wino_uses_x3(4, 512) == 0
    finally:
        assert lib.ssd_tune_set_wino_x3(-1) == 0
    # the default comes from the environment, read once per process: on unless SSD_WINO_X3=0
    code = ("import sys; sys.path.insert(0, %r); from objectdetection_ssd_amd import _lib; print(_lib.load().ssd_wino_uses_x3(4, 512))"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for env, want in (({}, "1"), ({"SSD_WINO_X3": "0"}, "0"), ({"SSD_WINO_X3": "1"}, "1")):
        e = {k: v for k, v in os.environ.items() if k != "SSD_WINO_X3"}
        e.update(env)
        assert subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=120).stdout.strip() == want
    # limb filter tensor: K x pad128(rows) elements per plane, 3 limbs of 2 bytes
    assert lib.ssd_gemm_x3_weights_bytes(512, 512, 36) == 36 * 512 * 512 * 6
    assert lib.ssd_gemm_x3_weights_bytes(100, 512, 36) == 36 * 512 * 128 * 6
    assert lib.ssd_gemm_x3_weights_bytes(150, 1024, 1) == 1024 * 256 * 6
    assert lib.ssd_gemm_x3_weights_bytes(100, 48, 1) == 0                  # K must be a multiple of 32
    assert lib.ssd_gemm_x3_weights_bytes(0, 512, 1) == 0
    assert lib.ssd_gemm_x3_split_weights(None, OK_PTR, 128, 256, 1, None) == -3
    assert lib.ssd_gemm_x3_split_weights(OK_PTR, ODD_PTR, 128, 256, 1, None) == -5
    assert lib.ssd_gemm_x3_split_weights(OK_PTR, OK_PTR, 128, 40, 1, None) == -1
    assert lib.ssd_gemm_planes_x3(None, OK_PTR, OK_PTR, 128, 256, 128, 128, 1, None) == -3
    assert lib.ssd_gemm_planes_x3(OK_PTR, OK_PTR, ODD_PTR, 128, 256, 128, 128, 1, None) == -5
    assert lib.ssd_gemm_planes_x3(OK_PTR, OK_PTR, OK_PTR, 128, 48, 128, 128, 1, None) == -1          # K % 32
    assert lib.ssd_gemm_planes_x3(OK_PTR, OK_PTR, OK_PTR, 128, 256, 300, 100, 1, None) == -1         # more columns than the padded filter rows
    assert lib.ssd_gemm_planes_x3(OK_PTR, OK_PTR, OK_PTR, 0, 256, 128, 128, 1, None) == -1
    assert lib.ssd_gemm_planes_f32(None, OK_PTR, OK_PTR, 128, 256, 128, 128, 1, None) == -3
    # weight job with limb outputs: forward K = Ci = 256 as limbs (bit 0), backward K = co_pad = 128 as f32
    job = _lib.WeightJob()
    job.w0 = job.w1 = OK_PTR; job.out_fwd = OK_PTR; job.out_bwd = OK_PTR
    job.co0 = job.co = 128; job.ci = 256; job.taps = 9; job.co_pad = 128; job.kind = 0
    job.pad0 = 0
    assert lib.ssd_weight_job_blocks(C.byref(job)) == (128 * 256 + 256 * 128 + 255) // 256
    job.pad0 = 1
    assert lib.ssd_weight_job_blocks(C.byref(job)) == (128 * 256 // 8 + 256 * 128 + 255) // 256
    job.pad0 = 3
    assert lib.ssd_weight_job_blocks(C.byref(job)) == (128 * 256 // 8 + 256 * 128 // 8 + 255) // 256


def test_asm_guard_flags_a_fragment_register_touched_before_its_wait():
    """build.py's guard for the inline-asm `ds_read` idiom (asm_guard.py): a copy of a read's destination before the covering
    `s_waitcnt lgkmcnt` is a violation; after it, or after a counted wait that retires that read, it is not; compiler-issued LDS
    operations count as queue entries."""
    from objectdetection_ssd_amd import asm_guard as G
    head = ["\t.type\tk1,@function", "k1:"]
    rd = lambda dst, addr: ["\t;;#ASMSTART", f"\tds_read_b128 {dst}, {addr}", "\t;;#ASMEND"]
    ok = head + rd("v[8:11]", "v0") + rd("v[12:15]", "v1") + ["\ts_waitcnt lgkmcnt(1)", "\tv_mov_b32_e32 v40, v9",
                                                               "\ts_waitcnt lgkmcnt(0)", "\tv_mfma_f32_32x32x16_bf16 a[0:15], v[8:11], v[12:15], a[0:15]", "\ts_endpgm"]
    assert G.check_assembly(ok) == []
    early = head + rd("v[8:11]", "v0") + rd("v[12:15]", "v1") + ["\ts_waitcnt lgkmcnt(1)", "\tv_mov_b32_e32 v40, v13", "\ts_endpgm"]
    bad = G.check_assembly(early)
    assert len(bad) == 1 and bad[0][0] == "k1" and "v13" in bad[0][2] and "v[12:15]" in bad[0][3]
    # a compiler-visible ds_read issued after the asm read keeps it in flight under lgkmcnt(1)
    mixed = head + rd("v[8:11]", "v0") + ["\tds_read_b64 v[20:21], v2", "\ts_waitcnt lgkmcnt(1)", "\tv_add_f32_e32 v30, v8, v8",
                                          "\ts_waitcnt vmcnt(0)", "\tv_add_f32_e32 v31, v20, v20", "\ts_endpgm"]
    assert G.check_assembly(mixed) == []                     # lgkmcnt(1) retired the asm read; the compiler read is the compiler's business
    spill = head + rd("v[8:11]", "v0") + ["\tscratch_store_dwordx4 off, v[8:11], s0", "\ts_waitcnt lgkmcnt(0)", "\ts_endpgm"]
    assert len(G.check_assembly(spill)) == 1
    # the raw simm16 form of a wait: 0xC07F = lgkmcnt(0) with the other counters open
    raw = head + rd("v[8:11]", "v0") + ["\ts_waitcnt 0xc07f", "\tv_mov_b32_e32 v40, v9", "\ts_endpgm"]
    assert G.check_assembly(raw) == []


def test_oracle_pinned_to_its_own_stored_tensors_is_the_unpinned_oracle():
    """`ssd300_forward(pinned=...)` (the rounding-pinned comparison of the bf16-tensor mode): pinned to the activations of its OWN bf16
    run the oracle reports zero distance at every tensor and returns the same outputs and gradients; a perturbed stored tensor shows up
    in the report at that tensor and nowhere upstream."""
    import ssd_oracle as O
    params = O.ssd300_random_params(3)
    x = torch.randn(1, 3, 300, 300, generator=torch.Generator().manual_seed(4))
    boxes, classes = [torch.tensor([[.1, .2, .6, .7], [.5, .4, .9, .95]])], [torch.tensor([3., 11.])]

    def run(pinned=None):
        P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        acts = {}
        loc, conf = O.ssd300_forward(x, P, operand_round="bf16", store_round=True, acts=acts, pinned=pinned)
        a1, a2 = O.multibox_loss_torch(loc, conf, boxes, classes)
        (a1 + a2).backward()
        return loc.detach(), conf.detach(), {k: v.grad.clone() for k, v in P.items()}, acts
    loc0, conf0, g0, acts = run()
    trunk = {n for n in acts if n.startswith(("a1", "a2", "a3", "a4", "a5")) or n == "n4_3"}
    pin = {"fwd": dict(acts), "bwd": {}, "bf16": trunk, "report": {}}
    loc1, conf1, g1, _ = run(pin)
    assert torch.equal(loc0, loc1) and torch.equal(conf0, conf1)
    assert set(pin["report"]) == {n + ":fwd" for n in acts}
    # its own unrounded result is within half a bf16 spacing of what it stored (trunk) / equal to it (f32 tensors)
    assert all(v <= (0.5 if k.split(":")[0] in trunk else 0.0) for k, v in pin["report"].items()), pin["report"]
    assert max(v for k, v in pin["report"].items() if k.split(":")[0] in trunk) > 0.45
    # (the gradients pass through an unrounded identity where the unpinned run rounds them: equal forward, close backward)
    for k in g0:
        assert float((g0[k] - g1[k]).norm() / g0[k].norm().clamp_min(1e-30)) <= 3e-2, k
    bad = dict(acts)
    bad["a3_2"] = acts["a3_2"] * 1.01                                   # 2.6 bf16 spacings off
    pin2 = {"fwd": bad, "bwd": {}, "bf16": trunk, "report": {}}
    run(pin2)
    # a3_2 itself is off, a3_3 was computed from the perturbed a3_2 and is off too; a4_1 sees the stored (clean) a3_3 / p3 again
    assert pin2["report"]["a3_2:fwd"] > 1.0 and pin2["report"]["a3_3:fwd"] > 1.0
    assert pin2["report"]["a3_1:fwd"] <= 0.5 and pin2["report"]["a4_1:fwd"] <= 0.5
