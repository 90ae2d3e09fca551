"""N > 1 path on CPU: world_size-2 gloo run of the flat-buffer gradient exchange (ddp.py) and of the
data-parallel loss normalisation checked against the oracle at the global batch."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

import ssd_oracle as O
from helpers import synth_gt


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Conv2d(3, 5, 3)          # 135 + 5 params: slots are padded to multiples of 4
        self.b = nn.Conv2d(5, 7, 1)
        self.g = nn.Parameter(torch.ones(1, 6, 1, 1))
        self._engine = types.SimpleNamespace(names=["a.weight", "a.bias", "b.weight", "b.bias", "g"], _wcache={"x": 1})


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from objectdetection_ssd_amd.ddp import FlatSGDDataParallel
        torch.manual_seed(0)
        net = _Tiny()
        if rank == 1:                               # ranks start different; broadcast must fix that
            with torch.no_grad():
                for p in net.parameters():
                    p.add_(1.0)
        tr = FlatSGDDataParallel(net, lr=0.1)
        assert tr.overlap, "more than one rank: the overlapped exchange is the default"
        assert tr.names == ["a.weight", "b.weight", "g", "a.bias", "b.bias"]      # weights | biases segments
        assert tr.n_w == 136 + 36 + 8 and tr.n == tr.n_w + 8 + 8
        tr.broadcast_parameters(0)
        assert net._engine._wcache == {}
        # --- data-parallel loss: each rank handles its own images with un-normalised sums ----------------
        boxes, classes = synth_gt(np.random.default_rng(5), 4)
        r = np.random.default_rng(6)
        loc = r.standard_normal((4, 8732, 4), dtype=np.float32)
        conf = r.standard_normal((4, 8732, 21), dtype=np.float32)
        sl = slice(2 * rank, 2 * rank + 2)
        mine = O.multibox_loss(loc[sl], conf[sl], boxes[sl], classes[sl])
        n_pos = float(mine["n_pos"])
        g = torch.Generator().manual_seed(100 + rank)
        for p in tr.params:                          # stand-in for backward of the un-normalised loss
            p.grad = torch.randn(p.shape, generator=g)
        local = [p.grad.clone() for p in tr.params]
        tr.reduce_gradients(torch.tensor(n_pos))
        # --- the overlapped exchange (slices all-reduced as the backward fills them) gives the same buffer ---------------
        net2 = _Tiny()
        net2._engine.grad_sink = None
        tr2 = FlatSGDDataParallel(net2, lr=0.1, overlap=True, bucket_bytes=256)       # 256 B buckets: several slices even here
        assert net2._engine.grad_sink is not None and len(tr2._bucket_rng) >= 2
        assert tr2._bucket_rng[0][0] == 0 and tr2._bucket_rng[-1][1] == tr2.n_w
        tr2.zero_grad()
        by_name = dict(zip(tr.names, local))
        for nm in ["g", "b.bias", "b.weight", "a.bias", "a.weight"]:              # the order a backward pass produces them in
            net2._engine.grad_sink(nm, by_name[nm])
        tr2.reduce_gradients(torch.tensor(n_pos))
        assert torch.equal(tr2.flat_grad[:tr2.n + 1], tr.flat_grad[:tr.n + 1]), "overlapped exchange differs"
        assert float(tr2.inv_npos) == float(tr.inv_npos)
        tr2.zero_grad()
        net2._engine.grad_sink("g", by_name["g"])
        try:
            tr2.reduce_gradients(torch.tensor(n_pos))
            raise AssertionError("incomplete backward must be refused")
        except RuntimeError as e:
            assert "did not deliver" in str(e)
        sums = torch.tensor([float(mine["loc_loss"]) * n_pos, float(mine["conf_loss"]) * n_pos, n_pos], dtype=torch.float64)
        dist.all_reduce(sums)
        gathered = [None] * world
        dist.all_gather_object(gathered, [t.numpy() for t in local])
        q.put((rank, tr.flat_param.clone().numpy(), [v.clone().numpy() for v in tr.grad_views], float(tr.inv_npos),
               float(tr.flat_grad[tr.n]), sums.numpy(), gathered))
    finally:
        dist.destroy_process_group()


def _torch_sgd_kernel(param, grad, buf, lr, momentum, weight_decay, grad_scale=None, first_step=False):
    """TEST stand-in for the GPU-only `ssd_sgd_momentum` kernel so that the optimizer's bookkeeping can be rehearsed on CPU
    tensors under gloo (the kernel itself is held against torch.optim.SGD bit for bit in the -m gpu tests)."""
    g = grad * grad_scale if grad_scale is not None else grad.clone()
    g = g + weight_decay * param
    if first_step:
        buf.copy_(g)
    else:
        buf.mul_(momentum).add_(g)
    param.sub_(lr * buf)


def _resume_worker(rank, world, port, q, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from objectdetection_ssd_amd import ddp
        ddp.ops.sgd_momentum_ = _torch_sgd_kernel

        def grads(tr, step):
            g = torch.Generator().manual_seed(1000 * step + rank)
            for p in tr.params:
                p.grad = torch.randn(p.shape, generator=g)

        def one_step(tr, step):
            tr.zero_grad()
            grads(tr, step)
            tr.reduce_gradients(torch.tensor(3.0 + rank))
            tr.step()

        torch.manual_seed(0)
        a = ddp.FlatSGDDataParallel(_Tiny(), lr=0.1)
        a.broadcast_parameters(0)
        assert isinstance(a, torch.optim.Optimizer) and len(a.param_groups) == 2
        assert [g["lr"] for g in a.param_groups] == [0.2, 0.1]                      # train.py:53: biases at 2x lr come first
        assert a.state_dict()["state"] == {}                                        # no momentum before the first step (as torch)
        sched = torch.optim.lr_scheduler.StepLR(a, step_size=2, gamma=0.1)          # train.py:57 accepts it
        for s_ in range(3):
            one_step(a, s_)
            sched.step()
        assert abs(a.param_groups[1]["lr"] - 0.01) < 1e-12 and abs(a.param_groups[0]["lr"] - 0.02) < 1e-12
        for g in a.param_groups:                                                   # train_function.py:29-30 resets the lr on resume
            g["lr"] = 0.05
        ckpt = os.path.join(tmp, f"ckpt{rank}.pt")
        torch.save({"opt": a.state_dict(), "par": a.flat_param.clone()}, ckpt)
        # the interrupted run: a fresh model + optimizer restored from the checkpoint
        torch.manual_seed(1)
        b = ddp.FlatSGDDataParallel(_Tiny(), lr=123.0, momentum=0.5)
        sd = torch.load(ckpt, weights_only=True)
        with torch.no_grad():
            b.flat_param.copy_(sd["par"])
        b.load_state_dict(sd["opt"])
        assert [g["lr"] for g in b.param_groups] == [0.05, 0.05] and b.param_groups[1]["momentum"] == 0.9
        assert b._has_momentum and torch.equal(b.flat_mom, a.flat_mom) and b.steps == 3
        one_step(a, 3)
        one_step(b, 3)
        same = torch.equal(a.flat_param, b.flat_param) and torch.equal(a.flat_mom, b.flat_mom)
        # a restore WITHOUT momentum would have restarted it: the two runs must then differ (guards against a vacuous comparison)
        c = ddp.FlatSGDDataParallel(_Tiny(), lr=0.05)
        with torch.no_grad():
            c.flat_param.copy_(sd["par"])
        for g in c.param_groups:
            g["lr"] = 0.05
        one_step(c, 3)
        q.put((rank, same, not torch.equal(c.flat_param, a.flat_param), a.flat_param.clone().numpy()))
    finally:
        dist.destroy_process_group()


def test_data_parallel_optimizer_resumes_with_its_momentum_world2(tmp_path):
    """train_function.py:27-30,114-120: `optimizer.state_dict()` is saved every epoch and restored on resume, the lr is forced
    back, a StepLR sits on top.  FlatSGDDataParallel must survive that round trip: one step after save -> fresh process state ->
    load equals the uninterrupted run bit for bit, on both ranks, and both ranks still hold identical parameters."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_resume_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(o[1] for o in out), "resumed run differs from the uninterrupted one"
    assert all(o[2] for o in out), "momentum-less restart did not differ: the comparison proves nothing"
    assert np.array_equal(out[0][3], out[1][3])


def test_flat_allreduce_and_global_normalisation_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, par0, g0, inv0, np0, sums0, gath), (_, par1, g1, inv1, np1, sums1, _) = out
    assert np.array_equal(par0, par1)                                 # broadcast made the replicas identical
    for a, b in zip(g0, g1):
        assert np.array_equal(a, b)                                   # every rank holds the same reduced gradient
    for k, a in enumerate(g0):
        np.testing.assert_allclose(a, gath[0][k] + gath[1][k], rtol=1e-6)     # = sum over ranks
    # the one extra element carries n_pos: global normaliser, identical on both ranks
    boxes, classes = synth_gt(np.random.default_rng(5), 4)
    r = np.random.default_rng(6)
    loc = r.standard_normal((4, 8732, 4), dtype=np.float32)
    conf = r.standard_normal((4, 8732, 21), dtype=np.float32)
    ref = O.multibox_loss(loc, conf, boxes, classes)                  # the reference's loss at the GLOBAL batch
    assert np0 == np1 == float(ref["n_pos"])
    assert abs(inv0 - 1.0 / ref["n_pos"]) < 1e-9 and inv0 == inv1
    np.testing.assert_allclose(sums0[0] / sums0[2], ref["loc_loss"], rtol=1e-5)
    np.testing.assert_allclose(sums0[1] / sums0[2], ref["conf_loss"], rtol=1e-5)
    # per-image gradients: un-normalised local gradient * 1/n_pos_global == global-batch gradient
    for rank in range(2):
        sl = slice(2 * rank, 2 * rank + 2)
        mine = O.multibox_loss(loc[sl], conf[sl], boxes[sl], classes[sl])
        np.testing.assert_allclose(mine["dloc"] * mine["n_pos"] * inv0, ref["dloc"][sl], rtol=1e-5, atol=1e-9)
        # hard-negative mining is per image (Losses.py:189-194), so the selection does not depend on the shard
        np.testing.assert_allclose(mine["dconf"] * mine["n_pos"] * inv0, ref["dconf"][sl], rtol=1e-5, atol=1e-9)


# ---- world sizes 4 and 8: bucket boundaries and arrival order of the overlapped exchange ------------------------------------
class _Wide(nn.Module):
    """Nine weight tensors of uneven sizes (slots pad to multiples of 4) so that 1 KB buckets cut the weight segment in several
    places, some exactly on a parameter boundary, some mid-way through a run of small tensors."""

    def __init__(self):
        super().__init__()
        sizes = [(7, 5, 3, 3), (64, 3, 1, 1), (3, 3, 3, 3), (33, 8, 1, 1), (5, 5, 1, 1), (128, 2, 1, 1), (1, 6, 1, 1), (16, 16, 1, 1), (9, 1, 1, 1)]
        names = []
        for i, sh in enumerate(sizes):
            c = nn.Conv2d(sh[1], sh[0], (sh[2], sh[3]))
            setattr(self, f"l{i}", c)
            names += [f"l{i}.weight", f"l{i}.bias"]
        self._engine = types.SimpleNamespace(names=names, _wcache={})


def _order_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from objectdetection_ssd_amd.ddp import FlatSGDDataParallel
        torch.manual_seed(0)
        a, b, c = _Wide(), _Wide(), _Wide()
        plain = FlatSGDDataParallel(a, lr=0.1, overlap=False)             # the single all-reduce: the reference of this test
        over = FlatSGDDataParallel(b, lr=0.1, overlap=True, bucket_bytes=1024)
        tiny = FlatSGDDataParallel(c, lr=0.1, overlap=True, bucket_bytes=4)            # every weight its own bucket
        assert len(over._bucket_rng) >= 3 and len(tiny._bucket_rng) == len(tiny.w_names)
        for tr in (over, tiny):                                                        # buckets tile the weight segment exactly
            assert tr._bucket_rng[0][0] == 0 and tr._bucket_rng[-1][1] == tr.n_w
            assert all(x[1] == y[0] for x, y in zip(tr._bucket_rng, tr._bucket_rng[1:]))
        # the engine hooks are installed in both modes; destinations are views of the flat buffer
        for net, tr in ((a, plain), (b, over)):
            assert net._engine.grad_sink is not None and net._engine.sink_owns_grads and net._engine.sink_early == tr.overlap
            dst = net._engine.grad_out(("l1.weight",))
            assert dst.data_ptr() == tr._view["l1.weight"].data_ptr() and dst.numel() == 192
            assert net._engine.grad_out(("l0.weight", "l1.weight")) is None           # 315 elements pad to 316: not contiguous
            assert net._engine.grad_out(("l1.weight", "l2.weight")).numel() == 192 + 81
            assert net._engine.grad_out(("nope",)) is None
        g = torch.Generator().manual_seed(500 + rank)
        local = {n: torch.randn(p.shape, generator=g) for n, p in zip(plain.names, plain.params)}
        n_pos = torch.tensor(2.0 + rank)
        # plain: half of the gradients are written in place (the engine path), half arrive as foreign tensors
        plain.zero_grad()
        for i, n in enumerate(plain.names):
            if i % 2 == 0:
                a._engine.grad_out((n,)).copy_(local[n].reshape(-1))
                a._engine.grad_sink(n, plain._view[n])
            else:
                a._engine.grad_sink(n, local[n])
        plain.reduce_gradients(n_pos)
        # overlapped: three different arrival orders (a backward's reverse order, forward order, a per-rank-independent shuffle --
        # every rank must issue the collectives of the buckets in the same order, so the shuffle is seeded identically)
        results, slices = [], []
        for tr, net in ((over, b), (tiny, c)):
            for order in (list(reversed(tr.names)), list(tr.names), [tr.names[i] for i in torch.randperm(len(tr.names), generator=torch.Generator().manual_seed(9)).tolist()]):
                tr.zero_grad()
                for n in order:
                    net._engine.grad_sink(n, local[n])
                tr.reduce_gradients(n_pos)
                # (a ring all-reduce adds the ranks' contributions in an order that depends on where an element sits in the reduced
                # buffer, so slices and the whole buffer agree to rounding, not bit for bit, once more than two ranks add up)
                results.append(torch.allclose(tr.flat_grad[:tr.n + 1], plain.flat_grad[:plain.n + 1], rtol=1e-5, atol=1e-6)
                               and float(tr.inv_npos) == float(plain.inv_npos))
                slices.append(tr.flat_grad[:tr.n].clone().numpy())
        # a second delivery of one gradient within a step is refused
        over.zero_grad()
        b._engine.grad_sink("l0.bias", local["l0.bias"])
        try:
            b._engine.grad_sink("l0.bias", local["l0.bias"])
            twice = False
        except RuntimeError:
            twice = True
        expect_npos = sum(2.0 + r for r in range(world))
        q.put((rank, results, twice, float(plain.flat_grad[plain.n]) == expect_npos, plain.flat_grad[:plain.n].clone().numpy(),
               torch.cat([local[n].reshape(-1) for n in plain.names]).numpy(), slices))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_overlapped_exchange_bucket_boundaries_and_order_world_4_and_8(world):
    """The N = 4 and N = 8 ranks of the driver's scaling runs, rehearsed under gloo: whatever the bucket size and the order the
    gradients arrive in, the overlapped exchange leaves the same flat buffer as the single all-reduce, on every rank, and that
    buffer holds the sum over ranks; gradients written in place and gradients handed over as tensors mix freely."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_order_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = sum(o[5] for o in out)
    for rank, results, twice, npos_ok, flat, _, slices in out:
        assert all(results), (rank, results)
        assert twice and npos_ok
        assert np.array_equal(flat, out[0][4])                          # every rank holds the same reduced buffer, bit for bit
        for mine, first in zip(slices, out[0][6]):
            assert np.array_equal(mine, first)                          # ... in the overlapped exchange too, whatever the arrival order
    # the flat buffer pads every slot to 4 floats: compare parameter by parameter
    from objectdetection_ssd_amd.ddp import FlatSGDDataParallel
    ref = FlatSGDDataParallel(_Wide(), lr=0.1)
    pos = 0
    for i, sz in enumerate(ref._sizes):
        np.testing.assert_allclose(out[0][4][ref._offs[i]:ref._offs[i] + sz], total[pos:pos + sz], rtol=1e-5, atol=1e-6)
        pos += sz


def _bf16_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gc
        from objectdetection_ssd_amd.ddp import FlatSGDDataParallel
        torch.manual_seed(0)
        a, b, c = _Wide(), _Wide(), _Wide()
        exact = FlatSGDDataParallel(a, lr=0.1, overlap=False)
        over16 = FlatSGDDataParallel(b, lr=0.1, overlap=True, bucket_bytes=1024, grad_dtype=torch.bfloat16)
        one16 = FlatSGDDataParallel(c, lr=0.1, overlap=False, grad_dtype=torch.bfloat16)
        g = torch.Generator().manual_seed(900 + rank)
        local = {n: torch.randn(p.shape, generator=g) * (10.0 ** float(torch.randint(-3, 3, (1,), generator=g))) for n, p in zip(exact.names, exact.params)}
        n_pos = torch.tensor(301.0 + 2 * rank)                       # not representable in bf16 once summed: must travel as f32
        for tr, net in ((exact, a), (over16, b), (one16, c)):
            tr.zero_grad()
            for n in reversed(tr.names):
                net._engine.grad_sink(n, local[n])
            tr.reduce_gradients(n_pos)
        gathered = [None] * world
        dist.all_gather_object(gathered, {n: t.numpy() for n, t in local.items()})
        # lifetime of the hooks: close() takes them off, a dropped optimizer takes them off, a replaced one is left alone
        d = _Wide()
        t1 = FlatSGDDataParallel(d, lr=0.1, overlap=False)
        assert d._engine.grad_sink is not None and d._engine.sink_owns_grads
        t1.close()
        closed = d._engine.grad_sink is None and d._engine.grad_out is None and not d._engine.sink_owns_grads
        t2 = FlatSGDDataParallel(d, lr=0.1, overlap=False)
        t3 = FlatSGDDataParallel(d, lr=0.1, overlap=False)          # replaces t2's hooks
        del t2
        gc.collect()
        kept = d._engine.grad_sink is not None and d._engine.grad_sink._token is t3._token
        del t3
        gc.collect()
        dropped = d._engine.grad_sink is None
        q.put((rank, exact.flat_grad[:exact.n + 1].clone().numpy(), over16.flat_grad[:over16.n + 1].clone().numpy(),
               one16.flat_grad[:one16.n + 1].clone().numpy(), gathered, (closed, kept, dropped)))
    finally:
        dist.destroy_process_group()


def test_bf16_gradient_payload_rounding_and_hook_lifetime_world2():
    """`FlatSGDDataParallel(grad_dtype=torch.bfloat16)` (the 52.6 MB payload of BASELINE configs[2]): weight gradients are rounded once
    to bf16, summed by the collective, and widened into the f32 buffer -- every reduced weight gradient is a bf16 value within bf16
    rounding of the exact sum, equal on both ranks and equal between the overlapped and the single exchange; the bias segment and
    the positive-prior count are the exact f32 sums.  Plus the life cycle of the engine hooks."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bf16_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from objectdetection_ssd_amd.ddp import FlatSGDDataParallel
    ref = FlatSGDDataParallel(_Wide(), lr=0.1, overlap=False)
    for rank, exact, over16, one16, gathered, life in out:
        assert all(life), life
        assert np.array_equal(over16, out[0][2]) and np.array_equal(one16, out[0][3])        # both ranks hold the same buffers
        assert np.array_equal(over16, one16)                                                    # slices or one collective: same rounding points
        assert exact[ref.n] == over16[ref.n] == 301.0 + 303.0                                   # the count is exact
        assert np.array_equal(exact[ref.n_w:ref.n], over16[ref.n_w:ref.n])                      # biases travel as f32
        for i, n in enumerate(ref.names[:len(ref.w_names)]):
            lo, sz = ref._offs[i], ref._sizes[i]
            got = torch.from_numpy(over16[lo:lo + sz])
            g0, g1 = (torch.from_numpy(gathered[r][n]).reshape(-1) for r in range(2))
            assert torch.equal(got, got.bfloat16().float()), n                                  # what arrived is a bf16 value
            emu = (g0.bfloat16().float() + g1.bfloat16().float()).bfloat16().float()            # round, add, round
            assert torch.equal(got, emu), n
            bound = 2.0 ** -7 * (g0.abs() + g1.abs()) + 1e-30      # unit roundoff 2^-8, twice (inputs, sum)
            assert bool(((got - torch.from_numpy(exact[lo:lo + sz])).abs() <= bound).all()), n


def test_bench_rendezvous_at_eight_ranks():
    """`python bench.py --gpus 8 --rendezvous-only`: the launch path of the driver's N = 8 run (self-started ranks, process group,
    one all-reduce) without touching a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--backend", "gloo", "--rendezvous-only"],
                       capture_output=True, text=True, timeout=900, env=dict(env, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout
    assert json.loads(line[0]) == {"rendezvous": True, "n_gpus": 8, "backend": "gloo", "ranks_counted": 8}


def test_data_parallel_optimizer_loads_a_single_gpu_checkpoint():
    """train_function.py:27,116: the checkpoint's `optimizer_state_dict` comes from `torch.optim.SGD` over train.py:44-55's groups --
    every requires_grad parameter in named_parameters() order, the dead `model.classifier.*` included.  FlatSGDDataParallel re-indexes
    it by name: each parameter's momentum lands in ITS slot of the flat momentum buffer."""
    from objectdetection_ssd_amd import Model
    from objectdetection_ssd_amd.ddp import FlatSGDDataParallel
    torch.manual_seed(0)
    net = Model.SSD_300()
    biases = [p for n, p in net.named_parameters() if p.requires_grad and n.endswith(".bias")]
    rest = [p for n, p in net.named_parameters() if p.requires_grad and not n.endswith(".bias")]
    ref = torch.optim.SGD([{"params": biases, "lr": 2e-4}, {"params": rest}], lr=1e-4, momentum=0.9, weight_decay=5e-4)
    names = {id(p): n for n, p in net.named_parameters()}
    for gi, grp in enumerate(ref.param_groups):                   # a momentum buffer that encodes (group, position): value = name hash
        for p in grp["params"]:
            if names[id(p)].startswith("model.classifier"):
                continue                                           # never receives a gradient: no state, as in a real run
            ref.state[p]["momentum_buffer"] = torch.full_like(p, float(sum(map(ord, names[id(p)])) % 997))
    sd = ref.state_dict()
    assert [len(g["params"]) for g in sd["param_groups"]] == [38, 39]
    dp = FlatSGDDataParallel(Model.SSD_300(), lr=1.0)
    assert [len(g["params"]) for g in dp.param_groups] == [35, 36]
    dp.load_state_dict(sd)
    assert dp._has_momentum and dp.param_groups[0]["lr"] == 2e-4 and dp.param_groups[1]["lr"] == 1e-4
    for n, mv in zip(dp.names, dp.mom_views):
        assert float(mv.flatten()[0]) == float(sum(map(ord, n)) % 997) and bool((mv == mv.flatten()[0]).all()), n
    # its own layout still round-trips
    dp2 = FlatSGDDataParallel(Model.SSD_300(), lr=1.0)
    dp2.load_state_dict(dp.state_dict())
    assert torch.equal(dp2.flat_mom, dp.flat_mom)
