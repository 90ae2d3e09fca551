"""Measure, on an MI355X, how far every parameter gradient of the train step is from its references, per tensor and per
engine, and write the table the tests use as bars (tests/golden/grad_bars.json; bar = max(2 x measured, floor)).

    python tools/grad_bars.py [out.json]        (GPU box; default gpurun_out/grad_bars.json -- copy it to tests/golden/)

This table is a REGRESSION RECORD of unpinned runs (two independent evaluations whose ReLU / max-pool / hard-negative decisions
differ in a few places out of ~1e8, each a discrete jump of one gradient path).  The bar that does not depend on the code under
test is `tests/test_gpu_path.py::test_train_step_gradients_vs_decision_pinned_f64_oracle`: the f64 oracle follows the HIP
forward's own decisions and every tensor must be within a FIXED 1e-5 (direct) / 5e-5 (Winograd) of it (`grad_measure.PINNED_BAR`).

Metrics (71 tensors each):
  f64_wino / f64_direct     relative L2 to an f64 CPU evaluation of the oracle network (bs 2, all six scales positive)
  cpu32_f64                 the f32 CPU oracle's own distance to f64 on that input (context: what "the reference" achieves)
  gold_wino / gold_direct   | ||g|| - ||g_ref|| | / ||g_ref|| against the reference's f32 CPU gradients (tests/golden/network.npz)
  gold_elem_wino / _direct  relative L2 to the six gradient tensors of the reference that the fixture holds in full
  wd32                      relative L2 between the two engines on bench.py's batch of 32
  bf16_oracle               bf16-operand mode against the oracle's bf16-operand train step (bs 2)
  bf16_act / f32_act        relative L2 of every activation of the forward to the oracle's, layer by layer (bf16 / f32 operands)
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import ssd_oracle as O  # noqa: E402
import grad_measure as M  # noqa: E402
from objectdetection_ssd_amd import Model  # noqa: E402


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "grad_bars.json")
    z = np.load(os.path.join(ROOT, "tests", "golden", "network.npz"))
    params = O.ssd300_random_params(int(z["param_seed"]))
    net = Model.SSD_300()
    named = dict(net.named_parameters())
    with torch.no_grad():
        for k, v in params.items():
            named[k].copy_(v)
    net = net.to(M.DEV)
    table = {"device": torch.cuda.get_device_name(0), "floor_rel": M.FLOOR_REL, "floor_norm": M.FLOOR_NORM}
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    _, _, _, _, g64 = M.f64_oracle_grads(params)
    _, _, _, _, g32 = M.f64_oracle_grads(params, dtype=torch.float32)
    table["cpu32_f64"] = {k: M.rel_l2(g32[k], g64[k]) for k in g64}
    gx, gc, gb = M.golden_case(z)
    gold = dict(zip([str(n) for n in z["grad_names"]], z["grad_l2"]))
    for eng in M.ENGINES:
        M.set_engine(net, eng)
        *_, g = M.gpu_f64_case(net)
        table["f64_" + eng] = {k: M.rel_l2(g[k], g64[k]) for k in g64}
        *_, gg = M.train_step(net, gx, gc, gb)
        table["gold_" + eng] = {k: abs(float(gg[k].double().norm()) - float(r)) / max(float(r), 1e-30) for k, r in gold.items()}
        table["gold_elem_" + eng] = {f[2:]: M.rel_l2(gg[f[2:]], torch.from_numpy(z[f])) for f in z.files if f.startswith("g_")}
    xb, cb, bb = M.bench_batch()
    res = {}
    for eng in M.ENGINES:
        M.set_engine(net, eng)
        res[eng] = M.train_step(net, xb, cb, bb)
    table["wd32"] = {k: M.rel_l2(res["wino"][4][k], res["direct"][4][k]) for k in res["direct"][4]}
    table["wd32_out"] = {"loc": float((res["wino"][0] - res["direct"][0]).abs().max() / res["direct"][0].abs().max().clamp_min(1)),
                         "conf": float((res["wino"][1] - res["direct"][1]).abs().max() / res["direct"][1].abs().max().clamp_min(1))}
    del res
    lo, co, a1, a2, gbo = M.f64_oracle_grads(params, operand_round="bf16", dtype=torch.float32, store_round=net._engine.bf16_tensors)
    M.set_engine(net, "wino", "bf16")
    loc, conf, l1, l2, g = M.gpu_f64_case(net)
    M.set_engine(net, "wino", "f32")
    table["bf16_oracle"] = {k: M.rel_l2(g[k], gbo[k]) for k in gbo}
    table["bf16_act"], bl, bc = M.layerwise_forward_distance(net, params, "bf16")
    table["f32_act"], fl, fc = M.layerwise_forward_distance(net, params, "f32")
    with torch.no_grad():
        lo32, co32 = O.ssd300_forward(torch.from_numpy(M.f64_case()[0]), params)
    table["bf16_mode_noise"] = {"loc": float((lo - lo32).abs().max() / lo32.abs().max().clamp_min(1)),
                                "conf": float((co - co32).abs().max() / co32.abs().max().clamp_min(1))}
    print("layerwise bf16:", {k: "%.1e" % v for k, v in table["bf16_act"].items()}, bl, bc)
    print("layerwise f32 :", {k: "%.1e" % v for k, v in table["f32_act"].items()}, fl, fc)
    print("bf16 mode noise (oracle bf16 vs oracle f32):", table["bf16_mode_noise"])
    table["bf16_oracle_out"] = {"loc": float((loc.cpu() - lo).abs().max() / lo.abs().max().clamp_min(1)),
                                "conf": float((conf.cpu() - co).abs().max() / co.abs().max().clamp_min(1)),
                                "loc_loss": abs(l1 - a1) / max(1.0, abs(a1)), "conf_loss": abs(l2 - a2) / max(1.0, abs(a2))}
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    for k in sorted(g64):
        print("%-28s f64: wino %.2e direct %.2e cpu32 %.2e | gold: wino %.2e direct %.2e | wd32 %.2e | bf16 %.2e" % (
            k, table["f64_wino"][k], table["f64_direct"][k], table["cpu32_f64"][k], table["gold_wino"].get(k, -1),
            table["gold_direct"].get(k, -1), table["wd32"][k], table["bf16_oracle"][k]))
    print("wd32_out", table["wd32_out"], "bf16_oracle_out", table["bf16_oracle_out"])


if __name__ == "__main__":
    main()
