import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import ssd_oracle as O
from helpers import synth_gt
from objectdetection_ssd_amd import Model, Losses
lr, bs = 1e-3, 2
x = np.random.default_rng(31).standard_normal((bs, 3, 300, 300), dtype=np.float32)
boxes, classes = synth_gt(np.random.default_rng(32), bs)
params = O.ssd300_random_params(5)
def groups(named):
    b = [p for n, p in named if n.endswith(".bias")]; o = [p for n, p in named if not n.endswith(".bias")]
    return b, o
P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
bc, oc = groups(P.items())
opt_c = torch.optim.SGD([{"params": bc, "lr": 2 * lr}, {"params": oc}], lr=lr, momentum=0.9, weight_decay=5e-4)
cnn = Model.SSD_300()
named = dict(cnn.named_parameters())
with torch.no_grad():
    for k, v in params.items(): named[k].copy_(v)
cnn = cnn.cuda().train()
named = dict(cnn.named_parameters())
bg, og = groups([(n, named[n]) for n in cnn._engine.names])
opt_g = torch.optim.SGD([{"params": bg, "lr": 2 * lr}, {"params": og}], lr=lr, momentum=0.9, weight_decay=5e-4)
xt = torch.from_numpy(x)
for it in range(2):
    opt_c.zero_grad()
    loc, conf = O.ssd300_forward(xt, P)
    l1, l2 = O.multibox_loss_torch(loc, conf, [torch.from_numpy(b) for b in boxes], [torch.from_numpy(c) for c in classes])
    (l1 + l2).backward()
    opt_g.zero_grad()
    lg, cg = cnn(xt.cuda())
    m1, m2 = Losses.ssd((lg, cg), [torch.from_numpy(c).cuda() for c in classes], [torch.from_numpy(b).cuda() for b in boxes])
    (m1 + m2).backward()
    print(f"it {it}: cpu {l1.item():.5f} {l2.item():.5f} gpu {m1.item():.5f} {m2.item():.5f}")
    worst = []
    for k in cnn._engine.names:
        gc, gg = P[k].grad, named[k].grad.cpu()
        worst.append((float((gc - gg).norm() / gc.norm().clamp_min(1e-20)), k, float(gc.norm())))
    worst.sort(reverse=True)
    print("  grad rel diff worst:", [(round(a, 5), b, round(c, 4)) for a, b, c in worst[:4]])
    v0 = {k: named[k]._version for k in cnn._engine.names}
    opt_c.step(); opt_g.step()
    dv = {named[k]._version - v0[k] for k in cnn._engine.names}
    print("  version bumps on gpu params:", dv)
    worst = []
    for k in cnn._engine.names:
        worst.append((float((P[k].detach() - named[k].detach().cpu()).abs().max()), k))
    worst.sort(reverse=True)
    print("  param abs diff worst:", [(round(a, 7), b) for a, b in worst[:4]])
