"""RCCL smoke of the data-parallel step with ONE rank (the pool gives one GPU per call): the process group bench.py forms at N > 1
(backend nccl = RCCL, device_id), the overlapped sliced all-reduce of ddp.FlatSGDDataParallel in f32 and bf16 payload, barrier, the
int32 MAX all-reduce of bench.py's spin-up.  One rank moves no bytes between GPUs; it does run every RCCL call the N > 1 path makes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
from objectdetection_ssd_amd import Model, Losses
from objectdetection_ssd_amd.ddp import FlatSGDDataParallel
import bench

class Forced(FlatSGDDataParallel):
    """the N > 1 branches with one rank: every collective is issued (and is the identity)"""
    world = property(lambda self: 2)


dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for payload in (torch.float32, torch.bfloat16):
    torch.manual_seed(0)
    net = Model.SSD_300().to(dev).train()
    tr = Forced(net, lr=1e-4, momentum=0.9, weight_decay=5e-4, overlap=True, grad_dtype=payload, time_exchange=True)
    tr.broadcast_parameters(0)
    x, classes, boxes = bench.synth_batch(4, 1234, dev)
    for it in range(3):
        tr.zero_grad()
        loc, conf = net(x)
        l1, l2, n_pos = Losses.ssd((loc, conf), classes, boxes, norm_mode=1, with_n_pos=True)
        (l1 + l2).backward()
        tr.reduce_and_step(n_pos)
    dist.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([1], device=dev, dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    print(f"payload {payload}: loss {(float(l1.detach()) + float(l2.detach())) / float(n_pos):.5f}, exposed exchange {tr.exposed_exchange_ms(last=2)} ms per step, "
          f"backend {dist.get_backend()}", flush=True)
    tr.close()
dist.destroy_process_group()
print("rccl single-rank path ok")
