"""GPU box: how long does the host need to ISSUE one training step?  Measured on an EMPTY queue (synchronise, issue one step, stop the
clock before it finishes): with a deep backlog the launch calls block on queue space and the figure degenerates into the GPU's step time.
    [FP_PLAN=0|1] python scripts/host_overhead.py [kitti|matterport]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from footprints_amd.model_manager import ModelManager
from footprints_amd.training.train import TrainStep, synthetic_batch
wl = sys.argv[1] if len(sys.argv) > 1 else "kitti"
B, H, W = (12, 192, 640) if wl == "kitti" else (4, 512, 640)
mm = ModelManager()
ts = TrainStep(mm.model, mm.optimiser)
batch = synthetic_batch(B, H, W, "cuda")
for _ in range(6):
    ts(batch)
torch.cuda.synchronize()
issue, total = [], []
for _ in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    issue.append((t1 - t0) * 1e3)
    total.append((t2 - t0) * 1e3)
issue.sort(); total.sort()
print("%s plan=%s: host issue of one step on an empty queue: median %.2f ms (min %.2f); step start-to-finish %.2f ms; launch plan nodes %s" % (
    wl, ts.use_plan, issue[len(issue) // 2], issue[0], total[len(total) // 2], [n for *_, n in ts._plans.values()]))
