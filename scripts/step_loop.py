"""GPU box: N full training steps of one workload and nothing else (the process rocprofv3 --pmc / --kernel-trace passes wrap when
per-step kernel figures are wanted without bench.py's extra legs).

    [FP_SERIAL=1] python scripts/step_loop.py [kitti|matterport] [steps] [warmup]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from footprints_amd.model_manager import ModelManager                     # noqa: E402
from footprints_amd.training.train import SEED, TrainStep, synthetic_batch   # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "kitti"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 1
B, H, W = (12, 192, 640) if wl == "kitti" else (4, 512, 640)
torch.manual_seed(SEED)
mm = ModelManager(use_cuda=True)
ts = TrainStep(mm.model, mm.optimiser)
batch = synthetic_batch(B, H, W, "cuda")
for _ in range(warmup):
    ts(batch)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    ts(batch)
e1.record()
torch.cuda.synchronize()
print("%s: %d steps, %.3f ms/step, loss %.5f" % (wl, steps, e0.elapsed_time(e1) / steps, float(ts.losses[20])))
