"""GPU box: does the mere existence of an RCCL process group slow the single-GPU training step?  (profiles/round2_notes.md, "Open")
    python scripts/dp_env_probe.py [0|1]     1 = create the process group and issue one all-reduce first"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "1":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    dist.init_process_group("nccl", rank=0, world_size=1)
    dist.all_reduce(torch.ones(1024, device="cuda"))
    torch.cuda.synchronize()
from footprints_amd.model_manager import ModelManager                     # noqa: E402
from footprints_amd.training.train import SEED, TrainStep, synthetic_batch   # noqa: E402

torch.manual_seed(SEED)
mm = ModelManager(use_cuda=True)
ts = TrainStep(mm.model, mm.optimiser)                                     # NOT distributed
batch = synthetic_batch(12, 192, 640, "cuda")
for _ in range(6):
    ts(batch)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ts(batch)
e1.record()
torch.cuda.synchronize()
print("process group %s: %.3f ms/step" % ("present" if len(sys.argv) > 1 and sys.argv[1] == "1" else "absent", e0.elapsed_time(e1) / 20))
