"""GPU box: training-step time per side-stream layout (FP_STREAM_LAYOUT = pool index of aux, wg, dwg0, dwg1), single-GPU and with the
forced world-of-one data-parallel branch (fp_comm transport, all-reduces on FP_DP_COMM_STREAM).  One fresh process per point.
    python scripts/stream_layout_sweep.py [layout ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAYOUTS = sys.argv[1:] or ["0,1,2,2", "0,1,2,3", "0,1,1,2", "0,0,1,2", "0,1,2,1", "0,1,2,0", "0,1,0,2", "0,1,1,1", "0,0,1,1"]
CODE = r"""
import os, sys, torch
sys.path.insert(0, %r)
from footprints_amd.model_manager import ModelManager
from footprints_amd.training.train import SEED, TrainStep, synthetic_batch
torch.manual_seed(SEED)
mm = ModelManager(use_cuda=True)
ts = TrainStep(mm.model, mm.optimiser, distributed=bool(int(os.environ.get("FP_DP_FORCE", "0"))))
batch = synthetic_batch(12, 192, 640, "cuda")
for _ in range(8):
    ts(batch)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    ts(batch)
e1.record()
torch.cuda.synchronize()
print("MS %%.3f" %% (e0.elapsed_time(e1) / 30))
""" % ROOT

for lay in LAYOUTS:
    row = []
    for dp, comm in ((0, ""), (1, "dwg0"), (1, "own")):
        env = dict(os.environ, FP_STREAM_LAYOUT=lay, FP_DP_FORCE=str(dp), FP_DP_TRANSPORT="rccl", FP_DP_COMM_STREAM=comm or "dwg0")
        p = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=300)
        ms = [l.split()[1] for l in p.stdout.splitlines() if l.startswith("MS")]
        row.append(ms[0] if ms else "FAIL(%s)" % (p.stderr or "")[-200:].replace("\n", "|"))
    print("LAYOUT aux,wg,dwg0,dwg1=%-8s single %s  dp(comm on dwg0) %s  dp(comm own stream) %s" % (lay, row[0], row[1], row[2]), flush=True)
