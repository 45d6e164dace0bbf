"""GPU box: how long does the host need to ISSUE one training step (no synchronisation) vs the GPU to run it?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from footprints_amd.model_manager import ModelManager
from footprints_amd.training.train import TrainStep, synthetic_batch
mm = ModelManager()
ts = TrainStep(mm.model, mm.optimiser)
batch = synthetic_batch(12, 192, 640, "cuda")
for _ in range(5):
    ts(batch)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    ts(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue %.2f ms/step, total %.2f ms/step (GPU drains %.2f ms after the last issue)" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, (t2 - t1) * 1e3))
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    ts(batch)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
