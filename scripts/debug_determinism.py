"""GPU box: two fresh 2-step runs of the full-size train step; report which state_dict entries differ bitwise."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from footprints_amd.model_manager import ModelManager
from footprints_amd.training.train import TrainStep, synthetic_batch
from oracle import restatement as R
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_network import _load_state

P, Bf = R.make_state(tag="fs")
batch = synthetic_batch(12, 192, 640, "cuda")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
runs = []
for r in range(3):
    mm = ModelManager()
    _load_state(mm.model, P, Bf)
    ts = TrainStep(mm.model, mm.optimiser)
    for _ in range(steps):
        ts(batch)
    torch.cuda.synchronize()
    runs.append(({k: v.clone() for k, v in mm.model.state_dict().items()}, ts.eng.flat_grad.clone(), list(ts.eng.live_names), list(ts.eng.offsets)))
for r in (1, 2):
    bad = [(k, (runs[0][0][k].float() - runs[r][0][k].float()).abs().max().item()) for k in runs[0][0] if not torch.equal(runs[0][0][k], runs[r][0][k])]
    print("run 0 vs %d: %d differing state entries" % (r, len(bad)), bad[:12])
    g0, g1 = runs[0][1], runs[r][1]
    names, offs = runs[0][2], runs[0][3]
    gb = []
    for n, o, nxt in zip(names, offs, offs[1:] + [g0.numel()]):
        if not torch.equal(g0[o:nxt], g1[o:nxt]):
            gb.append((n, (g0[o:nxt] - g1[o:nxt]).abs().max().item()))
    print("   differing gradient tensors: %d" % len(gb), gb[:12])

