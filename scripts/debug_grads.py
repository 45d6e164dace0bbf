"""Debug helper (GPU box): per-parameter gradient error of the HIP engine vs the CPU oracle, in backward order."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import restatement as R
from oracle.cpu_threads import effective_cores
from footprints_amd import FootprintNetwork
from footprints_amd.training.losses import LossManager

torch.set_num_threads(min(effective_cores(), 32))
Bn, Hn, Wn = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (2, 192, 640)
P, B = R.make_state(tag="full")
cpu_batch = R.make_batch(Bn, Hn, Wn, tag="full")
tr = R.OracleTrainer(P, B)
out_ref, l_ref = tr.forward_backward(cpu_batch)
model = FootprintNetwork(pretrained=False)
model.load_state_dict({**P, **B})
model.cuda().train()
batch = {k: v.cuda() for k, v in cpu_batch.items()}
out = model(batch["image"])
losses = LossManager((0.1, 100), 0.25, compute_viz=False)(out, batch)
losses["loss"].backward()
rows = []
for n, p in model.named_parameters():
    g = tr.P[n].grad
    if g is None:
        continue
    d = p.grad.cpu().double() - g.double()
    e = (d.abs().max() / g.double().abs().max().clamp_min(1e-30)).item()
    l2 = (d.norm() / g.double().norm().clamp_min(1e-30)).item()
    rows.append((n, e, l2))
for n, e, l2 in reversed(rows):
    if "encoder" in n:
        print("%-50s max/max %.2e   L2rel %.2e %s" % (n, e, l2, "  <<<<" if l2 > 1e-3 else ""))
# ReLU mask flips between the engine forward and the oracle forward (the non-smooth points of the network)
rec = []
R.footprint_network(cpu_batch["image"], {k: v for k, v in P.items()}, {k: v.clone() for k, v in B.items()}, True, record=rec)
eng = model.engine()
flips = []
for i, blk in enumerate(eng.saved["blocks"]):
    o = blk["out"].permute(0, 3, 1, 2).cpu()
    flips.append(int(((o > 0) != (rec[i] > 0)).sum()))
print("ReLU mask flips per block output (engine vs oracle):", flips, " of ", [int(r.numel()) for r in rec])
