"""Debug helper (GPU box): gradient wrt every encoder block output, engine vs oracle."""
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
Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
rec = []
out_ref = R.footprint_network(cpu_batch["image"], Pg, {k: v.clone() for k, v in B.items()}, True, record=rec)
l_ref, _ = R.loss_manager(out_ref, cpu_batch)
l_ref["loss"].backward()
model = FootprintNetwork(pretrained=False)
model.load_state_dict({**P, **B})
model.cuda().train()
eng = model.engine()
cap = {}
eng.debug_hook = lambda i, d: cap.__setitem__(i, {k: d[k].clone() for k in ("dout", "dz2", "dz1", "g")})
batch = {k: v.cuda() for k, v in cpu_batch.items()}
out = model(batch["image"])
LossManager((0.1, 100), 0.25, compute_viz=False)(out, batch)["loss"].backward()
torch.cuda.synchronize()
nchw = lambda t: t.permute(0, 3, 1, 2).contiguous().cpu()
rel = lambda a, b: ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()
for i in range(15, -1, -1):
    print("block %2d  dout vs oracle grad(out_i): %.2e   shape %s" % (i, rel(nchw(cap[i]["dout"]), rec[i].grad), tuple(rec[i].shape)))

# ---- extra: is the buffer handed from block 14 to block 13 stable? and is a CPU replay of block 14 == oracle? ----
import torch.nn.functional as F
cap2 = {}
eng.debug_hook = lambda i, d: cap2.__setitem__(i, {k: (v.clone() if torch.is_tensor(v) else None) for k, v in d.items() if k != "B"})
out = model(batch["image"])
LossManager((0.1, 100), 0.25, compute_viz=False)(out, batch)["loss"].backward()
torch.cuda.synchronize()
print("dnext@hook14 vs dout@hook13: %.2e" % rel(nchw(cap2[14]["dnext"]), nchw(cap2[13]["dout"])))
print("dnext@hook14 vs oracle grad(out13): %.2e" % rel(nchw(cap2[14]["dnext"]), rec[13].grad))
print("dout@hook13  vs oracle grad(out13): %.2e" % rel(nchw(cap2[13]["dout"]), rec[13].grad))
print("dnext@hook15 vs oracle grad(out14): %.2e" % rel(nchw(cap2[15]["dnext"]), rec[14].grad))
print("dout@hook14  vs oracle grad(out14): %.2e" % rel(nchw(cap2[14]["dout"]), rec[14].grad))
