"""Debug helper (GPU box): engine feature gradients dF[k] (total gradient wrt encoder features) vs the oracle."""
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
out_ref, feats_ref = R.footprint_network(cpu_batch["image"], Pg, {k: v.clone() for k, v in B.items()}, True, return_features=True)
for f in feats_ref:
    f.retain_grad()
l_ref, _ = R.loss_manager(out_ref, cpu_batch)
l_ref["loss"].backward()
model = FootprintNetwork(pretrained=False)
model.load_state_dict({**P, **B})
model.cuda().train()
eng = model.engine()
cap = {}
eng.debug_hook = lambda i, d: cap.__setitem__(i, d["dout"].clone()) if i in (15, 12, 6, 2) else None
batch = {k: v.cuda() for k, v in cpu_batch.items()}
out = model(batch["image"])
LossManager((0.1, 100), 0.25, compute_viz=False)(out, batch)["loss"].backward()
torch.cuda.synchronize()
nchw = lambda t: t.permute(0, 3, 1, 2).contiguous().cpu()
for i, k in ((15, 4), (12, 3), (6, 2), (2, 1)):
    a, b = nchw(cap[i]).double(), feats_ref[k].grad.double()
    d = (a - b)
    print("dF[%d] (block %d dout): max rel err %.2e ; per-channel mean(err)/max %.2e ; shape %s" % (
        k, i, (d.abs().max() / b.abs().max()).item(), (d.mean((0, 2, 3)).abs().max() / b.abs().max()).item(), tuple(a.shape)))
    # where is the error located?
    e = d.abs().amax(1)[0]
    rows = e.amax(1); cols = e.amax(0)
    print("   err by row :", " ".join("%.1e" % v for v in rows.tolist()[:24]))
    print("   err by col :", " ".join("%.1e" % v for v in cols.tolist()[:24]))
