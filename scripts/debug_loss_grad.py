"""GPU box: the fused loss kernel's d loss / d pred against torch autograd of the oracle's loss on the SAME predictions.
    python scripts/debug_loss_grad.py B H W [tag]"""
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from footprints_amd import FootprintNetwork, ops             # noqa: E402
from oracle import restatement as R                          # noqa: E402

Bn, Hn, Wn = (int(v) for v in sys.argv[1:4])
tag = sys.argv[4] if len(sys.argv) > 4 else "anch"
P, B = R.make_state(tag=tag)
cpu_batch = R.make_batch(Bn, Hn, Wn, tag="%s%d" % (tag, Hn))
model = FootprintNetwork(pretrained=False)
model.load_state_dict({**P, **B})
model.cuda().train()
batch = {k: v.cuda() for k, v in cpu_batch.items()}
with torch.no_grad():
    out = model(batch["image"])
preds = [out[k].contiguous() for k in R.SCALES]
losses = torch.empty(21, device="cuda")
dp = [torch.empty_like(p) for p in preds]
tg = {k: batch[k].contiguous().float() for k in ("visible_ground", "all_ground", "depth", "ground_depth", "moving_object_mask", "depth_mask")}
ops.loss_fwd_bwd(preds, tg, losses, dp, (0.1, 100.0), 0.25)
torch.cuda.synchronize()
for dt in (torch.float32, torch.float64):
    cp = OrderedDict((k, p.cpu().to(dt).requires_grad_(True)) for k, p in zip(R.SCALES, preds))
    l, _ = R.loss_manager(cp, OrderedDict((k, v.to(dt)) for k, v in cpu_batch.items()))
    l["loss"].backward()
    print("reference dtype", dt)
    for si, k in enumerate(R.SCALES):
        g, r = dp[si].cpu().double(), cp[k].grad.double()
        for ch in range(4):
            d = (g[:, ch] - r[:, ch]).abs()
            rel = (g[:, ch] - r[:, ch]).norm() / r[:, ch].norm()
            if rel > 1e-5:
                idx = torch.nonzero(d > 0.1 * d.max())
                print("  %s ch%d rel L2 %.2e  max|d| %.3e  max|ref| %.3e  px with |d| > 0.1 max: %d" % (k, ch, rel, d.max(), r[:, ch].abs().max(), len(idx)))
                for b, y, x in idx[:6].tolist():
                    o = float(preds[si][b, ch, y, x])
                    dd = 1.0 / (0.01 + 9.99 * o)
                    t = float(cpu_batch["ground_depth" if ch == 3 else "depth"][b, y, x])
                    print("     (%d,%d,%d) pred %.9g depth(pred) %.9g target %.9g  e %.3e  gpu %.6e ref %.6e" % (b, y, x, o, dd, t, dd - t, g[b, ch, y, x], r[b, ch, y, x]))
            else:
                print("  %s ch%d rel L2 %.2e" % (k, ch, rel))
