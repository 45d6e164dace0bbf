"""GPU box: where does a train-step gradient leave the float64 truth?  One fwd + loss + bwd at (B, H, W) on the engine, the CPU
oracle in fp32 and in fp64; prints per-stage errors of features, outputs, loss gradients and parameter gradients.

    python scripts/debug_parity_stage.py B H W [tag]
"""
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import restatement as R                      # noqa: E402
from oracle.cpu_threads import effective_cores           # noqa: E402
from tests.parity import chan_relerr, rel_l2              # noqa: E402

torch.set_num_threads(min(effective_cores(), 32))
Bn, Hn, Wn = (int(v) for v in sys.argv[1:4])
tag = sys.argv[4] if len(sys.argv) > 4 else "anch"
P, B = R.make_state(tag=tag)
cpu_batch = R.make_batch(Bn, Hn, Wn, tag="%s%d" % (tag, Hn))
if tag == "natural":          # the low-pass / saturated images and 2^16-range BatchNorm parameters of tests/test_gpu_parity_fullsize.py
    from tests.test_gpu_parity_fullsize import _natural_batch, _wide_range_state
    P, B = _wide_range_state()
    cpu_batch = _natural_batch(Bn, Hn, Wn)


def oracle(dtype):
    Pd = OrderedDict((k, v.detach().clone().to(dtype).requires_grad_(True)) for k, v in P.items())
    Bd = OrderedDict((k, v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in B.items())
    batch = OrderedDict((k, v.to(dtype)) for k, v in cpu_batch.items())
    out, feats = R.footprint_network(batch["image"], Pd, Bd, True, return_features=True)
    for t in list(out.values()) + feats:
        t.retain_grad()
    losses, _ = R.loss_manager(out, batch)
    losses["loss"].backward()
    return Pd, out, feats, losses


P32, o32, f32, l32 = oracle(torch.float32)
P64, o64, f64, l64 = oracle(torch.float64)

from footprints_amd import FootprintNetwork                  # noqa: E402
from footprints_amd.training.losses import LossManager      # noqa: E402
model = FootprintNetwork(pretrained=False)
model.load_state_dict({**P, **B})
model.cuda().train()
batch = {k: v.cuda() for k, v in cpu_batch.items()}
out = model(batch["image"])
for v in out.values():
    v.retain_grad()
losses = LossManager((0.1, 100), 0.25, compute_viz=False)(out, batch)
losses["loss"].backward()
torch.cuda.synchronize()
eng = model.engine()
S = eng.saved
print("features (rel L2 vs fp64):   gpu        cpu32")
for i, f in enumerate(S["feats"]):
    g = f.permute(0, 3, 1, 2)
    print("  f%d %-22s %.2e   %.2e" % (i, tuple(g.shape), rel_l2(g, f64[i].detach()), rel_l2(f32[i].detach(), f64[i].detach())))
dF = [eng._bufs["dF%d" % i][:f.numel()].view(f.shape).permute(0, 3, 1, 2) for i, f in enumerate(S["feats"])]
print("feature gradients from the decoders + encoder (rel L2 vs fp64; dF buffers hold the TOTAL gradient of each feature):")
for i in range(5):
    print("  dF%d  gpu %.2e   cpu32 %.2e" % (i, rel_l2(dF[i], f64[i].grad), rel_l2(f32[i].grad, f64[i].grad)))
print("outputs per channel (max|d|/max|ref| vs fp64):")
for k in R.SCALES:
    print("  %s gpu %s   cpu32 %s" % (k, ["%.1e" % e for e in chan_relerr(out[k], o64[k])], ["%.1e" % e for e in chan_relerr(o32[k], o64[k])]))
    print("      min/max of the depth channels (sigmoid): %.3e %.3e" % (float(o64[k][:, 2:].min()), float(o64[k][:, 2:].max())))
print("loss gradient d loss / d outputs per channel (rel L2 vs fp64):")
for k in R.SCALES:
    g, c, r = out[k].grad, o32[k].grad, o64[k].grad
    print("  %s gpu %s   cpu32 %s" % (k, ["%.1e" % rel_l2(g[:, ch], r[:, ch]) for ch in range(4)], ["%.1e" % rel_l2(c[:, ch], r[:, ch]) for ch in range(4)]))
print("21 losses: max rel err gpu %.2e cpu32 %.2e" % (
    max(abs(float(losses[k]) - float(l64[k])) / max(abs(float(l64[k])), 1e-3) for k in R.LOSS_KEYS),
    max(abs(float(l32[k]) - float(l64[k])) / max(abs(float(l64[k])), 1e-3) for k in R.LOSS_KEYS)))
print("parameter gradients (rel L2 vs fp64), every tensor:")
for n, p in model.named_parameters():
    if P64[n].grad is None:
        continue
    eg, ec = rel_l2(p.grad, P64[n].grad), rel_l2(P32[n].grad, P64[n].grad)
    flag = " <<<" if eg > 3 * max(ec, 2e-5) else ""
    print("  %-58s gpu %.2e cpu32 %.2e%s" % (n, eg, ec, flag))
