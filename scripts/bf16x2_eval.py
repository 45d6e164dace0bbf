"""GPU box: the opt-in two-term inference mode (FP_EPI_BF16X2) against the exact path and the CPU oracle: output error and forward time.
    python scripts/bf16x2_eval.py [B H W]"""
import os, sys, time
from collections import OrderedDict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from footprints_amd import FootprintNetwork
from oracle import restatement as R
from oracle.cpu_threads import effective_cores
from tests.parity import chan_relerr
torch.set_num_threads(min(effective_cores(), 32))
B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (12, 192, 640)
P, Bf = R.make_state(tag="bf2")
img = R.make_batch(B, H, W, tag="bf2")["image"]
m = FootprintNetwork(pretrained=False)
m.load_state_dict({**P, **Bf})
m.cuda().eval()
x = img.cuda()
res = {}
for mode in ("exact", "bf16x2"):
    m.inference_precision = mode
    with torch.no_grad():
        for _ in range(3):
            out = m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            out = m(x)
        torch.cuda.synchronize()
        res[mode] = ({k: v.clone() for k, v in out.items()}, (time.perf_counter() - t0) / 10 / B * 1e3)
with torch.no_grad():
    ref = R.footprint_network(img[:2].double(), OrderedDict((k, v.double()) for k, v in P.items()),
                              OrderedDict((k, v.double() if v.is_floating_point() else v.clone()) for k, v in Bf.items()), False)
for mode in ("exact", "bf16x2"):
    out, ms = res[mode]
    errs = {k: max(chan_relerr(out[k][:2], ref[k])) for k in out}
    flips = sum(int(((out[k][:2, :2].cpu() > 0) != (ref[k][:, :2] > 0)).sum()) for k in out)
    print("%-7s forward %.4f ms/img   worst per-channel error vs the float64 oracle: %s   mask bits differing from the oracle (logit > 0): %d of %d" % (
        mode, ms, {k: "%.1e" % e for k, e in errs.items()}, flips, 4 * 2 * 2 * H * W))
