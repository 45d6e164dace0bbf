"""Debug helper (GPU box): replay one encoder BasicBlock's backward on the CPU from the engine's own saved tensors."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import restatement as R
from oracle.cpu_threads import effective_cores
from footprints_amd import FootprintNetwork
from footprints_amd.training.losses import LossManager

torch.set_num_threads(min(effective_cores(), 32))
Bn, Hn, Wn = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (2, 192, 640)
P, B = R.make_state(tag="full")
cpu_batch = R.make_batch(Bn, Hn, Wn, tag="full")
model = FootprintNetwork(pretrained=False)
model.load_state_dict({**P, **B})
model.cuda().train()
eng = model.engine()
cap = {}


def hook(i, d):
    if i in (15, 14):
        cap[i] = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in d.items() if k != "B"}
        cap[i]["B"] = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in d["B"].items()}


eng.debug_hook = hook
batch = {k: v.cuda() for k, v in cpu_batch.items()}
out = model(batch["image"])
losses = LossManager((0.1, 100), 0.25, compute_viz=False)(out, batch)
losses["loss"].backward()
torch.cuda.synchronize()
nchw = lambda t: t.permute(0, 3, 1, 2).contiguous().cpu()
rel = lambda a, b: ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()
names = {15: "encoder.layer4.2", 14: "encoder.layer4.1"}
for i in (15, 14):
    c = cap[i]
    pre = names[i]
    x = nchw(c["B"]["x"]).requires_grad_(True)
    w1, w2 = P[pre + ".conv1.weight"], P[pre + ".conv2.weight"]
    z1 = F.conv2d(x, w1, None, 1, 1); z1.retain_grad()
    a1 = F.relu(F.batch_norm(z1, None, None, P[pre + ".bn1.weight"], P[pre + ".bn1.bias"], True, 0.1, 1e-5)); a1.retain_grad()
    z2 = F.conv2d(a1, w2, None, 1, 1); z2.retain_grad()
    o = F.relu(F.batch_norm(z2, None, None, P[pre + ".bn2.weight"], P[pre + ".bn2.bias"], True, 0.1, 1e-5) + x)
    print("block", i, "fwd: z1 %.2e a1 %.2e z2 %.2e out %.2e" % (rel(nchw(c["B"]["z1"]), z1), rel(nchw(c["B"]["a1"]), a1),
                                                              rel(nchw(c["B"]["z2"]), z2), rel(nchw(c["B"]["out"]), o)))
    o.backward(nchw(c["dout"]))
    print("   bwd: dz2 %.2e da1 %.2e dz1 %.2e dx %.2e   (g vs dout*mask %.2e)" % (
        rel(nchw(c["dz2"]), z2.grad), rel(nchw(c["da1"]) * (a1 > 0), a1.grad), rel(nchw(c["dz1"]), z1.grad),
        rel(nchw(c["dnext"]), x.grad), rel(nchw(c["g"]), nchw(c["dout"]) * (o > 0))))
