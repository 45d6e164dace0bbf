"""GPU box: one fp_conv_wgrad_bf3 shape in a loop (PMC / A-B timing).   python scripts/wgrad_one.py C Cout H W [N] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from footprints_amd import ops, _lib as L      # noqa: E402

C, Co, H, W = (int(v) for v in sys.argv[1:5])
N = int(sys.argv[5]) if len(sys.argv) > 5 else 12
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
x = torch.rand(N, H, W, C, device="cuda") - 0.5
dz = torch.rand(N, H, W, Co, device="cuda") - 0.5
dw = torch.empty(Co, C, 3, 3, device="cuda")
db = torch.empty(Co, device="cuda")
d = ops.make_desc(N, H, W, H, W, C, 0, Co, 3, 1, 1, L.GATHER_FWD_REFLECT)
am = None
if os.environ.get("WG_HP", "1") != "0":            # fp16-pair operands (the engine's default); WG_HP=0: exact bf16x3 split
    am = (ops.amax_f32(x, ops.new_slot()), ops.amax_f32(dz, ops.new_slot()))
for _ in range(3):
    ops.conv_wgrad_bf3(d, x, dz, dw, 0, db=db, amax=am)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    ops.conv_wgrad_bf3(d, x, dz, dw, 0, db=db, amax=am)
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) / reps * 1e3
fl = 2.0 * N * H * W * C * Co * 9
nprod = 6 if am is None else 3
print("wgrad %s %d->%d @%dx%dx%d: %.1f us  %.1f TF/s fp32-equivalent  (%.3f of the %d-product roof)" % (
    "bf16x3" if am is None else "fp16-pair", C, Co, H, W, N, us, fl / us / 1e6, fl / us / 1e6 / (2500 / nprod), nprod))
