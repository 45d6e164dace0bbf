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
for _ in range(3):
    ops.conv_wgrad_bf3(d, x, dz, dw, 0, db=db)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    ops.conv_wgrad_bf3(d, x, dz, dw, 0, db=db)
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) / reps * 1e3
fl = 2.0 * N * H * W * C * Co * 9
print("wgrad_bf3 %d->%d @%dx%dx%d: %.1f us  %.1f TF/s fp32-equivalent  (%.3f of the bf16x6 roof)" % (C, Co, H, W, N, us, fl / us / 1e6, fl / us / 1e6 / (2500 / 6)))
