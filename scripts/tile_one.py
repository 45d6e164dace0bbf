"""GPU box: one fp_conv3x3_bf3 shape in a loop (PMC / A-B timing).   python scripts/tile_one.py Cin Cout H W [N] [reps] [mode: fwd|dgrad]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from footprints_amd import ops, _lib as L      # noqa: E402

C, Co, H, W = (int(v) for v in sys.argv[1:5])
N = int(sys.argv[5]) if len(sys.argv) > 5 else 12
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
mode = sys.argv[7] if len(sys.argv) > 7 else "fwd"
x = torch.rand(N, H, W, C, device="cuda") - 0.5
w = (torch.rand(Co, C, 3, 3, device="cuda") - 0.5) * 0.1
y = torch.empty(N, H, W, Co, device="cuda")
b = torch.zeros(Co, device="cuda")
dg = mode == "dgrad"
wp3 = ops.pack_conv_weight_bf3(w if not dg else w.permute(1, 0, 2, 3).contiguous(), torch.empty(ops.packed_weight_elems_bf3(Co, C, 3, False), device="cuda"), False)
d = ops.make_desc(N, H, W, H, W, C, 0, Co, 3, 1, 1, L.GATHER_DGRAD_REFLECT if dg else L.GATHER_FWD_REFLECT, act=0 if (dg or mode == "fwdnoact") else L.ACT_ELU)
run = (lambda: ops.conv3x3_bf3(d, x, wp3, y, actsrc=x if C == Co else None)) if dg else (lambda: ops.conv3x3_bf3(d, x, wp3, y, bias=b))
if dg and C == Co:
    d.epi = L.EPI_ACTGRAD_ELU
for _ in range(3):
    run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    run()
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) / reps * 1e3
fl = 2.0 * N * H * W * C * Co * 9
print("conv3x3_bf3 %s %d->%d @%dx%dx%d: %.1f us  %.1f TF/s fp32-equivalent  (%.3f of the bf16x6 roof)" % (mode, C, Co, H, W, N, us, fl / us / 1e6, fl / us / 1e6 / (2500 / 6)))
