"""What does `v_pk_fma_f32 D, S0, S1, S2 op_sel:[0,1,0]` compute when it is wrong?  (profiles/round4_notes.md section 12)
head_wgrad A/B builds (FP_LIB): FP_HEAD_WGRAD_FMA=4 multiplies by the pair S1 = (0, x), =6 by S1 = (3x, x), both through op_sel:[0,1,0] (low and
high result read S1's HIGH half).  With x = 1 and dZ = 1 everywhere every partial sum is an exact small integer (the number of contributing
pixels), so a result that read S1's LOW half instead shows as a deficit of exactly k (build 4: k products were 0) or an excess of exactly 2k
(build 6: k products were 3).  Run alone and next to the bf16-MFMA tile convolution on a second stream."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from footprints_amd import ops, _lib as L
N, H, W, C = 12, 192, 640, 32
x = torch.ones(N, H, W, C, device="cuda")
dz = torch.ones(N, H, W, 2, device="cuda")
dw = torch.empty(2, C, 3, 3, device="cuda"); db = torch.empty(2, device="cuda")
xin = torch.rand(N, 96, 320, 64, device="cuda"); w = torch.rand(64, 64, 3, 3, device="cuda") * 0.05
wp3 = ops.pack_conv_weight_bf3(w, torch.empty(ops.packed_weight_elems_bf3(64, 64, 3), device="cuda"))
y = torch.empty(N, 96, 320, 64, device="cuda")
d = ops.make_desc(N, 96, 320, 96, 320, 64, 0, 64, 3, 1, 1, L.GATHER_FWD_REFLECT, act=L.ACT_ELU)
side = torch.cuda.Stream()
per, nblk = 9 * C * 2 + 2, 1024


def parts(conc):
    if conc:
        with torch.cuda.stream(side):
            for _ in range(3):
                ops.conv3x3_bf3(d, xin, wp3, y)
    ops.head_wgrad(x, dz, dw, db)
    torch.cuda.synchronize()
    return ops.workspace(1, x.device)[:nblk * per * 4].view(torch.float32).view(nblk, per)[:, :9 * C * 2].clone().double()


ref = parts(False)
assert torch.equal(ref, parts(False)), "not reproducible alone"
assert float((ref - ref.round()).abs().max()) == 0.0
print("alone: partial sums are integers, max %d, bit-stable" % int(ref.max()))
for it in range(4):
    p = parts(True)
    delta = p - ref
    bad = delta != 0
    lo = bad.view(nblk, 9, C, 2)[..., 0].sum().item(), bad.view(nblk, 9, C, 2)[..., 1].sum().item()
    vals = delta[bad]
    print("run %d next to the bf16-MFMA convolution: %d of %d partial sums differ (output channel 0 / low result: %d, channel 1 / high result: %d); "
          "integer deltas: %s; min %d max %d; histogram of the 6 most frequent: %s"
          % (it, int(bad.sum()), bad.numel(), lo[0], lo[1], bool((vals == vals.round()).all()), int(vals.min()) if vals.numel() else 0,
             int(vals.max()) if vals.numel() else 0, sorted(((int(v), int(c)) for v, c in zip(*torch.unique(vals, return_counts=True))), key=lambda t: -t[1])[:6]))
