"""GPU box: fused vs three-launch BatchNorm backward on the same inputs -- where do they differ?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from footprints_amd import ops      # noqa: E402

M, C = int(sys.argv[1]) if len(sys.argv) > 1 else 368640, int(sys.argv[2]) if len(sys.argv) > 2 else 64
g = torch.Generator().manual_seed(3)
r = lambda *s: torch.rand(*s, generator=g)
z = ((r(M, C) * 2 - 1) * (0.25 + 4 * r(1, C)) + (r(1, C) * 6 - 3)).cuda()
gamma, beta = (r(C) + 0.5).cuda(), (r(C) * 2 - 1).cuda()
res, dy = (r(M, C) * 2 - 1).cuda(), (r(M, C) * 2 - 1).cuda()
outs = {}
for fused in (False, True):
    ops._BN_FUSED = fused
    rm, rv, nbt = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), torch.zeros((), dtype=torch.int64, device="cuda")
    mean, invstd, scale, shift = (torch.empty(C, device="cuda") for _ in range(4))
    y = torch.empty_like(z)
    if fused:
        ops.bn_train_fused(z, y, gamma, beta, rm, rv, nbt, mean, invstd, scale, shift, residual=res, relu=True)
    else:
        ops.bn_train_stats(z, gamma, beta, rm, rv, nbt, mean, invstd, scale, shift)
        ops.bn_apply(z, scale, shift, y, residual=res, relu=True)
    dz, gout, dgam, dbet = torch.empty_like(z), torch.empty_like(z), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.bn_bwd(dy, y, z, mean, invstd, gamma, dz, dgam, dbet, g_out=gout)
    torch.cuda.synchronize()
    outs[fused] = dict(y=y, mean=mean, invstd=invstd, dz=dz, gout=gout, dgam=dgam, dbet=dbet)
a, b = outs[False], outs[True]
for k in a:
    d = (a[k].double() - b[k].double()).abs()
    print("%-7s max|diff| %.3e  (max|ref| %.3e)" % (k, d.max().item(), a[k].abs().max().item()))
d = (a["dz"].double() - b["dz"].double()).abs()
rows = (d.amax(dim=1) > 1e-4 * a["dz"].abs().max()).nonzero().flatten()
print("rows with wrong dz: %d of %d; first %s last %s" % (rows.numel(), M, rows[:8].tolist(), rows[-8:].tolist()))
cols = (d.amax(dim=0) > 1e-4 * a["dz"].abs().max()).nonzero().flatten()
print("channels with wrong dz: %s" % cols.tolist()[:64])
