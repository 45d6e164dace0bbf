"""GPU box: time individual fp_conv_igemm / fp_conv_wgrad shapes with HIP events (run under rocprofv3 --pmc for counters).

    python scripts/conv_microbench.py [igemm|wgrad] [reps]
"""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from footprints_amd import ops, _lib as L

which = sys.argv[1] if len(sys.argv) > 1 else "igemm"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N = 12
# (name, gather, OH, OW, IH, IW, C0, C1, Nout, K, stride)
SHAPES = [
    ("o41 fwd up2 64->32 @192x640", L.GATHER_FWD_REFLECT_UP2, 192, 640, 192, 640, 64, 0, 32, 3, 1),
    ("o42 fwd 32->32 @192x640", L.GATHER_FWD_REFLECT, 192, 640, 192, 640, 32, 0, 32, 3, 1),
    ("b4.post1 fwd up2cat 64+64->64 @96x320", L.GATHER_FWD_REFLECT_UP2, 96, 320, 96, 320, 64, 64, 64, 3, 1),
    ("b4.post2 fwd 64->64 @96x320", L.GATHER_FWD_REFLECT, 96, 320, 96, 320, 64, 0, 64, 3, 1),
    ("l1 fwd zero 64->64 @48x160", L.GATHER_FWD_ZERO, 48, 160, 48, 160, 64, 0, 64, 3, 1),
    ("l2 fwd zero 128->128 @24x80", L.GATHER_FWD_ZERO, 24, 80, 24, 80, 128, 0, 128, 3, 1),
    ("l3 fwd zero 256->256 @12x40", L.GATHER_FWD_ZERO, 12, 40, 12, 40, 256, 0, 256, 3, 1),
    ("l4 fwd zero 512->512 @6x20", L.GATHER_FWD_ZERO, 6, 20, 6, 20, 512, 0, 512, 3, 1),
]
dev = "cuda"
for name, gather, OH, OW, IH, IW, C0, C1, Nout, K, stride in SHAPES:
    up2 = gather == L.GATHER_FWD_REFLECT_UP2
    src0 = torch.rand((N, IH // 2, IW // 2, C0) if up2 else (N, IH, IW, C0), device=dev) - 0.5
    src1 = (torch.rand((N, IH, IW, C1), device=dev) - 0.5) if C1 else None
    w = (torch.rand((Nout, C0 + C1, K, K), device=dev) - 0.5) * 0.1
    d = ops.make_desc(N, OH, OW, IH, IW, C0, C1, Nout, K, stride, K // 2, gather, act=L.ACT_ELU if gather != L.GATHER_FWD_ZERO else 0)
    flops = 2.0 * N * OH * OW * Nout * K * K * (C0 + C1)
    if which == "igemm":
        wp = torch.empty(ops.packed_weight_elems(Nout, C0 + C1, K), device=dev)
        ops.pack_conv_weight(w, wp)
        y = torch.empty((N, OH, OW, Nout), device=dev)
        b = torch.zeros(Nout, device=dev)
        run = lambda: ops.conv_igemm(d, src0, src1, wp, y, bias=b if gather != L.GATHER_FWD_ZERO else None)
    else:
        d.act = 0
        dz = torch.rand((N, OH, OW, Nout), device=dev) - 0.5
        dw = torch.empty_like(w)
        run = lambda: ops.conv_wgrad(d, src0, src1, dz, dw)
    run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        run()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / reps * 1e3
    print("%-42s %9.1f us  %6.1f TF/s  (%.2f GF)" % (name, us, flops / us / 1e6, flops / 1e9), flush=True)
