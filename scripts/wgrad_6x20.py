import os, sys, torch
sys.path.insert(0, "/root/repo")
from footprints_amd import ops, _lib as L
for (C, Co, H, W) in ((512, 512, 6, 20), (512, 256, 6, 20), (256, 256, 6, 20)):
    N = 12
    x = torch.rand(N, H, W, C, device="cuda") - 0.5
    dz = torch.rand(N, H, W, Co, device="cuda") - 0.5
    dw = torch.empty(Co, C, 3, 3, device="cuda")
    d = ops.make_desc(N, H, W, H, W, C, 0, Co, 3, 1, 1, L.GATHER_FWD_ZERO)
    fl = 2.0 * N * H * W * C * Co * 9
    for name, run in (("bf3", (lambda: ops.conv_wgrad_bf3(d, x, dz, dw, 0)) if ops.conv_wgrad_bf3_supported(d) else None), ("fp32", lambda: ops.conv_wgrad(d, x, None, dz, dw))):
        if run is None:
            print(C, Co, name, "unsupported"); continue
        for _ in range(3): run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): run()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 20 * 1e3
        print("%d->%d @6x20 %s: %.1f us %.1f TF" % (C, Co, name, us, fl / us / 1e6))
    ref = dw.clone()
