import sys, torch
sys.path.insert(0, "/root/repo")
from footprints_amd import ops, _lib as L
N, dev = 12, "cuda"
for (H, W) in ((48, 160), (96, 320)):
    for Cin in (16, 32, 64, 128, 256):
        Cout = 64
        x = torch.rand(N, H, W, Cin, device=dev) - 0.5
        w = (torch.rand(Cout, Cin, 3, 3, device=dev) - 0.5) * 0.1
        b = torch.zeros(Cout, device=dev)
        y = torch.empty(N, H, W, Cout, device=dev)
        d = ops.make_desc(N, H, W, H, W, Cin, 0, Cout, 3, 1, 1, L.GATHER_FWD_REFLECT, act=L.ACT_ELU)
        wp3 = ops.pack_conv_weight_bf3(w, torch.empty(ops.packed_weight_elems_bf3(Cout, Cin, 3), device=dev))
        run = lambda: ops.conv3x3_bf3(d, x, wp3, y, bias=b)
        run(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): run()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 20 * 1e3
        print("%dx%d Cin %3d (%2d chunks): %7.1f us" % (H, W, Cin, Cin // 16, us), flush=True)
