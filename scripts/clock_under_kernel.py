"""GPU box: the shader clock the chip sustains under ONE kernel in a loop (fp_clock_probe before / after: d s_memtime / d s_memrealtime per XCD).
   python scripts/clock_under_kernel.py  -> a table: kernel, time per launch, sustained clock"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from footprints_amd import ops, _lib as L      # noqa: E402

lib = L.load()
khz = int(lib.fp_wall_clock_khz())


def clock_of(run, seconds=1.5):
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    pr = torch.zeros((2, 16), dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib.fp_clock_probe(pr[0].data_ptr(), st)
    s.record()
    n, t0 = 0, time.time()
    while time.time() - t0 < seconds:
        for _ in range(50):
            run()
        n += 50
        torch.cuda.synchronize()
    e.record()
    lib.fp_clock_probe(pr[1].data_ptr(), st)
    torch.cuda.synchronize()
    p = pr.cpu().view(2, 8, 2).double()
    mhz = [float((p[1, x, 0] - p[0, x, 0]) / (p[1, x, 1] - p[0, x, 1]) * khz / 1e3) for x in range(8) if p[1, x, 1] > p[0, x, 1]]
    return s.elapsed_time(e) / n * 1e3, sum(mhz) / len(mhz)


def conv(C, Co, H, W, N=12):
    x = torch.rand(N, H, W, C, device="cuda") - 0.5
    w = (torch.rand(Co, C, 3, 3, device="cuda") - 0.5) * 0.1
    y = torch.empty(N, H, W, Co, device="cuda")
    b = torch.zeros(Co, device="cuda")
    wp3 = ops.pack_conv_weight_bf3(w, torch.empty(ops.packed_weight_elems_bf3(Co, C, 3, False), device="cuda"), False)
    d = ops.make_desc(N, H, W, H, W, C, 0, Co, 3, 1, 1, L.GATHER_FWD_REFLECT, act=L.ACT_ELU)
    return lambda: ops.conv3x3_bf3(d, x, wp3, y, bias=b)


big = torch.empty(256 << 20, dtype=torch.float32, device="cuda")
rows = [("idle-ish: 1 GB fill (HBM-bound, no MFMA)", lambda: big.fill_(1.0)),
        ("tile forward 64 -> 64 @ 96 x 320 x 12 (exact)", conv(64, 64, 96, 320)),
        ("tile forward 32 -> 32 @ 192 x 640 x 12 (exact)", conv(32, 32, 192, 640)),
        ("tile forward 256 -> 256 @ 12 x 40 x 12 (exact)", conv(256, 256, 12, 40))]
print("%-56s %10s %12s" % ("kernel in a loop (~1.5 s)", "us/launch", "shader MHz"))
for name, run in rows:
    us, mhz = clock_of(run)
    print("%-56s %10.1f %12.0f" % (name, us, mhz))
