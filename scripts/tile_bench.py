"""GPU box: the fp16-pair tile convolution (fp_conv3x3_hp) over the shapes of a KITTI step, one line per shape: time, fraction of the dense
fp16 MFMA peak of the three executed products, relative L2 error against float64 on the first repetition.
   [FP_LIB=...] python scripts/tile_bench.py [tag] [reps]      (shapes: decoder forward with bias + ELU, reflection data gradient with ELU')"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from footprints_amd import ops, _lib as L      # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("FP_LIB", "default"))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
SHAPES = [  # Cin, Cout, H, W, N, mode
    (64, 64, 96, 320, 12, "fwd"), (64, 64, 96, 320, 12, "dgrad"), (128, 64, 96, 320, 12, "fwd"), (64, 128, 96, 320, 12, "dgrad"),
    (64, 32, 192, 640, 12, "fwd"), (32, 32, 192, 640, 12, "fwd"), (32, 32, 192, 640, 12, "dgrad"),
    (64, 64, 48, 160, 12, "fwd"), (64, 64, 48, 160, 12, "dgrad"), (128, 128, 24, 80, 12, "fwd"), (256, 256, 12, 40, 12, "fwd"),
    (512, 512, 6, 20, 12, "fwd"), (64, 64, 48, 160, 12, "enc"),
]
if os.environ.get("TILE_SHAPES"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["TILE_SHAPES"].split(",")]


def timeit(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for (C, Co, H, W, N, mode) in SHAPES:
    torch.manual_seed(1)
    x = torch.randn(N, H, W, C, device="cuda")
    x = x * (torch.rand_like(x) > 0.3)
    w = torch.randn(Co, C, 3, 3, device="cuda") * 0.1
    b = torch.randn(Co, device="cuda") * 0.1
    dg = mode == "dgrad"
    slot_w = torch.zeros(ops.amax_elems(), dtype=torch.int32, device="cuda")
    if dg:      # the forward convolution maps Co -> C channels with V = w^T [C, Co, 3, 3]; x plays its output gradient
        V = w.permute(1, 0, 2, 3).contiguous()
        wph = ops.pack_conv_weight_hp(V, torch.empty(ops.packed_weight_elems_hp(C, Co, 3, True), device="cuda"), slot_w, True)
    else:
        wph = ops.pack_conv_weight_hp(w, torch.empty(ops.packed_weight_elems_hp(Co, C, 3, False), device="cuda"), slot_w, False)
    slot_x = ops.amax_f32(x, torch.zeros(ops.amax_elems(), dtype=torch.int32, device="cuda"))
    slot_y = torch.zeros(ops.amax_elems(), dtype=torch.int32, device="cuda")
    y = torch.empty(N, H, W, Co, device="cuda")
    if mode == "fwd":
        d = ops.make_desc(N, H, W, H, W, C, 0, Co, 3, 1, 1, L.GATHER_FWD_REFLECT, act=L.ACT_ELU)
        run = lambda: ops.conv3x3_hp(d, x, wph, y, slot_x, slot_w, amax_out=slot_y, bias=b)
    elif mode == "enc":       # encoder forward: zero padding, no bias / activation (BatchNorm follows)
        d = ops.make_desc(N, H, W, H, W, C, 0, Co, 3, 1, 1, L.GATHER_FWD_ZERO, act=0)
        run = lambda: ops.conv3x3_hp(d, x, wph, y, slot_x, slot_w, amax_out=slot_y)
    else:
        act_src = torch.randn(N, H, W, Co, device="cuda")
        d = ops.make_desc(N, H, W, H, W, C, 0, Co, 3, 1, 1, L.GATHER_DGRAD_REFLECT, act=0)
        d.epi = L.EPI_ACTGRAD_ELU
        run = lambda: ops.conv3x3_hp(d, x, wph, y, slot_x, slot_w, amax_out=slot_y, actsrc=act_src)
    run()
    torch.cuda.synchronize()
    err = float("nan")
    if N * H * W * Co <= 12 * 96 * 320 * 64:
        x64, w64 = x.double().permute(0, 3, 1, 2), w.double()
        if mode == "fwd":
            ref = F.elu(F.conv2d(F.pad(x64, (1, 1, 1, 1), mode="reflect"), w64, b.double())).permute(0, 2, 3, 1)
        elif mode == "enc":
            ref = F.conv2d(x64, w64, padding=1).permute(0, 2, 3, 1)
        else:
            xin = torch.zeros(N, Co, H, W, dtype=torch.float64, device="cuda", requires_grad=True)
            out = F.conv2d(F.pad(xin, (1, 1, 1, 1), mode="reflect"), w64.permute(1, 0, 2, 3).contiguous())
            out.backward(x64)
            s = act_src.double()
            ref = xin.grad.permute(0, 2, 3, 1) * torch.where(s > 0, torch.ones_like(s), s + 1)
        err = ((y.double() - ref).norm() / ref.norm()).item()
        del ref, x64
    digest = ""
    if os.environ.get("TILE_HASH"):      # bit-for-bit comparison of two builds: hash of the output and the published amax
        import hashlib
        digest = "  sha1 %s amax %08x" % (hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12], int(slot_y.max()) & 0xffffffff)
    us = timeit(run)
    fl = 2.0 * N * H * W * C * Co * 9
    print("%-10s %-5s %3d->%3d @%3dx%3dx%2d  %7.1f us  %6.1f TF/s fp32-equiv  frac %.3f  relL2 %.2e" % (
        tag, mode, C, Co, H, W, N, us, fl / us / 1e6, 3 * fl / us / 1e6 / 2500.0, err) + digest, flush=True)
