"""GPU box: time individual fp_conv_igemm / fp_conv_wgrad shapes with HIP events (run under rocprofv3 --pmc for counters).

    python scripts/conv_microbench.py [igemm|wgrad] [reps]
"""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from footprints_amd import ops, _lib as L
L.LIB_PATH = __import__("os").environ.get("FP_LIB", L.LIB_PATH)

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
for name, gather, OH, OW, IH, IW, C0, C1, Nout, K, stride in (SHAPES if which in ("igemm", "wgrad") else []):
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
        if which == "igemm" and not up2 and ops.conv3x3_bf3_supported(d):
            wp3 = ops.pack_conv_weight_bf3(w, torch.empty(ops.packed_weight_elems_bf3(Nout, C0, K), device=dev))
            run3 = lambda: ops.conv3x3_bf3(d, src0, wp3, y, bias=b if gather != L.GATHER_FWD_ZERO else None)
            run3()
            torch.cuda.synchronize()
            s3, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s3.record()
            for _ in range(reps):
                run3()
            e3.record()
            torch.cuda.synchronize()
            us3 = s3.elapsed_time(e3) / reps * 1e3
            print("%-42s %9.1f us  %6.1f TF/s  (bf16x3 split)" % (name, us3, flops / us3 / 1e6), flush=True)
    else:
        d.act = 0
        dz = torch.rand((N, OH, OW, Nout), device=dev) - 0.5
        dw = torch.empty_like(w)
        run = lambda: ops.conv_wgrad(d, src0, src1, dz, dw)
        if not up2 and ops.conv_wgrad_bf3_supported(d):
            run3 = lambda: ops.conv_wgrad_bf3(d, src0, dz, dw)
            run3()
            torch.cuda.synchronize()
            s3, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s3.record()
            for _ in range(reps):
                run3()
            e3.record()
            torch.cuda.synchronize()
            us3 = s3.elapsed_time(e3) / reps * 1e3
            print("%-42s %9.1f us  %6.1f TF/s  (bf16x3 split)" % (name, us3, flops / us3 / 1e6), flush=True)
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

UP2 = [("b1.post1 256+256->256 @12x40", 6, 20, 256, 256, 256), ("b2.post1 128+128->128 @24x80", 12, 40, 128, 128, 128),
       ("b3.post1 64+64->64 @48x160", 24, 80, 64, 64, 64), ("b4.post1 64+64->64 @96x320", 48, 160, 64, 64, 64),
       ("o41 64->32 @192x640", 96, 320, 64, 0, 32)]
if which == "up2":
    # every upsample conv of the two decoders: old fused-gather path vs phase decomposition (+ skip half at hi-res)
    for name, h, w_, C0, C1, Nout in UP2:
        lo = torch.rand((N, h, w_, C0), device=dev) - 0.5
        sk = (torch.rand((N, 2 * h, 2 * w_, C1), device=dev) - 0.5) if C1 else None
        wt = (torch.rand((Nout, C0 + C1, 3, 3), device=dev) - 0.5) * 0.1
        b = torch.zeros(Nout, device=dev)
        y = torch.empty((N, 2 * h, 2 * w_, Nout), device=dev)
        wp = ops.pack_conv_weight(wt, torch.empty(ops.packed_weight_elems(Nout, C0 + C1, 3), device=dev))
        wph = ops.pack_up2_weight(wt, torch.empty(ops.up2_packed_weight_elems(Nout, C0), device=dev), 0, C0)
        d_old = ops.make_desc(N, 2 * h, 2 * w_, 2 * h, 2 * w_, C0, C1, Nout, 3, 1, 1, L.GATHER_FWD_REFLECT_UP2, act=L.ACT_ELU)
        old = lambda: ops.conv_igemm(d_old, lo, sk, wp, y, bias=b)
        if C1:
            wsk = ops.pack_conv_weight_slice(wt, torch.empty(ops.packed_weight_elems(Nout, C1, 3), device=dev), C0, C1)
            d_sk = ops.make_desc(N, 2 * h, 2 * w_, 2 * h, 2 * w_, C1, 0, Nout, 3, 1, 1, L.GATHER_FWD_REFLECT)
            def new():
                ops.conv_igemm(d_sk, sk, None, wsk, y)
                ops.conv_up2_phase_fwd(lo, wph, b, y, act=L.ACT_ELU, addend=y)
        else:
            new = lambda: ops.conv_up2_phase_fwd(lo, wph, b, y, act=L.ACT_ELU)
        wph3 = ops.pack_up2_weight_bf3(wt, torch.empty(ops.up2_packed_weight_elems(Nout, C0) * 3 // 2, device=dev), 0, C0)
        runs = [("old", old), ("phase", new), ("ph-bf3", lambda: ops.conv_up2_phase_fwd_bf3(lo, wph3, b, y, act=L.ACT_ELU, addend=y if C1 else None))]
        if C1:
            runs += [("skip", lambda: ops.conv_igemm(d_sk, sk, None, wsk, y)),
                     ("ph-only", lambda: ops.conv_up2_phase_fwd(lo, wph, b, y, act=L.ACT_ELU, addend=y))]
        for tag, run in runs:
            run()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                run()
            e.record()
            torch.cuda.synchronize()
            print("%-32s %-6s %9.1f us" % (name, tag, s.elapsed_time(e) / reps * 1e3), flush=True)

if which == "up2bwd":
    # dgrad / wgrad of the upsample convs as a 4x4 stride-2 pad-3 conv over dZ on the (h+2) x (w+2) extended low-res grid
    for name, h, w_, C0, C1, Nout in UP2:
        dz = torch.rand((N, 2 * h, 2 * w_, Nout), device=dev) - 0.5
        ext = torch.empty((N, h + 2, w_ + 2, C0), device=dev)
        wt = (torch.rand((C0, Nout, 4, 4), device=dev) - 0.5) * 0.1
        wp = ops.pack_conv_weight(wt, torch.empty(ops.packed_weight_elems(C0, Nout, 4), device=dev))
        d = ops.make_desc(N, h + 2, w_ + 2, 2 * h, 2 * w_, Nout, 0, C0, 4, 2, 3, L.GATHER_FWD_ZERO)
        xv = torch.empty((N, 2 * h, 2 * w_, C0 + C1), device=dev)
        wo = (torch.rand((Nout, C0 + C1, 3, 3), device=dev) - 0.5) * 0.1
        wpo = ops.pack_conv_weight_dgrad(wo, torch.empty(ops.packed_weight_elems(Nout, C0 + C1, 3, True), device=dev))
        wp3d = ops.pack_up2_weight_dgrad_bf3(wo, torch.empty(ops.up2_packed_weight_elems(C0, Nout) * 3 // 2, device=dev), 0, C0)
        d_old = ops.make_desc(N, 2 * h, 2 * w_, 2 * h, 2 * w_, Nout, 0, C0 + C1, 3, 1, 1, L.GATHER_DGRAD_REFLECT)
        dlow = torch.empty((N, h, w_, C0), device=dev)
        dsk = torch.empty((N, 2 * h, 2 * w_, C1), device=dev) if C1 else None
        lowx = torch.rand((N, h + 2, w_ + 2, C0), device=dev)
        dk4 = torch.empty((C0, Nout, 4, 4), device=dev)
        dwo = torch.empty_like(wo)
        lo = torch.rand((N, h, w_, C0), device=dev)
        sk = torch.rand((N, 2 * h, 2 * w_, C1), device=dev) if C1 else None
        d_wold = ops.make_desc(N, 2 * h, 2 * w_, 2 * h, 2 * w_, C0, C1, Nout, 3, 1, 1, L.GATHER_FWD_REFLECT_UP2)
        runs = [("dgrad-old", lambda: (ops.conv_igemm(d_old, dz, None, wpo, xv), ops.up2cat_bwd(xv, N, h, w_, C0, C1, dlow, ylow=lo, dskip=dsk))),
                ("dgrad-4x4s2", lambda: ops.conv_igemm(d, dz, None, wp, ext)),
                ("dgrad-ph-bf3", lambda: ops.conv_up2_phase_dgrad_bf3(dz, wp3d, ext)),
                ("wgrad-old", lambda: ops.conv_wgrad(d_wold, lo, sk, dz, dwo)),
                ("wgrad-4x4s2", lambda: ops.conv_wgrad(d, dz, None, lowx, dk4))]
        if ops.up2_phase_wgrad_supported(N, h, w_, C0, Nout):
            runs.append(("wgrad-phase", lambda: ops.conv_up2_phase_wgrad(lo, dz, dwo, 0)))
            if hasattr(ops._lib.load(), "fp_conv_up2_phase_wgrad_bf3"):
                runs.append(("wgrad-phase-bf3", lambda: ops.conv_up2_phase_wgrad(lo, dz, dwo, 0, bf3=True)))
        if C1:
            wps = ops.pack_conv_weight_dgrad(wo[:, C0:].contiguous(), torch.empty(ops.packed_weight_elems(Nout, C1, 3, True), device=dev))
            d_s = ops.make_desc(N, 2 * h, 2 * w_, 2 * h, 2 * w_, Nout, 0, C1, 3, 1, 1, L.GATHER_DGRAD_REFLECT)
            d_ws = ops.make_desc(N, 2 * h, 2 * w_, 2 * h, 2 * w_, C1, 0, Nout, 3, 1, 1, L.GATHER_FWD_REFLECT)
            dws = torch.empty((Nout, C1, 3, 3), device=dev)
            runs += [("dgrad-skip", lambda: ops.conv_igemm(d_s, dz, None, wps, dsk)),
                     ("wgrad-skip", lambda: ops.conv_wgrad(d_ws, sk, None, dz, dws))]
        for tag, run in runs:
            run()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                run()
            e.record()
            torch.cuda.synchronize()
            print("%-32s %-12s %9.1f us" % (name, tag, s.elapsed_time(e) / reps * 1e3), flush=True)
