"""HBM-bound kernels of the 192x640 decoder tail (heads, loss, Adam, max-pool) timed alone: algorithmic bytes / launch duration
against the 8 TB/s HBM3E peak.  FP_LIB=<other .so> runs the same table on another build (A/B inside one gpurun call)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from footprints_amd import ops
from footprints_amd.training.train import synthetic_batch

PEAK = 8000.0   # GB/s


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def row(name, us, nbytes):
    gbs = nbytes / us / 1e3
    print("%-44s %9.1f us %9.1f MB %8.0f GB/s  %5.1f %% of HBM peak" % (name, us, nbytes / 1e6, gbs, 100 * gbs / PEAK), flush=True)


def main():
    N = 12
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: torch.rand(*s, device=dev, generator=g) - 0.5
    for (h, w, Cin, scale) in [(192, 640, 32, 1), (96, 320, 64, 2), (48, 160, 64, 4), (24, 80, 128, 8)]:
        x = rnd(N, h, w, Cin)
        wt, b = rnd(2, Cin, 3, 3), rnd(2)
        low = torch.empty(N, h, w, 2, device=dev)
        out = torch.zeros(N, 4, h * scale, w * scale, device=dev)
        gout = rnd(N, 4, h * scale, w * scale)
        dz = torch.empty(N, h, w, 2, device=dev)
        dx = torch.empty(N, h, w, Cin, device=dev)
        dw, db = torch.empty(2, Cin, 3, 3, device=dev), torch.empty(2, device=dev)
        tag = "%dx%d C%d s%d" % (h, w, Cin, scale)
        xb, lb, ob = x.numel() * 4, low.numel() * 4, out.numel() * 2
        row("head_fwd " + tag, timeit(lambda: ops.head_fwd(x, wt, b, low, True)), xb + lb)
        row("head_upsample " + tag, timeit(lambda: ops.head_upsample(low, out, scale, 2)), lb + ob)
        row("head_upsample_bwd " + tag, timeit(lambda: ops.head_upsample_bwd(gout, low, dz, scale, 2, True)), ob + 2 * lb)
        row("head_dgrad(+elu) " + tag, timeit(lambda: ops.head_dgrad(dz, wt, dx, elu_src=x)), 2 * xb + lb)
        row("head_dgrad " + tag, timeit(lambda: ops.head_dgrad(dz, wt, dx)), xb + lb)
        row("head_wgrad " + tag, timeit(lambda: ops.head_wgrad(x, dz, dw, db)), xb + lb)
    # loss forward + backward on the four full-resolution outputs
    batch = synthetic_batch(N, 192, 640, dev)
    preds = [rnd(N, 4, 192, 640) for _ in range(4)]
    dpreds = [torch.empty_like(p) for p in preds]
    losses = torch.empty(21, device=dev)
    pb = preds[0].numel() * 4
    row("loss_fwd_bwd 4 scales", timeit(lambda: ops.loss_fwd_bwd(preds, batch, losses, dpreds)), 8 * pb + 6 * pb // 4)
    n = 31012944
    p, gr, m, v = (torch.zeros(n, device=dev) for _ in range(4))
    row("adam_step 31.0 M", timeit(lambda: ops.adam_step(p, gr, m, v, 1e-4, 0.9, 0.999, 1e-8, 1)), 7 * n * 4)
    xin = rnd(N, 96, 320, 64)
    y = torch.empty(N, 48, 160, 64, device=dev)
    am = torch.empty(N, 48, 160, 64, device=dev, dtype=torch.uint8)
    row("maxpool_fwd 96x320 C64", timeit(lambda: ops.maxpool_fwd(xin, y, am)), xin.numel() * 4 + y.numel() * 5)
    row("maxpool_bwd 96x320 C64", timeit(lambda: ops.maxpool_bwd(y, am, xin)), xin.numel() * 4 + y.numel() * 5)


if __name__ == "__main__":
    main()
