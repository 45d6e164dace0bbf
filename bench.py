"""bench.py -- headline benchmark of BASELINE.json: training images/sec at 192x640 bs=12 on 1..8 MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                                   (N > 1, one rank per GPU, RCCL)

A "step" = one full training step of the hot path on one per-GPU batch of 12 synthetic 192x640 images:
forward + fused loss + backward + (bucketed gradient all-reduce when N > 1) + fused Adam, all inputs resident
in HBM before the timed region (configs[2] of BASELINE.json; configs[1], the inference-only forward, is reported
beside it as fwd_ms_per_img).  Weak scaling: every rank keeps batch 12.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      dominant kernel family = the convolution forward + data-gradient launches: fp_conv3x3_bf3 (halo-tile kernel with
                fp32 operands split EXACTLY into three bf16 terms, six bf16 MFMA products, fp32 accumulation -- error below the
                fp32 MFMA's own), its phase-decomposed upsample variant, and fp_conv_igemm (fp32 MFMA) for the rest.
                achieved = sum of ALGORITHMIC FLOPs of the launches (the dense convolution of the reference graph, also where
                the phase decomposition executes 4/9 of it) / sum of their durations, measured with HIP events recorded around
                every launch, on the stream it is launched on, during the first timed steps.  `peak` is the native fp32 MFMA
                peak (the dtype is f32); `peak_bf16x6` = dense bf16 peak / 6 is the split kernels' own roof.
                The timed steps run up to five streams concurrently, so a launch shares the chip and its event-to-event
                duration is longer than its exclusive duration; `achieved_exclusive` / `frac_exclusive` are the same quantity
                from extra steps (outside the timed region) with concurrency switched off -- the kernels' own speed.
  step_conv_tflops  the reference graph's conv FLOPs of one step (fwd + dgrad + wgrad) over the measured step time.
  decoder_backward  SURVEY.md section 8(d): both decoders' backward alone (d loss / d outputs -> d loss / d features + all decoder
                weight gradients) timed with HIP events, as achieved_hbm = 7.018 GB / t against 8 TB/s and achieved_mfma =
                1023.9 GFLOP / t against the fp32 MFMA peak (the convolutions are MFMA-bound: F/B 72-755 vs a ridge of ~20).
  step_ms       median / p10 / p90 of the GPU-side step durations inside the timed region.
  cpu_baseline  the CPU oracle (a restatement of the reference's PyTorch CPU path, kind "port") timed on this
                box's host cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
BF16X6_PEAK_TFLOPS = 2500.0 / 6   # dense bf16 MFMA peak / 6 products per fp32-equivalent multiply-add (the split kernels' own roof)
HBM_PEAK_GBS = 8000.0             # same guide: HBM3E peak
DEC_BWD_GB, DEC_BWD_GFLOP = 7.018, 1023.9   # decoder backward, both decoders, KITTI bs=12 (fused-minimum bytes; dgrad + wgrad FLOPs)
B, H, W = 12, 192, 640


def network_conv_gflop(n, h, w):
    """algorithmic GFLOP of ONE training step's convolutions (reference graph: every conv counted as the dense
    Conv2d the reference runs, 2*MACs, forward + data-gradient + weight-gradient; the stem has no data-gradient)."""
    def conv(oh, ow, cin, cout, k):
        return 2.0 * n * oh * ow * cout * cin * k * k
    fwd = conv(h // 2, w // 2, 3, 64, 7)
    tot = 2 * fwd                                            # stem: forward + weight gradient
    res, cin = (h // 4, w // 4), 64
    for cout, blocks, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)):
        for b in range(blocks):
            s = stride if b == 0 else 1
            res = (res[0] // s, res[1] // s)
            f = conv(res[0], res[1], cin, cout, 3) + conv(res[0], res[1], cout, cout, 3)
            if s != 1 or cin != cout:
                f += conv(res[0], res[1], cin, cout, 1)
            fwd += f
            tot += 3 * f
            cin = cout
    dec = 0.0
    r = (h // 32, w // 32)
    for ci, co in ((512, 256), (256, 128), (128, 64), (64, 64)):
        dec += conv(r[0], r[1], ci, co, 3) + conv(r[0], r[1], co, co, 3)          # pre_concat_conv
        r = (r[0] * 2, r[1] * 2)
        dec += conv(r[0], r[1], 2 * co, co, 3) + conv(r[0], r[1], co, co, 3)      # post_concat_conv on cat[up, skip]
    dec += conv(h, w, 64, 32, 3) + conv(h, w, 32, 32, 3)                          # outconv4 ConvBlock at full resolution
    dec += sum(conv(h // s, w // s, c, 2, 3) for s, c in ((8, 128), (4, 64), (2, 64), (1, 32)))   # 2-channel heads
    fwd += 2 * dec
    tot += 3 * 2 * dec
    return fwd / 1e9, tot / 1e9


class KernelTimer:
    """HIP-event bracket around every launch of one kernel family (events go on torch's current stream, which
    is the stream the C ABI launches on)."""

    def __init__(self):
        self.records = []

    def wrap(self, fn, flops_of, key_of=None, name=""):
        def wrapped(first, *a, **k):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = fn(first, *a, **k)
            e.record()
            key = key_of(first, *a, **k) if key_of is not None else name + " " + desc_key(first)
            self.records.append((s, e, flops_of(first, *a, **k), key))
            return r
        return wrapped

    def summary(self):
        ms = sum(r[0].elapsed_time(r[1]) for r in self.records)
        fl = sum(r[2] for r in self.records)
        return len(self.records), ms, fl

    def table(self):
        """Per distinct launch shape: count, mean microseconds, algorithmic TFLOP/s."""
        agg = {}
        for s, e, f, key in self.records:
            a = agg.setdefault(key, [0, 0.0, f])
            a[0] += 1
            a[1] += s.elapsed_time(e)
        rows = [{"shape": k, "launches": n, "avg_us": round(ms / n * 1e3, 1), "gflop": round(f / 1e9, 3),
                 "tflops": round(f * n / (ms * 1e-3) / 1e12, 1) if ms > 0 else 0.0, "total_ms": round(ms, 3)} for k, (n, ms, f) in agg.items()]
        return sorted(rows, key=lambda r: -r["total_ms"])


GATHER_NAMES = ("fwd_zero", "fwd_reflect", "fwd_up2", "dgrad_zero", "dgrad_reflect", "stem")


def desc_key(d):
    return "%s N%d %dx%d<-%dx%d C%d+%d->%d k%d s%d" % (GATHER_NAMES[d.gather], d.N, d.OH, d.OW, d.IH, d.IW, d.C0, d.C1, d.Nout,
                                                      d.KH, d.stride)


def conv_flops(d, *a, **k):
    taps = 1 if d.gather == 5 else d.KH * d.KW
    kk = 147 if d.gather == 5 else d.C0 + d.C1
    return 2.0 * d.N * d.OH * d.OW * d.Nout * taps * kk


def phase_fwd_flops(low, wphase, bias, y, *a, **k):
    """algorithmic FLOPs of the dense 3x3 conv over the x2-upsampled tensor that the phase kernel replaces"""
    N, h, w, C0 = low.shape
    return 2.0 * N * (2 * h) * (2 * w) * y.shape[3] * 9 * C0


def phase_dgrad_flops(dz, wpacked, ext, *a, **k):
    """algorithmic FLOPs of the dense data gradient (3x3 over the hi-res grid) the phase kernel replaces"""
    N, H2, W2, Cout = dz.shape
    return 2.0 * N * H2 * W2 * Cout * 9 * ext.shape[3]


def phase_wgrad_flops(low, dz, *a, **k):
    N, h, w, C0 = low.shape
    return 2.0 * N * (2 * h) * (2 * w) * dz.shape[3] * 9 * C0


def phase_key(tag):
    return lambda low, *a, **k: "%s N%d low %dx%d C%d" % (tag, low.shape[0], low.shape[1], low.shape[2], low.shape[3])


# op name -> (flop function, key function): the forward + data-gradient convolution family (roofline) and the weight gradients
FWD_DGRAD_OPS = {"conv_igemm": (conv_flops, None), "conv3x3_bf3": (conv_flops, None),
                 "conv_up2_phase_fwd": (phase_fwd_flops, phase_key("phase_fwd")),
                 "conv_up2_phase_fwd_bf3": (phase_fwd_flops, phase_key("phase_fwd_bf3")),
                 "conv_up2_phase_dgrad_bf3": (phase_dgrad_flops, lambda dz, *a, **k: "phase_dgrad_bf3 N%d dz %dx%d C%d" % (
                     dz.shape[0], dz.shape[1], dz.shape[2], dz.shape[3]))}
WGRAD_OPS = {"conv_wgrad": (conv_flops, None), "conv_wgrad_slice": (conv_flops, None), "conv_wgrad_bf3": (conv_flops, None),
             "conv_up2_phase_wgrad": (phase_wgrad_flops, phase_key("phase_wgrad"))}


class Instrument:
    """swap a family of footprints_amd.ops entry points for event-timed wrappers, and back"""

    def __init__(self, ops, table):
        self.ops, self.table = ops, table
        self.orig = {n: getattr(ops, n) for n in table}
        self.timer = KernelTimer()

    def install(self):
        for n, (ff, kf) in self.table.items():
            setattr(self.ops, n, self.timer.wrap(self.orig[n], ff, kf, n))

    def remove(self):
        for n, f in self.orig.items():
            setattr(self.ops, n, f)


def _pick_threads():
    """Fastest intra-op thread count among {8, 16, 32, 64, all effective cores} on a representative conv fwd+bwd
    (os.cpu_count() over-reports under cgroup quotas, and oneDNN stops scaling long before 256 threads)."""
    import torch.nn.functional as F
    from oracle.cpu_threads import effective_cores
    eff = effective_cores()
    x = torch.rand(2, 64, 96, 320)
    w = torch.rand(64, 64, 3, 3, requires_grad=True)
    best, best_t = 1, float("inf")
    for n in sorted({min(c, eff) for c in (8, 16, 32, 64, eff)}):
        torch.set_num_threads(n)
        F.conv2d(x, w, padding=1).sum().backward()
        t0 = time.time()
        for _ in range(2):
            F.conv2d(x, w, padding=1).sum().backward()
        dt = time.time() - t0
        if dt < best_t:
            best, best_t = n, dt
    return best, eff


def cpu_baseline(sample_b=2, steps=2):
    from oracle import restatement as R
    cores, eff = _pick_threads()
    torch.set_num_threads(cores)
    P, Bf = R.make_state(tag="bench")
    tr = R.OracleTrainer(P, Bf)
    tr.step(R.make_batch(1, H, W, tag="bench.warm"))
    batch = R.make_batch(sample_b, H, W, tag="bench.cpu")
    t0 = time.time()
    for _ in range(steps):
        tr.step(batch)
    dt = (time.time() - t0) / steps
    # what the shipped trainer does (reference training/train.py:12-14 pins OMP/MKL to one thread): one batch-1 step
    torch.set_num_threads(1)
    t1 = time.time()
    tr.step(R.make_batch(1, H, W, tag="bench.warm"))
    dt1 = time.time() - t1
    torch.set_num_threads(cores)
    return {"value": round(sample_b / dt, 4), "unit": "img/s", "cores": cores, "kind": "port",
            "single_thread": {"value": round(1.0 / dt1, 4), "unit": "img/s", "cores": 1,
                              "sample": "1 full train step at batch 1 with torch.set_num_threads(1), the reference trainer's own setting"},
            "sample": "%d timed full train steps (fwd+loss+bwd+Adam) of the CPU oracle at %dx%d, batch %d of the 12-image "
                      "workload, torch.set_num_threads(%d) = fastest of {8,16,32,64,%d} on this host (%d effective cores), "
                      "after 1 warm-up step" % (steps, H, W, sample_b, cores, eff, eff),
            "s_per_step": round(dt, 3), "host_cores": eff}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--dump-kernels", type=str, default=None, help="write the per-launch-shape timing tables (JSON) here")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from footprints_amd import ops
    from footprints_amd.model_manager import ModelManager
    from footprints_amd.parallel import broadcast_state
    from footprints_amd.training.train import SEED, TrainStep, synthetic_batch

    torch.manual_seed(SEED)
    mm = ModelManager(use_cuda=True, learning_rate=1e-4)          # random-init weights (no checkpoints offline)
    if distributed:
        broadcast_state(mm.model)
    step = TrainStep(mm.model, mm.optimiser, distributed=distributed)
    batch = synthetic_batch(B, H, W, "cuda", seed=SEED + rank)     # per-rank shard, resident in HBM

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(batch)
    inst = winst = None
    if rank == 0 and not args.no_kernel_events:
        inst = Instrument(ops, FWD_DGRAD_OPS)
        inst.install()
        if args.dump_kernels:
            winst = Instrument(ops, WGRAD_OPS)
            winst.install()
    ev_steps = min(args.steps, 4)          # event brackets on the first steps of the timed region only (host cost of
    barrier()                              # ~300 event records per step would otherwise perturb `value`)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # one record per step: GPU-side step durations
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        if i == ev_steps:
            for x in (inst, winst):
                if x is not None:
                    x.remove()
        step(batch)
        marks[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    for x in (inst, winst):
        if x is not None:
            x.remove()
    timer = inst.timer if inst is not None else None
    wtimer = winst.timer if winst is not None else None
    # kernel-exclusive pass (outside the timed region): same steps, one stream, so launches do not overlap
    xtimer = None
    if timer is not None and step.eng.concurrent:
        step.eng.concurrent = False
        step(batch)
        xinst = Instrument(ops, FWD_DGRAD_OPS)
        xinst.install()
        torch.cuda.synchronize()
        for _ in range(min(3, args.steps)):
            step(batch)
        torch.cuda.synchronize()
        xinst.remove()
        xtimer = xinst.timer
        step.eng.concurrent = True
    t = torch.tensor([dt], device="cuda", dtype=torch.float64)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    final_loss = float(step.losses[20])

    # decoder backward alone: both decoders, from d loss / d outputs to d loss / d features plus all decoder
    # weight gradients, HIP events around repeated runs on one saved forward (all five streams, joined inside the bracket)
    dec_bwd = None
    if rank == 0:
        eng = step.eng
        eng.forward(batch["image"], training=True, save_for_backward=True, outputs=step.outputs)
        ops.loss_fwd_bwd(step.outputs, batch, step.losses, step.dpreds, step.depth_range, step.prior)
        for _ in range(2):
            eng.decoders_backward(step.dpreds)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            eng.decoders_backward(step.dpreds)
        e1.record()
        torch.cuda.synchronize()
        dec_ms = e0.elapsed_time(e1) / 10
        dec_bwd = {"ms": round(dec_ms, 3), "algorithmic_gb": DEC_BWD_GB, "algorithmic_gflop": DEC_BWD_GFLOP,
                   "achieved_hbm_gbs": round(DEC_BWD_GB / dec_ms * 1e3, 1), "frac_hbm_peak": round(DEC_BWD_GB / dec_ms * 1e3 / HBM_PEAK_GBS, 4),
                   "achieved_tflops": round(DEC_BWD_GFLOP / dec_ms, 2), "frac_f32_mfma_peak": round(DEC_BWD_GFLOP / dec_ms / MFMA_F32_PEAK_TFLOPS, 4),
                   "note": "fused-minimum bytes and reference-graph FLOPs of both decoders' backward (SURVEY.md section 8d), KITTI bs=12; "
                           "time = HIP events around 10 repetitions after 2 warm-ups"}

    # forward-only latency (configs[1]): eval-mode, no_grad, batch 12
    mm.model.eval()
    with torch.no_grad():
        for _ in range(2):
            mm.model(batch["image"])
        torch.cuda.synchronize()
        f0 = time.perf_counter()
        for _ in range(5):
            mm.model(batch["image"])
        torch.cuda.synchronize()
        fwd_ms_img = (time.perf_counter() - f0) / 5 / B * 1e3

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        out = {"metric": "training images/sec at 192x640 bs=12", "value": round(world * B * args.steps / dt, 2), "unit": "img/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "arithmetic": "fp32 tensors and accumulation; 3x3 stride-1 convs (fwd, dgrad, wgrad) multiply exactly split operands "
                             "(x = h + m + l in bf16, 6 of 9 bf16 MFMA products: error <= fp32 MFMA); remaining convs native fp32 MFMA",
               "config": {"workload": "KITTI 192x640 bs=12 full train step (fwd+loss+bwd+Adam), random-init weights, synthetic RGB + masks",
                          "per_gpu_batch": B, "global_batch": B * world, "height": H, "width": W,
                          "parallelism": "dp%d" % world if world > 1 else "single"},
               "fwd_ms_per_img": round(fwd_ms_img, 4), "final_loss": round(final_loss, 5),
               "step_ms": {"median": round(step_ms[len(step_ms) // 2], 3), "p10": round(step_ms[len(step_ms) // 10], 3),
                           "p90": round(step_ms[min(len(step_ms) - 1, (len(step_ms) * 9) // 10)], 3),
                           "note": "GPU-side durations between per-step HIP events inside the timed region (rank 0)"},
               "decoder_backward": dec_bwd}
        gf_fwd, gf_step = network_conv_gflop(B, H, W)
        # whole-step figure: the reference graph's conv FLOPs (fwd + dgrad + wgrad) over the measured step time, i.e. including
        # every non-conv kernel, launch gap and the FLOPs the nearest-x2 phase decomposition does not execute
        out["step_conv_tflops"] = {"algorithmic_gflop_per_step": round(gf_step, 1), "tflops": round(gf_step / ms_per_step, 2),
                                   "frac_of_f32_mfma_peak": round(gf_step / ms_per_step / MFMA_F32_PEAK_TFLOPS, 4),
                                   "fwd_algorithmic_gflop": round(gf_fwd, 1), "fwd_tflops": round(gf_fwd / (fwd_ms_img * B), 2)}
        if timer is not None:
            n, ms, fl = timer.summary()
            ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                               "traffic_note": "PMC FETCH_SIZE x2 / WRITE_SIZE per launch = 1.0-1.04x the algorithmic input / output bytes "
                                               "(profiles/round1_pmc_hbm_conv_bf3.txt; collected with rocprofv3 --pmc, not inside bench.py)",
                               "peak_bf16x6": BF16X6_PEAK_TFLOPS, "frac_bf16x6": round(ach / BF16X6_PEAK_TFLOPS, 4),
                               "kernel": "convolution forward + data-gradient launches: fp_conv3x3_bf3 (conv3x3_tile_bf3_kernel: fp32 operands split "
                                         "exactly into 3 bf16 terms, 6 v_mfma_f32_32x32x16_bf16 products, fp32 accumulate), fp_conv_up2_phase_fwd_bf3, "
                                         "fp_conv_igemm (igemm_kernel, v_mfma_f32_32x32x2_f32: stride 2, 1x1, 4x4/2 phase dgrad, stem)",
                               "launches_per_step": n // max(ev_steps, 1), "avg_launch_us": round(ms / max(n, 1) * 1e3, 2),
                               "algorithmic_gflop_per_launch": round(fl / max(n, 1) / 1e9, 3),
                               "kernel_ms_per_step": round(ms / max(ev_steps, 1), 3), "event_steps": ev_steps,
                               "concurrent_streams": bool(step.eng.concurrent)}
            if xtimer is not None:
                xn, xms, xfl = xtimer.summary()
                xach = xfl / (xms * 1e-3) / 1e12 if xms > 0 else 0.0
                out["roofline"].update({"achieved_exclusive": round(xach, 2), "frac_exclusive": round(xach / MFMA_F32_PEAK_TFLOPS, 4),
                                        "frac_exclusive_bf16x6": round(xach / BF16X6_PEAK_TFLOPS, 4),
                                        "avg_launch_us_exclusive": round(xms / max(xn, 1) * 1e3, 2)})
        if args.dump_kernels and timer is not None:
            with open(args.dump_kernels, "w") as fh:
                json.dump({"igemm": timer.table(), "wgrad": wtimer.table()}, fh, indent=1)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
