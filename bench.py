"""bench.py -- headline benchmark of BASELINE.json: training images/sec at 192x640 bs=12 on 1..8 MI355X.

    python bench.py --gpus N --steps K --warmup W                                (N > 1: spawns its N ranks itself, see below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                                   (N > 1, one rank per GPU, RCCL)

Started plainly with --gpus N > 1 (no WORLD_SIZE in the environment) it re-executes itself through torch.distributed.run with
N ranks on 127.0.0.1 and a free port, passes the ranks' output through and exits with their status: rank 0 still prints the ONE
JSON line.  The ranks rendezvous over a gloo group (host plumbing: unique-id exchange, barriers, the max over ranks of the timed
region); the gradient buckets travel over RCCL through this library's own fp_comm_* entry points (footprints_amd/parallel.py).  On a box with fewer GPUs than ranks the ranks share the GPUs and exchange over gloo (RCCL refuses two
ranks per device): a functional dry run of the launch path, flagged "shared_gpu" in the line, never a performance number.

A "step" = one full training step of the hot path on one per-GPU batch of 12 synthetic 192x640 images:
forward + fused loss + backward + (bucketed gradient all-reduce when N > 1) + fused Adam, all inputs resident
in HBM before the timed region (configs[2] of BASELINE.json; configs[1], the inference-only forward, is reported
beside it as fwd_ms_per_img).  Weak scaling: every rank keeps batch 12.  Rank 0 prints ONE JSON line.

`--workload matterport` runs configs[4] (512x640 bs=4) instead of the KITTI step; the metric name carries the workload.

Output (round 6): the LAST stdout line is a compact record (<= 4 KB: the contract keys, `roofline`, `cpu_baseline` and a handful of
scalars -- compact_record()); the full record described below is written to bench_detail.json (repo root, and gpurun_out/ when present).
--sustain and --other-format are opt-in; the default run is one timed region + the instrumented passes + the bounded CPU baseline.

Objects in the full record (bench_detail.json):
  roofline      the DOMINANT kernel of the step = the convolution entry point with the largest exclusive time per step, found by an
                extra pass outside the timed region with concurrency switched off (one stream; HIP events on that stream around
                every convolution launch of 3 steps).  achieved = MFMA FLOPs the kernel EXECUTES per launch / its average launch
                duration: for the exactly split bf16x3 kernels that is 6 bf16 products per multiply-add (x = h + m + l), priced
                against the dense bf16 MFMA peak (2.5 PFLOP/s); the fp32-MFMA kernels (igemm / stem / flattened wgrad) are priced
                against the fp32 matrix peak (157.3 TFLOP/s).  Phase-decomposed upsample convs count the 4/9 of the dense conv
                they execute.  `fp32_equiv_tflops` is the same launch in multiply-adds of the reference graph.  `traffic` = HBM
                bytes per launch from the committed PMC passes (profiles/round2_pmc_hbm_*.json: FETCH_SIZE x2 per the gfx950
                note of MI355X_MICROARCH.md, WRITE_SIZE as reported) next to the algorithmic bytes, or null when not collected.
                `groups` lists every convolution kernel family of the step the same way.
                `mfma_random_operand_peak` (round 6): what v_mfma_f32_32x32x16_bf16 sustains on this chip, measured for 0.7 s beside the run
                with pseudo-random operands and nothing else in the loop (fp_mfma_probe): ~1.85 PFLOP/s at ~1.8 GHz -- the nominal peak `frac` is
                quoted against is reached on constant data only (2.46 PFLOP/s at 2.39 GHz); `frac_of_it` = achieved / that.
  step_conv_tflops  the reference graph's conv FLOPs of one step (fwd + dgrad + wgrad) over the measured step time.
  decoder_backward  SURVEY.md section 8(d): both decoders' backward alone (d loss / d outputs -> d loss / d features + all decoder
                weight gradients) timed with HIP events, as achieved_hbm = algorithmic GB / t against 8 TB/s and achieved_mfma =
                GFLOP / t against the fp32 MFMA peak and the bf16x6 roof (the convolutions are MFMA-bound: F/B 72-755 vs a ridge of ~20).
  step_ms       median / p10 / p90 of the GPU-side step durations inside the timed region.
  cpu_baseline  the CPU oracle (a restatement of the reference's PyTorch CPU path, kind "port") timed on this box's host cores:
                1 warm-up + 3 timed full train steps of the same workload at the same batch size, plus the eval-mode forward in
                ms per image (rank 0, N = 1 only).
  dtype / operand formats (round 5)   `value`, `roofline` and every other top-level figure are measured in the DEFAULT operand format, the
                exact bf16x3 split (footprints_amd/_format.py: every fp32 operand bit is kept, six MFMA products; the arithmetic the
                reference's fp32 step is compared with).  `fp16_pair` is the same command re-run in a child process in the OPT-IN format
                (FP_OPERANDS=fp16_pair: 22 significant bits, three products) with the caller's --steps / --warmup / --sustain and the same
                instrumentation (value, ms_per_step, step_ms, roofline with its own kernel events and counter file, kernels, decoder_backward,
                sustained) -- a faster mode below fp32 that has to be asked for, never the headline.  Started with FP_OPERANDS=fp16_pair the
                roles swap and `dtype` says so.
  sustained     --sustain S (opt-in since round 6; default 0): the step looped for S seconds of wall clock after the timed region: img/s over the whole span,
                per-second img/s, step_ms percentiles, and the GPU's sclk / power / busy % sampled from sysfs every 100 ms (what a trainer
                that runs for hours sees; the timed region above is 0.2-0.6 s).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from footprints_amd._format import DTYPE_LABEL, FORMATS, format_env, operand_format      # noqa: E402  (no torch / HIP import)

MFMA_F32_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
BF16X6_PEAK_TFLOPS = 2500.0 / 6   # dense bf16 MFMA peak / 6 products per fp32-equivalent multiply-add (the split kernels' own roof)
HBM_PEAK_GBS = 8000.0             # same guide: HBM3E peak
BF16_PEAK_TFLOPS = 2500.0         # same guide: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16)
# decoder backward, both decoders (fused-minimum bytes; dgrad + wgrad FLOPs of the reference graph)
WORKLOADS = {"kitti": dict(B=12, H=192, W=640, dec_gb=7.018, dec_gflop=1023.9,
                           name="KITTI 192x640 bs=12 full train step (fwd+loss+bwd+Adam), random-init weights, synthetic RGB + masks"),
             "matterport": dict(B=4, H=512, W=640, dec_gb=6.247, dec_gflop=910.1,
                                name="Matterport 512x640 bs=4 full train step (fwd+loss+bwd+Adam), random-init weights, synthetic RGB + masks")}
B, H, W = 12, 192, 640            # set from --workload in main()


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


def step_algorithmic_bytes(n, h, w):
    """fused-minimum HBM bytes of ONE training step, fp32, by SURVEY.md section 8(d)'s per-convolution rule applied to every convolution of
    the network (VERDICT r5 "Next" 7): forward 4 (|X_unique| + |Y| + |W| + |b|), backward 4 (2 |Y| + 2 |X_unique| + 2 (|W| + |b|)) -- read dY, Y, X,
    W; write dX, dW -- with padding / upsampling / concatenation never materialised and BatchNorm / ReLU / residual / max-pool fused into
    their neighbours (the encoder's element-wise passes cost nothing extra at the fused minimum: that is what makes it a floor).  Heads count
    their low-resolution output.  Plus the loss (reads 4 scales x 4 channels of predictions and the target maps, writes the prediction
    gradients) and Adam (reads p, g, m, v; writes p, m, v).  Checks: decoders forward / backward and the loss reproduce SURVEY's 3.544 /
    7.018 / 0.224 GB at 12x192x640 (tests/test_bench_launch_cpu.py).  Returns GB per family and the total."""
    def conv(xu, y, cin, cout, k, bias):
        wb = cout * cin * k * k + (cout if bias else 0)
        return 4.0 * (xu + y + wb), 4.0 * (2 * y + 2 * xu + 2 * wb)
    ef = eb = 0.0
    f, b = conv(n * h * w * 3, n * (h // 2) * (w // 2) * 64, 3, 64, 7, False)            # stem (its input gradient is not needed: counted anyway, 3 channels)
    ef, eb = ef + f, eb + b
    res, cin = (h // 4, w // 4), 64
    for cout, blocks, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)):
        for bi in range(blocks):
            s_ = stride if bi == 0 else 1
            rin = res
            res = (res[0] // s_, res[1] // s_)
            px_in, px = n * rin[0] * rin[1], n * res[0] * res[1]
            for (xu, ci, k) in ((px_in * cin, cin, 3), (px * cout, cout, 3)):
                f, b = conv(xu, px * cout, ci, cout, k, False)
                ef, eb = ef + f, eb + b
            if s_ != 1 or cin != cout:
                f, b = conv(px_in * cin, px * cout, cin, cout, 1, False)
                ef, eb = ef + f, eb + b
            cin = cout
    df = db = 0.0
    r = (h // 32, w // 32)
    for ci, co in ((512, 256), (256, 128), (128, 64), (64, 64)):
        px = n * r[0] * r[1]
        for (xu, c_in) in ((px * ci, ci), (px * co, co)):                                   # pre_concat_conv
            f, b = conv(xu, px * co, c_in, co, 3, True)
            df, db = df + f, db + b
        r = (r[0] * 2, r[1] * 2)
        px2 = n * r[0] * r[1]
        f, b = conv(px * co + px2 * co, px2 * co, 2 * co, co, 3, True)                      # post_concat_conv 1 on cat[up2(low), skip]: low read at ITS resolution
        df, db = df + f, db + b
        f, b = conv(px2 * co, px2 * co, co, co, 3, True)
        df, db = df + f, db + b
    px_lo, px_hi = n * r[0] * r[1], n * h * w
    f, b = conv(px_lo * 64, px_hi * 32, 64, 32, 3, True)                                    # outconv4 ConvBlock: up2(64) -> 32 -> 32 at full resolution
    df, db = df + f, db + b
    f, b = conv(px_hi * 32, px_hi * 32, 32, 32, 3, True)
    df, db = df + f, db + b
    for s_, c in ((8, 128), (4, 64), (2, 64), (1, 32)):                                     # 2-channel heads at their own resolution
        px = n * (h // s_) * (w // s_)
        f, b = conv(px * c, px * 2, c, 2, 3, True)
        df, db = df + f, db + b
    df, db = 2 * df, 2 * db                                                                 # two decoders
    loss = 4.0 * n * h * w * (4 * 4 + 4 * 4 + 7) / 1.0                                      # read 16 prediction planes, write 16 gradient planes, read 7 target maps
    params = 31012944
    adam = 4.0 * params * 7
    gb = lambda v: round(v / 1e9, 3)
    out = {"encoder_fwd": gb(ef), "encoder_bwd": gb(eb), "decoders_fwd": gb(df), "decoders_bwd": gb(db), "loss": gb(loss), "adam": gb(adam)}
    out["total"] = gb(ef + eb + df + db + loss + adam)
    return out


class KernelTimer:
    """HIP-event bracket around every launch of a set of convolution entry points (events go on torch's current stream, which
    is the stream the C ABI launches on; used with concurrency switched off, so a bracket is that launch's exclusive time)."""

    def __init__(self):
        self.records = []

    def wrap(self, fn, spec, opname):
        def wrapped(first, *a, **k):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = fn(first, *a, **k)
            e.record()
            g = spec["group"](first, *a, **k) if callable(spec["group"]) else spec["group"]
            self.records.append((s, e, spec["dense"](first, *a, **k), spec["exec"](first, *a, **k), spec["bytes"](first, *a, **k), g,
                                 spec["key"](first, *a, **k) if spec.get("key") else opname + " " + desc_key(first)))
            return r
        return wrapped

    def groups(self, steps):
        """per kernel family: launches / step, exclusive ms / step, executed and reference-graph GFLOP per launch"""
        agg = {}
        for s, e, dense, ex, by, g, _ in self.records:
            a = agg.setdefault(g, [0, 0.0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += s.elapsed_time(e)
            a[2] += dense
            a[3] += ex
            a[4] += by
        rows = []
        for g, (n, ms, dense, ex, by) in agg.items():
            info = GROUPS[g]
            products = float(info.get("products", 6 if info["bf16x3"] else 1))
            peak = BF16_PEAK_TFLOPS if products > 1 else MFMA_F32_PEAK_TFLOPS
            ach = products * ex / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            label = {6.0: "bf16 x6 products (exact 3-way split)", 4.0: "fp16 x4 products (scaled fp16 pairs, 22 significant bits)",
                     3.0: "fp16 x3 products (scaled fp16 pairs, 22 significant bits: hh + hm + mh)", 1.0: "fp32"}[products]
            rows.append({"kernel": info["kernel"], "entry_point": g, "mfma": label,
                         "launches_per_step": round(n / steps, 1), "exclusive_ms_per_step": round(ms / steps, 3),
                         "avg_launch_us": round(ms / n * 1e3, 2), "executed_mfma_gflop_per_launch": round(products * ex / n / 1e9, 3),
                         "achieved": round(ach, 1), "peak": peak, "frac": round(ach / peak, 4),
                         "fp32_equiv_tflops": round(dense / (ms * 1e-3) / 1e12, 1) if ms > 0 else 0.0,
                         "algorithmic_mb_per_launch": round(by / n / 1e6, 2)})
        return sorted(rows, key=lambda r: -r["exclusive_ms_per_step"])

    def table(self):
        """Per distinct launch shape: count, mean microseconds, reference-graph TFLOP/s."""
        agg = {}
        for s, e, dense, ex, by, g, key in self.records:
            a = agg.setdefault(key, [0, 0.0, dense, g])
            a[0] += 1
            a[1] += s.elapsed_time(e)
        rows = [{"shape": k, "group": g, "launches": n, "avg_us": round(ms / n * 1e3, 1), "gflop": round(f / 1e9, 3),
                 "tflops": round(f * n / (ms * 1e-3) / 1e12, 1) if ms > 0 else 0.0, "total_ms": round(ms, 3)} for k, (n, ms, f, g) in agg.items()]
        return sorted(rows, key=lambda r: -r["total_ms"])


GATHER_NAMES = ("fwd_zero", "fwd_reflect", "fwd_up2", "dgrad_zero", "dgrad_reflect", "stem")


def desc_key(d):
    return "%s N%d %dx%d<-%dx%d C%d+%d->%d k%d s%d" % (GATHER_NAMES[d.gather], d.N, d.OH, d.OW, d.IH, d.IW, d.C0, d.C1, d.Nout,
                                                      d.KH, d.stride)


def conv_flops(d, *a, **k):
    taps = 1 if d.gather == 5 else d.KH * d.KW
    kk = 147 if d.gather == 5 else d.C0 + d.C1
    return 2.0 * d.N * d.OH * d.OW * d.Nout * taps * kk


def conv_bytes(d, *a, **k):
    """fused-minimum bytes of one forward / data-gradient / weight-gradient launch: every distinct fp32 element once"""
    taps = 49 if d.gather == 5 else d.KH * d.KW
    cin = 3 if d.gather == 5 else d.C0 + d.C1
    x = d.N * d.IH * d.IW * cin
    if d.gather == 2:                                            # nearest-x2 of C0 low-res channels + C1 skip channels
        x = d.N * (d.IH // 2) * (d.IW // 2) * d.C0 + d.N * d.IH * d.IW * d.C1
    return 4.0 * (x + d.N * d.OH * d.OW * d.Nout + taps * cin * d.Nout)


# nearest-x2 phase decomposition: a 3x3 conv over the upsampled tensor = four 2x2 convs over the low-res tensor (4/9 of the MACs)
def phase_fwd_dense(low, wphase, bias, y, *a, **k):
    N, h, w, C0 = low.shape
    return 2.0 * N * (2 * h) * (2 * w) * y.shape[3] * 9 * C0


def phase_fwd_bytes(low, wphase, bias, y, *a, **k):
    return 4.0 * (low.numel() + y.numel() * (2 if k.get("addend") is not None else 1) + 9 * low.shape[3] * y.shape[3])


def phase_dgrad_dense(dz, wpacked, ext, *a, **k):
    N, H2, W2, Cout = dz.shape
    return 2.0 * N * H2 * W2 * Cout * 9 * ext.shape[3]


def phase_dgrad_exec(dz, wpacked, ext, *a, **k):
    """4 phases x 2x2 taps at every position of the (h+2) x (w+2) extended low-res grid"""
    N, he, we, C0 = ext.shape
    return 2.0 * N * he * we * 16 * dz.shape[3] * C0


def phase_dgrad_bytes(dz, wpacked, ext, *a, **k):
    return 4.0 * (dz.numel() + ext.numel() + 9 * dz.shape[3] * ext.shape[3])


def phase_wgrad_dense(low, dz, *a, **k):
    N, h, w, C0 = low.shape
    return 2.0 * N * (2 * h) * (2 * w) * dz.shape[3] * 9 * C0


def phase_wgrad_bytes(low, dz, dw, *a, **k):
    return 4.0 * (low.numel() + dz.numel() + 9 * low.shape[3] * dz.shape[3])


def _frac(f, frac):
    return lambda *a, **k: f(*a, **k) * frac


def phase_key(tag):
    return lambda low, *a, **k: "%s N%d low %dx%d C%d" % (tag, low.shape[0], low.shape[1], low.shape[2], low.shape[3])


# entry point (footprints_amd.ops name) -> how its launches are counted.  `exec` = multiply-adds x2 the kernel really executes
# (the phase kernels run 4/9 of the dense conv), `dense` = the reference graph's conv, `bytes` = fused-minimum HBM bytes.
HP_PRODUCTS = 3      # csrc/fp_common.h: FP_HP_PRODUCTS of the default build
FMT = operand_format()              # "exact" (default) | "fp16_pair" (opt-in): footprints_amd/_format.py
HP_ON = FMT == "fp16_pair" and os.environ.get("FP_NO_BF3", "0") == "0"      # mirrors footprints_amd.engine._HP
PROFILE_ROUND = 6                   # profiles/round<N>_*: the counter files bench.py may attach
GROUPS = {
    "conv3x3_bf3": dict(kernel="conv3x3_tile_bf3_kernel (+ splitk_reduce_kernel on small grids)", bf16x3=True),
    "conv3x3_hp": dict(kernel="conv3x3_tile_bf3_kernel<..., HP> (fp16-pair operands; + splitk_reduce_kernel / amax_kernel on small grids)", bf16x3=False,
                       products=HP_PRODUCTS),
    "conv_up2_phase_fwd_hp": dict(kernel="up2_phase_fwd_bf3_kernel<..., 2> (fp16-pair operands)", bf16x3=False, products=HP_PRODUCTS),
    "conv_up2_phase_dgrad_hp": dict(kernel="up2_phase_dgrad_bf3_kernel<..., 2> (fp16-pair operands)", bf16x3=False, products=HP_PRODUCTS),
    "conv_wgrad_hp": dict(kernel="wgrad3x3_hp_pf_kernel<MODE, 2> (fp16-pair operands, two-deep register prefetch ring; + wgrad_reduce_bias[_t]_kernel)", bf16x3=False,
                          products=HP_PRODUCTS),
    "conv_up2_phase_wgrad_hp": dict(kernel="wgrad_up2_phase_bf3_kernel<2> (fp16-pair operands; + its sum / un-collapse / bias reduce launches)",
                                    bf16x3=False, products=HP_PRODUCTS),
    "conv_igemm": dict(kernel="igemm_kernel / stem_tile_kernel (7x7 stem, 4x4/2 phase dgrad of small levels; stride 2 / 1x1 with FP_HP_IGEMM=0)", bf16x3=False),
    "conv_igemm_bf3": dict(kernel="igemm_hp_kernel<TN, 3> (exactly split bf16x3 operands: 3x3 stride 2, 1x1, their data gradients; + splitk_reduce_kernel on small grids)",
                           bf16x3=True),
    "conv_igemm_hp": dict(kernel="igemm_hp_kernel (fp16-pair operands: 3x3 stride 2, 1x1, their data gradients; + splitk_reduce_kernel on small grids)",
                          bf16x3=False, products=HP_PRODUCTS),
    "conv_stem_hp": dict(kernel="stem_tile_hp_kernel / stem_wgrad_hp_kernel (the 7x7 / 2 stem and its weight gradient with fp16-pair operands; + wgrad_reduce_wide_kernel)",
                         bf16x3=False, products=HP_PRODUCTS),
    "conv_up2_phase_fwd_bf3": dict(kernel="up2_phase_fwd_bf3_kernel", bf16x3=True),
    "conv_up2_phase_fwd": dict(kernel="up2_phase_fwd_kernel", bf16x3=False),
    "conv_up2_phase_dgrad_bf3": dict(kernel="up2_phase_dgrad_bf3_kernel", bf16x3=True),
    "conv_wgrad_bf3": dict(kernel="wgrad3x3_hp_pf_kernel<MODE, 2, 3> (exact bf16x3 operands on the prefetch ring; nearest-x2 gather form: wgrad3x3_bf3_v3_kernel; "
                                  "+ wgrad_reduce_bias[_t]_kernel of the same entry point)", bf16x3=True),
    "conv_up2_phase_wgrad_bf3": dict(kernel="wgrad_up2_phase_bf3_kernel (+ its sum / un-collapse / bias reduce launches)", bf16x3=True),
    "conv_up2_phase_wgrad": dict(kernel="wgrad_up2_phase_kernel", bf16x3=False),
    "conv_wgrad": dict(kernel="wgrad_kernel / wgrad3x3_tile_kernel (fp32 MFMA: stem, stride 2, 1x1, shapes the bf16x3 kernel rejects)", bf16x3=False),
}
# main kernel symbol(s) of each entry point (the name before the template arguments, as fp_ktime_row / rocprofv3 print it)
GROUP_SYMBOL = {
    "conv3x3_hp": ("conv3x3_tile_bf3_kernel",), "conv3x3_bf3": ("conv3x3_tile_bf3_kernel",),
    "conv_wgrad_hp": ("wgrad3x3_hp_pf_kernel", "wgrad3x3_bf3_v3_kernel"), "conv_wgrad_bf3": ("wgrad3x3_hp_pf_kernel", "wgrad3x3_bf3_v3_kernel"),
    "conv_up2_phase_fwd_hp": ("up2_phase_fwd_bf3_kernel",), "conv_up2_phase_fwd_bf3": ("up2_phase_fwd_bf3_kernel",),
    "conv_up2_phase_dgrad_hp": ("up2_phase_dgrad_bf3_kernel",), "conv_up2_phase_dgrad_bf3": ("up2_phase_dgrad_bf3_kernel",),
    "conv_up2_phase_wgrad_hp": ("wgrad_up2_phase_bf3_kernel",), "conv_up2_phase_wgrad_bf3": ("wgrad_up2_phase_bf3_kernel",),
    "conv_stem_hp": ("stem_tile_hp_kernel", "stem_wgrad_hp_kernel"),
    "conv_igemm": ("igemm_kernel", "stem_tile_kernel"), "conv_igemm_hp": ("igemm_hp_kernel",), "conv_igemm_bf3": ("igemm_hp_kernel",), "conv_wgrad": ("wgrad_kernel", "wgrad3x3_tile_kernel", "stem_wgrad_tile_kernel"),
}
CONV_OPS = {
    "conv_igemm": dict(group="conv_igemm", dense=conv_flops, exec=conv_flops, bytes=conv_bytes),
    "conv_igemm_hp": dict(group="conv_igemm_hp", dense=conv_flops, exec=conv_flops, bytes=conv_bytes),
    "conv_igemm_bf3": dict(group="conv_igemm_bf3", dense=conv_flops, exec=conv_flops, bytes=conv_bytes),
    "conv_stem_hp": dict(group="conv_stem_hp", dense=conv_flops, exec=conv_flops, bytes=conv_bytes),
    "conv_stem_wgrad_hp": dict(group="conv_stem_hp", dense=conv_flops, exec=conv_flops, bytes=conv_bytes),
    "conv3x3_bf3": dict(group="conv3x3_bf3", dense=conv_flops, exec=conv_flops, bytes=conv_bytes),
    "conv3x3_hp": dict(group="conv3x3_hp", dense=conv_flops, exec=conv_flops, bytes=conv_bytes),
    "conv_up2_phase_fwd": dict(group="conv_up2_phase_fwd", dense=phase_fwd_dense, exec=_frac(phase_fwd_dense, 4.0 / 9.0), bytes=phase_fwd_bytes,
                               key=phase_key("phase_fwd")),
    "conv_up2_phase_fwd_bf3": dict(group="conv_up2_phase_fwd_bf3", dense=phase_fwd_dense, exec=_frac(phase_fwd_dense, 4.0 / 9.0),
                                   bytes=phase_fwd_bytes, key=phase_key("phase_fwd_bf3")),
    "conv_up2_phase_dgrad_bf3": dict(group="conv_up2_phase_dgrad_bf3", dense=phase_dgrad_dense, exec=phase_dgrad_exec, bytes=phase_dgrad_bytes,
                                     key=lambda dz, *a, **k: "phase_dgrad_bf3 N%d dz %dx%d C%d" % (dz.shape[0], dz.shape[1], dz.shape[2], dz.shape[3])),
    "conv_wgrad": dict(group="conv_wgrad", dense=conv_flops, exec=conv_flops, bytes=conv_bytes),
    "conv_wgrad_slice": dict(group="conv_wgrad", dense=conv_flops, exec=conv_flops, bytes=conv_bytes),
    "conv_wgrad_bf3": dict(group=lambda d, *a, **k: "conv_wgrad_hp" if k.get("amax") is not None else "conv_wgrad_bf3", dense=conv_flops,
                           exec=conv_flops, bytes=conv_bytes),
    "conv_up2_phase_fwd_hp": dict(group="conv_up2_phase_fwd_hp", dense=phase_fwd_dense, exec=_frac(phase_fwd_dense, 4.0 / 9.0),
                                  bytes=phase_fwd_bytes, key=phase_key("phase_fwd_hp")),
    "conv_up2_phase_dgrad_hp": dict(group="conv_up2_phase_dgrad_hp", dense=phase_dgrad_dense, exec=phase_dgrad_exec, bytes=phase_dgrad_bytes,
                                    key=lambda dz, *a, **k: "phase_dgrad_hp N%d dz %dx%d C%d" % (dz.shape[0], dz.shape[1], dz.shape[2], dz.shape[3])),
    "conv_up2_phase_wgrad_hp": dict(group="conv_up2_phase_wgrad_hp", dense=phase_wgrad_dense, exec=_frac(phase_wgrad_dense, 4.0 / 9.0),
                                    bytes=phase_wgrad_bytes, key=phase_key("phase_wgrad_hp")),
    "conv_up2_phase_wgrad": dict(group=lambda low, dz, dw, *a, **k: "conv_up2_phase_wgrad_bf3" if k.get("bf3") else "conv_up2_phase_wgrad",
                                 dense=phase_wgrad_dense, exec=_frac(phase_wgrad_dense, 4.0 / 9.0), bytes=phase_wgrad_bytes, key=phase_key("phase_wgrad")),
}


class Instrument:
    """swap the convolution entry points of footprints_amd.ops for event-timed wrappers, and back"""

    def __init__(self, ops, table):
        self.ops, self.table = ops, table
        self.orig = {n: getattr(ops, n) for n in table}
        self.timer = KernelTimer()

    def install(self):
        for n, spec in self.table.items():
            setattr(self.ops, n, self.timer.wrap(self.orig[n], spec, n))

    def remove(self):
        for n, f in self.orig.items():
            setattr(self.ops, n, f)


def kernel_source_digest():
    """sha256 over the sources of the dominant kernel (csrc/conv3x3_tile_bf3.hip + fp_common.h): ties a committed counter file to the build
    it was measured on without needing .git (the GPU box receives a snapshot without it)"""
    import hashlib
    h = hashlib.sha256()
    for name in ("conv3x3_tile_bf3.hip", "fp_common.h"):
        with open(os.path.join(ROOT, "footprints_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_traffic(workload, entry_point):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (scripts/profile_session_r6.sh -> profiles/round6_pmc_hbm_*.json;
    the counters need rocprofv3 around the process, so they cannot be collected inside this one), or None.  The file records the digest of
    the kernel source it was measured on: a file from another build is NOT attached (its figure would describe a different kernel)."""
    name = "round%d_pmc_hbm_%s_%s.json" % (PROFILE_ROUND, workload, FMT)
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            doc = json.load(fh)
    except (OSError, ValueError):
        return None, "no counter file for this round and operand format (profiles/%s)" % name
    t = doc.get(entry_point)
    if not t:
        return None, "counter file has no entry for " + entry_point
    have, want = doc.get("kernel_source_sha16"), kernel_source_digest()
    if have != want:
        return None, "counter file was measured on another build of the kernel (source digest %s, this tree %s): not attached" % (have, want)
    return t, None


def load_step_counters(workload):
    """step-level MFMA busy % and HBM bytes from the committed counters-only passes of THIS build and operand format
    (scripts/pmc_step.py -> profiles/round6_pmc_step_<workload>_<format>.json; digest-checked like load_traffic), or (None, why)"""
    name = "round%d_pmc_step_%s_%s.json" % (PROFILE_ROUND, workload, FMT)
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            doc = json.load(fh)
    except (OSError, ValueError):
        return None, "no step-level counter file for this round and operand format (profiles/%s)" % name
    have, want = doc.get("kernel_source_sha16"), kernel_source_digest()
    if have != want:
        return None, "step-level counter file was measured on another build of the kernel (source digest %s, this tree %s): not attached" % (have, want)
    return doc, None


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


def cpu_baseline():
    """BASELINE.md section 4 / SURVEY.md section 8d, bounded to ~20 s of CPU work: 1 warm-up + 2 timed full train steps (fwd + loss + bwd +
    Adam, the span of training/train.py:149-159) of the CPU oracle on the SAME workload and batch size as the GPU line, then the eval-mode
    forward in ms per image; plus the shipped trainer's own setting (one thread, training/train.py:12-14) on a batch-2 step."""
    from oracle import restatement as R
    cores, eff = _pick_threads()
    torch.set_num_threads(cores)
    P, Bf = R.make_state(tag="bench")
    tr = R.OracleTrainer(P, Bf)
    batch = R.make_batch(B, H, W, tag="bench.cpu")
    tr.step(batch)                                   # warm-up at the timed shape (oneDNN primitive caches, allocator)
    times = []
    for _ in range(2):
        t0 = time.time()
        tr.step(batch)
        times.append(time.time() - t0)
    dt = sum(times) / len(times)
    with torch.no_grad():
        R.footprint_network(batch["image"], tr.P, tr.B, training=False)
        f0 = time.time()
        R.footprint_network(batch["image"], tr.P, tr.B, training=False)
        fwd_ms_img = (time.time() - f0) / B * 1e3
    torch.set_num_threads(1)                          # the shipped trainer's own setting (training/train.py:12-14), bounded sample: batch 2
    b1 = min(2, B)
    small = R.make_batch(b1, H, W, tag="bench.cpu1")
    tr.step(small)
    t1 = time.time()
    tr.step(small)
    dt1 = time.time() - t1
    torch.set_num_threads(cores)
    return {"value": round(B / dt, 4), "unit": "img/s", "cores": cores, "kind": "port",
            "eval_fwd_ms_per_img": round(fwd_ms_img, 2),
            "single_thread": {"value": round(b1 / dt1, 4), "unit": "img/s", "cores": 1, "batch": b1, "s_per_step": round(dt1, 2),
                              "sample": "1 timed full train step at batch %d after 1 warm-up step with torch.set_num_threads(1), the reference "
                                        "trainer's own setting (BASELINE.md section 4)" % b1},
            "sample": "mean of 2 timed train steps (fwd+loss+bwd+Adam) of the CPU oracle at %dx%d bs=%d after 1 warm-up; %d threads = fastest of "
                      "{8,16,32,64,%d}" % (H, W, B, cores, eff),
            "s_per_step": round(dt, 3), "s_per_step_each": [round(t, 3) for t in times], "host_cores": eff}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`python bench.py --gpus N` started plainly: run the N ranks through torch.distributed.run on this node and pass their output
    through (rank 0 prints the ONE JSON line); the exit status is theirs."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def other_format_leg(fmt, args):
    """the same command in a child process in the OTHER operand format (fixed when footprints_amd.engine is imported): caller's workload,
    --steps, --warmup and --sustain, same instrumentation (kernel events, roofline, decoder backward, sustained) -> its JSON line, trimmed"""
    env = dict(os.environ)
    env.update(format_env(fmt))
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--sustain", str(args.sustain), "--no-cpu-baseline", "--no-loader"]
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if not line:
            return {"error": (p.stderr or "")[-400:]}
        doc = json.loads(line[-1])
        if doc.get("detail"):                                      # the child's full record (its stdout line is the compact one)
            with open(os.path.join(ROOT, doc["detail"])) as fh:
                doc = json.load(fh)
    except Exception as e:      # the headline must not die with a side leg
        return {"error": repr(e)}
    keep = ("value", "unit", "steps", "warmup", "ms_per_step", "dtype", "operand_format", "arithmetic", "fwd_ms_per_img", "final_loss", "step_ms",
            "sustained", "decoder_backward", "step_conv_tflops", "roofline")
    leg = {k: doc[k] for k in keep if k in doc}
    if "kernels" in doc:
        leg["kernels"] = {"serial": doc["kernels"]["serial"][:12], "concurrent": doc["kernels"]["concurrent"][:12]}
    return leg


class GpuSampler:
    """sclk (MHz), socket power (W) and busy % of one GPU from sysfs every `period` seconds on a host thread (no subprocess per sample:
    rocm-smi takes ~0.3 s a call).  Files that do not exist on a box are skipped; `summary()` says which were read."""

    def __init__(self, index=0, period=0.1):
        import glob
        import threading
        self.period, self.rows, self._stop = period, [], threading.Event()
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/gpu_busy_percent"))
        self.dev = os.path.dirname(cards[index]) if index < len(cards) else None
        hw = sorted(glob.glob(os.path.join(self.dev, "hwmon", "hwmon*"))) if self.dev else []
        self.hw = hw[0] if hw else None
        self._thr = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _read(path):
        try:
            with open(path) as fh:
                return fh.read()
        except OSError:
            return None

    def _sample(self):
        sclk = power = busy = None
        if self.hw:
            v = self._read(os.path.join(self.hw, "freq1_input"))
            sclk = float(v) / 1e6 if v and v.strip().isdigit() else None
            for f in ("power1_average", "power1_input"):
                v = self._read(os.path.join(self.hw, f))
                if v and v.strip().isdigit():
                    power = float(v) / 1e6
                    break
        if self.dev:
            if sclk is None:
                v = self._read(os.path.join(self.dev, "pp_dpm_sclk")) or ""
                for ln in v.splitlines():
                    if ln.strip().endswith("*"):
                        try:
                            sclk = float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
                        except (IndexError, ValueError):
                            pass
            v = self._read(os.path.join(self.dev, "gpu_busy_percent"))
            busy = float(v) if v and v.strip().isdigit() else None
        return (time.perf_counter(), sclk, power, busy)

    def _run(self):
        while not self._stop.is_set():
            self.rows.append(self._sample())
            self._stop.wait(self.period)

    def start(self):
        self._thr.start()
        return self

    def stop(self):
        self._stop.set()
        self._thr.join(timeout=2.0)

    def summary(self):
        def stat(i, nd=1):
            v = sorted(r[i] for r in self.rows if r[i] is not None)
            if not v:
                return None
            return {"mean": round(sum(v) / len(v), nd), "p10": round(v[len(v) // 10], nd), "median": round(v[len(v) // 2], nd),
                    "p90": round(v[min(len(v) - 1, len(v) * 9 // 10)], nd), "samples": len(v)}
        return {"sclk_mhz": stat(1), "power_w": stat(2), "busy_percent": stat(3), "period_s": self.period,
                "source": "sysfs: %s (hwmon freq1_input / power1_average, gpu_busy_percent)" % (self.dev or "no amdgpu device node found")}


def sustained_leg(step, batch, seconds, world, local_rank, n_steps=None):
    """loop the training step for `seconds` of wall clock (what a trainer sees: clocks and power settle over seconds, the timed region lasts
    a fraction of one); per-step HIP events, per-second img/s, sysfs samples of the GPU this rank runs on.  n_steps: run exactly that many
    steps instead (data-parallel runs: every step contains collectives, so all ranks must agree on the count -- a rank that stopped on its
    own clock would leave the others waiting in an all-reduce)"""
    from footprints_amd import _lib
    lib = _lib.load()
    sampler = GpuSampler(index=local_rank).start()
    # the chip's own counters (fp_clock_probe: shader cycles + constant-rate ticks per XCD) every 32 steps: d cycles / d ticks x wall-clock
    # rate = the shader clock really sustained in between, independent of what the SMU's sysfs nodes report
    probes = torch.zeros((4096, 16), dtype=torch.int64, device="cuda")
    n_probe = 0

    def probe():
        nonlocal n_probe
        if n_probe < probes.shape[0]:
            _lib.check(lib.fp_clock_probe(probes[n_probe].data_ptr(), torch.cuda.current_stream().cuda_stream), "fp_clock_probe")
            n_probe += 1
    marks = [torch.cuda.Event(enable_timing=True)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    probe()
    host_t = []
    while True:
        step(batch)
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append(e)
        host_t.append(time.perf_counter() - t0)
        if (len(marks) - 1) % 32 == 0:
            probe()
        if (n_steps is not None and len(marks) - 1 >= n_steps) or (n_steps is None and host_t[-1] >= seconds) or len(marks) > 200000:
            break
        if len(marks) % 64 == 0:
            marks[-32].synchronize()          # keep the host at most ~32 steps ahead: the loop ends on GPU time, not on queue depth
    probe()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sampler.stop()
    khz = int(lib.fp_wall_clock_khz())
    pr = probes[:n_probe].cpu().view(n_probe, 8, 2).double()
    clocks = []
    if khz > 0 and n_probe >= 2:
        for i in range(n_probe - 1):
            per_xcd = [(pr[i + 1, x, 0] - pr[i, x, 0]) / (pr[i + 1, x, 1] - pr[i, x, 1]) * khz / 1e3
                       for x in range(8) if pr[i, x, 1] > 0 and pr[i + 1, x, 1] > pr[i, x, 1]]
            if per_xcd:
                clocks.append(float(sum(per_xcd) / len(per_xcd)))
    cs = sorted(clocks)
    shader_clock = ({"mean": round(sum(cs) / len(cs), 1), "p10": round(cs[len(cs) // 10], 1), "median": round(cs[len(cs) // 2], 1),
                     "p90": round(cs[min(len(cs) - 1, len(cs) * 9 // 10)], 1), "intervals": len(cs), "wall_clock_khz": khz,
                     "how": "fp_clock_probe every 32 steps: d s_memtime / d s_memrealtime x wall-clock rate, mean over the XCDs, per interval"}
                    if cs else None)
    n = len(marks) - 1
    ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(n)]
    ends, acc = [], 0.0
    for v in ms:
        acc += v
        ends.append(acc / 1e3)
    per_sec = []
    for sec in range(int(ends[-1])):
        k = sum(1 for e in ends if sec <= e < sec + 1)
        per_sec.append(round(k * B * world, 1))
    srt = sorted(ms)
    return {"seconds": round(dt, 2), "steps": n, "img_per_s": round(world * B * n / dt, 2), "ms_per_step": round(dt / n * 1e3, 3),
            "step_ms": {"p10": round(srt[n // 10], 3), "median": round(srt[n // 2], 3), "p90": round(srt[min(n - 1, n * 9 // 10)], 3), "max": round(srt[-1], 3)},
            "img_per_s_each_second": per_sec, "shader_clock_mhz": shader_clock, "gpu": sampler.summary(),
            "note": "the same step looped for --sustain seconds right after the timed region (rank 0's GPU sampled every 100 ms)"}


def kernel_table(lib, steps):
    """rows of the fp_ktime_* collection: per kernel symbol launches / step, total ms / step, average microseconds"""
    import ctypes
    n = lib.fp_ktime_end()
    rows = []
    name = ctypes.create_string_buffer(2048)
    cnt, ms = ctypes.c_int64(), ctypes.c_double()
    for i in range(max(n, 0)):
        if lib.fp_ktime_row(i, name, 2048, ctypes.byref(cnt), ctypes.byref(ms)) == 0:
            pretty = name.value.decode().replace("(anonymous namespace)::", "")
            pretty = pretty[5:] if pretty.startswith("void ") else pretty
            rows.append({"kernel": pretty, "launches_per_step": round(cnt.value / steps, 2), "ms_per_step": round(ms.value / steps, 4),
                         "avg_us": round(ms.value / max(cnt.value, 1) * 1e3, 2)})
    return sorted(rows, key=lambda r: -r["ms_per_step"])


def mfma_random_operand_peak(lib, seconds=0.7, iters=4000):
    """what v_mfma_f32_32x32x16_bf16 sustains on THIS chip, right now, with operands that toggle like real activations (fp_mfma_probe mode 1:
    nothing but MFMAs, three waves per SIMD): launches of ~5 ms back to back for `seconds`, the rate and the shader clock of the last third.
    Round 6 (profiles/round6_mfma_sustained_clock.txt): 2.46 PFLOP/s at 2.39 GHz on constant data, 1.85 PFLOP/s at 1.81 GHz on random data --
    the clock the chip holds falls with the toggling of the operand buses, so the nominal dense peak `roofline.peak` is priced against is a
    constant-data figure; this is the same instruction's ceiling on data, measured beside the run."""
    out = torch.empty(768 * 256, device="cuda")
    clk = torch.zeros(2, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    khz = int(lib.fp_wall_clock_khz())
    flop = float(lib.fp_mfma_probe_flop(iters))
    rows = []
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.fp_mfma_probe(out.data_ptr(), clk.data_ptr(), iters, 1, st)
        e1.record()
        e1.synchronize()
        c = clk.cpu()
        rows.append((flop / e0.elapsed_time(e1) / 1e9, float(c[0]) / max(float(c[1]), 1.0) * khz / 1e3 if khz > 0 else None))
    tail = rows[-max(1, len(rows) // 3):]
    tf = sum(r[0] for r in tail) / len(tail)
    mhz = [r[1] for r in tail if r[1]]
    return {"tflops": round(tf, 1), "shader_mhz": round(sum(mhz) / len(mhz), 0) if mhz else None, "launches": len(rows),
            "how": "fp_mfma_probe mode 1 (pseudo-random bf16 operands, MFMAs only) back to back for %.1f s; mean of the last third" % seconds}


COMPACT_LIMIT = 4096              # bytes: the driver parses the LAST stdout line; round 5's 34.6 KB line came back unparsed


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_record(out, detail_path=None):
    """the ONE stdout line: the contract keys of the bench record and a handful of scalars (<= COMPACT_LIMIT bytes); everything else
    `main` collects (kernel tables, groups, notes, side legs) goes to the sidecar file named in `detail`.  Pure function of `out`
    (tests/test_bench_launch_cpu.py builds it from a synthetic result without a GPU)."""
    rec = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling"))
    rec["vs_baseline"] = out.get("vs_baseline")
    fmt = out.get("operand_format", "exact")
    rec["dtype"] = "f32" if fmt == "exact" else "f32 tensors; %s operands (22 significant bits, below fp32: opt-in)" % fmt
    rec.update(_pick(out, ("data",)))
    cfg = out.get("config", {})
    rec["config"] = _pick(cfg, ("workload", "per_gpu_batch", "global_batch", "height", "width", "parallelism"))
    if "operand_format" in out:
        rec["config"]["operand_format"] = out["operand_format"]
    ge = cfg.get("gradient_exchange")
    if ge:
        rec["config"]["gradient_exchange"] = _pick(ge, ("transport", "buckets", "overlap_with_backward", "in_launch_plan", "rccl_ranks"))
        ex = ge.get("exposed_communication")
        if ex:
            rec["config"]["gradient_exchange"].update(_pick(ex, ("exposed_ms", "step_ms_with_exchange", "step_ms_without_exchange")))
    rf = out.get("roofline")
    if rf:
        r = _pick(rf, ("bound", "achieved", "peak", "unit", "frac"))
        r["traffic"] = rf.get("traffic")
        r.update(_pick(rf, ("traffic_ratio", "algorithmic_mb_per_launch", "avg_kernel_us", "launches_per_step", "kernel_only_ms_per_step",
                            "fp32_equiv_tflops")))
        r["kernel"] = str(rf.get("kernel_symbol") or rf.get("kernel", ""))[:64]
        mp = rf.get("mfma_random_operand_peak") or {}
        if mp.get("tflops"):
            r["mfma_random_operand_peak"] = _pick(mp, ("tflops", "shader_mhz", "frac_of_it"))
        sc = rf.get("step_counters") or {}
        if sc.get("hbm_bytes_per_step"):
            r["step"] = _pick(sc, ("mfma_busy_fraction_of_serial_kernel_time", "hbm_bytes_per_step", "hbm_fraction_of_peak", "kernel_launches_per_step"))
        rec["roofline"] = r
    sb = out.get("step_bytes")
    if sb:
        rec["step"] = _pick(sb, ("algorithmic_gb", "hbm_gb", "traffic_ratio", "kernel_launches"))
    cb = out.get("cpu_baseline")
    if cb:
        c = _pick(cb, ("value", "unit", "cores", "kind", "s_per_step", "eval_fwd_ms_per_img", "host_cores"))
        c["sample"] = str(cb.get("sample", ""))[:160]
        st = cb.get("single_thread")
        if st:
            c["single_thread"] = _pick(st, ("value", "cores", "batch"))
        rec["cpu_baseline"] = c
    rec.update(_pick(out, ("fwd_ms_per_img", "final_loss", "rccl_ranks")))
    if out.get("step_ms"):
        rec["step_ms"] = _pick(out["step_ms"], ("median", "p10", "p90"))
    db = out.get("decoder_backward")
    if db:
        rec["decoder_backward"] = _pick(db, ("ms", "algorithmic_gb", "frac_hbm_peak", "achieved_tflops", "frac_bf16x6_roof"))
    sct = out.get("step_conv_tflops")
    if sct:
        rec["step_conv_tflops"] = sct.get("tflops")
    if out.get("sustained"):
        rec["sustained"] = _pick(out["sustained"], ("img_per_s", "seconds"))
    for fmt in FORMATS:
        if isinstance(out.get(fmt), dict) and "value" in out[fmt]:
            rec[fmt] = _pick(out[fmt], ("value", "ms_per_step"))
    if out.get("device_data_path"):
        rec["device_data_path"] = _pick(out["device_data_path"], ("img_per_s",))
    if out.get("shared_gpu"):
        rec["shared_gpu"] = True
    if detail_path:
        rec["detail"] = detail_path
    line = json.dumps(rec, separators=(",", ":"))
    if len(line) > COMPACT_LIMIT:                    # never expected: drop the optional scalars rather than print an unparseable line
        for k in ("device_data_path", "sustained", "step_conv_tflops", "step_ms", "final_loss", "decoder_backward", "step"):
            rec.pop(k, None)
        line = json.dumps(rec, separators=(",", ":"))
    assert len(line) <= COMPACT_LIMIT, len(line)
    return line


def write_detail(out):
    """the full record (what rounds 1-5 printed as the line) -> bench_detail.json beside bench.py, and under gpurun_out/ when that exists
    (merged back from the GPU box); returns the path written, relative to the repo root, or None"""
    written = None
    for d in (os.path.join(ROOT, "gpurun_out"), ROOT):
        if not os.path.isdir(d):
            continue
        try:
            with open(os.path.join(d, "bench_detail.json"), "w") as fh:
                json.dump(out, fh, indent=1)
            written = written or os.path.relpath(os.path.join(d, "bench_detail.json"), ROOT)
        except OSError:
            pass
    return written


def main():
    global B, H, W
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # >= 50 timed steps after >= 10 warm-up steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="kitti", help="kitti = BASELINE configs[2] (the metric's config); "
                    "matterport = configs[4] (512x640 bs=4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="N = 1 only: run the data-parallel branch anyway (RCCL communicator, "
                    "bucketed all-reduces in a world of one rank) -- a dry run of the code path the N > 1 launches take")
    ap.add_argument("--dump-kernels", type=str, default=None, help="write the per-launch-shape timing tables (JSON) here")
    ap.add_argument("--no-loader", action="store_true", help="skip the extra leg that feeds the step from the device-side data path")
    ap.add_argument("--other-format", dest="other_format", action="store_true",
                    help="also run the child-process leg in the other operand format (default run: the opt-in fp16 pairs); off by default "
                         "(VERDICT r5 #1: the default run is the record, not the lab notebook)")
    ap.add_argument("--no-other-format", "--no-exact-split", dest="other_format", action="store_false", help="(default) skip that leg")
    ap.add_argument("--sustain", type=float, default=0.0, help="seconds of the sustained-throughput leg after the timed region (default 0: skip)")
    ap.add_argument("--leg", choices=["train-only"], default=None, help="internal: time the training step only and print a small JSON object")
    ap.add_argument("--dry-run-dist", action="store_true", help="N > 1: spawn / rendezvous / one collective / ONE JSON line, no GPU "
                    "work (what the CPU test of the launch path runs)")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    B, H, W = wl["B"], wl["H"], wl["W"]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:          # started plainly: become the launcher of N ranks
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py --gpus %d: WORLD_SIZE is %d (torch.distributed.run --nproc-per-node must equal --gpus)" % (args.gpus, world))
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    shared_gpu = world > 1 and have < world                       # a dry run of the launch path on a smaller box (or none)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if os.environ.get("NCCL_DEBUG", "").upper() in ("", "VERSION"):
        os.environ["NCCL_DEBUG"] = "WARN"                          # RCCL's version banner goes to stdout, where the ONE JSON line belongs
    if world > 1:
        # host-side group (gloo): rendezvous, unique-id exchange, barriers, max over ranks -- nothing of it touches the GPU.  The gradient
        # buckets travel over RCCL through the library's own fp_comm_* entry points (one rank per GPU); ranks that share a GPU (a smaller
        # box: RCCL refuses two ranks per device) exchange over the gloo group instead.  If fp_comm_init fails on any rank, all ranks
        # agree to fall back to the gloo group as well (footprints_amd.parallel.get_communicator) and the line says so.
        dist.init_process_group("gloo", rank=rank, world_size=world)
        os.environ.setdefault("FP_DP_TRANSPORT", "torch" if shared_gpu else "rccl")      # read when footprints_amd.parallel is imported
    if args.dry_run_dist:
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t)
            dist.barrier()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_sum": float(t.item()), "gpus_visible": have, "shared_gpu": shared_gpu,
                              "backend": str(dist.get_backend()) if world > 1 else None}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if have == 0:
        raise SystemExit("bench.py needs a GPU: the product has no CPU path (use --dry-run-dist to exercise the launch path alone)")
    torch.cuda.set_device(local_rank % have)
    distributed = world > 1 or args.force_dist
    if args.force_dist:
        os.environ["FP_DP_FORCE"] = "1"                            # read when footprints_amd.parallel is imported (below)

    from footprints_amd import _lib, ops
    from footprints_amd.model_manager import ModelManager
    from footprints_amd.parallel import broadcast_state, destroy_communicators
    from footprints_amd.training.train import SEED, TrainStep, synthetic_batch

    torch.manual_seed(SEED)
    mm = ModelManager(use_cuda=True, learning_rate=1e-4)          # random-init weights (no checkpoints offline)
    if world > 1:
        broadcast_state(mm.model)
    step = TrainStep(mm.model, mm.optimiser, distributed=distributed)
    batch = synthetic_batch(B, H, W, "cuda", seed=SEED + rank)     # per-rank shard, resident in HBM

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(batch)
    barrier()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # one record per step: GPU-side step durations
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        step(batch)
        marks[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    t = torch.tensor([dt], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                   # host tensor: the gloo side of the group
    dt = float(t.item())
    final_loss = float(step.losses[20])
    if args.leg == "train-only":
        if rank == 0:
            print(json.dumps({"img_per_s": round(world * B * args.steps / dt, 2), "ms_per_step": round(dt / args.steps * 1e3, 3),
                              "steps": args.steps, "warmup": args.warmup, "final_loss": round(final_loss, 5),
                              "operand_format": FMT}), flush=True)
        return

    # data-parallel only: what the gradient exchange adds to a step -- the same step WITH its bucket all-reduces and WITHOUT them (a second
    # TrainStep on the same engine with no reducer), alternating blocks of 10 steps, medians of the GPU-side step durations.  In a world of
    # one (--force-dist) it prices the branch itself (event edges, the communicator's launches); with N ranks it is the exposed, i.e.
    # un-overlapped, communication.  (The replicas' weights drift apart during the blocks without exchange: nothing after this reads them.)
    exposed = None
    if step.reducer is not None:
        step_nc = TrainStep(mm.model, mm.optimiser, distributed=False)
        mm.optimiser.grad_scale = step.reducer.grad_scale

        def block(fn, n=10):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
            ev[0].record()
            for i in range(n):
                fn(batch)
                ev[i + 1].record()
            barrier()
            return [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
        for fn in (step_nc, step):
            for _ in range(4):
                fn(batch)
        barrier()
        with_ms, without_ms = [], []
        for _ in range(3):
            without_ms += block(step_nc)
            with_ms += block(step)
        med = lambda v: sorted(v)[len(v) // 2]
        tt = torch.tensor([med(with_ms), med(without_ms)], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        exposed = {"step_ms_with_exchange": round(float(tt[0]), 3), "step_ms_without_exchange": round(float(tt[1]), 3),
                   "exposed_ms": round(float(tt[0] - tt[1]), 3),
                   "how": "3 x (10 steps without the bucket all-reduces, 10 steps with them), medians of per-step HIP-event durations, max over ranks"}
        del step_nc

    sustained = None
    if args.sustain > 0:
        # every rank loops (the steps contain the all-reduces): with collectives in the step the count comes from the timed region's
        # rank-maximum step time, identical on all ranks
        n_sus = max(8, int(args.sustain / (dt / args.steps))) if distributed else None
        sustained = sustained_leg(step, batch, args.sustain, world, local_rank % have, n_steps=n_sus)
        if world > 1:
            dist.barrier()

    # kernel-exclusive pass (outside the timed region): same steps on ONE stream, HIP events around every convolution launch
    # (every rank runs it when data-parallel: the steps contain the bucket all-reduces; rank 0 reports)
    xtimer, xsteps, ktable, ktable_conc = None, 3, None, None
    if not args.no_kernel_events:
        was, was_plan = step.eng.concurrent, step.use_plan
        step.eng.concurrent, step.use_plan = False, False            # eager, one stream: the instrumented wrappers see every launch
        step(batch)
        xinst = Instrument(ops, CONV_OPS)
        xinst.install()
        torch.cuda.synchronize()
        lib = _lib.load()
        lib.fp_ktime_begin()                                         # HIP events on the launch stream around EVERY kernel launch
        for _ in range(xsteps):
            step(batch)
        ktable = kernel_table(lib, xsteps)                           # synchronises; per kernel symbol, like rocprofv3 --kernel-trace --stats
        xinst.remove()
        xtimer = xinst.timer
        step.eng.concurrent, step.use_plan = was, was_plan
        # the default schedule (five streams, recorded plan) through the same events: what each kernel takes when it shares the chip
        for _ in range(3):
            step(batch)
        torch.cuda.synchronize()
        lib.fp_ktime_begin()
        for _ in range(xsteps):
            step(batch)
        ktable_conc = kernel_table(lib, xsteps)

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
        gb, gf = wl["dec_gb"], wl["dec_gflop"]
        dec_bwd = {"ms": round(dec_ms, 3), "algorithmic_gb": gb, "algorithmic_gflop": gf,
                   "achieved_hbm_gbs": round(gb / dec_ms * 1e3, 1), "frac_hbm_peak": round(gb / dec_ms * 1e3 / HBM_PEAK_GBS, 4),
                   "achieved_tflops": round(gf / dec_ms, 2), "frac_f32_mfma_peak": round(gf / dec_ms / MFMA_F32_PEAK_TFLOPS, 4),
                   "frac_bf16x6_roof": round(gf / dec_ms / BF16X6_PEAK_TFLOPS, 4),
                   "note": "fused-minimum bytes and reference-graph FLOPs of both decoders' backward (SURVEY.md section 8d) for this workload; "
                           "time = HIP events around 10 repetitions after 2 warm-ups; the convolutions are MFMA-bound (72-755 flop/B), the "
                           "phase decomposition executes 4/9 of the upsample convs' FLOPs"}

    # the same step fed by the device-side data path (outside the timed region, reported beside `value`):
    # host samples (resized uint8 images + float64 label maps, as the file readers deliver them) -> pinned double-buffered H2D ->
    # flip / ColorJitter / ToTensor / label algebra kernels on a copy stream -> TrainStep.  Decode + resize stay host-side.
    loader_leg = None
    if rank == 0 and not distributed and not args.no_loader and args.workload == "kitti":
        import random as _random
        import numpy as np
        from footprints_amd.datasets import DeviceBatchAssembler, DeviceLoader
        nrng = np.random.default_rng(SEED)
        pool = []
        for _ in range(2 * B):
            maps = {"visible_ground": nrng.random((H, W)), "ground_depth": nrng.random((H, W)) * 30 * (nrng.random((H, W)) < 0.5),
                    "depth_mask": (nrng.random((H, W)) < 0.1).astype(np.float64), "disparity": nrng.random((H, W)) * 60,
                    "moving_objects": (nrng.random((H, W)) < 0.05).astype(np.float64)}
            pool.append((nrng.integers(0, 256, (H, W, 3), dtype=np.uint8), maps))
        nb = max(6, min(args.steps, 12))
        source = [[pool[(i * B + j) % len(pool)] for j in range(B)] for i in range(nb + 2)]
        asm = DeviceBatchAssembler(B, H, W, dataset="kitti", stream=None if os.environ.get("FP_LOADER_OWN_STREAM") else step.eng.dwg[0])
        it = iter(DeviceLoader(source, asm, is_train=True, rng=_random.Random(SEED)))
        for _ in range(2):
            step(next(it))
        torch.cuda.synchronize()
        l0 = time.perf_counter()
        n_l = 0
        for bt in it:
            step(bt)
            n_l += 1
        torch.cuda.synchronize()
        l_ms = (time.perf_counter() - l0) / max(n_l, 1) * 1e3
        # the assembly kernels alone (copy stream idle otherwise): one batch, HIP events on the assembler's stream
        prm = [__import__("footprints_amd.datasets", fromlist=["draw_augmentation"]).draw_augmentation(True, _random.Random(i)) for i in range(B)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(asm.stream)
        sl = asm.submit(source[0], prm)
        e1.record(asm.stream)
        torch.cuda.synchronize()
        asm.release(sl)
        h2d_mb = (B * H * W * 3 + 5 * B * H * W * 8) / 1e6
        loader_leg = {"ms_per_step": round(l_ms, 3), "img_per_s": round(B / l_ms * 1e3, 1), "steps": n_l,
                      "h2d_plus_assembly_ms_per_batch": round(e0.elapsed_time(e1), 3), "h2d_mb_per_batch": round(h2d_mb, 1),
                      "note": "same train step, batches produced by footprints_amd.datasets.DeviceLoader (pinned double-buffered H2D of uint8 images + "
                              "float64 label maps, then fp_assemble_images / fp_assemble_labels on a copy stream under the previous step); the host side "
                              "of this leg only memcpy's pre-decoded samples into pinned memory (decode / resize are dataset plumbing, out of scope)"}

    # forward-only latency (configs[1]): eval-mode, no_grad, same batch
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
        # opt-in two-term inference mode of the tile convolutions (NOT exact: ~2e-5 of the channel max, tests/test_gpu_network.py)
        mm.model.inference_precision = "bf16x2"
        for _ in range(2):
            mm.model(batch["image"])
        torch.cuda.synchronize()
        f0 = time.perf_counter()
        for _ in range(5):
            mm.model(batch["image"])
        torch.cuda.synchronize()
        fwd_ms_img_bf2 = (time.perf_counter() - f0) / 5 / B * 1e3
        mm.model.inference_precision = "exact"

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        out = {"metric": "training images/sec at %dx%d bs=%d" % (H, W, B), "value": round(world * B * args.steps / dt, 2), "unit": "img/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": DTYPE_LABEL[FMT], "operand_format": FMT, "data": "synthetic",
               "comparability": ("`value` is measured in the fp32-faithful default format (exact bf16x3 operands).  Rounds 2-4 reported the fp16-pair format as "
                                 "`value` (round 4: 1065 img/s) and this format as the side leg `exact_split` (round 4: 768 img/s); the fp16-pair format is now "
                                 "the side leg `fp16_pair`" if FMT == "exact" else
                                 "started with FP_OPERANDS=fp16_pair: `value` is the OPT-IN 22-bit format, the fp32-faithful default is the side leg `exact`"),
               "arithmetic": ("fp32 tensors and accumulation; 3x3 stride-1 convs (fwd, dgrad, wgrad) multiply operands split into scaled fp16 "
                              "pairs (x * 2^k = h + m, per-tensor k from the tensor's largest magnitude, 22 significant bits, three fp16 MFMA "
                              "products hh + hm + mh, fp32 accumulate: measured error vs float64 equal to the exact bf16x3 split's and to fp32 "
                              "MIOpen's), the 7x7 stem, the stride-2 3x3 and the 1x1 convs likewise; the few remaining shapes native fp32 MFMA -- "
                              "OPT-IN format, below fp32" if HP_ON else
                              "fp32 tensors and accumulation; 3x3 stride-1 convs (fwd, dgrad, wgrad) multiply exactly split operands "
                              "(x = h + m + l in bf16, 6 of 9 bf16 MFMA products: error <= fp32 MFMA); remaining convs native fp32 MFMA"),
               "config": {"workload": wl["name"], "per_gpu_batch": B, "global_batch": B * world, "height": H, "width": W,
                          "parallelism": ("dp%d" % world if world > 1 else "single") + (" (forced data-parallel branch, world of one)" if args.force_dist and world == 1 else "")},
               "fwd_ms_per_img": round(fwd_ms_img, 4),
               "fwd_ms_per_img_optin_bf16x2": {"value": round(fwd_ms_img_bf2, 4), "note": "model.inference_precision = 'bf16x2': two bf16 terms per operand "
                                               "in the 3x3 tile convolutions, three MFMA products; outputs within ~2e-5 of the channel max (bar 1e-4), not the default"},
               "final_loss": round(final_loss, 5),
               "step_ms": {"median": round(step_ms[len(step_ms) // 2], 3), "p10": round(step_ms[len(step_ms) // 10], 3),
                           "p90": round(step_ms[min(len(step_ms) - 1, (len(step_ms) * 9) // 10)], 3),
                           "note": "GPU-side durations between per-step HIP events inside the timed region (rank 0)"},
               "sustained": sustained, "decoder_backward": dec_bwd, "device_data_path": loader_leg}
        gf_fwd, gf_step = network_conv_gflop(B, H, W)
        # whole-step figure: the reference graph's conv FLOPs (fwd + dgrad + wgrad) over the measured step time, i.e. including
        # every non-conv kernel, launch gap and the FLOPs the nearest-x2 phase decomposition does not execute
        out["step_conv_tflops"] = {"algorithmic_gflop_per_step": round(gf_step, 1), "tflops": round(gf_step / ms_per_step, 2),
                                   "frac_of_f32_mfma_peak": round(gf_step / ms_per_step / MFMA_F32_PEAK_TFLOPS, 4),
                                   "frac_of_bf16x6_roof": round(gf_step / ms_per_step / BF16X6_PEAK_TFLOPS, 4),
                                   "fwd_algorithmic_gflop": round(gf_fwd, 1), "fwd_tflops": round(gf_fwd / (fwd_ms_img * B), 2)}
        if xtimer is not None:
            groups = xtimer.groups(xsteps)
            # kernel-only durations (fp_ktime_*: events around the kernel launch itself, per kernel symbol) next to each entry
            # point's bracket, which also contains its helper launches (split-K / partial-sum reduces, amax reductions)
            for g in groups:
                sym = GROUP_SYMBOL.get(g["entry_point"])
                rows = [r for r in (ktable or []) if sym and r["kernel"].split("<")[0].split("(")[0] in sym]
                if rows:
                    k_ms = sum(r["ms_per_step"] for r in rows)
                    k_n = sum(r["launches_per_step"] for r in rows)
                    g["kernel_symbol"] = "/".join(sym)
                    g["kernel_only_ms_per_step"] = round(k_ms, 3)
                    g["kernel_launches_per_step"] = round(k_n, 1)
                    g["avg_kernel_us"] = round(k_ms / max(k_n, 1e-9) * 1e3, 2)
                    ex_gflop_step = g["executed_mfma_gflop_per_launch"] * g["launches_per_step"]
                    g["achieved_kernel_only"] = round(ex_gflop_step / k_ms, 1) if k_ms > 0 else 0.0      # GFLOP / ms = TFLOP/s
                    g["frac_kernel_only"] = round(g["achieved_kernel_only"] / g["peak"], 4)
            dom = groups[0]
            tr, tr_why = load_traffic(args.workload, dom["entry_point"])
            ach = dom.get("achieved_kernel_only", dom["achieved"])
            tr_bytes = (tr["fetch_bytes_per_launch"] + tr["write_bytes_per_launch"]) if tr else None
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": dom["peak"], "unit": "TFLOP/s", "frac": round(ach / dom["peak"], 4),
                               "traffic": tr_bytes,                 # HBM bytes per launch (read + write) from the PMC passes, or null
                               "traffic_ratio": round(tr_bytes / (dom["algorithmic_mb_per_launch"] * 1e6), 3) if tr_bytes else None,
                               "traffic_detail": tr if tr else {"attached": False, "why": tr_why},
                               "kernel": dom["kernel"], "entry_point": "fp_" + dom["entry_point"], "mfma": dom["mfma"],
                               "launches_per_step": dom["launches_per_step"], "avg_launch_us": dom["avg_launch_us"],
                               "avg_kernel_us": dom.get("avg_kernel_us"), "kernel_launches_per_step": dom.get("kernel_launches_per_step"),
                               "kernel_only_ms_per_step": dom.get("kernel_only_ms_per_step"),
                               "entry_point_bracket": {"achieved": dom["achieved"], "frac": dom["frac"],
                                                       "note": "the same FLOPs over the entry point's event bracket, which also contains its helper "
                                                               "launches (split-K reduce, amax reduction)"},
                               "exclusive_ms_per_step": dom["exclusive_ms_per_step"],
                               "executed_mfma_gflop_per_launch": dom["executed_mfma_gflop_per_launch"],
                               "fp32_equiv_tflops": dom["fp32_equiv_tflops"], "algorithmic_mb_per_launch": dom["algorithmic_mb_per_launch"],
                               "how": "dominant = largest exclusive time per step among the convolution entry points; %d extra steps with "
                                      "concurrency off (one stream) outside the timed region; achieved = executed MFMA FLOPs of the entry point's "
                                      "launches (3 fp16 products per multiply-add for scaled fp16 pairs, 6 bf16 products for the exact bf16 "
                                      "split) / summed duration of its main kernel's launches, each measured with HIP events recorded on the "
                                      "launch stream right around the kernel launch (fp_ktime_*: the per-kernel figure of rocprofv3 "
                                      "--kernel-trace --stats, see `kernels`)" % xsteps,
                               "reading": "frac prices EXECUTED MFMA FLOPs against the nominal dense peak, so it falls whenever products are removed "
                                          "from the split (exact bf16x3: 6 per multiply-add, frac 0.34 at 115 fp32-equivalent TFLOP/s; scaled fp16 pairs: "
                                          "3, frac 0.24 at 150): compare fp32_equiv_tflops across operand formats; over a whole step the chip sustains ~1.9 of "
                                          "its 2.4 GHz (`sustained.shader_clock_mhz`, from its own counters), i.e. ~0.8 of the nominal peak is attainable",
                               "conv_exclusive_ms_per_step": round(sum(g["exclusive_ms_per_step"] for g in groups), 3),
                               "groups": groups}
            # the same instruction's ceiling on toggling data, measured beside the run (the nominal peak is a constant-data figure)
            try:
                mp = mfma_random_operand_peak(lib)
                out["roofline"]["mfma_random_operand_peak"] = dict(mp, frac_of_it=round(ach / mp["tflops"], 4))
            except Exception as ex:                                # a measurement aid must not take the headline down
                out["roofline"]["mfma_random_operand_peak"] = {"error": repr(ex)[:200]}
            sc, sc_why = load_step_counters(args.workload)
            if sc:
                hb = sc["hbm_bytes_per_step"]["total"]
                out["roofline"]["step_counters"] = {
                    "mfma_busy_fraction_of_serial_kernel_time": sc["mfma"]["busy_fraction_of_serial_kernel_time"],
                    "hbm_bytes_per_step": hb, "hbm_gb_per_s": round(hb / ms_per_step / 1e6, 1),
                    "hbm_fraction_of_peak": round(hb / ms_per_step / 1e6 / HBM_PEAK_GBS, 4),
                    "kernel_launches_per_step": sc.get("kernel_launches_per_step"),
                    "how": "rocprofv3 counters-only passes over scripts/step_loop.py on this build (scripts/pmc_step.py, profiles/round%d_pmc_step_%s_%s.json): "
                           "MFMA busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs) summed over every kernel of a step; HBM bytes = "
                           "FETCH_SIZE x 2 + WRITE_SIZE per step, over THIS line's ms_per_step" % (PROFILE_ROUND, args.workload, FMT)}
            else:
                out["roofline"]["step_counters"] = {"attached": False, "why": sc_why}
            # whole-step waste ratio (VERDICT r5 "Next" 7): counter traffic of one step over the fused-minimum bytes of the whole step
            alg = step_algorithmic_bytes(B, H, W)
            hb_gb = round(sc["hbm_bytes_per_step"]["total"] / 1e9, 3) if sc else None
            out["step_bytes"] = {"algorithmic_gb": alg["total"], "hbm_gb": hb_gb, "traffic_ratio": round(hb_gb / alg["total"], 3) if hb_gb else None,
                                 "kernel_launches": sc.get("kernel_launches_per_step") if sc else None, "algorithmic_by_family_gb": alg,
                                 "note": "algorithmic = SURVEY.md section 8(d)'s per-convolution fused-minimum rule over every convolution of the "
                                         "network + loss + Adam (bench.py step_algorithmic_bytes); hbm = FETCH_SIZE x 2 + WRITE_SIZE of one step from the "
                                         "committed counters-only passes of this build (null when not collected for this build)"}
            out["kernels"] = {"serial": (ktable or [])[:24], "concurrent": (ktable_conc or [])[:24],
                              "note": "per kernel symbol, from HIP events around every kernel launch of %d steps (fp_ktime_*): `serial` = one stream, "
                                      "eager launches (exclusive durations); `concurrent` = the default five-stream schedule replayed from the "
                                      "recorded plan (a launch shares the chip with its neighbours)" % xsteps}
        if args.dump_kernels and xtimer is not None:
            with open(args.dump_kernels, "w") as fh:
                json.dump({"groups": xtimer.groups(xsteps), "shapes": xtimer.table()}, fh, indent=1)
        if shared_gpu:
            out["shared_gpu"] = {"gpus_visible": have, "note": "fewer GPUs than ranks: the ranks share them and exchange gradients over gloo -- a "
                                                               "functional dry run of the launch path, NOT a performance number"}
        if step.reducer is not None:
            comm = step.reducer.comm
            ar_rows = [r for r in (ktable_conc or []) if r["kernel"].startswith("rccl_allreduce")]
            out["config"]["gradient_exchange"] = {"transport": step.reducer.transport, "buckets": len(step.reducer.buckets),
                                                  "overlap_with_backward": bool(step.reducer.overlap),
                                                  "in_launch_plan": bool(step.reducer.plan_recordable and step.use_plan),
                                                  # from the communicator itself (ncclCommCount), not from WORLD_SIZE
                                                  "rccl_ranks": comm.count() if (comm is not None and hasattr(comm, "count")) else 0,
                                                  "allreduce_us": [{"bucket": r["kernel"], "per_step": r["launches_per_step"], "avg_us": r["avg_us"]}
                                                                   for r in ar_rows]}
            out["config"]["gradient_exchange"]["exposed_communication"] = exposed
            out["rccl_ranks"] = out["config"]["gradient_exchange"]["rccl_ranks"]
            if step.reducer.transport == "rccl" and not shared_gpu and out["rccl_ranks"] != world:
                raise SystemExit("bench.py --gpus %d: the RCCL communicator counts %d ranks (ncclCommCount), expected %d -- not a valid "
                                 "data-parallel measurement" % (args.gpus, out["rccl_ranks"], world))
            if world > 1 and not shared_gpu and step.reducer.transport != "rccl":
                raise SystemExit("bench.py --gpus %d: one GPU per rank but the gradient exchange runs over %r instead of RCCL" % (args.gpus, step.reducer.transport))
        if world == 1 and not args.force_dist and args.other_format:
            del step, mm                                               # free this process's arena before the child builds its own
            torch.cuda.empty_cache()
            other = [f for f in FORMATS if f != FMT][0]
            out[other] = {"what": "the same command re-run in a child process with FP_OPERANDS=%s (%s): same steps / warm-up / sustain, same "
                                  "instrumentation; `value` above is NOT this leg" % (other, DTYPE_LABEL[other]), **other_format_leg(other, args)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    destroy_communicators()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        try:                                                       # C-side stdio of the libraries first: the JSON line is the last line on stdout
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(compact_record(out, write_detail(out)), flush=True)


if __name__ == "__main__":
    main()
