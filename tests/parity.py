"""Shared parity helpers for the -m gpu tests (test infrastructure).

The fp64-anchored rule: the CPU oracle run in float64 is the truth; the CPU oracle run in float32 (the arithmetic the
reference itself uses, training/train.py:150-156 on torch CPU ops) shows how far a correct fp32 implementation sits from it.
The HIP engine -- another fp32 implementation with a different summation order -- has to be as close to the truth as the
reference's own arithmetic is:   err(GPU vs fp64) <= FACTOR * max(err_cpu32(tensor), median err_cpu32 of the tensor's stage)
per parameter tensor (relative L2), with a floor far below every other tolerance of the suite for tensors where both errors are
at round-off level.  No hand-picked constant per layer: ill-conditioned tensors (train-mode BatchNorm over few samples, ReLU
masks next to 0) get exactly the slack the reference's own fp32 arithmetic needs on the same inputs.

Why FACTOR = 4 and the stage median (measured on the MI355X box, profiles/round2_notes.md): over 196 tensors x 3 shapes the
ratio err_gpu / err_cpu32 has median 0.90-1.15 -- the two fp32 implementations are equally far from the truth -- but a heavy
tail: the encoder is piecewise linear (ReLU after train-mode BN), and ONE activation within fp32 round-off of 0 that the two
implementations resolve differently moves a whole BatchNorm channel's statistics (ratios up to 3.3 on layer4 tensors at
12x192x640; and a tensor on which the CPU happened to be lucky, 4e-6 against its neighbours' 1e-4..2e-3, reads as 25).  The stage
median keeps one lucky CPU tensor from setting its own bound; the factor covers the flip tail.  A real defect shows as
hundreds (the test that introduced the rule flagged a single |.|-kink sign flip in the depth loss at ratio 200-700).

The loss has its own kink: log(|d(o) - depth| + 1) (losses.py:95-107) flips the sign of a pixel's gradient when the predicted
depth crosses the target.  A pixel with |d - depth| within reach of fp32 output error is undetermined in ANY fp32
implementation and, sitting at the steepest point of the loss, carries ~30x a typical pixel's gradient: one such pixel among
327 680 moved every depth-decoder gradient by 1.5e-4 (1x512x640).  `tie_free_batch` removes those pixels from the valid masks
(depth / ground_depth := 0 where the float64 prediction is within TIE of the target at any scale) for all three runs alike.
Round 3 added a second band for ONE test: as wide as the depth moves under an OUTPUT perturbation of the parity tolerance itself
(TIE_SIGMA = 1e-4 on the sigmoid: (hi - lo) d^2 * 1e-4 metres).  With saturated predictions (sigmoid ~ 0, d ~ 100 m: the
natural-statistics case) a relative band of 1 % is 1 m while two conforming implementations may differ by 10 m there.  Round 4 (ADVICE
r3): that band is OPT-IN (`tie_sigma=TIE_SIGMA`, used by the natural-statistics / wide-range test only); the standard cases run with the
1 % band alone, as in round 2, and every caller asserts an upper bound on the number of pixels it removed.

Round 5 (profiles/round5_notes.md section 1) adds the decision-forced evaluation: every case is ALSO compared with the float64 oracle
evaluated under the engine's own ReLU decisions (`decision_forced_report`).  Against that truth every tensor has to pass the per-tensor
rule, the errors have ABSOLUTE bounds (FORCED_MAX_ERR / FORCED_MEDIAN_ERR), and the upper median gate applies to whichever of the two
medians is smaller: a median above 1.5 against the plain float64 run that drops below it once the decisions are imposed is a draw of the
ReLU lottery (89 of 104 693 760 decisions at 12x192x640), not a drift of the arithmetic.

The gates are fixed (round 4; a change needs a written reason under profiles/): per tensor the rule above; per case the median of
err(GPU) / err(CPU fp32) inside [0.4, 1.5] ([0.2, 1.5] for the natural-statistics case, whose bound comes from five fp32 runs).  The lower
end is not a defect -- split operands with exact products are MORE accurate than an fp32 accumulation chain, a median of 0.5 says the
engine sits twice as close to float64 as the CPU path -- it is there so that a change of the distribution in either direction gets
looked at.  KINK_MAX_FRACTION bounds the masked pixels (measured: 0.03-0.09 % standard, 0.76 % natural)."""
from collections import OrderedDict

import torch

FACTOR = 4.0
FLOOR = 2e-5          # relative L2; both implementations at fp32 round-off
TIE = 1e-2            # |predicted depth - target| (metres, relative to max(depth, 1)) below which the L1 kink is undetermined in fp32
MEDIAN_GATE = (0.4, 1.5)           # median err(GPU) / err(CPU fp32) per case (see the module docstring)
MEDIAN_GATE_NATURAL = (0.2, 1.5)
# Round 5: absolute bounds on err(GPU) against the float64 oracle evaluated under the engine's own ReLU decisions (decision_forced_report) -- with
# the decisions taken out, what is left is arithmetic, and that has an absolute size: measured at 12x192x640 max 1.8e-4 / median 1.3e-6 (exact
# bf16x3 operands) and 5.0e-4 / 9.1e-7 (fp16 pairs) over 196 tensors (profiles/round5_parity_ratios.md).  The maximum sits on the BatchNorm
# affine gradients of the deepest layers (sums of ~1e3 terms that cancel to ~1e-2 of their magnitude: the amplification of section 8 of
# profiles/round4_notes.md).  The bounds leave a factor 2-4 over the worst measured case.
FORCED_MAX_ERR = 1e-3              # measured over 4 sizes x 2 formats: 1.3e-5 .. 5.0e-4 (profiles/round5_parity_ratios.md)
FORCED_MEDIAN_ERR = 1e-5           # measured: 9.1e-7 .. 2.6e-6
FORCED_MAX_ERR_NATURAL = 5e-3      # the natural-statistics / wide-range case (BatchNorm gammas spread over 2^16): measured 5.6e-4
FORCED_MEDIAN_ERR_NATURAL = 2e-4   # measured 3.4e-5
# Round 6 (VERDICT r5 "Next" 2 / ADVICE r5): the imposed decisions themselves are bounded -- without that the forced oracle could absorb a real
# masking error of the engine.  Every ReLU decision of the engine that differs from the float64 run's must sit where float64 |z| is at
# round-off distance from zero (relative to the RMS of z's channel), every differing max-pool winner must tie the true maximum to the same
# degree, and the number of differing decisions is a vanishing fraction of all decisions.  Two bounds on the distance:
#   * absolute, FLIP_MAX_DISTANCE = 1e-4: the parity contract's own tolerance (north_star: tensors within 1e-4) applied to the pre-activation --
#     an element whose float64 value lies within 1e-4 of its channel's RMS from zero is one that two conforming implementations may resolve
#     differently; a mis-masked element of ordinary magnitude sits at ~1 (tests/test_format_and_decisions_cpu.py);
#   * relative to the reference's arithmetic: the CPU fp32 run's OWN decisions are imposed on the float64 oracle the same way
#     (`reference_flip_stats`), and the engine's worst distance may not exceed FACTOR x the CPU fp32 run's (floor FLIP_DISTANCE_FLOOR = 1e-5, the
#     judge's figure, VERDICT r5 "Next" 2) -- the same anchoring as the gradient rule: round-off grows with depth (first measurement, round 6:
#     2.0e-5 at the 29th ReLU, layer4, 1x512x640), and the reference's own fp32 arithmetic shows by how much.
# Measured: profiles/round6_parity_ratios.md (84-89 of 104 693 760 decisions at 12x192x640 = 8e-7).
FLIP_MAX_FRACTION = 5e-6           # of all ReLU (resp. max-pool) decisions
FLIP_MAX_DISTANCE = 1e-4           # |z64| / RMS(channel) at a flipped ReLU; (max - imposed winner) / RMS(channel) at a flipped pool window
FLIP_DISTANCE_FLOOR = 1e-5         # below this the relative rule does not bind (both implementations at round-off)
KINK_MAX_FRACTION = 0.005          # of the 2 B H W depth-target pixels, standard cases
KINK_MAX_FRACTION_NATURAL = 0.02
TIE_SIGMA = 1e-4      # (opt-in) ... and the output tolerance of the parity contract itself (north_star: depth / mask tensors within 1e-4): a sigmoid
                      # output may move by this much between two conforming implementations, which moves the predicted depth
                      # d = 1 / (lo + (hi - lo) sigma) by (hi - lo) d^2 * 1e-4 -- 10 m at d = 100 m: far pixels sit on the kink for any target


def rel_l2(t, ref64):
    ref64 = ref64.double()
    return ((t.detach().double().cpu() - ref64).norm() / ref64.norm().clamp_min(1e-300)).item()


def chan_relerr(got, ref):
    """per-channel max|got - ref| / max|ref| of [B,C,H,W] tensors -> list of C floats (a channel whose magnitude is far below the
    tensor's max cannot hide behind it)"""
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    d = (got - ref).abs().amax(dim=(0, 2, 3))
    s = ref.abs().amax(dim=(0, 2, 3)).clamp_min(1e-30)
    return (d / s).tolist()


def oracle_grads(P, B, cpu_batch, dtype, fix_batch=None, perturb=0.0, seed=0, record=None, relu_decisions=None):
    """one train-mode fwd + loss + bwd of the CPU oracle in `dtype` -> (outputs, losses, {name: grad}, trainer, batch used).
    fix_batch(batch, outputs) -> batch: applied between the forward and the loss (the outputs do not depend on the targets).
    perturb > 0: the image is multiplied by (1 + perturb * u), u uniform in [-1, 1) (seeded) -- "another conforming fp32
    implementation": a perturbation at round-off level draws a different set of ReLU-kink decisions (see fp32_spread).
    record (optional list): receives every encoder BasicBlock output with its gradient retained (oracle/restatement.py resnet_encoder)
    relu_decisions (optional oracle.restatement.ReluDecisions): the encoder is evaluated with imposed ReLU decisions instead of its own
    and / or its decisions are recorded (round 5, see `decision_forced_report`)"""
    from oracle import restatement as R
    Pd = OrderedDict((k, v.to(dtype)) for k, v in P.items())
    Bd = OrderedDict((k, v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in B.items())
    tr = R.OracleTrainer(Pd, Bd)
    if perturb > 0.0:
        g = torch.Generator().manual_seed(1234 + seed)
        cpu_batch = OrderedDict(cpu_batch)
        cpu_batch["image"] = cpu_batch["image"] * (1.0 + perturb * (torch.rand(cpu_batch["image"].shape, generator=g) * 2 - 1))
    out = R.footprint_network(cpu_batch["image"].to(dtype), tr.P, tr.B, True, record=record, relu_decisions=relu_decisions)
    used = cpu_batch if fix_batch is None else fix_batch(cpu_batch, out)
    losses, _ = R.loss_manager(out, OrderedDict((k, v.to(dtype)) for k, v in used.items()))
    for p in tr.P.values():
        p.grad = None
    losses["loss"].backward()
    grads = OrderedDict((k, p.grad) for k, p in tr.P.items())
    return {k: v.detach() for k, v in out.items()}, losses, grads, tr, used


def stage_of(name):
    """encoder.layerK / mask_decoder.blockK / depth_decoder.outconvK ...: the granularity at which conditioning is homogeneous"""
    p = name.split(".")
    return ".".join(p[:2])


def anchored_report(gpu, cpu32, ref64, factor=FACTOR, floor=FLOOR, spread=(), cpu_ref64=None):
    """gpu / cpu32 / ref64: {name: tensor or None}.  Returns (failures, rows) with rows = (ratio, name, err_gpu, err_cpu).
    cpu_ref64 (optional): the truth the fp32 CPU runs are measured against when it differs from the engine's (decision_forced_report:
    the engine against the float64 oracle under the ENGINE's ReLU decisions, the CPU runs against the float64 oracle's own).
    spread: further fp32 CPU runs on inputs perturbed at round-off level (oracle_grads(perturb=...)): a tensor's fp32 error is then
    the LARGEST over all fp32 runs -- where the network is chaotic (extreme BatchNorm scales over a few hundred samples: one ReLU
    decision within round-off of zero moves a whole channel's statistics) one fp32 run is a single draw of a heavy-tailed lottery,
    and the bound has to come from how far conforming fp32 implementations scatter, not from one of them."""
    errs = {}
    for n, r in ref64.items():
        if r is None:
            assert gpu.get(n) is None, "%s: the oracle has no gradient here, the engine produced one" % n
            continue
        assert gpu.get(n) is not None, "%s: missing gradient" % n
        rc = r if cpu_ref64 is None else cpu_ref64[n]
        ec = max([rel_l2(cpu32[n], rc)] + [rel_l2(sp[n], rc) for sp in spread])
        errs[n] = (rel_l2(gpu[n], r), ec)
    stages = {}
    for n, (_, ec) in errs.items():
        stages.setdefault(stage_of(n), []).append(ec)
    med = {s: sorted(v)[len(v) // 2] for s, v in stages.items()}
    rows, bad = [], []
    for n, (eg, ec) in errs.items():
        bound = max(factor * max(ec, med[stage_of(n)]), floor)
        rows.append((eg / max(ec, 1e-30), n, eg, ec))
        if not eg <= bound:
            bad.append("%s gpu %.2e cpu32 %.2e stage median %.2e bound %.2e" % (n, eg, ec, med[stage_of(n)], bound))
    rows.sort(reverse=True)
    return bad, rows


def decision_forced_report(P, B, cpu_batch, decisions, gpu, cpu32, ref64, spread=()):
    """Round 5 (VERDICT r4 "Next" 1b): separate an implementation's DECISIONS from its ARITHMETIC.

    The encoder is piecewise linear: 33 ReLUs behind train-mode BatchNorms.  An activation within fp32 round-off of zero lands on either
    side of its ReLU depending on the summation order of whoever computes it, and a flipped element moves a whole BatchNorm channel's
    gradient sums (profiles/round4_notes.md section 8: ONE element of 737 280 moved `encoder.layer4.2.bn2.weight` by 1e-4) -- the single-run
    rule of anchored_report then compares one draw of that lottery (the engine's) with another (the CPU fp32 run's).  Here the float64
    oracle is evaluated once more with the ENGINE's own ReLU decisions -- and its max-pool winners: two window elements within round-off of
    each other are the same kind of decision, and they reach the stem's weight gradient -- imposed (oracle/restatement.py `_relu`, `_maxpool`; forward values are
    unchanged to ~1e-7, a flipped element being ~0 on either side; the backward pass follows the engine's masks).  Against THAT truth the
    engine's gradients carry arithmetic error only, and every tensor has to pass the same bound as before: a tensor that fails the
    single-run rule but passes here differs from float64 by decisions at round-off distance from zero, which no fp32 implementation
    determines; one that fails here too is a defect.  Returns (failures, rows, grads of the forced oracle, flip statistics): the last is
    what `assert_decisions_at_roundoff` bounds -- the imposed run measures, for every decision it was told to take against its own float64
    judgement, how far from the boundary the float64 value sits."""
    from oracle.restatement import ReluDecisions
    imposed = ReluDecisions(impose=decisions["relu"], pool_impose=decisions.get("pool")) if isinstance(decisions, dict) else ReluDecisions(impose=decisions)
    g64f = oracle_grads(P, B, cpu_batch, torch.float64, relu_decisions=imposed)[2]
    bad, rows = anchored_report(gpu, cpu32, g64f, spread=spread, cpu_ref64=ref64)
    pool = decisions.get("pool") if isinstance(decisions, dict) else None
    stats = {"relu_flips": imposed.relu_flips, "relu_decisions": sum(m.numel() for m in imposed.taken),
             "relu_flip_worst_distance": imposed.relu_flip_worst, "relu_flip_worst_where": imposed.relu_flip_where,
             "pool_flips": imposed.pool_flips, "pool_decisions": 0 if pool is None else pool.numel(), "pool_flip_worst_distance": imposed.pool_flip_worst}
    return bad, rows, g64f, stats


def reference_flip_stats(P, B, cpu_batch, relu_decisions32):
    """the same measurement for the REFERENCE's arithmetic: the CPU fp32 run's ReLU decisions imposed on the float64 oracle (its max-pool
    winners are not recorded: the pool decides itself) -> how many differ from float64 and how far from zero the worst one sits"""
    from oracle.restatement import ReluDecisions
    imposed = ReluDecisions(impose=relu_decisions32)
    oracle_grads(P, B, cpu_batch, torch.float64, relu_decisions=imposed)
    return {"relu_flips": imposed.relu_flips, "relu_decisions": sum(m.numel() for m in imposed.taken),
            "relu_flip_worst_distance": imposed.relu_flip_worst, "relu_flip_worst_where": imposed.relu_flip_where}


def fp32_forward_decisions(P, B, cpu_batch):
    """the ReLU decisions of the CPU oracle's fp32 forward (train-mode BatchNorm on copies of the buffers, no gradients): what
    `ReluDecisions(probe=...)` measures against a float64 run -- the reference-arithmetic anchor of the flip bound at the price of one fp32
    forward instead of a float64 run of its own (`reference_flip_stats`, kept for the CPU tests, gives the same numbers to ~1e-7)"""
    from oracle import restatement as R
    rec = R.ReluDecisions()
    with torch.no_grad():
        R.footprint_network(cpu_batch["image"].float(), OrderedDict((k, v.float()) for k, v in P.items()),
                            OrderedDict((k, v.clone()) for k, v in B.items()), True, relu_decisions=rec)
    return rec.taken


def probe_flip_stats(dec64):
    """the statistics a float64 recording run collected about its probe (see fp32_forward_decisions), in reference_flip_stats's format"""
    return {"relu_flips": dec64.probe_flips, "relu_decisions": sum(m.numel() for m in dec64.taken),
            "relu_flip_worst_distance": dec64.probe_flip_worst, "relu_flip_worst_where": dec64.probe_flip_where}


def assert_decisions_at_roundoff(stats, tag="", max_fraction=None, max_distance=None, reference=None):
    """the bound on what decision_forced_report imposed (round 6): few, and each at round-off distance from its boundary in float64;
    reference (optional, `reference_flip_stats`): additionally no further from the boundary than FACTOR x the CPU fp32 run's own worst flip"""
    max_fraction = FLIP_MAX_FRACTION if max_fraction is None else max_fraction
    max_distance = FLIP_MAX_DISTANCE if max_distance is None else max_distance
    if reference is not None:
        rel = max(FACTOR * reference["relu_flip_worst_distance"], FLIP_DISTANCE_FLOOR)
        assert stats["relu_flip_worst_distance"] <= rel, ("%s: the engine's worst flipped ReLU decision sits at %.2e x RMS from zero in float64, the CPU fp32 "
                                                           "run's own worst one at %.2e (bound: %g x that, floor %.0e)" % (
            tag, stats["relu_flip_worst_distance"], reference["relu_flip_worst_distance"], FACTOR, FLIP_DISTANCE_FLOOR))
        assert stats["relu_flips"] <= max(FACTOR * reference["relu_flips"], 8), "%s: %d ReLU decisions of the engine differ from float64, %d of the CPU fp32 run" % (
            tag, stats["relu_flips"], reference["relu_flips"])
    assert stats["relu_flips"] <= max(max_fraction * stats["relu_decisions"], 2), "%s: %d of %d ReLU decisions differ from float64 (bound %.1e)" % (
        tag, stats["relu_flips"], stats["relu_decisions"], max_fraction)
    assert stats["relu_flip_worst_distance"] <= max_distance, ("%s: a ReLU decision imposed on the float64 oracle sits at |z64| = %.2e x its channel's RMS "
                                                                "(ReLU #%s, channel %s): not a round-off tie (bound %.1e)" % (
        tag, stats["relu_flip_worst_distance"], *(stats["relu_flip_worst_where"] or (None, None)), max_distance))
    assert stats["pool_flips"] <= max(max_fraction * max(stats["pool_decisions"], 1), 2), "%s: %d of %d max-pool winners differ from float64" % (
        tag, stats["pool_flips"], stats["pool_decisions"])
    assert stats["pool_flip_worst_distance"] <= max_distance, "%s: an imposed max-pool winner is %.2e x RMS below the float64 maximum (bound %.1e)" % (
        tag, stats["pool_flip_worst_distance"], max_distance)


def count_decision_flips(decisions, decisions_ref):
    """number of ReLU decisions that differ between two runs, total number"""
    decisions = decisions["relu"] if isinstance(decisions, dict) else decisions
    flips = sum(int((a != b).sum()) for a, b in zip(decisions, decisions_ref))
    return flips, sum(a.numel() for a in decisions)


def tie_free_batch(cpu_batch, out64, depth_range=(0.1, 100.0), tie=TIE, tie_sigma=0.0):
    """copy of the batch with the |.|-kink pixels of the depth losses removed from the valid masks (see the module docstring);
    out64: the float64 oracle's outputs (they do not depend on the targets).  Returns (batch, number of pixels removed)."""
    batch = OrderedDict((k, v.clone()) for k, v in cpu_batch.items())
    lo, hi = 1.0 / depth_range[1], 1.0 / depth_range[0]
    removed = 0
    for ch, key in ((2, "depth"), (3, "ground_depth")):
        t = batch[key].double()
        tiebrk = torch.zeros_like(t, dtype=torch.bool)
        for o in out64.values():
            d = 1.0 / (lo + (hi - lo) * o[:, ch].detach().double())
            band = torch.maximum(tie * d.clamp_min(1.0), (hi - lo) * d * d * tie_sigma)
            tiebrk |= ((d - t).abs() <= band) & (t > 0)
        removed += int(tiebrk.sum())
        batch[key][tiebrk] = 0.0
    batch["all_ground"] = ((batch["ground_depth"] + batch["visible_ground"]) > 0).float()      # kitti_dataset.py:114-122
    return batch, removed
