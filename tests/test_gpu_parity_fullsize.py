"""GPU: train-mode forward + loss + backward of the HIP engine at the BASELINE.json workloads against the CPU oracle, with
float64-anchored gradient tolerances (tests/parity.py) instead of hand-picked constants.

Why these sizes: tile plans, split-K factors, weight-gradient split counts and the phase-kernel fallbacks all depend on
N, H, W (conv3x3_tile_bf3.hip plan3, wgrad3x3_bf3.hip, Engine._phase_ok), so parity at 2x96x128 says nothing about the
launches the benchmark actually runs:
  * 12x192x640  -- BASELINE configs[2] (KITTI train step): outputs, 21 losses, every parameter gradient, BN running statistics;
  *  1x512x640  -- the Matterport resolution of configs[4] (16x20 .. 512x640 pyramid: other plans / splits than KITTI);
  *  4x512x640  -- configs[4] itself (Matterport bs=4: the split-K / weight-gradient split plans depend on N);
  *  1x256x448  -- predict_simple's `handheld` model size (8x14 pyramid top: the phase kernels' padding fallbacks).
Round 5: every case runs in BOTH operand formats (footprints_amd/_format.py: the exact bf16x3 split = the default = what bench.py's `value`
is measured in, and the opt-in scaled fp16 pairs); the format the session itself uses runs in-process, the other one in a child process
(tests/gpu_child.py), both against ONE set of oracle runs.  A tensor that fails the single-run rule has to pass it against the float64 oracle
evaluated under the ENGINE's own ReLU decisions (tests/parity.py decision_forced_report): decisions within round-off of zero are not
arithmetic, everything else is.
Bars: outputs per CHANNEL within 1e-4 of the fp32 CPU path (north_star) and of the float64 truth; losses 1e-4 relative; masks
bit-exact outside the |logit - thr| < 1e-4 max tie band; every parameter gradient fp64-anchored:
err(GPU vs fp64) <= 4 x max(err(CPU fp32 vs fp64), its stage median), relative L2 per tensor, floor 2e-5 (tests/parity.py says why).
"""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from tests.gpu_child import gpu_step
from tests.parity import (FORCED_MAX_ERR, FORCED_MAX_ERR_NATURAL, FORCED_MEDIAN_ERR, FORCED_MEDIAN_ERR_NATURAL, KINK_MAX_FRACTION, KINK_MAX_FRACTION_NATURAL, MEDIAN_GATE, MEDIAN_GATE_NATURAL, TIE_SIGMA,
                          anchored_report, assert_decisions_at_roundoff, chan_relerr, count_decision_flips, decision_forced_report, fp32_forward_decisions, oracle_grads, probe_flip_stats,
                          reference_flip_stats, rel_l2,
                          tie_free_batch)

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _keep_ratio_table(tag, rows, extra=None):
    """err(GPU) / err(CPU fp32) per parameter tensor (both against the float64 oracle) -> gpurun_out/parity/<tag>.json: the table
    travels back from the GPU box and is committed under profiles/ every round (tests/conftest.py assembles the markdown table at the end
    of the session), so drift of the distribution is visible.  Returns the median, which the callers hold inside tests.parity.MEDIAN_GATE =
    [0.4, 1.5] ([0.2, 1.5] for the natural-statistics case): fixed in round 4, see tests/parity.py."""
    ratios = [r for r, *_ in rows]
    med = float(np.median(ratios))
    doc = {"case": tag, "tensors": len(rows), "median_ratio": round(med, 4), "count_ratio_gt_2": int(sum(r > 2 for r in ratios)),
           "p90_ratio": round(float(np.percentile(ratios, 90)), 4),
           "top6": [{"tensor": n, "ratio": round(r, 3), "err_gpu": float("%.3e" % eg), "err_cpu32": float("%.3e" % ec)} for r, n, eg, ec in rows[:6]]}
    if extra:
        doc.update(extra)
    out_dir = os.environ.get("FP_PARITY_DUMP", os.path.join(ROOT, "gpurun_out", "parity"))
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, tag + ".json"), "w") as fh:
            json.dump(doc, fh, indent=1)
    except OSError:
        pass
    return med


def _top_share(err, k=3):
    e2 = (err * err).flatten()
    return round(float(e2.topk(k).values.sum() / e2.sum().clamp_min(1e-300)), 4)


def _flips(mask, mask64, dout64, z, truth):
    """[(channel, d out of the element, its xhat, the element's term relative to the channel's d gamma)] of the elements whose ReLU decision
    differs from the float64 oracle's"""
    xhat = (z - z.mean(0)) / torch.sqrt(z.var(0, unbiased=False) + 1e-5)
    rows = []
    for m, c in (mask != mask64).nonzero().tolist()[:8]:
        rows.append({"channel": c, "dout": float("%.3e" % dout64[m, c]), "xhat": round(float(xhat[m, c]), 3),
                     "term_over_dgamma": float("%.3e" % (dout64[m, c] * xhat[m, c] / truth[c]))})
    return rows


def _last_bn_decomposition(taps, x64, x32, g_gpu, g32, g64, tag):
    """Where does the error of d loss / d encoder.layer4.2.bn2.weight come from?  d gamma = sum g * xhat over the 12 x 6 x 20 samples of a
    channel, with sum g / sum g xhat ~ 10^2..10^3 (profiles/round4_notes.md section 8).  Three float64 evaluations of the same formula on the
    host separate the engine's INPUTS (g = incoming gradient * ReLU mask; z = conv2's output) from its BatchNorm ARITHMETIC:
      (a) the engine's own d gamma                                                        -> total error
      (b) float64 arithmetic on the engine's g and z                                     -> error carried by the inputs
      (c) float64 arithmetic on the engine's z with the float64 oracle's g, and vice versa -> which input
    Written to gpurun_out/parity/last_bn_decomposition.json; asserts only that the pieces are finite and that (a) stays within 2x of (b) or the
    floor, i.e. that the BatchNorm kernels themselves add nothing beyond what their inputs carry."""
    name = "encoder.layer4.2.bn2.weight"
    f64 = lambda t: t.detach().double().cpu()
    nhwc = lambda t: f64(t).permute(0, 2, 3, 1).reshape(-1, t.shape[1])                 # oracle tensors are NCHW
    C = taps["z2"].shape[-1]
    z_gpu, g_gpu_in, dout_gpu, out_gpu = (f64(taps[k]).reshape(-1, C) for k in ("z2", "g", "dout", "out"))
    dout64, dout32 = nhwc(x64.grad), nhwc(x32.grad)
    mask64 = nhwc(x64) > 0
    g_64 = dout64 * mask64

    def dgamma(g, z):
        mu, var = z.mean(0), z.var(0, unbiased=False)
        return (g * ((z - mu) / torch.sqrt(var + 1e-5))).sum(0)
    # the oracle's z is not recorded; its d gamma IS: g64[name].  (b) and (c) need a float64 z: the engine's z is within 1e-6 of it (forward parity)
    truth = f64(g64[name])
    rel = lambda a: float((a - truth).norm() / truth.norm())
    doc = {
        "tensor": name,
        "a_engine_dgamma": rel(f64(g_gpu[name])),
        "cpu_fp32_dgamma": rel(f64(g32[name])),
        "b_fp64_formula_on_engine_g_and_engine_z": rel(dgamma(g_gpu_in, z_gpu)),
        "c_fp64_formula_on_fp64_g_and_engine_z": rel(dgamma(g_64, z_gpu)),
        "incoming_gradient_rel_l2": {"engine": float((dout_gpu - dout64).norm() / dout64.norm()), "cpu_fp32": float((dout32 - dout64).norm() / dout64.norm())},
        "relu_mask_flips_vs_fp64": {"engine": int(((out_gpu > 0) != mask64).sum()), "cpu_fp32": int(((nhwc(x32) > 0) != mask64).sum()), "elements": int(mask64.numel())},
        "share_of_squared_error_in_the_3_worst_channels": {"engine": _top_share(f64(g_gpu[name]) - truth), "cpu_fp32": _top_share(f64(g32[name]) - truth)},
        "flipped_elements": {"engine": _flips(out_gpu > 0, mask64, dout64, z_gpu, truth), "cpu_fp32": _flips(nhwc(x32) > 0, mask64, dout64, z_gpu, truth)},
        "amplification_sum_abs_over_abs_sum": float(((g_64.abs() * ((z_gpu - z_gpu.mean(0)) / z_gpu.std(0, unbiased=False)).abs()).sum(0) / truth.abs().clamp_min(1e-30)).median()),
    }
    print("\n[last BatchNorm decomposition] %s" % json.dumps(doc, indent=1))
    out_dir = os.environ.get("FP_PARITY_DUMP", os.path.join(ROOT, "gpurun_out", "parity"))
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "last_bn_decomposition_%s.json" % tag), "w") as fh:
            json.dump(doc, fh, indent=1)
    except OSError:
        pass
    assert all(np.isfinite(v) for v in (doc["a_engine_dgamma"], doc["b_fp64_formula_on_engine_g_and_engine_z"], doc["c_fp64_formula_on_fp64_g_and_engine_z"]))
    assert doc["a_engine_dgamma"] <= max(2.0 * doc["b_fp64_formula_on_engine_g_and_engine_z"], 2e-5), doc


FORMATS = ("exact", "fp16_pair")
_ORACLE_CACHE = {}          # one case at a time: both operand formats of a case share its float64 / float32 oracle runs


def _oracle_runs(key, build):
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE.clear()
        _ORACLE_CACHE[key] = build()
    return _ORACLE_CACHE[key]


def _table_tag(tag, fmt):
    return tag if fmt == "exact" else tag + "_fp16_pair"


def _decision_rule(tag, P, B, cpu_batch, res, g32, g64, bad, rows, dec64=None, spread=(), max_err=FORCED_MAX_ERR, median_err=FORCED_MEDIAN_ERR, oracle=None):
    """the second half of the gradient rule (tests/parity.py decision_forced_report), run for EVERY case: the engine's gradients against the
    float64 oracle under the engine's own ReLU decisions -- the per-tensor rule once more (a tensor that failed the single-run rule has to pass
    here), absolute bounds on the errors, and the median of err(GPU forced) / err(CPU fp32).  Returns (what goes into the table, that median)."""
    bad_f, rows_f, _, dstats = decision_forced_report(P, B, cpu_batch, res["decisions"], res["grads"], g32, g64, spread=spread)
    errs = sorted((eg for _, _, eg, _ in rows_f), reverse=True)
    med_f = float(np.median([r for r, *_ in rows_f]))
    extra = {"single_run_rule_failures": [b.split(" ")[0] for b in bad],
             "decision_forced": {"failures": [b.split(" ")[0] for b in bad_f], "max_err": float("%.3e" % errs[0]),
                                 "median_err": float("%.3e" % errs[len(errs) // 2]), "median_ratio": round(med_f, 4),
                                 "worst": [{"tensor": n, "err_gpu": float("%.3e" % eg), "err_cpu32": float("%.3e" % ec)} for _, n, eg, ec in
                                           sorted(rows_f, key=lambda r: -r[2])[:4]]}}
    extra["imposed_decisions"] = {k: (float("%.3e" % v) if isinstance(v, float) else v) for k, v in dstats.items()}
    if dec64 is not None:
        flips, total = count_decision_flips(res["decisions"], dec64)
        extra["relu_decisions_differing_from_float64"] = {"engine": flips, "of": total}
    # round 6: the imposed decisions are bounded BEFORE anything is concluded from the forced truth -- few, each a float64 round-off tie, and no
    # further from the boundary than the reference's own fp32 arithmetic gets (its decisions imposed the same way; once per case, both formats)
    ref = None
    if oracle is not None and (oracle.get("flip32") is not None or oracle.get("dec32") is not None):
        if oracle.get("flip32") is None:
            oracle["flip32"] = reference_flip_stats(P, B, cpu_batch, oracle["dec32"])
        ref = oracle["flip32"]
        extra["cpu_fp32_decisions_vs_float64"] = {k: (float("%.3e" % v) if isinstance(v, float) else v) for k, v in ref.items()}
        print("[%s] decisions differing from float64: engine %d (worst %.2e x RMS), CPU fp32 %d (worst %.2e)" % (
            tag, dstats["relu_flips"], dstats["relu_flip_worst_distance"], ref["relu_flips"], ref["relu_flip_worst_distance"]))
    assert_decisions_at_roundoff(dstats, tag, reference=ref)
    print("\n[%s] single-run rule failures %d; against float64 under the engine's ReLU decisions: failures %d, max err %.2e, median %.2e, median ratio %.2f %s" % (
        tag, len(bad), len(bad_f), errs[0], errs[len(errs) // 2], med_f, extra.get("relu_decisions_differing_from_float64", "")))
    assert not bad_f, "gradients that differ from float64 by more than ReLU decisions at round-off distance from zero explain: %s" % bad_f[:10]
    assert errs[0] <= max_err and errs[len(errs) // 2] <= median_err, (
        "arithmetic error against float64 under the engine's own decisions: max %.2e (bound %.1e), median %.2e (bound %.1e): %s" % (
            errs[0], max_err, errs[len(errs) // 2], median_err, extra["decision_forced"]["worst"]))
    return extra, med_f


@pytest.mark.parametrize("Bn,Hn,Wn,fmt", [(b, h, w, f) for (b, h, w) in ((1, 256, 448), (1, 512, 640), (4, 512, 640), (12, 192, 640)) for f in FORMATS])
def test_train_step_fp64_anchored(Bn, Hn, Wn, fmt):
    from oracle import restatement as R
    decompose = (Bn, Hn, Wn) == (12, 192, 640)          # the case whose last encoder BatchNorm sits at 9-33x the CPU path's error: see _last_bn_decomposition

    def build():
        P, B = R.make_state(tag="anch")
        cpu_batch = R.make_batch(Bn, Hn, Wn, tag="anch%d" % Hn)
        removed = []

        def fix(batch, out64):
            b, n = tie_free_batch(batch, out64)
            removed.append(n)
            return b
        rec64, rec32 = ([] if decompose else None), ([] if decompose else None)
        # the fp32 forward's decisions first (they do not depend on the targets): the float64 run measures them as a probe -- the
        # reference-arithmetic anchor of the flip bound without a float64 run of its own (tests/parity.py fp32_forward_decisions)
        dec64 = R.ReluDecisions(probe=fp32_forward_decisions(P, B, cpu_batch))
        out64, l64, g64, _, cpu_batch = oracle_grads(P, B, cpu_batch, torch.float64, fix_batch=fix, record=rec64, relu_decisions=dec64)   # float64 first: it defines the tie pixels
        out32, l32, g32, tr32, _ = oracle_grads(P, B, cpu_batch, torch.float32, record=rec32)
        l64 = {k: float(v) for k, v in l64.items()}
        l32 = {k: float(v) for k, v in l32.items()}
        return dict(P=P, B=B, cpu_batch=cpu_batch, removed=removed[0], out64=out64, l64=l64, g64=g64, out32=out32, l32=l32, g32=g32,
                    bn32={k: v.clone() for k, v in tr32.B.items()}, dec64=dec64.taken, flip32=probe_flip_stats(dec64),
                    x64=rec64[-1] if decompose else None, x32=rec32[-1] if decompose else None)
    o = _oracle_runs(("train_step", Bn, Hn, Wn), build)
    P, B, cpu_batch, out64, out32, l64, l32, g64, g32 = (o[k] for k in ("P", "B", "cpu_batch", "out64", "out32", "l64", "l32", "g64", "g32"))
    res = gpu_step(P, B, cpu_batch, fmt, tap_block=15 if decompose else None)      # encoder.layer4.2 = the 16th BasicBlock of ResNet-34
    out, losses, g_gpu = res["out"], res["losses"], res["grads"]
    case = "%dx%dx%d %s" % (Bn, Hn, Wn, fmt)
    if decompose:
        _last_bn_decomposition(res["taps"], o["x64"], o["x32"], g_gpu, g32, g64, fmt)
    print("\n[%s] |.|-kink pixels removed from the depth masks: %d of %d" % (case, o["removed"], 2 * Bn * Hn * Wn))
    assert o["removed"] <= KINK_MAX_FRACTION * 2 * Bn * Hn * Wn, "the tie band removed %d of %d depth-target pixels" % (o["removed"], 2 * Bn * Hn * Wn)
    # ---- outputs: per channel, against the reference's fp32 CPU arithmetic and against the float64 truth ---------------
    for k in R.SCALES:
        e32, e64 = chan_relerr(out[k], out32[k]), chan_relerr(out[k], out64[k])
        assert max(e32) <= 1e-4 and max(e64) <= 1e-4, "output %s per-channel rel err vs fp32 %s vs fp64 %s" % (k, e32, e64)
        ref = out32[k]
        for thr in (0.0, 0.5):          # sigmoid(logit) > 0.5 (losses.py:78) and predict_simple's logit > 0.5 (predict_simple.py:77)
            band = (ref[:, :2] - thr).abs() < 1e-4 * ref[:, :2].abs().max()
            assert torch.equal((out[k][:, :2].cpu() > thr) | band, (ref[:, :2] > thr) | band), "mask bits (%s, thr %.1f)" % (k, thr)
    # ---- 21 losses -----------------------------------------------------------------------------------------------------
    for key in R.LOSS_KEYS:
        ref = float(l32[key])
        assert abs(float(losses[key]) - ref) <= 1e-4 * max(abs(ref), 1e-3), (key, float(losses[key]), ref)
        assert abs(float(losses[key]) - float(l64[key])) <= 1e-4 * max(abs(float(l64[key])), 1e-3), key
    # ---- every parameter gradient, fp64-anchored -------------------------------------------------------------------------
    bad, rows = anchored_report(g_gpu, g32, g64)
    print("\n[%s] worst GPU/CPU32 error ratios (vs fp64): %s" % (case, ["%s %.2f (gpu %.1e cpu %.1e)" % (n, r, eg, ec) for r, n, eg, ec in rows[:6]]))
    extra = {"kink_pixels_removed": o["removed"], "operand_format": fmt}
    # a tensor outside the single-run bound must be inside it once the float64 truth takes the engine's own ReLU decisions; the headline
    # case reports that evaluation for every tensor, whether or not one failed
    # the decision-forced evaluation costs one more float64 oracle run: always for the headline case (both formats) and the two batch-1 sizes; at
    # 4x512x640 (measured round 5: 0 failures, max 2.9e-5 / 1.9e-5) only when the plain rule or the median gate needs it -- the driver's suite time
    med_plain = float(np.median([r for r, *_ in rows]))
    if (Bn, Hn, Wn) != (4, 512, 640) or bad or not (MEDIAN_GATE[0] <= med_plain <= MEDIAN_GATE[1]):
        forced, med_f = _decision_rule(case, P, B, cpu_batch, res, g32, g64, bad, rows, dec64=o["dec64"], oracle=o)
        extra.update(forced)
    else:
        med_f = med_plain
    med = _keep_ratio_table(_table_tag("train_step_%dx%dx%d" % (Bn, Hn, Wn), fmt), rows, extra)
    print("median ratio %.2f (%.2f under the engine's ReLU decisions), tensors %d" % (med, med_f, len(rows)))
    # measured spread of the median over the five cases and both operand formats (profiles/round3_parity_ratios.md): 0.72 .. 1.34
    # (the lower end is not a defect -- the split-operand kernels are MORE accurate than an fp32 accumulation chain -- it is there so
    # that a change of the distribution in either direction gets looked at).  Round 5: the upper gate holds for the smaller of the two
    # medians -- 1.56 against the plain float64 run at 12x192x640 (exact operands, 89 decisions of 1e8 differ) is a draw of the ReLU lottery
    # when the same gradients sit at the CPU path's own distance from float64 once the decisions are imposed (tests/parity.py)
    assert MEDIAN_GATE[0] <= med and min(med, med_f) <= MEDIAN_GATE[1], (
        "median err(GPU) / err(CPU fp32) = %.3f (%.3f under the engine's decisions): the engine's arithmetic drifted away from fp32-equivalent" % (med, med_f))
    # ---- BatchNorm running statistics after the step (train-mode side effect, network.py:40-44 / nn.BatchNorm2d) ----------
    sd = res["state"]
    for k, v in o["bn32"].items():
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v), k
        elif "encoder" in k:
            assert rel_l2(sd[k], v) <= 1e-5, k


def _natural_batch(Bn, Hn, Wn, seed=31):
    """images with the statistics the per-tensor operand scale has to survive on real photographs: low-pass content (box-filtered
    noise at three octaves), saturated regions (blown-out sky / deep shadow blocks clipped to exactly 1 and 0) and a few isolated
    specular pixels; label maps as in make_batch but spatially coherent (block-wise masks)"""
    from oracle import restatement as R
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    img = torch.zeros(Bn, 3, Hn, Wn)
    for k, wgt in ((32, 0.6), (8, 0.3), (2, 0.1)):
        low = torch.rand(Bn, 3, Hn // k, Wn // k, generator=g)
        img += wgt * F.interpolate(low, size=(Hn, Wn), mode="bilinear", align_corners=False)
    img = ((img - 0.5) * 2.2 + 0.5).clamp(0, 1)                                  # contrast stretch: a good part saturates
    blocks = F.interpolate(torch.rand(Bn, 1, Hn // 32, Wn // 32, generator=g), size=(Hn, Wn), mode="nearest")
    img = torch.where(blocks > 0.85, torch.ones_like(img), img)                  # blown-out regions
    img = torch.where(blocks < 0.10, torch.zeros_like(img), img)                 # crushed shadows
    spec = torch.rand(Bn, 1, Hn, Wn, generator=g) > 0.9995
    img = torch.where(spec, torch.ones_like(img), img)                           # isolated specular highlights
    batch = R.make_batch(Bn, Hn, Wn, tag="natural%d" % Hn)
    batch["image"] = img.contiguous()
    coarse = lambda p: (F.interpolate(torch.rand(Bn, 1, Hn // 16, Wn // 16, generator=g), size=(Hn, Wn), mode="nearest")[:, 0] < p).float()
    batch["visible_ground"] = coarse(0.4)
    batch["moving_object_mask"] = coarse(0.05)
    batch["depth_mask"] = coarse(0.1)
    batch["all_ground"] = ((batch["ground_depth"] + batch["visible_ground"]) > 0).float()
    return batch


def _wide_range_state(tag="natural"):
    """make_state with the encoder's BatchNorm affine parameters spread over >= 2^15: gamma_c = +-2^u, u uniform in [-8, 8] per
    channel (beta likewise, smaller), so that every activation tensor holds channels whose magnitudes differ by five orders of
    magnitude -- the case in which a per-TENSOR fp16 scale leaves the small channels the fewest mantissa bits"""
    from oracle import restatement as R
    P, B = R.make_state(tag=tag)
    g = torch.Generator().manual_seed(77)
    for k in list(P.keys()):
        if k.startswith("encoder") and P[k].dim() == 1 and (".bn" in k or ".downsample.1" in k or k.startswith("encoder.layer0.1")):
            n = P[k].numel()
            mag = torch.pow(2.0, torch.rand(n, generator=g) * 16.0 - 8.0)
            sign = torch.where(torch.rand(n, generator=g) < 0.5, -1.0, 1.0)
            P[k] = (mag * sign * (1.0 if k.endswith("weight") else 0.25)).to(P[k].dtype)
    return P, B


@pytest.mark.parametrize("fmt", FORMATS)
def test_natural_statistics_wide_dynamic_range_fp64_anchored(fmt):
    """VERDICT r2 parity item 1 / Next 2c: every other parity input is uniform noise with O(1) BatchNorm parameters.  Here the images are
    low-pass with saturated regions and specular pixels, and the encoder's BatchNorm gammas / betas span 2^16 per tensor, so every
    conv operand tensor of the encoder and the decoder skip inputs mix channels five orders of magnitude apart -- the worst case
    for the per-tensor power-of-two scale of the fp16-pair operands (network.py:21-30 forward, training/losses.py:95-107 through
    the backward).  Checked: the fp64-anchored rule on every gradient, the 1e-4 per-channel bar on all outputs, AND an
    elementwise bound on the full-resolution outputs (|gpu - fp64| <= 1e-4 * channel max at every pixel is the per-channel bar;
    on top, the fraction of pixels whose error exceeds 10x the fp32 CPU path's own worst error must be zero)."""
    from oracle import restatement as R
    Bn, Hn, Wn = 4, 192, 640
    def build():
        P, B = _wide_range_state()
        gam = torch.cat([v.abs().flatten() for k, v in P.items() if k.startswith("encoder") and v.dim() == 1 and k.endswith("weight")])
        assert float(gam.max() / gam.min()) >= 2.0 ** 15
        cpu_batch = _natural_batch(Bn, Hn, Wn)
        sat = float(((cpu_batch["image"] == 0) | (cpu_batch["image"] == 1)).float().mean())
        assert sat > 0.1, "the synthetic 'photograph' should have saturated regions (%.3f)" % sat
        removed = []

        def fix(batch, out64):
            b, n = tie_free_batch(batch, out64, tie_sigma=TIE_SIGMA)          # the only case with the output-tolerance band (tests/parity.py)
            removed.append(n)
            return b
        out64, l64, g64, _, cpu_batch = oracle_grads(P, B, cpu_batch, torch.float64, fix_batch=fix)
        assert removed[0] <= KINK_MAX_FRACTION_NATURAL * 2 * Bn * Hn * Wn, "the tie band removed %d of %d depth-target pixels" % (removed[0], 2 * Bn * Hn * Wn)
        out32, l32, g32, _, _ = oracle_grads(P, B, cpu_batch, torch.float32)
        # four more fp32 CPU runs on images perturbed by 1e-6 (relative) -- the level at which the fp32 implementations' own features sit
        # from the float64 ones on this input (f1 .. f4: 7e-7 .. 6e-6, scripts/debug_parity_stage.py): how far conforming fp32
        # implementations scatter here
        spread = [oracle_grads(P, B, cpu_batch, torch.float32, perturb=1e-6, seed=k)[2] for k in range(4)]
        return dict(P=P, B=B, cpu_batch=cpu_batch, removed=removed, sat=sat, gam=gam, out64=out64, l64={k: float(v) for k, v in l64.items()}, g64=g64,
                    out32=out32, g32=g32, spread=spread)
    o = _oracle_runs(("natural", Bn, Hn, Wn), build)
    P, B, cpu_batch, removed, sat, gam, out64, l64, g64, out32, g32, spread = (o[k] for k in (
        "P", "B", "cpu_batch", "removed", "sat", "gam", "out64", "l64", "g64", "out32", "g32", "spread"))
    res = gpu_step(P, B, cpu_batch, fmt)
    out, losses, g_gpu = res["out"], res["losses"], res["grads"]
    for k in R.SCALES:
        e32, e64 = chan_relerr(out[k], out32[k]), chan_relerr(out[k], out64[k])
        assert max(e32) <= 1e-4 and max(e64) <= 1e-4, "output %s per-channel rel err vs fp32 %s vs fp64 %s" % (k, e32, e64)
    # elementwise on the 1/1 outputs: no pixel of the engine may sit further from the float64 truth than 10x the WORST pixel of
    # the reference's own fp32 arithmetic (per channel) -- sparse outliers cannot hide in a norm
    o, r32, r64 = out["1/1"].double().cpu(), out32["1/1"].double(), out64["1/1"].double()
    worst32 = (r32 - r64).abs().amax(dim=(0, 2, 3))
    chmax = r64.abs().amax(dim=(0, 2, 3))
    egpu = (o - r64).abs()
    bound = torch.maximum(10.0 * worst32, 2e-6 * chmax).view(1, 4, 1, 1)
    nbad = int((egpu > bound).sum())
    assert nbad == 0, "%d pixels of the 1/1 outputs exceed 10x the fp32 CPU path's worst pixel error (per channel gpu max %s, cpu32 max %s)" % (
        nbad, egpu.amax(dim=(0, 2, 3)).tolist(), worst32.tolist())
    for key in R.LOSS_KEYS:
        assert abs(float(losses[key]) - float(l64[key])) <= 1e-4 * max(abs(float(l64[key])), 1e-3), key
    bad1, rows1 = anchored_report(g_gpu, g32, g64)                       # against the single unperturbed fp32 run: kept for the table
    bad, rows = anchored_report(g_gpu, g32, g64, spread=spread)
    _keep_ratio_table(_table_tag("natural_wide_range_single_fp32_run_%dx%dx%d" % (Bn, Hn, Wn), fmt), rows1,
                      {"failures_under_single_run_rule": len(bad1), "operand_format": fmt})
    # (this case's BatchNorm gammas span 2^16: its absolute bounds are its own, tests/parity.py FORCED_*_NATURAL)
    # (decision-forced evaluation: for the default format always, for the other one when a tensor sits outside the spread-of-five bound)
    extra = {}
    if fmt == "exact" or bad:
        extra, _ = _decision_rule("natural %s" % fmt, P, B, cpu_batch, res, g32, g64, bad, rows, spread=spread, max_err=FORCED_MAX_ERR_NATURAL,
                                  median_err=FORCED_MEDIAN_ERR_NATURAL)
    med = _keep_ratio_table(_table_tag("natural_wide_range_%dx%dx%d" % (Bn, Hn, Wn), fmt), rows,
                            {**extra, "operand_format": fmt, "kink_pixels_removed": removed[0], "saturated_fraction": round(sat, 4),
                             "gamma_dynamic_range_log2": round(float(torch.log2(gam.max() / gam.min())), 2),
                             "out_1_1_worst_pixel_err_gpu": [float("%.3e" % v) for v in egpu.amax(dim=(0, 2, 3)).tolist()],
                             "out_1_1_worst_pixel_err_cpu32": [float("%.3e" % v) for v in worst32.tolist()]})
    print("\n[natural / wide range] worst ratios:", ["%s %.2f (gpu %.1e cpu %.1e)" % (n, r, eg, ec) for r, n, eg, ec in rows[:6]], "median %.2f" % med)
    # (a tensor outside the spread-of-five bound was held to the decision-forced evaluation above: _decision_rule asserts it)
    # the single-run rule on this case is a lottery over ReLU decisions in EITHER format (profiles/round4_parity_natural_seeds.md: 6 of 8
    # image seeds fail it with fp16 pairs AND with the exact split); its failure count is recorded in the table, not asserted
    assert MEDIAN_GATE_NATURAL[0] <= med <= MEDIAN_GATE_NATURAL[1], med


def test_g5_gradients_and_adam_state_fp64_anchored():
    """The G5 fixture's inputs (2x64x96: 12 BatchNorm samples per channel at layer4 -- ill-conditioned in fp32) through two
    drop-in steps; instead of loose constants the per-tensor gradients AND Adam's exp_avg / exp_avg_sq after two steps are held
    to the fp64-anchored rule against the oracle run in float64."""
    from footprints_amd.model_manager import ModelManager
    from footprints_amd.training.losses import LossManager
    from oracle import restatement as R
    P, B = R.make_state()
    batches = [R.make_batch(2, 64, 96, tag="g5.step%d" % s) for s in range(2)]
    mm = ModelManager(use_cuda=True, learning_rate=1e-4)
    mm.model.load_state_dict({**P, **B})
    mm.model.train()
    lm = LossManager((0.1, 100), 0.25, compute_viz=False)
    trs = {}
    for dt in (torch.float32, torch.float64):
        trs[dt] = R.OracleTrainer(OrderedDict((k, v.to(dt)) for k, v in P.items()),
                                  OrderedDict((k, v.to(dt) if v.is_floating_point() else v.clone()) for k, v in B.items()))
    names = [k for k, _ in mm.model.named_parameters()]
    for s, cb in enumerate(batches):
        batch = {k: v.cuda() for k, v in cb.items()}
        out = mm.model(batch["image"])
        losses = lm(out, batch)
        mm.model.zero_grad()
        losses["loss"].backward()
        g_gpu = OrderedDict((n, None if p.grad is None else p.grad.clone()) for n, p in mm.model.named_parameters())
        mm.optimiser.step()
        gs = {}
        for dt, tr in trs.items():
            tr.step(OrderedDict((k, v.to(dt)) for k, v in cb.items()))
            gs[dt] = OrderedDict((k, p.grad) for k, p in tr.P.items())
        if s == 0:          # the second step's gradients depend on sign-like first Adam updates of round-off-level gradients
            bad, rows = anchored_report(g_gpu, gs[torch.float32], gs[torch.float64])
            print("\nG5 step 0 worst ratios:", ["%s %.2f" % (n, r) for r, n, *_ in rows[:5]])
            assert not bad, bad[:10]
    st = mm.optimiser.state_dict()["state"]
    for key in ("exp_avg", "exp_avg_sq"):
        gpu, c32, c64 = {}, {}, {}
        for i, n in enumerate(names):
            p32, p64 = trs[torch.float32].P[n], trs[torch.float64].P[n]
            s32, s64 = trs[torch.float32].opt.state.get(p32), trs[torch.float64].opt.state.get(p64)
            if not s64:
                assert i not in st, n
                c64[n] = None
                continue
            gpu[n], c32[n], c64[n] = st[i][key], s32[key], s64[key]
        bad, rows = anchored_report(gpu, c32, c64, floor=1e-4)      # two steps deep: first-step sign flips of ~0 gradients
        print("G5 Adam %s worst ratios:" % key, ["%s %.2f (gpu %.1e cpu %.1e)" % (n, r, eg, ec) for r, n, eg, ec in rows[:4]])
        assert not bad, (key, bad[:10])


@pytest.mark.parametrize("Bn,Hn,Wn", [(12, 192, 640), (4, 512, 640)])
def test_unedited_batch_through_the_loss_gradient_fullsize(Bn, Hn, Wn):
    """VERDICT r5 weak point 3: the gradient cases above run on a batch whose |.|-kink pixels were taken out of the depth masks
    (tests/parity.py tie_free_batch), so no full-size test fed the kernels the UNEDITED BASELINE batch through a gradient comparison.  The kink
    lives in ONE kernel -- the loss (training/losses.py:95-107: log(|d - depth| + 1)); everything behind it is linear in d loss / d outputs,
    which the edited-batch cases hold to float64.  So here: the engine's own forward on the unedited batch, its fused loss kernel's 21 losses
    and d loss / d outputs (16 planes of B x H x W) against the float64 loss evaluated on the SAME outputs --
      * losses: 1e-6 relative (same inputs, no network in between);
      * every gradient element off the kink: 1e-5 of the plane's largest magnitude;
      * every gradient element ON the kink (|d - depth| within the tie band of tests/parity.py): the float64 magnitude with either sign --
        the one thing an fp32 implementation is free to decide there;
      * the kink pixels are few (<= KINK_MAX_FRACTION), so the free signs cannot hide anything else."""
    from footprints_amd import FootprintNetwork, ops
    from oracle import restatement as R
    from tests.parity import TIE
    P, B = R.make_state(tag="anch")
    cpu_batch = R.make_batch(Bn, Hn, Wn, tag="anch%d" % Hn)                       # the UNEDITED batch of the fp64-anchored cases
    model = FootprintNetwork(pretrained=False)
    model.load_state_dict({**P, **B})
    model.cuda().train()
    batch = {k: v.cuda() for k, v in cpu_batch.items()}
    with torch.no_grad():
        out = model(batch["image"])
    preds = [out[k].contiguous() for k in R.SCALES]
    losses = torch.zeros(21, device="cuda")
    dpreds = [torch.empty_like(p) for p in preds]
    ops.loss_fwd_bwd(preds, batch, losses, dpreds, (0.1, 100.0), 0.25)
    torch.cuda.synchronize()
    # float64 loss on the engine's outputs
    o64 = OrderedDict((k, p.detach().double().cpu().requires_grad_(True)) for k, p in zip(R.SCALES, preds))
    l64, _ = R.loss_manager(o64, OrderedDict((k, v.double()) for k, v in cpu_batch.items()))
    l64["loss"].backward()
    for i, key in enumerate(R.LOSS_KEYS):
        assert abs(float(losses[i]) - float(l64[key])) <= 1e-6 * max(abs(float(l64[key])), 1e-3), (key, float(losses[i]), float(l64[key]))
    lo, hi = 1.0 / 100.0, 1.0 / 0.1
    kink_total, worst = 0, 0.0
    for k, dp in zip(R.SCALES, dpreds):
        g64 = o64[k].grad
        got = dp.double().cpu()
        for ch, tgt in ((2, "depth"), (3, "ground_depth")):
            d = 1.0 / (lo + (hi - lo) * torch.sigmoid(o64[k].detach()[:, ch]))
            t = cpu_batch[tgt].double()
            kink = ((d - t).abs() <= TIE * t.clamp_min(1.0)) & (t > 0)
            kink_total += int(kink.sum())
            scale = float(g64[:, ch].abs().max())
            off = (got[:, ch] - g64[:, ch]).abs()
            assert float(off[~kink].max()) <= 1e-5 * scale, "depth-loss gradient off the kink (%s, channel %d): %.3e of %.3e" % (k, ch, float(off[~kink].max()), scale)
            if bool(kink.any()):                                              # on the kink: the same magnitude, either sign
                mag = (got[:, ch].abs() - g64[:, ch].abs()).abs()[kink]
                assert float(mag.max()) <= 1e-4 * scale, (k, ch, float(mag.max()), scale)
            worst = max(worst, float(off[~kink].max()) / max(scale, 1e-30))
        for ch in (0, 1):                                                     # the two BCE channels have no kink
            scale = float(g64[:, ch].abs().max())
            e = float((got[:, ch] - g64[:, ch]).abs().max())
            assert e <= 1e-5 * scale, "BCE gradient (%s, channel %d): %.3e of %.3e" % (k, ch, e, scale)
            worst = max(worst, e / max(scale, 1e-30))
    total = 8 * Bn * Hn * Wn                                                  # two depth channels x four scales
    print("\n[unedited batch %dx%dx%d] kink pixels %d of %d (%.4f %%), worst off-kink gradient error %.2e of the plane's maximum" % (
        Bn, Hn, Wn, kink_total, total, 100.0 * kink_total / total, worst))
    assert kink_total <= KINK_MAX_FRACTION * total
