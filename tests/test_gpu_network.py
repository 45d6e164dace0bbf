"""GPU: the HIP engine end to end (through the nn.Module / LossManager / optimiser surface and the C ABI)
against (a) the committed golden fixtures produced by the reference's own code and (b) the CPU oracle.

Tolerances: 1e-4 relative to the tensor's max for activations / outputs / losses (north_star), golden digests of decoder-side
gradients 1e-3; thresholded masks bit-exact outside a documented |logit - threshold| < 1e-4 tie band.  Encoder-side gradients are
ill-conditioned in fp32 on small inputs (the G5 fixture has 12 BatchNorm samples per channel at layer4: the reference's own CPU
arithmetic in fp32 vs fp64 differs by percents there), so they are NOT held to constants: every parameter gradient is held to the
fp64-anchored rule of tests/parity.py (err(GPU) <= 4 x max(err(CPU fp32), stage median) against the oracle in float64) -- here at
2x96x128 and 2x192x640, in tests/test_gpu_parity_fullsize.py at the benchmark workloads and for G5's gradients and Adam moments.
"""
from collections import OrderedDict

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _load_state(model, P, B):
    model.load_state_dict({**P, **B})
    return model


def _new_model(P, B):
    from footprints_amd import FootprintNetwork
    return _load_state(FootprintNetwork(pretrained=False), P, B).cuda()


def relerr(got, ref):
    ref = ref.double().cpu()
    return ((got.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def _g2_diagnose(eng, dec, S, D, P, decname, sig, feats_nchw):
    """failure diagnostics of test_g2: every saved activation of the engine's decoder forward against the oracle's block functions on the CPU,
    and the same forward once more in this process (a transient first-touch problem shows as a correct second pass)"""
    import torch.nn.functional as F
    from oracle import restatement as R
    Pd = {k: v.detach() for k, v in P.items()}
    x = feats_nchw[4]
    print("\n[g2 diagnose] decoder %s" % decname)
    for bi in range(4):
        pre, post = "%s.block%d.pre_concat_conv" % (decname, bi + 1), "%s.block%d.post_concat_conv" % (decname, bi + 1)
        r1 = F.elu(F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), Pd[pre + ".conv1.weight"], Pd[pre + ".conv1.bias"]))
        r2 = F.elu(F.conv2d(F.pad(r1, (1, 1, 1, 1), mode="reflect"), Pd[pre + ".conv2.weight"], Pd[pre + ".conv2.bias"]))
        cat = torch.cat([F.interpolate(r2, scale_factor=2, mode="nearest"), feats_nchw[3 - bi]], 1)
        r3 = F.elu(F.conv2d(F.pad(cat, (1, 1, 1, 1), mode="reflect"), Pd[post + ".conv1.weight"], Pd[post + ".conv1.bias"]))
        r4 = F.elu(F.conv2d(F.pad(r3, (1, 1, 1, 1), mode="reflect"), Pd[post + ".conv2.weight"], Pd[post + ".conv2.bias"]))
        y1, y2, y3 = D["y"][bi]
        errs = [relerr(nchw(a), b) for a, b in ((y1, r1), (y2, r2), (y3, r3), (D["x"][bi], r4))]
        print("  block%d: pre1 %.2e pre2 %.2e post1 %.2e post2 %.2e  nan: %s" % (bi + 1, *errs, [bool(torch.isnan(t).any()) for t in (y1, y2, y3, D["x"][bi])]))
        x = r4
    outs2 = [torch.zeros((2, 4, 64, 96), device="cuda") for _ in range(4)]
    D2 = {}
    for _ in eng._decoder_forward(dec, S, outs2, D2):
        pass
    print("  second pass in the same process: y1 of block1 equal to the first pass: %s; max |diff| of the four outputs vs first pass: see below" %
          bool(torch.equal(D2["y"][0][0], D["y"][0][0])))
    print("  lazily packed fp32 layouts: used %d, fresh %d" % (len(eng._w32_used), len(eng._w32_fresh)))


# ----------------------------------------------------------------------------------------------------------
def test_g2_decoder_golden_through_engine():
    """SkipDecoder fwd + bwd (reference-authored golden vectors, real channel counts, 64x96 pyramid)."""
    from footprints_amd import FootprintNetwork
    from tests.golden.digest import compare, fill, load
    from tests.test_oracle_golden import G2_SHAPES, decoder_state
    gold = load("g2_decoder")
    for sig, decname in ((False, "mask_decoder"), (True, "depth_decoder")):
        tag = "dec.%s" % ("sig" if sig else "lin")
        model = FootprintNetwork(pretrained=False)
        P = decoder_state(prefix=decname)
        sd = model.state_dict()
        sd.update({k: v.detach() for k, v in P.items()})
        model.load_state_dict(sd)
        model.cuda()
        eng = model.engine()
        eng.refresh_packed(force=True)
        dec = eng.decoders[1 if sig else 0]
        feats_nchw = [fill("g2.feat%d" % i, s) for i, s in enumerate(G2_SHAPES)]
        feats = [nhwc(f) for f in feats_nchw]
        S = {"N": 2, "H": 64, "W": 96, "feats": feats, "dims": [tuple(f.shape[1:3]) for f in feats], "training": True}
        outs = [torch.zeros((2, 4, 64, 96), device="cuda") for _ in range(4)]
        D = {}
        for _ in eng._decoder_forward(dec, S, outs, D):       # generators: sections are interleaved across streams in Engine.forward
            pass
        c0 = dec.c0
        try:
            for k, o in zip(("1/8", "1/4", "1/2", "1/1"), outs):
                compare(gold, tag + ".out" + k, o[:, c0:c0 + 2])
        except AssertionError:
            _g2_diagnose(eng, dec, S, D, P, decname, sig, feats_nchw)      # prints where the forward left the oracle, then re-raises
            raise
        gouts = []
        for k in ("1/8", "1/4", "1/2", "1/1"):
            g = torch.zeros((2, 4, 64, 96), device="cuda")
            g[:, c0:c0 + 2] = fill("g2.g" + k, (2, 2, 64, 96)).cuda()
            gouts.append(g)
        dF = [torch.empty_like(f) for f in feats]
        for _ in eng._decoder_backward(dec, D, S, gouts, dF, first=True, acc=False):
            pass
        for i in range(5):
            compare(gold, tag + ".dfeat%d" % i, nchw(dF[i]))
        gv = dict(zip(eng.live_names, eng.grad_views))
        for name in ("block1.pre_concat_conv.conv1.weight", "block4.post_concat_conv.conv2.weight", "outconv1.conv1.weight",
                     "outconv4.0.conv1.weight", "outconv4.1.conv1.bias", "block2.post_concat_conv.conv1.bias"):
            compare(gold, tag + ".d." + name, gv[decname + "." + name])


def test_g3_network_forward_train_and_eval_golden():
    from oracle import restatement as R
    from tests.golden.digest import compare, load
    gold = load("g3_network")
    P, B = R.make_state()
    batch = R.make_batch(2, 64, 96)
    for mode in ("train", "eval"):
        model = _new_model(P, B)
        model.train(mode == "train")
        with torch.no_grad():
            o = model(batch["image"].cuda())
        assert list(o.keys()) == ["1/8", "1/4", "1/2", "1/1"]
        for k in o:
            assert tuple(o[k].shape) == (2, 4, 64, 96)
            compare(gold, "net.%s.out%s" % (mode, k), o[k])
        if mode == "train":
            sd = model.state_dict()
            rm = torch.cat([sd[k].flatten() for k in sd if k.endswith("running_mean") and "encoder" in k])
            rv = torch.cat([sd[k].flatten() for k in sd if k.endswith("running_var") and "encoder" in k])
            compare(gold, "net.train.running_mean", rm)
            compare(gold, "net.train.running_var", rv)
            nbt = [int(sd[k]) for k in sd if k.endswith("num_batches_tracked") and "encoder" in k]
            assert all(v == 1 for v in nbt)


def test_g5_two_train_steps_golden_dropin_surface():
    """model(x) -> LossManager -> zero_grad -> backward -> optimiser.step, exactly as train.py:150-156."""
    from footprints_amd.model_manager import ModelManager
    from footprints_amd.training.losses import LossManager
    from oracle import restatement as R
    from tests.golden.digest import compare, load
    gold = load("g5_train")
    P, B = R.make_state()
    mm = ModelManager(use_cuda=True, learning_rate=1e-4)
    _load_state(mm.model, P, B)
    model, opt = mm.model, mm.optimiser
    model.train()
    lm = LossManager((0.1, 100), 0.25)
    names = [k for k, _ in model.named_parameters()]
    assert names == list(gold["train.param_names"])
    for step in range(2):
        batch = {k: v.cuda() for k, v in R.make_batch(2, 64, 96, tag="g5.step%d" % step).items()}
        outputs = model(batch["image"])
        losses = lm(outputs, batch)
        assert len(losses) == 21 and len(outputs) == 24             # 4 outputs + 20 viz tensors (losses.py:90)
        model.zero_grad()
        losses["loss"].backward()
        g = dict(model.named_parameters())
        if step == 0:
            dead = [k for k in names if g[k].grad is None]
            assert dead == list(gold["train.dead_params"])
            # decoder-side gradient samples against the reference's own values; encoder-side gradients (fp32-ill-conditioned on this
            # 2x64x96 input) and Adam's moments: tests/test_gpu_parity_fullsize.py::test_g5_gradients_and_adam_state_fp64_anchored
            for k in ("mask_decoder.block1.pre_concat_conv.conv1.weight", "depth_decoder.outconv4.1.conv1.weight",
                      "depth_decoder.block4.post_concat_conv.conv1.weight"):
                compare(gold, "train.grad." + k, g[k].grad)
        opt.step()
        vals = np.array([float(losses[k]) for k in R.LOSS_KEYS])
        np.testing.assert_allclose(vals, gold["train.losses%d" % step], rtol=1e-4)
        sd = model.state_dict()
        ps = np.array([float(sd[k].double().sum()) for k in names])
        pa = gold["train.param_abs%d" % step]
        bad = np.abs(ps - gold["train.param_sums%d" % step]) > 1e-4 * np.maximum(pa, 1e-12)
        assert not bad.any(), [(names[i], ps[i], gold["train.param_sums%d" % step][i]) for i in np.nonzero(bad)[0][:5]]
    st = opt.state_dict()["state"]
    steps = np.array([float(st[i]["step"]) if i in st else -1.0 for i in range(len(names))])
    assert np.array_equal(steps, gold["train.adam_steps"])
    sd = model.state_dict()
    assert np.array_equal(np.array([int(sd[k]) for k in sd if k.endswith("num_batches_tracked")]), gold["train.nbt"])


def test_trainstep_fast_path_equals_dropin_path():
    from footprints_amd.model_manager import ModelManager
    from footprints_amd.training.losses import LossManager
    from footprints_amd.training.train import TrainStep
    from oracle import restatement as R
    P, B = R.make_state(tag="ts")
    batch = {k: v.cuda() for k, v in R.make_batch(2, 64, 64, tag="ts").items()}
    a, b = ModelManager(), ModelManager()
    _load_state(a.model, P, B)
    _load_state(b.model, P, B)
    ts = TrainStep(a.model, a.optimiser)
    lm = LossManager((0.1, 100), 0.25, compute_viz=False)
    for _ in range(2):
        la = ts(batch).clone()
        b.model.train()
        out = b.model(batch["image"])
        lb = lm(out, batch)
        b.model.zero_grad()
        lb["loss"].backward()
        b.optimiser.step()
        assert torch.equal(la[20], lb["loss"].detach())
    for (ka, va), (kb, vb) in zip(a.model.state_dict().items(), b.model.state_dict().items()):
        assert torch.equal(va, vb), ka                           # same kernels, same order => bit-identical


@pytest.mark.parametrize("Bn,Hn,Wn", [(2, 96, 128), (2, 192, 640)])
def test_full_train_step_against_oracle_and_masks(Bn, Hn, Wn):
    """One full fwd+loss+bwd at non-fixture sizes (incl. the KITTI resolution) through the drop-in surface against the CPU oracle:
    outputs per channel, 21 losses, thresholded masks bit-exact, every parameter gradient fp64-anchored (tests/parity.py)."""
    from footprints_amd.training.losses import LossManager
    from oracle import restatement as R
    from tests.gpu_child import engine_decisions
    from tests.parity import (anchored_report, assert_decisions_at_roundoff, chan_relerr, count_decision_flips, decision_forced_report, oracle_grads,
                              tie_free_batch)
    P, B = R.make_state(tag="full")
    cpu_batch = R.make_batch(Bn, Hn, Wn, tag="full")
    dec64 = R.ReluDecisions()
    out64, l64, g64, _, cpu_batch = oracle_grads(P, B, cpu_batch, torch.float64, fix_batch=lambda b, o: tie_free_batch(b, o)[0], relu_decisions=dec64)
    out_ref, l_ref, g32, _, _ = oracle_grads(P, B, cpu_batch, torch.float32)
    model = _new_model(P, B)
    model.train()
    batch = {k: v.cuda() for k, v in cpu_batch.items()}
    out = model(batch["image"])
    torch.cuda.synchronize()
    decisions = engine_decisions(model.engine())            # before backward: the saved activations are the engine's own buffers
    losses = LossManager((0.1, 100), 0.25)(out, batch)
    losses["loss"].backward()
    for k in R.SCALES:
        assert max(chan_relerr(out[k], out_ref[k])) <= 1e-4 and max(chan_relerr(out[k], out64[k])) <= 1e-4, k
        ref = out_ref[k]
        # masks: sigmoid(logit) > 0.5 (losses.py:78) == logit > 0; predict_simple thresholds the logit at 0.5 (quirk)
        for thr in (0.0, 0.5):
            got_m, ref_m = (out[k][:, :2].cpu() > thr), (ref[:, :2] > thr)
            band = (ref[:, :2] - thr).abs() < 1e-4 * ref[:, :2].abs().max()
            assert torch.equal(got_m | band, ref_m | band), "mask mismatch outside the tie band (%s, thr %.1f)" % (k, thr)
    for key in R.LOSS_KEYS:
        assert abs(float(losses[key]) - float(l_ref[key])) <= 1e-4 * max(1.0, abs(float(l_ref[key]))), key
    g_gpu = OrderedDict((n, p.grad) for n, p in model.named_parameters())
    # The encoder is piecewise linear (ReLU after train-mode BatchNorm over few samples at these sizes: 24 / 240 per channel at
    # layer4).  An activation within fp32 round-off of 0 that one fp32 implementation resolves differently from the float64 truth
    # moves a whole channel's statistics by ~1/samples and every gradient upstream with it -- for whichever implementation it
    # happens to: the gradient is not a continuous function of the arithmetic there.  Round 6 (VERDICT r5 "Next" 2): the encoder part of
    # the rule is never dropped.  With every decision equal to the float64 oracle's it is the plain rule; with a flip, the flips must be
    # FEW and each a float64 round-off tie (tests/parity.py assert_decisions_at_roundoff -- a mis-masked element fails here), and the
    # encoder gradients are then held to the same rule against the float64 oracle evaluated under the engine's decisions.
    flips, total = count_decision_flips(decisions, dec64.taken)
    bad, rows = anchored_report(g_gpu, g32, g64)
    if flips:
        bad_f, rows_f, _, dstats = decision_forced_report(P, B, cpu_batch, decisions, g_gpu, g32, g64)
        print("ReLU decisions differing from the float64 oracle: %d of %d -> imposed on the oracle: %s" % (flips, total, dstats))
        assert_decisions_at_roundoff(dstats, "%dx%dx%d" % (Bn, Hn, Wn))
        dec_bad = [b for b in bad if "decoder" in b.split(" ")[0]]
        assert not dec_bad, dec_bad[:10]                    # decoders sit behind the features: a round-off flip moves them by round-off
        bad, rows = bad_f, rows_f
    print("worst GPU/CPU32 error ratios vs fp64:", ["%s %.2f (gpu %.1e cpu %.1e)" % (n, r, eg, ec) for r, n, eg, ec in rows[:5]])
    assert not bad, bad[:10]


def test_g6_predict_simple_plumbing(tmp_path):
    from PIL import Image
    from footprints_amd.model_manager import ModelManager
    from footprints_amd.predict_simple import InferenceManager
    from oracle import filler, restatement as R
    from tests.golden.digest import compare, load
    gold = load("g6_predict")
    P, B = R.make_state()
    mm = ModelManager(is_inference=True)
    _load_state(mm.model, P, B)
    img = (filler.uniform("g6.image", (269, 477, 3)) * 255).astype(np.uint8)
    path = tmp_path / "synthetic.png"
    Image.fromarray(img, "RGB").save(path)
    im = InferenceManager("kitti", str(tmp_path / "pred"), model_manager=mm)
    im.predict(str(path))
    pred = np.load(tmp_path / "pred" / "outputs" / "synthetic.npy")
    assert pred.shape == (4, 192, 640) and pred.dtype == np.float32
    compare(gold, "predict.npy", torch.from_numpy(pred))
    ref_bits = np.unpackbits(gold["predict.mask_logit_gt_half"])[:192 * 640].reshape(192, 640).astype(bool)
    sample_scale = np.abs(gold["predict.npy#sample"]).max()
    band = np.abs(pred[1] - 0.5) < 1e-4 * sample_scale
    assert np.array_equal((pred[1] > 0.5) | band, ref_bits | band)
    assert (tmp_path / "pred" / "visualisations" / "synthetic.jpg").exists()


def test_checkpoint_roundtrip_reference_format(tmp_path):
    from footprints_amd.model_manager import ModelManager
    from oracle import restatement as R
    P, B = R.make_state(tag="ckpt")
    mm = ModelManager(save_folder=str(tmp_path))
    _load_state(mm.model, P, B)
    from footprints_amd.training.train import TrainStep
    batch = {k: v.cuda() for k, v in R.make_batch(2, 64, 64, tag="ckpt").items()}
    ts = TrainStep(mm.model, mm.optimiser)
    ts(batch)
    mm.save_model("weights_0")
    sd = torch.load(tmp_path / "weights_0" / "model.pth", map_location="cpu")
    assert [k for k in sd] == [s[0] for s in R.state_spec()]
    assert all(tuple(sd[s[0]].shape) == tuple(s[1]) for s in R.state_spec())
    osd = torch.load(tmp_path / "weights_0" / "optimiser.pth", map_location="cpu")
    assert len(osd["state"]) == 196 and len(osd["param_groups"][0]["params"]) == 268
    # the oracle (== reference format) can consume the checkpoint, and we can reload it
    P2 = OrderedDict((k, sd[k]) for k in P)
    ref_opt = torch.optim.Adam([p.clone().requires_grad_(True) for p in P2.values()], lr=1e-4)
    ref_opt.load_state_dict(osd)
    mm2 = ModelManager(save_folder=str(tmp_path))
    mm2.load_model(str(tmp_path / "weights_0"), load_optimiser=True)
    l1 = ts(batch).clone()
    l2 = TrainStep(mm2.model, mm2.optimiser)(batch)
    assert torch.equal(l1, l2)
    for va, vb in zip(mm.model.state_dict().values(), mm2.model.state_dict().values()):
        assert torch.equal(va, vb)


def test_cpu_input_is_refused():
    from footprints_amd import FootprintNetwork
    m = FootprintNetwork(pretrained=False).cuda()
    with pytest.raises(RuntimeError, match="no CPU compute path"):
        m(torch.zeros(1, 3, 64, 64))


# ----------------------------------------------------------------------------------------------------------
# BASELINE.json full sizes: size-independent properties
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,W", [(12, 192, 640), (4, 512, 640), (2, 256, 448)])
def test_full_size_properties(B, H, W):
    """KITTI bs=12 192x640, Matterport bs=4 512x640 and predict_simple's `handheld` resolution 256x448 (8x14 pyramid top): (1) bit-reproducible step, (2) eval forward of a batch ==
    per-image forwards (images are independent in eval mode; equal to fp32 round-off, not bitwise: small grids are
    split along K, so the summation grouping depends on the batch size), (3) image 0 against the CPU oracle (eval)."""
    from footprints_amd.model_manager import ModelManager
    from footprints_amd.training.train import TrainStep, synthetic_batch
    from oracle import restatement as R
    P, Bf = R.make_state(tag="fs")
    batch = synthetic_batch(B, H, W, "cuda")
    runs = []
    for _ in range(2):
        mm = ModelManager()
        _load_state(mm.model, P, Bf)
        ts = TrainStep(mm.model, mm.optimiser)
        l = [ts(batch).clone() for _ in range(2)]
        runs.append((l, [v.clone() for v in mm.model.state_dict().values()]))
    assert all(torch.equal(a, b) for a, b in zip(runs[0][0], runs[1][0]))
    assert all(torch.equal(a, b) for a, b in zip(runs[0][1], runs[1][1]))
    assert torch.isfinite(runs[0][0][1]).all() and float(runs[0][0][1][20]) < float(runs[0][0][0][20]) * 1.5
    model = _new_model(P, Bf)
    model.eval()
    with torch.no_grad():
        full = {k: v.clone() for k, v in model(batch["image"]).items()}
        for i in (0, B - 1):
            one = model(batch["image"][i:i + 1])
            for k in full:
                assert relerr(one[k][0], full[k][i]) <= 1e-5, (k, i)
        ref = R.footprint_network(batch["image"][:1].cpu(), P, OrderedDict((k, v.clone()) for k, v in Bf.items()), False)
    for k in full:
        assert relerr(full[k][:1], ref[k]) <= 1e-4, k


def test_trainstep_graph_replay_is_bit_identical_to_eager():
    """opt-in hipGraph replay of the whole step (TrainStep(graph=True)): same losses and weights as the eager schedule"""
    from footprints_amd.model_manager import ModelManager
    from footprints_amd.training.train import TrainStep, synthetic_batch
    from oracle import restatement as R
    P, Bf = R.make_state(tag="gr")
    batch = synthetic_batch(2, 96, 128, "cuda")
    res = []
    for graph in (False, True):
        mm = ModelManager()
        _load_state(mm.model, P, Bf)
        ts = TrainStep(mm.model, mm.optimiser, graph=graph)
        losses = [ts(batch).clone() for _ in range(5)]           # graph mode: 2 eager + capture + 2 replays
        res.append((losses, [v.clone() for v in mm.model.state_dict().values()]))
        if graph:
            assert ts._graph is not None
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)


def test_trainstep_launch_plan_replay_is_bit_identical_to_eager():
    """recorded launch plan (csrc/plan.cpp; TrainStep(plan=True), the default): 2 eager steps, 1 recording step, then replays from C --
    same losses and weights as issuing every step from Python, also when the batch alternates between two sets of input buffers
    (the double-buffered loader: one plan per set) and when an eval forward runs between replays"""
    from footprints_amd.model_manager import ModelManager
    from footprints_amd.training.train import TrainStep, synthetic_batch
    from oracle import restatement as R
    P, Bf = R.make_state(tag="plan")
    b0, b1 = synthetic_batch(2, 96, 128, "cuda", seed=3), synthetic_batch(2, 96, 128, "cuda", seed=4)
    res = []
    for plan in (False, True):
        mm = ModelManager()
        _load_state(mm.model, P, Bf)
        ts = TrainStep(mm.model, mm.optimiser, plan=plan)
        losses = []
        for i in range(9):
            losses.append(ts(b0 if i % 2 == 0 else b1).clone())
            if i == 6:
                mm.model.eval()
                with torch.no_grad():
                    ev = mm.model(b0["image"])["1/1"].clone()
                mm.model.train()
        res.append((losses, [v.clone() for v in mm.model.state_dict().values()], ev))
        if plan:
            assert len(ts._plans) == 2 and all(n > 500 for *_, n in ts._plans.values())
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
    assert torch.equal(res[0][2], res[1][2])


def test_eval_forward_with_folded_batchnorm_matches_unfolded():
    """inference fast path: encoder BN folded into the conv weights == conv -> BN launches, after training steps
    have moved the running statistics, and again after they move once more (the fold is rebuilt)"""
    from footprints_amd.model_manager import ModelManager
    from footprints_amd.training.train import TrainStep, synthetic_batch
    from oracle import restatement as R
    P, Bf = R.make_state(tag="fold")
    mm = ModelManager()
    _load_state(mm.model, P, Bf)
    batch = synthetic_batch(2, 192, 640, "cuda")
    ts = TrainStep(mm.model, mm.optimiser)
    eng = mm.model.engine()
    for rnd in range(2):
        ts(batch)
        mm.model.eval()
        with torch.no_grad():
            eng.fold_eval = True
            a = [v.clone() for v in mm.model(batch["image"]).values()]
            assert eng._fold_ready
            eng.fold_eval = False
            b = [v.clone() for v in mm.model(batch["image"]).values()]
            eng.fold_eval = True
        for x, y in zip(a, b):
            assert relerr(x, y) <= 2e-5, rnd
        mm.model.train()


def test_optin_bf16x2_inference_stays_inside_the_north_star_bar():
    """model.inference_precision = "bf16x2" (opt-in, eval / no_grad only): the 3x3 stride-1 tile convolutions round their operands to two
    bf16 terms (three MFMA products instead of six).  Not exact -- measured ~2e-5 of the channel max at 12x192x640 -- so the test holds
    it to the north star's own bar: outputs per channel within 1e-4 of the CPU oracle, thresholded masks equal outside the tie band; the
    default path stays ~1e-6, and a training forward ignores the switch (bit-identical step)."""
    from footprints_amd.model_manager import ModelManager
    from footprints_amd.training.train import TrainStep
    from oracle import restatement as R
    from tests.parity import chan_relerr
    P, Bf = R.make_state(tag="bf2")
    img = R.make_batch(2, 192, 640, tag="bf2")["image"]
    with torch.no_grad():
        ref = R.footprint_network(img, P, OrderedDict((k, v.clone()) for k, v in Bf.items()), False)
    model = _new_model(P, Bf)
    model.eval()
    errs = {}
    for mode in ("exact", "bf16x2"):
        model.inference_precision = mode
        with torch.no_grad():
            out = model(img.cuda())
        errs[mode] = max(max(chan_relerr(out[k], ref[k])) for k in out)
        assert errs[mode] <= 1e-4, (mode, errs[mode])
        for k in out:
            for thr in (0.0, 0.5):
                band = (ref[k][:, :2] - thr).abs() < 1e-4 * ref[k][:, :2].abs().max()
                assert torch.equal((out[k][:, :2].cpu() > thr) | band, (ref[k][:, :2] > thr) | band), (mode, k, thr)
    assert errs["exact"] <= 1e-5 < errs["bf16x2"]              # the switch really changed the arithmetic, and only there
    batch = {k: v.cuda() for k, v in R.make_batch(2, 96, 128, tag="bf2t").items()}
    res = []
    for mode in ("exact", "bf16x2"):
        mm = ModelManager()
        _load_state(mm.model, P, Bf)
        mm.model.inference_precision = mode
        ts = TrainStep(mm.model, mm.optimiser)
        res.append([ts(batch).clone() for _ in range(2)])
    assert all(torch.equal(a, b) for a, b in zip(*res))


def test_inference_scales_subset():
    from footprints_amd.model_manager import ModelManager
    from footprints_amd.training.train import synthetic_batch
    mm = ModelManager()
    mm.model.eval()
    img = synthetic_batch(2, 64, 96, "cuda")["image"]
    with torch.no_grad():
        full = {k: v.clone() for k, v in mm.model(img).items()}
        mm.model.inference_scales = ("1/1",)
        only = mm.model(img)
    assert list(only.keys()) == ["1/1"] and torch.equal(only["1/1"], full["1/1"])


def test_inference_manager_test_batch_matches_reference_postprocessing(tmp_path):
    """evaluation/inference.py:99-108 + inference_dataset.py:35-38: model(x)['1/1'] -> sigmoid on channels 0:2 -> float16"""
    from footprints_amd.evaluation.inference import InferenceManager
    from footprints_amd.model_manager import ModelManager
    from footprints_amd.training.train import synthetic_batch
    mm = ModelManager()
    img = synthetic_batch(2, 64, 96, "cuda")["image"]
    mm.model.eval()
    with torch.no_grad():
        ref = mm.model(img)["1/1"].clone()
    ref[:, 0:2] = torch.sigmoid(ref[:, 0:2])
    ref16 = ref.cpu().numpy().astype(np.float16)
    im = InferenceManager(model_manager=mm, save_path=str(tmp_path))
    got = im.test_batch({"image": img.cpu()})
    assert got.dtype == np.float16 and got.shape == ref16.shape
    assert np.array_equal(got[:, 2:], ref16[:, 2:])                                    # pure rounding: bit-exact
    ulp = np.abs(got[:, :2].view(np.int16).astype(np.int32) - ref16[:, :2].view(np.int16).astype(np.int32))
    assert ulp.max() <= 1                                                              # sigmoid: within one float16 ulp
    im.save_result("frame0", got[0])
    back = np.load(os.path.join(str(tmp_path), "frame0.npy"))
    assert back.dtype == np.float16 and np.array_equal(back, got[0])


@pytest.mark.parametrize("dt", [np.float16, np.float32])
def test_g7_device_metrics_match_reference_scores(dt):
    """evaluation/evaluate_model.py: per-image IoU / precision / recall / F1 from the device confusion counts are EXACTLY the
    reference's (same integers), depth errors within float32-summation noise of the reference's numpy means"""
    from footprints_amd.evaluation import evaluate_model as EM
    from tests.golden.metrics_inputs import DEPTH_KEYS, MASK_KEYS, N, metrics_inputs
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g7_metrics.npz"))
    pred, gt_kitti, gt_mp, free, gt_depth = metrics_inputs(dt)
    tag = np.dtype(dt).name
    for flavour, gt in (("kitti", gt_kitti), ("matterport", gt_mp)):
        scores = EM.mask_scores(pred, gt, free)
        for i in range(N):
            np.testing.assert_array_equal(np.array([scores[i]["freespace"][k] for k in MASK_KEYS], np.float64), g["%s.%s.freespace" % (tag, flavour)][i])
            np.testing.assert_array_equal(np.array([scores[i]["footprint"][k] for k in MASK_KEYS], np.float64), g["%s.%s.footprint" % (tag, flavour)][i])
        ref = {"freespace_iou": np.nanmean(g["%s.%s.freespace" % (tag, flavour)][:, 0]), "footprint_f1": np.nanmean(g["%s.%s.footprint" % (tag, flavour)][:, 3])}
        summ = EM.summarise(scores, "iou")
        assert summ["freespace_iou"] == ref["freespace_iou"] and summ["footprint_f1"] == ref["footprint_f1"]
    dscores = EM.depth_scores(pred, gt_depth)
    want = g["%s.depth" % tag]
    for i in range(N):
        got = np.array([dscores[i][k] for k in DEPTH_KEYS], np.float64)
        if np.isnan(want[i]).all():
            assert np.isnan(got).all()
        else:
            assert got[0] == want[i][0]                                       # a1 is a count ratio: exact
            np.testing.assert_allclose(got[1:], want[i][1:], rtol=1e-5)       # the reference's numpy means accumulate in float32, the device sums in float64
