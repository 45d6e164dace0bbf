"""CPU: the oracle restatement against the committed golden fixtures (reference outputs).

Runs without /root/reference (the fixtures were generated from it by
tests/golden/make_golden.py).  Tolerances: 1e-4 relative to the tensor's max
(north_star: "within 1e-4 rel fp32"); in practice the restatement is bit-exact.
"""
import os
from collections import OrderedDict

import numpy as np
import torch

from oracle import restatement as R
from tests.golden.digest import compare, fill, fill_value, load


def _convblock_state(tag, prefix, cin, cout):
    P = {}
    for key, shape in ((".conv1.weight", (cout, cin, 3, 3)), (".conv1.bias", (cout,)),
                       (".conv2.weight", (cout, cout, 3, 3)), (".conv2.bias", (cout,))):
        P[prefix + key] = fill_value(tag, (prefix + key).lstrip("."), shape).requires_grad_(True)
    return P


def test_g1_convblock():
    gold = load("g1_blocks")
    P = _convblock_state("g1.convblock", "", 16, 8)
    x = fill("g1.convblock.x", (2, 16, 6, 10)).requires_grad_(True)
    y = R.conv_block(x, P, "")
    (y * fill("g1.convblock.g", tuple(y.shape))).sum().backward()
    compare(gold, "convblock.y", y)
    compare(gold, "convblock.dx", x.grad)
    compare(gold, "convblock.dw1", P[".conv1.weight"].grad)
    compare(gold, "convblock.db2", P[".conv2.bias"].grad)


def test_g1_upcat():
    gold = load("g1_blocks")
    P = {}
    P.update(_convblock_state("g1.upcat", ".pre_concat_conv", 16, 8))
    P.update(_convblock_state("g1.upcat", ".post_concat_conv", 16, 8))
    x = fill("g1.upcat.x", (2, 16, 4, 6)).requires_grad_(True)
    s = fill("g1.upcat.skip", (2, 8, 8, 12)).requires_grad_(True)
    y = R.up_concat_block(x, s, P, "")
    (y * fill("g1.upcat.g", tuple(y.shape))).sum().backward()
    compare(gold, "upcat.y", y)
    compare(gold, "upcat.dx", x.grad)
    compare(gold, "upcat.dskip", s.grad)
    compare(gold, "upcat.post.dw1", P[".post_concat_conv.conv1.weight"].grad)
    compare(gold, "upcat.pre.dw2", P[".pre_concat_conv.conv2.weight"].grad)


def test_g1_outconv_all_scales():
    gold = load("g1_blocks")
    for scale in (1, 2, 4, 8):
        for sig in (False, True):
            P = {".conv1.weight": fill_value("g1.outconv", "conv1.weight", (2, 16, 3, 3)).requires_grad_(True),
                 ".conv1.bias": fill_value("g1.outconv", "conv1.bias", (2,)).requires_grad_(True)}
            x = fill("g1.outconv.x", (2, 16, 6, 10)).requires_grad_(True)
            y = R.out_conv_block(x, P, "", scale, sig)
            (y * fill("g1.outconv.g%d" % scale, tuple(y.shape))).sum().backward()
            tag = "outconv.s%d.%s" % (scale, "sig" if sig else "lin")
            compare(gold, tag + ".y", y)
            compare(gold, tag + ".dx", x.grad)
            compare(gold, tag + ".dw", P[".conv1.weight"].grad)
            compare(gold, tag + ".db", P[".conv1.bias"].grad)


def decoder_state(tag="g2.decoder", prefix="dec"):
    """SkipDecoder weights as make_golden.fill_module produced them (module-local key names)."""
    P = OrderedDict()
    for key, shape, kind in R.state_spec():
        if key.startswith("mask_decoder.") and kind in ("conv_w", "conv_b"):
            local = key[len("mask_decoder."):]
            P[prefix + "." + local] = fill_value(tag, local, shape).requires_grad_(True)
    return P


G2_SHAPES = [(2, 64, 32, 48), (2, 64, 16, 24), (2, 128, 8, 12), (2, 256, 4, 6), (2, 512, 2, 3)]


def test_g2_decoder():
    gold = load("g2_decoder")
    for sig in (False, True):
        tag = "dec.%s" % ("sig" if sig else "lin")
        P = decoder_state()
        feats = [fill("g2.feat%d" % i, s).requires_grad_(True) for i, s in enumerate(G2_SHAPES)]
        o = R.skip_decoder(feats, P, "dec", sig)
        loss = 0
        for k in o:
            loss = loss + (o[k] * fill("g2.g" + k, tuple(o[k].shape))).sum()
        loss.backward()
        for k in o:
            compare(gold, tag + ".out" + k, o[k])
        for i, f in enumerate(feats):
            compare(gold, tag + ".dfeat%d" % i, f.grad)
        for name in ("block1.pre_concat_conv.conv1.weight", "block4.post_concat_conv.conv2.weight",
                     "outconv1.conv1.weight", "outconv4.0.conv1.weight", "outconv4.1.conv1.bias",
                     "block2.post_concat_conv.conv1.bias"):
            compare(gold, tag + ".d." + name, P["dec." + name].grad)


def test_g3_network_train_eval():
    gold = load("g3_network")
    P, B = R.make_state()
    batch = R.make_batch(2, 64, 96)
    for mode in ("train", "eval"):
        B2 = OrderedDict((k, v.clone()) for k, v in B.items())
        with torch.no_grad():
            o = R.footprint_network(batch["image"], P, B2, mode == "train")
        assert list(o.keys()) == ["1/8", "1/4", "1/2", "1/1"]          # network.py:26-30 insertion order
        for k in o:
            assert tuple(o[k].shape) == (2, 4, 64, 96)
            compare(gold, "net.%s.out%s" % (mode, k), o[k])
        if mode == "train":
            rm = torch.cat([B2[k].flatten() for k in B2 if k.endswith("running_mean") and "encoder" in k])
            rv = torch.cat([B2[k].flatten() for k in B2 if k.endswith("running_var") and "encoder" in k])
            compare(gold, "net.train.running_mean", rm)
            compare(gold, "net.train.running_var", rv)


def g4_inputs():
    B, H, W = 2, 8, 16
    batch = R.make_batch(B, H, W, tag="g4")
    preds = OrderedDict()
    for k in R.SCALES:
        p = fill("g4.pred" + k, (B, 4, H, W), -3.0, 3.0)
        p[:, 2:] = torch.sigmoid(p[:, 2:])
        preds[k] = p.requires_grad_(True)
    return preds, batch


def test_g4_loss():
    gold = load("g4_loss")
    preds, batch = g4_inputs()
    losses, viz = R.loss_manager(preds, batch)
    assert list(losses.keys()) == R.LOSS_KEYS and len(losses) == 21 and len(viz) == 20
    losses["loss"].backward()
    vals = np.array([float(losses[k]) for k in R.LOSS_KEYS])
    np.testing.assert_allclose(vals, gold["loss.values"], rtol=2e-6)
    for k in R.SCALES:
        compare(gold, "loss.dpred" + k, preds[k].grad, rtol=1e-5)
        compare(gold, "loss.viz.ground_depth_masked" + k, viz[("ground_depth_masked", k)], rtol=1e-6)


def test_g5_two_train_steps():
    gold = load("g5_train")
    P, B = R.make_state()
    tr = R.OracleTrainer(P, B, lr=1e-4)
    names = [k for k in tr.P]
    assert names == list(gold["train.param_names"])
    for step in range(2):
        batch = R.make_batch(2, 64, 96, tag="g5.step%d" % step)
        if step == 0:
            _, losses = tr.forward_backward(batch)
            dead = [k for k in names if tr.P[k].grad is None]
            assert dead == list(gold["train.dead_params"]) and len(dead) == 72
            assert all(R.is_dead_param(k) for k in dead)
            gs = np.array([float(tr.P[k].grad.double().sum()) if tr.P[k].grad is not None else 0.0 for k in names])
            ga = gold["train.grad_abs"]
            assert np.all(np.abs(gs - gold["train.grad_sums"]) <= 1e-4 * np.maximum(ga, 1e-12))
            tr.opt.step()
        else:
            _, losses = tr.step(batch)
        vals = np.array([float(losses[k]) for k in R.LOSS_KEYS])
        np.testing.assert_allclose(vals, gold["train.losses%d" % step], rtol=1e-5)
        ps = np.array([float(tr.P[k].detach().double().sum()) for k in names])
        pa = gold["train.param_abs%d" % step]
        assert np.all(np.abs(ps - gold["train.param_sums%d" % step]) <= 1e-5 * np.maximum(pa, 1e-12))
    nbt = np.array([int(tr.B[k]) for k in tr.B if k.endswith("num_batches_tracked")])
    assert np.array_equal(nbt, gold["train.nbt"])


def test_state_spec_counts():
    spec = R.state_spec()
    assert len(spec) == 484
    params = [s for s in spec if s[2] in ("conv_w", "conv_b", "bn_w", "bn_b")]
    assert len(params) == 268
    n_all = sum(int(np.prod(s[1])) for s in params)
    n_live = sum(int(np.prod(s[1])) for s in params if not R.is_dead_param(s[0]))
    assert n_all == 31021392 and n_live == 31012944
    assert sum(1 for s in params if not R.is_dead_param(s[0])) == 196


def test_g7_metrics_oracle_reproduces_reference_scores():
    """oracle/metrics.py (restatement of evaluation/evaluate_model.py) == the reference's own per-image scores, exactly"""
    from oracle import metrics as M
    from tests.golden.metrics_inputs import DEPTH_KEYS, MASK_KEYS, N, metrics_inputs
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g7_metrics.npz"))
    for dt in (np.float16, np.float32):
        pred, gt_kitti, gt_mp, free, gt_depth = metrics_inputs(dt)
        tag = np.dtype(dt).name
        for flavour, gt in (("kitti", gt_kitti), ("matterport", gt_mp)):
            for i in range(N):
                s = M.score_image(pred[i], gt[i], free[i], "iou")
                np.testing.assert_array_equal(np.array([s["freespace"][k] for k in MASK_KEYS], np.float64), g["%s.%s.freespace" % (tag, flavour)][i])
                np.testing.assert_array_equal(np.array([s["footprint"][k] for k in MASK_KEYS], np.float64), g["%s.%s.footprint" % (tag, flavour)][i])
        for i in range(N):
            s = M.score_image(pred[i], gt_depth[i], None, "depth")
            np.testing.assert_array_equal(np.array([s[k] for k in DEPTH_KEYS], np.float64), g["%s.depth" % tag][i])
    assert np.isnan(g["float16.kitti.freespace"][4]).all() and np.isnan(g["float16.depth"][5]).all()      # the nan cases are in the fixture


def _g8_preds(tag, B=2, H=16, W=32):
    from tests.golden.digest import fill
    d = {}
    for k in R.SCALES:
        p = fill("g8.%s.pred%s" % (tag, k), (B, 4, H, W), -3.0, 3.0)
        p[:, 2:] = torch.sigmoid(p[:, 2:])
        d[k] = p
    return d


def test_g8_evaluator_restatement_matches_reference_evaluator():
    """training/evaluation.py:28-67 (fixture from the reference's own Evaluator): averaged train / val losses, reset semantics"""
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_evaluator.npz"))
    ev = R.OracleEvaluator((0.1, 100), 0.25)
    last = None
    for i in range(3):
        last = ev.compute_losses(R.make_batch(2, 16, 32, tag="g8.train%d" % i), _g8_preds("train%d" % i), "train", True)
    for i in range(2):
        assert ev.compute_losses(R.make_batch(2, 16, 32, tag="g8.val%d" % i), _g8_preds("val%d" % i), "val") is None
    np.testing.assert_allclose([float(last[k]) for k in R.LOSS_KEYS], gold["eval.last_batch"], rtol=1e-6)
    a = ev.get_averaged_losses("train", reset=False)
    b = ev.get_averaged_losses("train", reset=True)
    assert a == b and ev.get_averaged_losses("train") == {}
    np.testing.assert_allclose([a[k] for k in R.LOSS_KEYS], gold["eval.train_avg"], rtol=1e-6)
    v = ev.get_averaged_losses("val")
    np.testing.assert_allclose([v[k] for k in R.LOSS_KEYS], gold["eval.val_avg"], rtol=1e-6)


def test_g9_segmentor_restatement_matches_reference():
    """preprocessing/segmentation/network.py (fixture from the reference's own Segmentor + segmentation Evaluator): four logit maps,
    masked-BCE loss, dead parameters, gradient digests -- with and without the pyramid-pooling module"""
    from oracle import filler
    gold = load("g9_segmentor")
    B, H, W = 2, 64, 96
    image = torch.from_numpy(filler.uniform("g9:image", (B, 3, H, W)))
    gmask = torch.from_numpy(filler.bernoulli("g9:gmask", (B, H, W), 0.4))
    lmask = torch.from_numpy(filler.bernoulli("g9:lmask", (B, H, W), 0.7))
    for psp in (False, True):
        tag = "psp" if psp else "plain"
        P, Bf = R.make_seg_state(psp, tag="g9." + tag)
        assert list({**P, **Bf}.keys()) is not None
        P = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in P.items())
        outs = R.segmentor(image, P, OrderedDict((k, v.clone()) for k, v in Bf.items()), True, psp)
        for i, o in enumerate(outs):
            compare(gold, "seg.%s.out%d" % (tag, i), o)
        loss = R.seg_loss(outs, gmask, lmask, H, W)
        loss.backward()
        assert abs(float(loss) - float(gold["seg.%s.loss" % tag])) <= 1e-6 * abs(float(gold["seg.%s.loss" % tag]))
        names = [str(n) for n in gold["seg.%s.param_names" % tag]]
        assert [k for k in names if P[k].grad is None] == [str(n) for n in gold["seg.%s.dead" % tag]]
        gs = np.array([float(P[k].grad.double().sum()) if P[k].grad is not None else 0.0 for k in names])
        ga = gold["seg.%s.grad_abs" % tag]
        assert np.all(np.abs(gs - gold["seg.%s.grad_sums" % tag]) <= 1e-4 * np.maximum(ga, 1e-12))
        for k in [f[len("seg.%s.grad." % tag):].split("#")[0] for f in gold.files if f.startswith("seg.%s.grad." % tag)]:
            compare(gold, "seg.%s.grad.%s" % (tag, k), P[k].grad)
