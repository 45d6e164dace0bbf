"""Generate the committed golden fixtures by running the REFERENCE's own code.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py
Writes tests/golden/*.npz.  Inputs and weights are regenerated from
oracle/filler.py (counter hash) by name, so only outputs / digests are stored.

What is reference-authored arithmetic and what is not:
  * G1 blocks, G2 decoder, G4 loss            -> reference classes, directly.
  * G3 whole net, G5 train steps, G6 predict  -> reference FootprintNetwork /
    LossManager / Adam, but the ENCODER inside is oracle/standin_resnet.py
    (torchvision is absent) => encoder part "parity unpinned" by the reference.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import filler, ref_import, restatement as R  # noqa: E402
from tests.golden.digest import digest, GOLDEN_DIR, fill, fill_value      # noqa: E402


def fill_module(mod, tag):
    """Deterministic weights for a reference sub-module (keys = its own state_dict names)."""
    sd = mod.state_dict()
    mod.load_state_dict({k: fill_value(tag, k, tuple(v.shape), v.dtype == torch.int64) for k, v in sd.items()})


def g1_blocks(net):
    out = {}
    torch.manual_seed(0)
    # ConvBlock 16->8 @ 6x10
    m = net.ConvBlock(in_ch=16, out_ch=8, use_elu=True, use_bn=False)
    fill_module(m, "g1.convblock")
    x = fill("g1.convblock.x", (2, 16, 6, 10)).requires_grad_(True)
    y = m(x)
    g = fill("g1.convblock.g", tuple(y.shape))
    (y * g).sum().backward()
    out.update({"convblock.y": y, "convblock.dx": x.grad,
                "convblock.dw1": m.conv1.weight.grad, "convblock.db1": m.conv1.bias.grad,
                "convblock.dw2": m.conv2.weight.grad, "convblock.db2": m.conv2.bias.grad})
    # ConvUpsampleAndConcatBlock 16->8, x @4x6, skip @8x12
    m = net.ConvUpsampleAndConcatBlock(in_ch=16, out_ch=8, use_elu=True, use_bn=False)
    fill_module(m, "g1.upcat")
    x = fill("g1.upcat.x", (2, 16, 4, 6)).requires_grad_(True)
    s = fill("g1.upcat.skip", (2, 8, 8, 12)).requires_grad_(True)
    y = m(x, s)
    g = fill("g1.upcat.g", tuple(y.shape))
    (y * g).sum().backward()
    out.update({"upcat.y": y, "upcat.dx": x.grad, "upcat.dskip": s.grad,
                "upcat.pre.dw1": m.pre_concat_conv.conv1.weight.grad,
                "upcat.pre.dw2": m.pre_concat_conv.conv2.weight.grad,
                "upcat.post.dw1": m.post_concat_conv.conv1.weight.grad,
                "upcat.post.db1": m.post_concat_conv.conv1.bias.grad,
                "upcat.post.dw2": m.post_concat_conv.conv2.weight.grad})
    # OutConvBlock 16->2, scales 1/2/4/8, sigmoid on/off
    for scale in (1, 2, 4, 8):
        for sig in (False, True):
            m = net.OutConvBlock(in_ch=16, out_ch=2, scale=scale, apply_sigmoid=sig)
            fill_module(m, "g1.outconv")
            x = fill("g1.outconv.x", (2, 16, 6, 10)).requires_grad_(True)
            y = m(x)
            g = fill("g1.outconv.g%d" % scale, tuple(y.shape))
            (y * g).sum().backward()
            tag = "outconv.s%d.%s" % (scale, "sig" if sig else "lin")
            out.update({tag + ".y": y, tag + ".dx": x.grad,
                        tag + ".dw": m.conv1.weight.grad, tag + ".db": m.conv1.bias.grad})
    return {k: v.detach().numpy() for k, v in out.items()}


def g2_decoder(net):
    out = {}
    shapes = [(2, 64, 32, 48), (2, 64, 16, 24), (2, 128, 8, 12), (2, 256, 4, 6), (2, 512, 2, 3)]
    for sig in (False, True):
        tag = "dec.%s" % ("sig" if sig else "lin")
        m = net.SkipDecoder(apply_sigmoid=sig)
        fill_module(m, "g2.decoder")
        feats = [fill("g2.feat%d" % i, s).requires_grad_(True) for i, s in enumerate(shapes)]
        o = m(feats)
        loss = 0
        for k in o:
            loss = loss + (o[k] * fill("g2.g" + k, tuple(o[k].shape))).sum()
        loss.backward()
        for k in o:
            out.update(digest(tag + ".out" + k, o[k]))
        for i, f in enumerate(feats):
            out.update(digest(tag + ".dfeat%d" % i, f.grad))
        for name in ("block1.pre_concat_conv.conv1.weight", "block4.post_concat_conv.conv2.weight",
                     "outconv1.conv1.weight", "outconv4.0.conv1.weight", "outconv4.1.conv1.bias",
                     "block2.post_concat_conv.conv1.bias"):
            p = dict(m.named_parameters())[name]
            out.update(digest(tag + ".d." + name, p.grad))
    return out


def g3_network(net):
    out = {}
    P, B = R.make_state()
    batch = R.make_batch(2, 64, 96)
    for mode in ("train", "eval"):
        m = net.FootprintNetwork(pretrained=False)
        m.load_state_dict({**P, **B})
        m.train(mode == "train")
        with torch.no_grad():
            o = m(batch["image"])
        for k in o:
            out.update(digest("net.%s.out%s" % (mode, k), o[k], full_limit=1 << 17))
        if mode == "train":
            sd = m.state_dict()
            rm = torch.cat([sd[k].flatten() for k in sd if k.endswith("running_mean") and "encoder" in k])
            rv = torch.cat([sd[k].flatten() for k in sd if k.endswith("running_var") and "encoder" in k])
            out["net.train.running_mean"] = rm.numpy()
            out["net.train.running_var"] = rv.numpy()
    return out


def g4_loss(loss_mod):
    out = {}
    B, H, W = 2, 8, 16
    batch = R.make_batch(B, H, W, tag="g4")
    preds = {}
    for k in R.SCALES:
        p = fill("g4.pred" + k, (B, 4, H, W), -3.0, 3.0)
        p[:, 2:] = torch.sigmoid(p[:, 2:])            # depth channels are sigmoid outputs
        preds[k] = p.requires_grad_(True)
    lm = loss_mod.LossManager((0.1, 100), 0.25)
    d = dict(preds)
    losses = lm(d, batch)
    losses["loss"].backward()
    out["loss.values"] = np.array([float(losses[k]) for k in R.LOSS_KEYS], dtype=np.float64)
    for k in R.SCALES:
        out["loss.dpred" + k] = preds[k].grad.numpy()
        out["loss.viz.ground_depth_masked" + k] = d[("ground_depth_masked", k)].detach().numpy()
    return out


def g5_train(net, loss_mod):
    """Two optimiser steps exactly as training/train.py:150-156 + model_manager.py:27."""
    out = {}
    P, B = R.make_state()
    m = net.FootprintNetwork(pretrained=False)
    m.load_state_dict({**P, **B})
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    lm = loss_mod.LossManager((0.1, 100), 0.25)
    names = [k for k, _ in m.named_parameters()]
    for step in range(2):
        batch = R.make_batch(2, 64, 96, tag="g5.step%d" % step)
        outputs = m(batch["image"])
        losses = lm(outputs, batch)
        m.zero_grad()
        losses["loss"].backward()
        if step == 0:
            dead = [k for k, p in m.named_parameters() if p.grad is None]
            out["train.dead_params"] = np.array(dead)
            g = dict(m.named_parameters())
            out["train.grad_sums"] = np.array([float(g[k].grad.double().sum()) if g[k].grad is not None else 0.0 for k in names])
            out["train.grad_abs"] = np.array([float(g[k].grad.double().abs().sum()) if g[k].grad is not None else 0.0 for k in names])
            for k in ("encoder.layer0.0.weight", "encoder.layer4.2.conv2.weight", "encoder.layer2.0.downsample.0.weight",
                      "encoder.layer1.1.0.bn1.weight", "encoder.layer3.5.bn2.bias",
                      "mask_decoder.block1.pre_concat_conv.conv1.weight", "depth_decoder.outconv4.1.conv1.weight",
                      "depth_decoder.block4.post_concat_conv.conv1.weight"):
                out.update(digest("train.grad." + k, g[k].grad))
        opt.step()
        out["train.losses%d" % step] = np.array([float(losses[k]) for k in R.LOSS_KEYS], dtype=np.float64)
        sd = m.state_dict()
        out["train.param_sums%d" % step] = np.array([float(sd[k].double().sum()) for k in names])
        out["train.param_abs%d" % step] = np.array([float(sd[k].double().abs().sum()) for k in names])
    out["train.param_names"] = np.array(names)
    st = opt.state_dict()["state"]
    out["train.adam_steps"] = np.array([float(st[i]["step"]) if i in st else -1.0 for i in range(len(names))])
    out["train.exp_avg_abs"] = np.array([float(st[i]["exp_avg"].double().abs().sum()) if i in st else 0.0 for i in range(len(names))])
    out["train.exp_avg_sq_sum"] = np.array([float(st[i]["exp_avg_sq"].double().sum()) if i in st else 0.0 for i in range(len(names))])
    sd = m.state_dict()
    out["train.nbt"] = np.array([int(sd[k]) for k in sd if k.endswith("num_batches_tracked")])
    return out


def g6_predict(net):
    """predict_simple.py:51-68 plumbing: LANCZOS resize -> ToTensor -> eval forward -> [4,H,W] npy."""
    from PIL import Image
    out = {}
    img = (filler.uniform("g6.image", (269, 477, 3)) * 255).astype(np.uint8)   # same size as test_data/cyclist.jpg
    pil = Image.fromarray(img, "RGB")
    pil = pil.resize((640, 192), Image.LANCZOS)     # transforms.Resize((192,640), ANTIALIAS) predict_simple.py:41-42
    x = torch.from_numpy(np.asarray(pil).astype(np.float32) / 255.0).permute(2, 0, 1)[None].contiguous()  # ToTensor
    out.update(digest("predict.input", x))
    P, B = R.make_state()
    m = net.FootprintNetwork(pretrained=False)
    m.load_state_dict({**P, **B})
    m.eval()
    pred = m(x)["1/1"].data.cpu().numpy().squeeze(0)          # predict_simple.py:67-68 (no no_grad: quirk)
    out.update(digest("predict.npy", torch.from_numpy(pred), full_limit=0))
    out["predict.mask_logit_gt_half"] = np.packbits(pred[1] > 0.5)   # predict_simple.py:77 threshold on the LOGIT
    return out


def g7_metrics():
    """evaluation/evaluate_model.py: the per-image scores of evaluate() (:160-177) from the reference's own evaluate_mask /
    evaluate_depth / sigmoid_to_depth, on float16 predictions (what the inference pass saves) and on float32 ones."""
    from tests.golden.metrics_inputs import DEPTH_KEYS, MASK_KEYS, N, metrics_inputs
    em = ref_import.load_reference_metrics()
    out = {}
    for dt in (np.float16, np.float32):
        pred, gt_kitti, gt_mp, free, gt_depth = metrics_inputs(dt)
        tag = np.dtype(dt).name
        for flavour, gt in (("kitti", gt_kitti), ("matterport", gt_mp)):
            fs = np.zeros((N, 4)); fpt = np.zeros((N, 4))
            for i in range(N):
                p = pred[i][em.HIDDEN_GROUND]                                           # evaluate_model.py:161-163
                a = em.evaluate_mask(gt[i], p)                                          # :166
                b = em.evaluate_mask(1 - gt[i][free[i]], 1 - p[free[i]])                # :167
                fs[i] = [a[k] for k in MASK_KEYS]
                fpt[i] = [b[k] for k in MASK_KEYS]
            out["%s.%s.freespace" % (tag, flavour)] = fs
            out["%s.%s.footprint" % (tag, flavour)] = fpt
        dd = np.zeros((N, 4))
        for i in range(N):
            p = em.sigmoid_to_depth(pred[i][em.HIDDEN_DEPTH])                           # :171-172
            mask = gt_depth[i] > 0                                                      # :174
            r = em.evaluate_depth(gt_depth[i][mask], p[mask])                           # :175
            dd[i] = [r[k] for k in DEPTH_KEYS]
        out["%s.depth" % tag] = dd
    return out


def g8_evaluator():
    """training/evaluation.py:28-67: the reference's own Evaluator over three 'train' and two 'val' batches of random
    predictions: averaged losses without reset, with reset, and the per-batch values of the last call (return_batch_loss)."""
    import importlib
    ev_mod = importlib.import_module("footprints.training.evaluation")
    ev = ev_mod.Evaluator((0.1, 100), 0.25)
    out = {}
    B, H, W = 2, 16, 32

    def preds(tag):
        d = {}
        for k in R.SCALES:
            p = fill("g8.%s.pred%s" % (tag, k), (B, 4, H, W), -3.0, 3.0)
            p[:, 2:] = torch.sigmoid(p[:, 2:])
            d[k] = p
        return d
    last = None
    for i in range(3):
        last = ev.compute_losses(R.make_batch(B, H, W, tag="g8.train%d" % i), preds("train%d" % i), mode="train", return_batch_loss=True)
    out["eval.last_batch"] = np.array([float(last[k]) for k in R.LOSS_KEYS], dtype=np.float64)
    for i in range(2):
        r = ev.compute_losses(R.make_batch(B, H, W, tag="g8.val%d" % i), preds("val%d" % i), mode="val")
        assert r is None                                                               # evaluation.py:45-46
    a = ev.get_averaged_losses("train", reset=False)
    b = ev.get_averaged_losses("train", reset=True)
    c = ev.get_averaged_losses("train", reset=True)
    assert a == b and c == {}
    out["eval.train_avg"] = np.array([a[k] for k in R.LOSS_KEYS], dtype=np.float64)
    v = ev.get_averaged_losses("val", reset=True)
    out["eval.val_avg"] = np.array([v[k] for k in R.LOSS_KEYS], dtype=np.float64)
    return out


def g9_segmentor():
    """preprocessing/segmentation/network.py: the reference Segmentor (with and without pyramid pooling) at 2x64x96, train mode:
    the four logit maps, and -- through the segmentation trainer's masked BCE (segmentation/train.py:184-193, evaluation.py:39-58) --
    the loss and gradient digests.  The encoder inside is oracle/standin_resnet.py (torchvision absent), like G3 / G5."""
    import importlib
    seg = importlib.import_module("footprints.preprocessing.segmentation.network")
    ev = importlib.import_module("footprints.preprocessing.segmentation.evaluation")
    out = {}
    B, H, W = 2, 64, 96
    image = torch.from_numpy(filler.uniform("g9:image", (B, 3, H, W)))
    gmask = torch.from_numpy(filler.bernoulli("g9:gmask", (B, H, W), 0.4))
    lmask = torch.from_numpy(filler.bernoulli("g9:lmask", (B, H, W), 0.7))
    for psp in (False, True):
        tag = "psp" if psp else "plain"
        P, Bf = R.make_seg_state(psp, tag="g9." + tag)
        m = seg.Segmentor(pretrained=False, use_PSP=psp)
        m.load_state_dict({**P, **Bf})
        m.train()
        outputs = m(image)
        preds = {}
        for scale, o in enumerate(outputs):                               # segmentation/train.py:184-190
            out.update(digest("seg.%s.out%d" % (tag, scale), o))
            o = torch.nn.functional.interpolate(o, size=(H, W), mode="bilinear", align_corners=False)
            preds[("ground", scale)] = o.squeeze(1)
        loss = ev.Evaluator().compute_losses(preds, gmask, lmask)
        loss.backward()
        out["seg.%s.loss" % tag] = np.float64(loss.item())
        g = dict(m.named_parameters())
        names = [k for k in g]
        out["seg.%s.param_names" % tag] = np.array(names)
        out["seg.%s.dead" % tag] = np.array([k for k in names if g[k].grad is None])
        out["seg.%s.grad_sums" % tag] = np.array([float(g[k].grad.double().sum()) if g[k].grad is not None else 0.0 for k in names])
        out["seg.%s.grad_abs" % tag] = np.array([float(g[k].grad.double().abs().sum()) if g[k].grad is not None else 0.0 for k in names])
        for k in ("decoder.outconv4.1.conv1.weight", "decoder.outconv1.conv1.weight", "decoder.block1.pre_concat_conv.conv1.weight",
                  "decoder.block4.post_concat_conv.conv1.weight") + (("decoder.PSP.block4.reduce.weight", "decoder.PSP.block1.reduce.weight") if psp else ()):
            out.update(digest("seg.%s.grad.%s" % (tag, k), g[k].grad))
    return out


def _data_standins():
    """stand-ins for the third-party modules the reference's dataset code imports and this image lacks (see g10_data_path)"""
    import random
    import types
    import scipy.ndimage
    from oracle import data_path as D
    ref_import.load_reference()
    cv2 = sys.modules["cv2"]
    cv2.INTER_NEAREST, cv2.INTER_AREA = 0, 3

    def resize(a, size, interpolation=None):
        assert (a.shape[1], a.shape[0]) == tuple(size), "the fixture's files are written at the target resolution"
        return np.ascontiguousarray(a)
    cv2.resize = resize
    sk = types.ModuleType("skimage"); skm = types.ModuleType("skimage.measure")
    skm.label = lambda m: scipy.ndimage.label(m, structure=np.ones((3, 3)))[0]
    sk.measure = skm
    sys.modules["skimage"], sys.modules["skimage.measure"] = sk, skm
    tvt = sys.modules["torchvision.transforms"]

    class ColorJitter:
        def __init__(self, brightness, contrast, saturation, hue):
            assert (brightness, contrast, saturation, hue) == D.JITTER_RANGES
        @staticmethod
        def get_params(brightness, contrast, saturation, hue):
            order, factors = D.jitter_params(random)
            return lambda img: D.jitter_pil(img, order, factors)
        def __call__(self, img):
            return self.get_params(*D.JITTER_RANGES)(img)

    class ToTensor:
        def __call__(self, pic):
            return torch.from_numpy(np.asarray(pic).copy()).permute(2, 0, 1).contiguous().float().div(255)
    tvt.ColorJitter, tvt.ToTensor = ColorJitter, ToTensor
    sys.modules["torchvision"].transforms = tvt


def g10_data_path():
    """datasets/kitti_dataset.py:44-122 + datasets/footprint_dataset.py:55-105: the reference's own KITTIDataset.__getitem__ (is_train=True)
    on synthetic files written at the target resolution (so every resize is the identity), Python RNG seeded.  Absent third-party
    modules are stood in: cv2.resize (identity, asserted), skimage.measure.label (scipy.ndimage.label, 8-connectivity) and
    torchvision.transforms.{ColorJitter,ToTensor} (oracle/data_path.py's restatement of torchvision 0.4.2 on top of the real Pillow)."""
    import importlib
    import random
    import tempfile
    import types
    from PIL import Image
    import scipy.ndimage
    from oracle import data_path as D
    from tests.golden.data_inputs import N_SAMPLES, SEED, H, W, sample_inputs
    _data_standins()
    kd = importlib.import_module("footprints.datasets.kitti_dataset")
    out = {"flags": np.zeros((N_SAMPLES, 2), np.int64)}
    with tempfile.TemporaryDirectory() as tmp:
        raw, tr = os.path.join(tmp, "raw"), os.path.join(tmp, "train")
        names = []
        for i in range(N_SAMPLES):
            img, maps = sample_inputs(i)
            seq, frame = "seq", "%010d" % i
            os.makedirs(os.path.join(raw, seq, "image_02", "data"), exist_ok=True)
            Image.fromarray(img, "RGB").save(os.path.join(raw, seq, "image_02", "data", frame + ".jpg"), format="PNG")   # lossless; PIL sniffs the content
            for sub, key, leaf in (("ground_seg", "visible_ground", "data"), ("hidden_depths", "ground_depth", "data"), ("depth_masks", "depth_mask", "data"),
                                   ("moving_objects", "moving_objects", "data"), ("stereo_matching_disps", "disparity", None)):
                d = os.path.join(tr, sub, seq, "image_02", *( [leaf] if leaf else [] ))
                os.makedirs(d, exist_ok=True)
                np.save(os.path.join(d, frame + ".npy"), maps[key])
            names.append("%s %d l" % (seq, i))
        random.seed(SEED)
        ds = kd.KITTIDataset(raw, tr, names, H, W, no_depth_mask=False, moving_objects_method="ours", project_down_baseline=False, is_train=True)
        # the fixture's depth masks are isolated pixels: filter_depth_mask (footprint_dataset.py:95-105) must keep them all
        for i in range(N_SAMPLES):
            dm = sample_inputs(i)[1]["depth_mask"]
            assert np.array_equal(ds.filter_depth_mask(dm), dm)
        state = random.getstate()
        probe = random.Random(); probe.setstate(state)
        for i in range(N_SAMPLES):
            out["flags"][i] = D.sample_augmentation(True, probe)
            if out["flags"][i][1]:
                D.jitter_params(probe)
            item = ds[i]
            for k, v in item.items():
                out["%d.%s" % (i, k)] = v.numpy()
    return out


def g11_data_path_matterport():
    """datasets/matterport_dataset.py:33-110: the reference's MatterportDataset.__getitem__ (is_train=True) on synthetic files at the
    target resolution (RGB png, 16-bit depth png, npy label maps), same stand-ins and RNG discipline as G10"""
    import importlib
    import random
    import tempfile
    from PIL import Image
    from oracle import data_path as D
    from tests.golden.data_inputs import N_SAMPLES, SEED, H, W, sample_inputs_matterport
    _data_standins()
    md = importlib.import_module("footprints.datasets.matterport_dataset")
    out = {"flags": np.zeros((N_SAMPLES, 2), np.int64)}
    with tempfile.TemporaryDirectory() as tmp:
        raw, tr = os.path.join(tmp, "raw"), os.path.join(tmp, "train")
        names = []
        for i in range(N_SAMPLES):
            img, maps = sample_inputs_matterport(i)
            scan, pos, hh, dd = "scan", "p%02d" % i, "1", "2"
            os.makedirs(os.path.join(raw, scan, scan, "matterport_color_images"), exist_ok=True)
            os.makedirs(os.path.join(raw, scan, scan, "matterport_depth_images"), exist_ok=True)
            Image.fromarray(img, "RGB").save(os.path.join(raw, scan, scan, "matterport_color_images", "%s_i%s_%s.jpg" % (pos, hh, dd)), format="PNG")
            Image.fromarray(maps["depth_raw"].astype(np.uint16)).save(os.path.join(raw, scan, scan, "matterport_depth_images", "%s_d%s_%s.png" % (pos, hh, dd)))
            for sub, key in (("ground_seg", "visible_ground"), ("hidden_depth", "ground_depth"), ("depth_masks", "depth_mask")):
                d = os.path.join(tr, sub, scan, "data")
                os.makedirs(d, exist_ok=True)
                np.save(os.path.join(d, "%s_%s_%s.npy" % (pos, hh, dd)), maps[key])
            names.append("%s %s %s %s" % (scan, pos, hh, dd))
        random.seed(SEED)
        ds = md.MatterportDataset(raw, tr, names, H, W, no_depth_mask=False, is_train=True)
        for i in range(N_SAMPLES):
            dm = sample_inputs_matterport(i)[1]["depth_mask"]
            assert np.array_equal(ds.filter_depth_mask(dm), dm)
        probe = random.Random(); probe.setstate(random.getstate())
        for i in range(N_SAMPLES):
            out["flags"][i] = D.sample_augmentation(True, probe)
            if out["flags"][i][1]:
                D.jitter_params(probe)
            item = ds[i]
            for k, v in item.items():
                out["%d.%s" % (i, k)] = v.numpy()
    return out


def main():
    mods = ref_import.load_reference()
    assert mods is not None, "needs /root/reference"
    net, loss_mod = mods
    torch.set_num_threads(8)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    only = sys.argv[1:]
    for name, fn in (("g1_blocks", lambda: g1_blocks(net)), ("g2_decoder", lambda: g2_decoder(net)),
                     ("g3_network", lambda: g3_network(net)), ("g4_loss", lambda: g4_loss(loss_mod)),
                     ("g5_train", lambda: g5_train(net, loss_mod)), ("g6_predict", lambda: g6_predict(net)),
                     ("g7_metrics", g7_metrics), ("g8_evaluator", g8_evaluator),
                     ("g9_segmentor", g9_segmentor), ("g10_data_path", g10_data_path),
                     ("g11_data_path_matterport", g11_data_path_matterport)):
        if only and name not in only:
            continue
        d = fn()
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, **d)
        print("%-12s %4d arrays %8.1f KB" % (name, len(d), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
