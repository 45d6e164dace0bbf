"""CPU restatement of the Footprints hot path (functional, NCHW, plain torch ops).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Every function cites the
reference file:line it follows (paths relative to /root/reference).  The
restatement is *functional*: parameters and buffers live in flat dicts keyed by
the reference's ``state_dict`` names, so the same dict
feeds this oracle, the golden fixtures and the HIP product path.

Backward is obtained the way the reference obtains it -- torch autograd on the
CPU ops (footprints/training/train.py:153-156) -- and the optimiser is the same
``torch.optim.Adam`` the reference constructs (footprints/model_manager.py:27).
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import filler

SCALES = ("1/8", "1/4", "1/2", "1/1")
RESNET34_BLOCKS = (3, 4, 6, 3)          # torchvision resnet34 (network.py:10,38)
RESNET34_PLANES = (64, 128, 256, 512)


# --------------------------------------------------------------------------
# state_dict layout (network.py:40-44,69-80,109-113,169)
# --------------------------------------------------------------------------
def _bn_entries(prefix, c):
    return [(prefix + ".weight", (c,), "bn_w"), (prefix + ".bias", (c,), "bn_b"),
            (prefix + ".running_mean", (c,), "bn_rm"), (prefix + ".running_var", (c,), "bn_rv"),
            (prefix + ".num_batches_tracked", (), "bn_nbt")]


def encoder_block_prefixes():
    """[(prefix, cin, cout, stride, has_downsample)] for the 16 BasicBlocks, forward order."""
    out = []
    cin = 64
    for li, (nblk, planes) in enumerate(zip(RESNET34_BLOCKS, RESNET34_PLANES), start=1):
        for b in range(nblk):
            stride = 2 if (b == 0 and li > 1) else 1
            # layer1 = Sequential(maxpool, layer1) -> extra ".1." (network.py:41)
            prefix = "encoder.layer1.1.%d" % b if li == 1 else "encoder.layer%d.%d" % (li, b)
            out.append((prefix, cin, planes, stride, b == 0 and li > 1))
            cin = planes
    return out


def _convblock_entries(prefix, cin, cout):
    e = [(prefix + ".conv1.weight", (cout, cin, 3, 3), "conv_w"), (prefix + ".conv1.bias", (cout,), "conv_b")]
    e += _bn_entries(prefix + ".bn1", cout)            # dead: use_bn=False (network.py:69-72,128)
    e += [(prefix + ".conv2.weight", (cout, cout, 3, 3), "conv_w"), (prefix + ".conv2.bias", (cout,), "conv_b")]
    e += _bn_entries(prefix + ".bn2", cout)
    return e


def state_spec():
    """Ordered list of (key, shape, kind) -- the 484 state_dict entries of FootprintNetwork."""
    spec = [("encoder.layer0.0.weight", (64, 3, 7, 7), "conv_w")]
    spec += _bn_entries("encoder.layer0.1", 64)
    for prefix, cin, cout, stride, ds in encoder_block_prefixes():
        spec.append((prefix + ".conv1.weight", (cout, cin, 3, 3), "conv_w"))
        spec += _bn_entries(prefix + ".bn1", cout)
        spec.append((prefix + ".conv2.weight", (cout, cout, 3, 3), "conv_w"))
        spec += _bn_entries(prefix + ".bn2", cout)
        if ds:
            spec.append((prefix + ".downsample.0.weight", (cout, cin, 1, 1), "conv_w"))
            spec += _bn_entries(prefix + ".downsample.1", cout)
    for dec in ("mask_decoder", "depth_decoder"):
        for i, (cin, cout) in enumerate(((512, 256), (256, 128), (128, 64), (64, 64)), start=1):
            spec += _convblock_entries("%s.block%d.pre_concat_conv" % (dec, i), cin, cout)
            spec += _convblock_entries("%s.block%d.post_concat_conv" % (dec, i), 2 * cout, cout)
        for i, cin in ((1, 128), (2, 64), (3, 64)):
            spec += [("%s.outconv%d.conv1.weight" % (dec, i), (2, cin, 3, 3), "conv_w"),
                     ("%s.outconv%d.conv1.bias" % (dec, i), (2,), "conv_b")]
        spec += _convblock_entries("%s.outconv4.0" % dec, 64, 32)
        spec += [("%s.outconv4.1.conv1.weight" % dec, (2, 32, 3, 3), "conv_w"),
                 ("%s.outconv4.1.conv1.bias" % dec, (2,), "conv_b")]
    return spec


def is_dead_param(key):
    """Decoder bn1/bn2 affine params exist but are never used (network.py:128,134)."""
    return ("decoder" in key) and (".bn1." in key or ".bn2." in key)


def make_state(tag="w", dtype=torch.float32):
    """Deterministic 'random-init-like' state (counter-hash filler, not stored anywhere).

    Conv weights ~ U(-b, b) with b = sqrt(3/fan_in) (variance-preserving-ish so a
    58-layer forward neither explodes nor dies); biases small; BN gamma in
    [0.5,1.5], beta in [-0.2,0.2]; running stats non-trivial so eval mode is tested.
    """
    params, buffers = OrderedDict(), OrderedDict()
    for key, shape, kind in state_spec():
        name = tag + ":" + key
        if kind == "conv_w":
            fan_in = shape[1] * shape[2] * shape[3]
            b = float(np.sqrt(3.0 / fan_in))
            params[key] = torch.from_numpy(filler.uniform(name, shape, -b, b)).to(dtype)
        elif kind == "conv_b":
            params[key] = torch.from_numpy(filler.uniform(name, shape, -0.1, 0.1)).to(dtype)
        elif kind == "bn_w":
            params[key] = torch.from_numpy(filler.uniform(name, shape, 0.5, 1.5)).to(dtype)
        elif kind == "bn_b":
            params[key] = torch.from_numpy(filler.uniform(name, shape, -0.2, 0.2)).to(dtype)
        elif kind == "bn_rm":
            buffers[key] = torch.from_numpy(filler.uniform(name, shape, -0.3, 0.3)).to(dtype)
        elif kind == "bn_rv":
            buffers[key] = torch.from_numpy(filler.uniform(name, shape, 0.5, 2.0)).to(dtype)
        elif kind == "bn_nbt":
            buffers[key] = torch.zeros((), dtype=torch.int64)
    return params, buffers


def make_batch(B, H, W, tag="batch"):
    """Synthetic batch with the reference schema (datasets/footprint_dataset.py:55-65,
    datasets/kitti_dataset.py:114-122; distributions from SURVEY.md section 8d)."""
    u, bern = filler.uniform, filler.bernoulli
    t = OrderedDict()
    t["image"] = torch.from_numpy(u(tag + ":image", (B, 3, H, W)))
    t["visible_ground"] = torch.from_numpy(bern(tag + ":vg", (B, H, W), 0.4))
    t["depth"] = torch.from_numpy(u(tag + ":depth", (B, H, W), 0.0, 80.0) * bern(tag + ":dv", (B, H, W), 0.8))
    t["ground_depth"] = torch.from_numpy(u(tag + ":gdepth", (B, H, W), 0.0, 30.0) * bern(tag + ":gv", (B, H, W), 0.5))
    t["moving_object_mask"] = torch.from_numpy(bern(tag + ":mov", (B, H, W), 0.05))
    t["depth_mask"] = torch.from_numpy(bern(tag + ":dm", (B, H, W), 0.1))
    t["all_ground"] = ((t["ground_depth"] + t["visible_ground"]) > 0).float()
    return t


# --------------------------------------------------------------------------
# encoder: ResNet-34 restated from the published architecture
# (call site network.py:35-59; torchvision 0.4.2 resnet.py is third-party)
# --------------------------------------------------------------------------
def _bn(x, P, B, prefix, training, momentum=0.1, eps=1e-5):
    rm, rv = B[prefix + ".running_mean"], B[prefix + ".running_var"]
    y = F.batch_norm(x, rm, rv, P[prefix + ".weight"], P[prefix + ".bias"], training, momentum, eps)
    if training:
        B[prefix + ".num_batches_tracked"] += 1
    return y


class ReluDecisions:
    """The encoder's discrete decisions.  impose: list of boolean ReLU masks consumed in call order (None: the network decides itself);
    taken: the masks actually used; pool_impose (optional, int64 [N, C, OH, OW]): the 3x3 / stride 2 max-pool's winning window position
    ky * 3 + kx per output element (two window elements within round-off of each other are the same kind of decision as a ReLU at zero)."""

    def __init__(self, impose=None, pool_impose=None, probe=None):
        self.impose = None if impose is None else iter(impose)
        self.pool_impose = pool_impose
        self.taken = []
        # probe (optional, recording runs only): somebody else's ReLU masks in call order -- the run keeps its OWN decisions and only measures
        # how many of the probe's differ and how far from zero this run's value of the worst one sits (same statistics as for imposed masks:
        # the reference's fp32 decisions measured against the float64 run without a float64 run of their own, tests/parity.py)
        self.probe = None if probe is None else iter(probe)
        self.probe_flips = 0
        self.probe_flip_worst = 0.0
        self.probe_flip_where = None
        # round 6: how far from the decision boundary every IMPOSED decision that differs from this run's own one sits (this run being
        # the float64 oracle, that is the float64 value the other implementation resolved differently): an imposed mask is only
        # legitimate where |z| is at round-off level, an imposed pool winner only where it ties the true maximum (tests/parity.py bounds both)
        self.relu_flips = 0
        self.relu_flip_worst = 0.0        # max |z| / RMS of z's channel over the flipped elements
        self.relu_flip_where = None       # (index of the ReLU in call order, channel) of the worst one
        self.pool_flips = 0
        self.pool_flip_worst = 0.0        # max (true max - imposed winner) / RMS of the channel


def _maxpool(x, decisions):
    """encoder.maxpool (network.py:41: MaxPool2d(3, 2, 1)), or -- decisions.pool_impose given -- the same gather with imposed winners"""
    if decisions is None or decisions.pool_impose is None:
        return F.max_pool2d(x, 3, 2, 1)
    N, C, H, W = x.shape
    OH, OW = (H + 1) // 2, (W + 1) // 2
    idx = decisions.pool_impose
    assert tuple(idx.shape) == (N, C, OH, OW), (tuple(idx.shape), (N, C, OH, OW))
    patches = F.unfold(F.pad(x, (1, 1, 1, 1), value=float("-inf")), 3, stride=2).view(N, C, 9, OH, OW)   # padded positions never win
    out = patches.gather(2, idx.unsqueeze(2).to(torch.int64)).squeeze(2)
    with torch.no_grad():
        gap = patches.max(2).values - out                                # >= 0; > 0 where the imposed winner is not this run's maximum
        lose = gap > 0
        n = int(lose.sum())
        if n:
            rms = x.pow(2).mean(dim=(0, 2, 3), keepdim=True).sqrt().clamp_min(1e-300)
            decisions.pool_flips += n
            decisions.pool_flip_worst = max(decisions.pool_flip_worst, float((gap / rms)[lose].max()))
    return out


def _relu(x, decisions):
    """F.relu, or -- decisions.impose given -- x * mask with the NEXT imposed mask: the piecewise-linear network evaluated with somebody
    else's ReLU decisions (tests/parity.py: the float64 truth under the decisions a float32 implementation took; where the mask equals
    x > 0 this IS relu, forward and backward).  A ReluDecisions without `impose` only records what F.relu decided."""
    if decisions is None:
        return F.relu(x)
    if decisions.impose is None:
        own = (x > 0).detach()
        if decisions.probe is not None:
            with torch.no_grad():
                m = next(decisions.probe)
                flipped = m.to(torch.bool) != own
                n = int(flipped.sum())
                if n:
                    rms = x.pow(2).mean(dim=(0, 2, 3), keepdim=True).sqrt().clamp_min(1e-300)
                    dist = torch.where(flipped, x.abs() / rms, torch.zeros((), dtype=x.dtype))
                    worst = float(dist.max())
                    decisions.probe_flips += n
                    if worst > decisions.probe_flip_worst:
                        decisions.probe_flip_worst = worst
                        decisions.probe_flip_where = (len(decisions.taken), int(dist.amax(dim=(0, 2, 3)).argmax()))
        decisions.taken.append(own)
        return F.relu(x)
    m = next(decisions.impose)
    assert m.shape == x.shape, (tuple(m.shape), tuple(x.shape))
    with torch.no_grad():
        flipped = m.to(torch.bool) != (x > 0)
        n = int(flipped.sum())
        if n:
            rms = x.pow(2).mean(dim=(0, 2, 3), keepdim=True).sqrt().clamp_min(1e-300)
            dist = torch.where(flipped, x.abs() / rms, torch.zeros((), dtype=x.dtype))
            worst = float(dist.max())
            decisions.relu_flips += n
            if worst > decisions.relu_flip_worst:
                decisions.relu_flip_worst = worst
                decisions.relu_flip_where = (len(decisions.taken), int(dist.amax(dim=(0, 2, 3)).argmax()))
    decisions.taken.append(m)
    return x * m.to(x.dtype)


def _basic_block(x, P, B, prefix, stride, has_ds, training, decisions=None):
    out = F.conv2d(x, P[prefix + ".conv1.weight"], None, stride, 1)
    out = _relu(_bn(out, P, B, prefix + ".bn1", training), decisions)
    out = F.conv2d(out, P[prefix + ".conv2.weight"], None, 1, 1)
    out = _bn(out, P, B, prefix + ".bn2", training)
    if has_ds:
        idt = F.conv2d(x, P[prefix + ".downsample.0.weight"], None, stride, 0)
        idt = _bn(idt, P, B, prefix + ".downsample.1", training)
    else:
        idt = x
    return _relu(out + idt, decisions)


def resnet_encoder(image, P, B, training, record=None, relu_decisions=None):
    """network.py:48-59 -- normalise, layer0 (conv7x7/2+BN+ReLU), layer1 (maxpool + 3 blocks), layer2..4.
    record (optional list): receives every BasicBlock output (grad retained) for block-level debugging.
    relu_decisions (optional ReluDecisions; 33 boolean NCHW masks: stem, then (bn1's ReLU, block output ReLU) per BasicBlock): evaluate
    the encoder with imposed ReLU decisions instead of its own and / or record the decisions taken (see _relu)."""
    decisions = relu_decisions
    x = (image - 0.45) / 0.225                                            # network.py:50
    x = F.conv2d(x, P["encoder.layer0.0.weight"], None, 2, 3)
    x = _relu(_bn(x, P, B, "encoder.layer0.1", training), decisions)
    feats = [x]
    blocks = encoder_block_prefixes()
    bi = 0
    for li, nblk in enumerate(RESNET34_BLOCKS, start=1):
        if li == 1:
            x = _maxpool(x, decisions)                                    # encoder.maxpool (network.py:41)
        for _ in range(nblk):
            prefix, cin, cout, stride, ds = blocks[bi]
            x = _basic_block(x, P, B, prefix, stride, ds, training, decisions)
            if record is not None:
                if x.requires_grad:
                    x.retain_grad()
                record.append(x)
            bi += 1
        feats.append(x)
    return feats


# --------------------------------------------------------------------------
# decoder (network.py:62-183)
# --------------------------------------------------------------------------
def conv_block(x, P, prefix):
    """ConvBlock.forward network.py:124-138 with use_bn=False, ELU."""
    x = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), P[prefix + ".conv1.weight"], P[prefix + ".conv1.bias"])
    x = F.elu(x)
    x = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), P[prefix + ".conv2.weight"], P[prefix + ".conv2.bias"])
    return F.elu(x)


def up_concat_block(x, skip, P, prefix):
    """ConvUpsampleAndConcatBlock.forward network.py:151-158."""
    x = conv_block(x, P, prefix + ".pre_concat_conv")
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    x = torch.cat([x, skip], 1)
    return conv_block(x, P, prefix + ".post_concat_conv")


def out_conv_block(x, P, prefix, scale, apply_sigmoid):
    """OutConvBlock.forward network.py:174-183 (sigmoid BEFORE the bilinear upsample)."""
    x = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), P[prefix + ".conv1.weight"], P[prefix + ".conv1.bias"])
    if apply_sigmoid:
        x = torch.sigmoid(x)
    if scale != 1:
        x = F.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=False)
    return x


def skip_decoder(feats, P, prefix, apply_sigmoid):
    """SkipDecoder.forward network.py:82-101."""
    out = OrderedDict()
    x = up_concat_block(feats[4], feats[3], P, prefix + ".block1")
    x = up_concat_block(x, feats[2], P, prefix + ".block2")
    out["1/8"] = out_conv_block(x, P, prefix + ".outconv1", 8, apply_sigmoid)
    x = up_concat_block(x, feats[1], P, prefix + ".block3")
    out["1/4"] = out_conv_block(x, P, prefix + ".outconv2", 4, apply_sigmoid)
    x = up_concat_block(x, feats[0], P, prefix + ".block4")
    out["1/2"] = out_conv_block(x, P, prefix + ".outconv3", 2, apply_sigmoid)
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    x = conv_block(x, P, prefix + ".outconv4.0")
    out["1/1"] = out_conv_block(x, P, prefix + ".outconv4.1", 1, apply_sigmoid)
    return out


def footprint_network(image, P, B, training=True, return_features=False, record=None, relu_decisions=None):
    """FootprintNetwork.forward network.py:21-30."""
    feats = resnet_encoder(image, P, B, training, record, relu_decisions)
    m = skip_decoder(feats, P, "mask_decoder", False)       # network.py:18
    d = skip_decoder(feats, P, "depth_decoder", True)       # network.py:19
    out = OrderedDict((k, torch.cat([m[k], d[k]], 1)) for k in m)
    return (out, feats) if return_features else out


# --------------------------------------------------------------------------
# loss (training/losses.py:31-152, utils.py:36-42)
# --------------------------------------------------------------------------
def sigmoid_to_depth(disp, min_depth=0.1, max_depth=100):
    """utils.py:36-42."""
    min_disp = 1 / max_depth
    max_disp = 1 / min_depth
    return 1 / (min_disp + (max_disp - min_disp) * disp)


def _bce_logits(x, t):
    """BCEWithLogitsLoss(reduction='none') closed form (losses.py:114)."""
    return torch.clamp(x, min=0) - x * t + torch.log1p(torch.exp(-torch.abs(x)))


def loss_manager(predictions, targets, depth_range=(0.1, 100), prior=0.25):
    """LossManager.__call__ losses.py:31-92.  Returns (losses[21], viz[20]); does not mutate inputs."""
    min_d, max_d = depth_range
    losses, viz = OrderedDict(), OrderedDict()
    depth, gdepth = targets["depth"], targets["ground_depth"]
    valid_d = (depth > 0).float()                                   # losses.py:38
    valid_g = (gdepth > 0).float()                                  # losses.py:47
    vg, ag = targets["visible_ground"], targets["all_ground"]
    keep = 1 - targets["moving_object_mask"]                        # losses.py:44 (inverted)
    dm = targets["depth_mask"]
    total = 0
    for k, o in predictions.items():
        if not isinstance(k, str):
            continue
        # ch0: visible ground, plain BCE mean (losses.py:53-56,110-126)
        losses[("visible_ground", k)] = _bce_logits(o[:, 0], vg).mean()
        viz[("visible_ground", k)] = torch.sigmoid(o[:, 0])
        # ch1: ThreeClassLoss (losses.py:129-152)
        m = ((ag + dm) > 0).float()
        l1 = _bce_logits(o[:, 1], ag) * m * keep + prior * _bce_logits(o[:, 1], torch.zeros_like(ag)) * (1 - m)
        losses[("all_ground", k)] = l1.mean()
        viz[("all_ground", k)] = torch.sigmoid(o[:, 1])
        # ch2 / ch3: log-L1 on depth (losses.py:66-73,95-107); mean over ALL pixels
        d2 = sigmoid_to_depth(o[:, 2], min_d, max_d)
        losses[("depth", k)] = (torch.log(torch.abs(d2 - depth) + 1) * valid_d).mean()
        viz[("depth", k)] = d2
        d3 = sigmoid_to_depth(o[:, 3], min_d, max_d)
        losses[("ground_depth", k)] = (torch.log(torch.abs(d3 - gdepth) + 1) * valid_g).mean()
        viz[("ground_depth", k)] = d3
        viz[("ground_depth_masked", k)] = d3 * (viz[("all_ground", k)] > 0.5).float()   # losses.py:76-78
        losses[("loss", k)] = (losses[("depth", k)] + losses[("visible_ground", k)]
                               + losses[("all_ground", k)] + losses[("ground_depth", k)])  # losses.py:80-83
        total = total + losses[("loss", k)]
    losses["loss"] = total / 4                                      # losses.py:87-88
    return losses, viz


LOSS_KEYS = [(n, s) for s in SCALES for n in ("visible_ground", "all_ground", "depth", "ground_depth", "loss")] + ["loss"]


# --------------------------------------------------------------------------
# train step (training/train.py:150-156; model_manager.py:27)
# --------------------------------------------------------------------------
class OracleTrainer:
    """fwd + loss + zero_grad + backward + Adam on the functional restatement."""

    def __init__(self, params, buffers, lr=1e-4):
        self.P = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in params.items())
        self.B = OrderedDict((k, v.clone()) for k, v in buffers.items())
        # the reference hands *all* parameters to Adam; dead ones simply never get a grad
        self.opt = torch.optim.Adam(list(self.P.values()), lr=lr)

    def step(self, batch, training=True):
        out = footprint_network(batch["image"], self.P, self.B, training)
        losses, _ = loss_manager(out, batch)
        self.opt.zero_grad(set_to_none=True)
        losses["loss"].backward()
        self.opt.step()
        return out, losses

    def forward_backward(self, batch, training=True):
        out = footprint_network(batch["image"], self.P, self.B, training)
        losses, _ = loss_manager(out, batch)
        for p in self.P.values():
            p.grad = None
        losses["loss"].backward()
        return out, losses


# --------------------------------------------------------------------------
# Evaluator (training/evaluation.py:14-67)
# --------------------------------------------------------------------------
class OracleEvaluator:
    """compute_losses / get_averaged_losses bookkeeping of training/evaluation.py:28-67 over `loss_manager`."""

    def __init__(self, depth_range=(0.1, 100), prior=0.25):
        self.depth_range, self.prior = depth_range, prior
        self.acc = {"train": {}, "val": {}}

    def compute_losses(self, inputs, outputs, mode="train", return_batch_loss=False):
        losses, _ = loss_manager(outputs, inputs, self.depth_range, self.prior)
        if mode in self.acc:
            for k, v in losses.items():                                  # evaluation.py:38-43
                self.acc[mode].setdefault(k, []).append(v.detach().cpu())
        if return_batch_loss:                                            # evaluation.py:45-46
            return losses

    def get_averaged_losses(self, mode, reset=True):
        out = {k: float(torch.stack(v).mean().numpy()) for k, v in self.acc.get(mode, {}).items()}   # evaluation.py:52-55
        if reset and mode in self.acc:
            self.acc[mode] = {}
        return out


# --------------------------------------------------------------------------
# ground-segmentation network (preprocessing/segmentation/network.py:13-207): same encoder and decoder blocks,
# 1-channel heads at their own resolution, optional pyramid pooling
# --------------------------------------------------------------------------
def seg_state_spec(use_psp):
    """ordered (key, shape, kind) of Segmentor(use_PSP).state_dict()"""
    spec = [e for e in state_spec() if e[0].startswith("encoder.")]
    dec = "decoder"
    if use_psp:
        spec += [("%s.PSP.block%d.reduce.weight" % (dec, i), (128, 512, 1, 1), "conv_w") for i in (1, 2, 3, 4)]
    for i, (cin, cout) in enumerate(((1024 if use_psp else 512, 256), (256, 128), (128, 64), (64, 64)), start=1):
        spec += _convblock_entries("%s.block%d.pre_concat_conv" % (dec, i), cin, cout)
        spec += _convblock_entries("%s.block%d.post_concat_conv" % (dec, i), 2 * cout, cout)
    for i, cin in ((1, 128), (2, 64), (3, 64)):
        spec += [("%s.outconv%d.conv1.weight" % (dec, i), (1, cin, 3, 3), "conv_w"), ("%s.outconv%d.conv1.bias" % (dec, i), (1,), "conv_b")]
    spec += _convblock_entries("%s.outconv4.0" % dec, 64, 32)
    spec += [("%s.outconv4.1.conv1.weight" % dec, (1, 32, 3, 3), "conv_w"), ("%s.outconv4.1.conv1.bias" % dec, (1,), "conv_b")]
    return spec


def make_seg_state(use_psp, tag="seg", dtype=torch.float32):
    """deterministic state like make_state, for the Segmentor's keys"""
    params, buffers = OrderedDict(), OrderedDict()
    for key, shape, kind in seg_state_spec(use_psp):
        name = tag + ":" + key
        if kind == "conv_w":
            b = float(np.sqrt(3.0 / (shape[1] * shape[2] * shape[3])))
            params[key] = torch.from_numpy(filler.uniform(name, shape, -b, b)).to(dtype)
        elif kind == "conv_b":
            params[key] = torch.from_numpy(filler.uniform(name, shape, -0.1, 0.1)).to(dtype)
        elif kind == "bn_w":
            params[key] = torch.from_numpy(filler.uniform(name, shape, 0.5, 1.5)).to(dtype)
        elif kind == "bn_b":
            params[key] = torch.from_numpy(filler.uniform(name, shape, -0.2, 0.2)).to(dtype)
        elif kind == "bn_rm":
            buffers[key] = torch.from_numpy(filler.uniform(name, shape, -0.3, 0.3)).to(dtype)
        elif kind == "bn_rv":
            buffers[key] = torch.from_numpy(filler.uniform(name, shape, 0.5, 2.0)).to(dtype)
        elif kind == "bn_nbt":
            buffers[key] = torch.zeros((), dtype=torch.int64)
    return params, buffers


def psp_module(x, P, prefix):
    """PSP.forward segmentation/network.py:198-207 (PSPBlock :183-190)"""
    h, w = x.shape[2:]
    outs = {}
    for name, size in (("block1", 1), ("block2", 2), ("block3", 4), ("block4", 6)):
        y = F.conv2d(F.adaptive_avg_pool2d(x, (size, size)), P["%s.%s.reduce.weight" % (prefix, name)])
        outs[size] = F.interpolate(y, size=(h, w), mode="bilinear", align_corners=True)
    return torch.cat([x, outs[6], outs[4], outs[2], outs[1]], 1)


def segmentor(image, P, B, training=True, use_psp=False, record=None):
    """Segmentor.forward segmentation/network.py:20-25 -> [1/8, 1/4, 1/2, 1/1] logit maps, each [B,1,h,w] (:84-99);
    record: see resnet_encoder"""
    feats = resnet_encoder(image, P, B, training, record)
    d = "decoder"
    x = feats[4]
    if use_psp:
        x = psp_module(x, P, d + ".PSP")
    outs = []
    x = up_concat_block(x, feats[3], P, d + ".block1")
    x = up_concat_block(x, feats[2], P, d + ".block2")
    outs.append(out_conv_block(x, P, d + ".outconv1", 1, False))
    x = up_concat_block(x, feats[1], P, d + ".block3")
    outs.append(out_conv_block(x, P, d + ".outconv2", 1, False))
    x = up_concat_block(x, feats[0], P, d + ".block4")
    outs.append(out_conv_block(x, P, d + ".outconv3", 1, False))
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    x = conv_block(x, P, d + ".outconv4.0")
    outs.append(out_conv_block(x, P, d + ".outconv4.1", 1, False))
    return outs


def seg_loss(outputs, ground_mask, loss_mask, height, width):
    """segmentation/train.py:184-193 + segmentation/evaluation.py:39-58: predictions up-sized bilinearly (align_corners=False) to
    (height, width), per-image masked BCE-with-logits means, averaged over the four scales, then over the batch"""
    total = 0
    for out in outputs:
        pred = F.interpolate(out, size=(height, width), mode="bilinear", align_corners=False).squeeze(1)
        loss = F.binary_cross_entropy_with_logits(pred, ground_mask, reduction="none")
        valid = loss_mask.sum(dim=[1, 2])
        total = total + (loss * loss_mask).sum(dim=[1, 2]) / (valid + 1e-7)
    return (total / 4).mean()
