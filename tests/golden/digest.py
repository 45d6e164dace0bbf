"""Compact, comparison-friendly digests of large tensors for the golden fixtures.

A tensor with <= full_limit elements is stored whole; a larger one is stored as
float64 sum, float64 abs-sum and a strided sample (stride 61, coprime to every
power-of-two / small-factor dimension in the network, so the samples walk all
channels and positions).  The same function digests oracle / HIP results in tests.
"""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))
STRIDE = 61


def fill(name, shape, lo=-1.0, hi=1.0):
    from oracle import filler
    return torch.from_numpy(filler.uniform(name, shape, lo, hi))


def fill_value(tag, key, shape, is_int=False):
    """Deterministic value for state entry `key` of a block fixture (shared by the
    generator and the tests so both sides rebuild identical weights)."""
    if is_int:
        return torch.zeros(shape, dtype=torch.int64)
    if key.endswith("running_var"):
        return fill(tag + ":" + key, shape, 0.5, 2.0)
    if key.endswith("weight") and len(shape) == 4:
        b = float(np.sqrt(3.0 / (shape[1] * shape[2] * shape[3])))
        return fill(tag + ":" + key, shape, -b, b)
    return fill(tag + ":" + key, shape, -0.3, 0.3)


def digest(name, t, full_limit=1 << 15):
    t = t.detach().cpu().contiguous()
    flat = t.reshape(-1)
    if flat.numel() <= full_limit:
        return {name: t.numpy().copy()}
    d = flat.double()
    return {name + "#sum": np.float64(d.sum().item()), name + "#abs": np.float64(d.abs().sum().item()),
            name + "#sample": flat[::STRIDE].numpy().copy(), name + "#shape": np.array(t.shape)}


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)


def compare(gold, name, t, rtol=1e-4, atol_scale=1e-4):
    """Max-normalised error of tensor `t` against golden entry `name` (full or digest form).

    Returns err = max|a-b| / max|b| (tensor-scale relative error).  For digests also
    checks the float64 sum against abs-sum scale.  Asserts err <= rtol.
    """
    t = t.detach().cpu().contiguous()
    if name in gold.files:
        ref = torch.from_numpy(gold[name])
        assert tuple(ref.shape) == tuple(t.shape), (name, ref.shape, t.shape)
        scale = max(ref.abs().max().item(), 1e-30)
        err = (t.double() - ref.double()).abs().max().item() / scale
        assert err <= rtol, "%s: rel-to-max error %.3e > %.1e" % (name, err, rtol)
        return err
    shape = tuple(gold[name + "#shape"])
    assert shape == tuple(t.shape), (name, shape, t.shape)
    ref = torch.from_numpy(gold[name + "#sample"])
    got = t.reshape(-1)[::STRIDE]
    scale = max(ref.abs().max().item(), 1e-30)
    err = (got.double() - ref.double()).abs().max().item() / scale
    assert err <= rtol, "%s: sample rel-to-max error %.3e > %.1e" % (name, err, rtol)
    s_ref, a_ref = float(gold[name + "#sum"]), float(gold[name + "#abs"])
    s = t.double().sum().item()
    serr = abs(s - s_ref) / max(a_ref, 1e-30)
    assert serr <= atol_scale, "%s: sum error %.3e (abs-sum scale)" % (name, serr)
    return err
