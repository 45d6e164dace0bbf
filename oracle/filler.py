"""Deterministic counter-hash filler (splitmix64 -> uniform float32).

Test infrastructure (see oracle/__init__.py).  Weights and inputs for the
golden fixtures are regenerated from this ~20-line generator on both sides
(fixture generation here, parity tests on the GPU box) so that no 124 MB
weight blob has to be committed.  Pure numpy, bit-reproducible.
"""
import hashlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def name_seed(name):
    return int.from_bytes(hashlib.sha256(name.encode()).digest()[:8], "little")


def uniform(name, shape, lo=0.0, hi=1.0):
    """float32 array of `shape`, U[lo, hi), fully determined by `name`."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + np.uint64(name_seed(name))
        bits = _splitmix64(ctr)
    u = (bits >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))  # 24-bit mantissa
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def bernoulli(name, shape, p):
    return (uniform(name, shape) < p).astype(np.float32)
