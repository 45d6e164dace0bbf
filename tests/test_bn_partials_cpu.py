"""CPU: the arithmetic of the BatchNorm statistics that come out of the tile convolution's epilogue (csrc/conv3x3_tile_bf3.hip, round 3),
restated in float32 numpy: a lane's 32 values -> two-pass (count, mean, M2) -> Chan merge with the other half-wave -> with the other
M wave -> per pixel tile; bn_stats_final_kernel's merge over the tiles (lane l takes tiles l, l + 64, ..., then a shuffle tree).  The
property checked is the one the GPU test cannot show on benign data: mean and variance stay within 2e-6 of float64 when the channel
mean is a thousand standard deviations away from zero (sum / sum-of-squares partials would lose the variance entirely there)."""
import numpy as np
import pytest

f32 = np.float32


def wf_merge(a, b):
    n = f32(a[0] + b[0])
    if n == 0:
        return a
    d = f32(b[1] - a[1])
    f = f32(b[0] / n)
    mean = f32(a[1] + d * f)
    m2 = f32(a[2] + f32(b[2] + f32(f32(d * d) * f32(a[0] * f))))
    return (n, mean, m2)


def lane_stats(v):                       # two passes over the lane's registers
    v = v.astype(f32)
    cnt = f32(len(v))
    s = f32(0)
    for x in v:
        s = f32(s + x)
    mean = f32(s / cnt) if cnt > 0 else f32(0)
    m2 = f32(0)
    for x in v:
        d = f32(x - mean)
        m2 = f32(m2 + d * d)
    return (cnt, mean, m2)


def tile_partial(tile):                  # tile: 128 pixels of one channel = 2 M waves x 2 half-waves x 32 registers' worth... 4 x 32 values
    lanes = [lane_stats(tile[i * 32:(i + 1) * 32]) for i in range(4)]
    w0 = wf_merge(lanes[0], lanes[1])    # half-waves of M wave 0
    w1 = wf_merge(lanes[2], lanes[3])
    return wf_merge(w0, w1)              # M waves in order


def final_merge(parts):                  # bn_stats_final_kernel: lane l merges partials l, l + 64, ...; then offsets 32, 16, ... 1
    lanes = []
    for l in range(64):
        w = (f32(0), f32(0), f32(0))
        for b in range(l, len(parts), 64):
            w = wf_merge(w, parts[b])
        lanes.append(w)
    o = 32
    while o > 0:
        lanes = [wf_merge(lanes[l], lanes[l + o]) if l + o < 64 else lanes[l] for l in range(64)]
        o >>= 1
    return lanes[0]


@pytest.mark.parametrize("mean,std,tiles", [(0.0, 1.0, 180), (3.0, 1.0, 720), (1000.0, 1.0, 180), (-250.0, 0.03, 96), (1e-3, 1e-6, 64)])
def test_tile_welford_partials_match_float64(mean, std, tiles):
    rng = np.random.default_rng(17)
    x = (rng.standard_normal(tiles * 128) * std + mean).astype(f32)
    n, m, m2 = final_merge([tile_partial(x[t * 128:(t + 1) * 128]) for t in range(tiles)])
    xd = x.astype(np.float64)
    assert n == tiles * 128
    assert abs(m - xd.mean()) <= 2e-6 * max(abs(xd.mean()), xd.std())
    assert abs(m2 / n - xd.var()) <= 2e-6 * xd.var() + 1e-30
    # what plain sums would have given in float32 (the form the kernel does NOT use): shown to fail where the mean dominates
    if abs(mean) >= 1000 * std:
        s, q = f32(0), f32(0)
        for v in x[:4096]:
            s = f32(s + v)
            q = f32(q + v * v)
        naive = q / f32(4096) - (s / f32(4096)) ** 2
        assert abs(naive - xd[:4096].var()) > 1e-2 * xd[:4096].var()


def test_empty_lanes_and_ragged_tiles_merge_cleanly():
    """a lane whose pixels all lie outside the image contributes (0, 0, 0); merging it on either side changes nothing"""
    a, z = (f32(7), f32(1.5), f32(0.25)), (f32(0), f32(0), f32(0))
    assert wf_merge(a, z) == a and wf_merge(z, a) == a and wf_merge(z, z) == z
