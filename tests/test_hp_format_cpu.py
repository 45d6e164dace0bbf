"""CPU: the numerical contract of the fp16-pair operand format (footprints_amd/csrc/fp_common.h), restated in
numpy: scale selection from the amax bit pattern, the split, its error bounds over the whole fp32 range, and the accuracy of the
three-product dot product against float64 next to a plain float32 dot product.  The GPU kernels are checked against float64 in
tests/test_gpu_hp.py; this file pins the arithmetic they implement."""
import numpy as np

TARGET_ACT, TARGET_W = 12, 11          # FP_HP_TARGET_ACT / FP_HP_TARGET_W


def hp_exponent(amax, target):
    """fp_hp_exponent: k with amax * 2^k in [2^target, 2^(target+1)); 0 for an all-zero tensor"""
    bits = np.float32(amax).view(np.uint32)
    return 0 if bits == 0 else target - (int(bits >> 23) - 127)


def split(x, k):
    """x * 2^k = h + m (+ error): h = fp16(x * 2^k), m = fp16(x * 2^k - h), both round-to-nearest-even like v_cvt_f16_f32"""
    xs = np.ldexp(x.astype(np.float32), k).astype(np.float32)
    h = xs.astype(np.float16)
    m = (xs - h.astype(np.float32)).astype(np.float16)
    return h, m


def test_scale_selection_maps_amax_into_the_target_binade():
    rng = np.random.default_rng(0)
    for amax in list(np.exp2(rng.uniform(-100, 100, 200)).astype(np.float32)) + [np.float32(1.0), np.float32(65504.0), np.float32(3e38)]:
        for target in (TARGET_ACT, TARGET_W):
            k = hp_exponent(amax, target)
            v = np.ldexp(np.float64(amax), k)
            assert 2.0 ** target <= v < 2.0 ** (target + 1), (amax, k, v)
    assert hp_exponent(np.float32(0.0), TARGET_ACT) == 0
    # a subnormal amax: the exponent field is 0 -> k = target + 127; the scaled value is still finite and below the fp16 range limit
    sub = np.float32(1e-40)
    assert np.isfinite(np.ldexp(np.float64(sub), hp_exponent(sub, TARGET_ACT))) and np.ldexp(np.float64(sub), hp_exponent(sub, TARGET_ACT)) < 65504


def test_split_never_overflows_and_carries_22_bits():
    rng = np.random.default_rng(1)
    for scale in (1e-30, 1e-8, 1.0, 3e4, 1e30):
        x = (rng.standard_normal(200000) * scale).astype(np.float32)
        x[::7] *= np.exp2(-rng.uniform(0, 24, x[::7].shape)).astype(np.float32)        # heavy tail towards small magnitudes
        k = hp_exponent(np.abs(x).max(), TARGET_ACT)
        h, m = split(x, k)
        assert np.isfinite(h.astype(np.float32)).all() and np.isfinite(m.astype(np.float32)).all()
        rec = np.ldexp(h.astype(np.float64) + m.astype(np.float64), -k)
        err = np.abs(rec - x.astype(np.float64))
        amax = np.abs(x).max().astype(np.float64)
        # relative 2^-22 wherever the low term is a normal fp16 number; below that an absolute floor of half a subnormal ulp / 2^k
        bound = np.maximum(np.abs(x.astype(np.float64)) * 2.0 ** -22, 2.0 ** -25 * 2.0 ** -k)
        assert (err <= bound).all(), (scale, float((err / bound).max()))
        assert 2.0 ** -25 * 2.0 ** -k <= amax * 2.0 ** -37          # the floor, relative to the tensor's largest element
        # the rms error is ~2^-23.5 of the values that have a normal low term
        big = np.abs(x) > amax * 2.0 ** -10
        assert np.sqrt(np.mean((err[big] / np.abs(x[big])) ** 2)) < 2.0 ** -23


def _dot_errors(nprod, K, rng, heavy):
    a = rng.standard_normal((64, K)).astype(np.float32)
    a *= (rng.random((64, K)) > 0.3)                                                      # ReLU-like sparsity
    w = (rng.standard_normal((K, 32)) * 0.05).astype(np.float32)
    if heavy:
        a *= np.exp2(-rng.uniform(0, 16, a.shape)).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64)
    ka, kw = hp_exponent(np.abs(a).max(), TARGET_ACT), hp_exponent(np.abs(w).max(), TARGET_W)
    ah, am = (t.astype(np.float32) for t in split(a, ka))
    wh, wm = (t.astype(np.float32) for t in split(w, kw))
    # every fp16 x fp16 product is exact in fp32; the MFMA accumulates in fp32: emulate with a float32 running sum over k in chunks of 16
    acc = np.zeros((64, 32), np.float32)
    terms = [(am, wh), (ah, wm), (ah, wh)] if nprod == 3 else [(am, wm), (am, wh), (ah, wm), (ah, wh)]
    for k0 in range(0, K, 16):
        for x_, y_ in terms:
            acc = (acc + (x_[:, k0:k0 + 16].astype(np.float64) @ y_[k0:k0 + 16].astype(np.float64)).astype(np.float32)).astype(np.float32)
    hp = np.ldexp(acc.astype(np.float64), -(ka + kw))
    f32 = np.zeros((64, 32), np.float32)
    for k0 in range(0, K, 16):
        f32 = (f32 + (a[:, k0:k0 + 16] @ w[k0:k0 + 16]).astype(np.float32)).astype(np.float32)
    n = np.linalg.norm(ref)
    return np.linalg.norm(hp - ref) / n, np.linalg.norm(f32.astype(np.float64) - ref) / n


def test_three_product_dot_product_is_as_accurate_as_float32():
    rng = np.random.default_rng(2)
    for K, heavy in ((576, False), (2304, False), (4608, False), (2304, True)):
        e3, e32 = _dot_errors(3, K, rng, heavy)
        e4, _ = _dot_errors(4, K, rng, heavy)
        assert e3 <= 2.0 * e32 + 1e-7, (K, heavy, e3, e32)          # the accumulation, not the operand format, sets the error
        assert e3 <= 1.25 * e4 + 2e-8, (K, heavy, e3, e4)           # the dropped mm products are below the operands' own rounding
