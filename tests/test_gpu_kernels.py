"""GPU: every HIP kernel of libfootprints_hip.so, called through the C ABI, against a plain PyTorch fp32 CPU
reference of the same op (the oracle's building blocks).  Tolerance 1e-4 relative to the tensor's max
(north_star: "within 1e-4 rel fp32"); index/byte outputs (argmax, counters) bit-exact.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _ops():
    from footprints_amd import ops, _lib
    return ops, _lib


def rnd(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * (hi - lo) + lo


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def relerr(got, ref):
    ref = ref.double()
    return ((got.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def check(got, ref, what, tol=TOL):
    e = relerr(got, ref)
    assert e <= tol, "%s: rel-to-max error %.3e > %.1e" % (what, e, tol)
    return e


def pack(w, dgrad=False, stem=False):
    ops, _ = _ops()
    Cout, Cin, K, _k = w.shape
    n = ops.packed_weight_elems(Cout, Cin, K, dgrad, stem)
    wp = torch.empty(n, device="cuda")
    wd = w.contiguous().cuda()
    return ops.pack_conv_weight_dgrad(wd, wp) if dgrad else ops.pack_conv_weight(wd, wp, stem)


# ----------------------------------------------------------------------------------------------------------
# forward convolutions
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,H,W,Cin,Cout,K,stride", [
    (2, 8, 12, 16, 8, 3, 1), (2, 9, 13, 16, 24, 3, 2), (3, 16, 20, 64, 64, 3, 1), (2, 12, 40, 64, 128, 3, 2),
    (2, 12, 40, 64, 128, 1, 2), (1, 6, 20, 512, 512, 3, 1), (2, 48, 160, 64, 64, 3, 1), (1, 24, 80, 128, 256, 3, 2)])
def test_conv_fwd_zero(N, H, W, Cin, Cout, K, stride):
    ops, L = _ops()
    pad = K // 2
    x, w = rnd((N, Cin, H, W), 1), rnd((Cout, Cin, K, K), 2, -0.1, 0.1)
    ref = F.conv2d(x, w, None, stride, pad)
    OH, OW = ref.shape[2:]
    y = torch.empty((N, OH, OW, Cout), device="cuda")
    d = ops.make_desc(N, OH, OW, H, W, Cin, 0, Cout, K, stride, pad, L.GATHER_FWD_ZERO)
    ops.conv_igemm(d, nhwc(x), None, pack(w), y)
    check(nchw(y), ref, "conv_fwd_zero")


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 6, 10, 16, 8), (2, 2, 3, 512, 256), (1, 24, 80, 128, 64), (1, 64, 96, 64, 32),
                                            (2, 12, 40, 256, 128), (1, 192, 640, 32, 32)])
def test_conv_fwd_reflect_bias_elu(N, H, W, Cin, Cout):
    ops, L = _ops()
    x, w, b = rnd((N, Cin, H, W), 3), rnd((Cout, Cin, 3, 3), 4, -0.1, 0.1), rnd((Cout,), 5)
    ref = F.elu(F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w, b))
    y = torch.empty((N, H, W, Cout), device="cuda")
    d = ops.make_desc(N, H, W, H, W, Cin, 0, Cout, 3, 1, 1, L.GATHER_FWD_REFLECT, act=L.ACT_ELU)
    ops.conv_igemm(d, nhwc(x), None, pack(w), y, bias=b.cuda())
    check(nchw(y), ref, "conv_fwd_reflect")


@pytest.mark.parametrize("N,h,w,C0,C1,Cout", [(2, 4, 6, 8, 8, 8), (1, 6, 20, 256, 256, 256), (1, 48, 160, 64, 64, 64),
                                              (1, 32, 48, 64, 0, 32), (2, 3, 5, 16, 4, 12)])
def test_conv_fwd_up2_concat(N, h, w, C0, C1, Cout):
    ops, L = _ops()
    lo, w_ = rnd((N, C0, h, w), 6), rnd((Cout, C0 + C1, 3, 3), 7, -0.1, 0.1)
    b = rnd((Cout,), 8)
    up = F.interpolate(lo, scale_factor=2, mode="nearest")
    skip = rnd((N, C1, 2 * h, 2 * w), 9) if C1 else None
    xin = torch.cat([up, skip], 1) if C1 else up
    ref = F.elu(F.conv2d(F.pad(xin, (1, 1, 1, 1), mode="reflect"), w_, b))
    y = torch.empty((N, 2 * h, 2 * w, Cout), device="cuda")
    d = ops.make_desc(N, 2 * h, 2 * w, 2 * h, 2 * w, C0, C1, Cout, 3, 1, 1, L.GATHER_FWD_REFLECT_UP2, act=L.ACT_ELU)
    ops.conv_igemm(d, nhwc(lo), nhwc(skip) if C1 else None, pack(w_), y, bias=b.cuda())
    check(nchw(y), ref, "conv_fwd_up2cat")


@pytest.mark.parametrize("N,H,W", [(2, 64, 96), (1, 192, 640), (3, 34, 50)])
def test_conv_stem(N, H, W):
    ops, L = _ops()
    img, w = rnd((N, 3, H, W), 10, 0.0, 1.0), rnd((64, 3, 7, 7), 11, -0.1, 0.1)
    ref = F.conv2d((img - 0.45) / 0.225, w, None, 2, 3)
    OH, OW = ref.shape[2:]
    y = torch.empty((N, OH, OW, 64), device="cuda")
    d = ops.make_desc(N, OH, OW, H, W, 3, 0, 64, 7, 2, 3, L.GATHER_STEM)
    ops.conv_igemm(d, img.cuda(), None, pack(w, stem=True), y)
    check(nchw(y), ref, "conv_stem")
    # the statistics side output (fp_aux.bn_part): (count, mean, M2) per 8 x 16 pixel tile and channel out of the stem's epilogue -> the same
    # BatchNorm coefficients as the reduction over the stored tensor; one-shot; ragged tiles at the right / bottom edge count only real pixels
    tiles = N * ((OH + 7) // 8) * ((OW + 15) // 16)
    part = torch.full((tiles * 64 * 3 + 5,), float("nan"), device="cuda")
    cell = ops.bn_stats_out(part)
    y2 = torch.empty_like(y)
    ops.conv_igemm(d, img.cuda(), None, pack(w, stem=True), y2, bn_out=cell)
    torch.cuda.synchronize()
    assert cell.nblk == tiles and torch.equal(y, y2)
    used = tiles * 64 * 3
    assert not bool(torch.isnan(part[:used]).any()) and bool(torch.isnan(part[used:]).all())
    assert float(part[:used].view(tiles, 64, 3)[:, :, 0].sum(0).min()) == float(N * OH * OW) == float(part[:used].view(tiles, 64, 3)[:, :, 0].sum(0).max())
    g, b = rnd((64,), 12, 0.5, 1.5).cuda(), rnd((64,), 13).cuda()
    outs = [[torch.zeros(64, device="cuda") for _ in range(4)] for _ in range(2)]
    rms, rvs = [torch.zeros(64, device="cuda") for _ in range(2)], [torch.ones(64, device="cuda") for _ in range(2)]
    nbt = [torch.zeros((), dtype=torch.int64, device="cuda") for _ in range(2)]
    ops.bn_train_stats_partials(part, tiles, 64, g, b, rms[0], rvs[0], nbt[0], *outs[0])
    ops.bn_train_stats(y.view(-1, 64), g, b, rms[1], rvs[1], nbt[1], *outs[1])
    yd = y.double().cpu().view(-1, 64)
    mean, invstd = yd.mean(0), 1.0 / torch.sqrt(yd.var(0, unbiased=False) + 1e-5)
    for k, want in enumerate((mean, invstd, g.double().cpu() * invstd, b.double().cpu() - mean * g.double().cpu() * invstd)):
        for o in outs:
            assert float((o[k].double().cpu() - want).abs().max() / want.abs().max()) < 2e-6, k
    assert torch.allclose(rms[0], rms[1], rtol=1e-6, atol=1e-7) and torch.allclose(rvs[0], rvs[1], rtol=1e-6, atol=1e-7)
    cell = ops.bn_stats_out(part)                                  # a launch that cannot emit (bias) reports 0
    d.epi = L.EPI_BIAS
    ops.conv_igemm(d, img.cuda(), None, pack(w, stem=True), y2, bias=b, bn_out=cell)
    assert cell.nblk == 0


# ----------------------------------------------------------------------------------------------------------
# nearest-x2 phase decomposition (conv_up2_phase.hip): upsample -> [cat skip] -> reflect-pad conv3x3 -> ELU
# ----------------------------------------------------------------------------------------------------------
def _up2_ref(lo, skip, w, b):
    up = F.interpolate(lo, scale_factor=2, mode="nearest")
    xin = torch.cat([up, skip], 1) if skip is not None else up
    return F.conv2d(F.pad(xin, (1, 1, 1, 1), mode="reflect"), w, b)


@pytest.mark.parametrize("N,h,w,C0,Cout", [(2, 24, 80, 64, 64), (2, 96, 320, 64, 32), (1, 5, 7, 32, 16), (2, 1, 1, 16, 16), (2, 9, 17, 20, 72)])
def test_up2_phase_fwd_bf3(N, h, w, C0, Cout):
    ops, L = _ops()
    wt, b = rnd((Cout, C0, 3, 3), 304, -0.1, 0.1), rnd((Cout,), 305)
    lo, prev = rnd((N, C0, h, w), 306), rnd((N, Cout, 2 * h, 2 * w), 307)
    ref = F.elu(_up2_ref(lo.double(), None, wt.double(), b.double()) + prev.double())
    wph = ops.pack_up2_weight_bf3(wt.cuda(), torch.empty(ops.up2_packed_weight_elems(Cout, C0) * 3 // 2, device="cuda"), 0, C0)
    y = nhwc(prev)
    ops.conv_up2_phase_fwd_bf3(nhwc(lo), wph, b.cuda(), y, act=L.ACT_ELU, addend=y)
    check(nchw(y), ref, "up2 phase fwd bf3", 3e-6)      # the collapsed weights are fp32 sums of the 3x3 taps: one extra rounding


@pytest.mark.parametrize("N,h,w,C0,C1,Cout", [
    (2, 6, 20, 256, 256, 256), (2, 12, 40, 128, 128, 128), (3, 24, 80, 64, 64, 64), (2, 48, 160, 32, 64, 32),
    (2, 96, 320, 16, 0, 16), (1, 5, 7, 32, 0, 16), (2, 1, 1, 16, 0, 16), (1, 3, 2, 24, 40, 48), (2, 9, 17, 20, 0, 72)])
def test_up2_phase_fwd(N, h, w, C0, C1, Cout):
    ops, L = _ops()
    wt, b = rnd((Cout, C0 + C1, 3, 3), 300, -0.1, 0.1), rnd((Cout,), 301)
    lo = rnd((N, C0, h, w), 302)
    skip = rnd((N, C1, 2 * h, 2 * w), 303) if C1 else None
    ref = F.elu(_up2_ref(lo, skip, wt, b))
    wd = wt.cuda()
    wph = ops.pack_up2_weight(wd, torch.empty(ops.up2_packed_weight_elems(Cout, C0), device="cuda"), 0, C0)
    y = torch.empty((N, 2 * h, 2 * w, Cout), device="cuda")
    if C1:                                           # skip half first (no bias / act), phase kernel adds in place
        wsk = ops.pack_conv_weight_slice(wd, torch.empty(ops.packed_weight_elems(Cout, C1, 3, False, False), device="cuda"), C0, C1)
        d = ops.make_desc(N, 2 * h, 2 * w, 2 * h, 2 * w, C1, 0, Cout, 3, 1, 1, L.GATHER_FWD_REFLECT)
        ops.conv_igemm(d, nhwc(skip), None, wsk, y)
        ops.conv_up2_phase_fwd(nhwc(lo), wph, b.cuda(), y, act=L.ACT_ELU, addend=y)
    else:
        ops.conv_up2_phase_fwd(nhwc(lo), wph, b.cuda(), y, act=L.ACT_ELU)
    check(nchw(y), ref, "up2 phase fwd")


def test_up2_phase_dgrad_bf3_case(N, h, w, C0, Cout):
    """bf16x3 phase data-gradient kernel: same ext grid as the 4x4 stride-2 fp32 convolution, then fold -> d(low) vs float64"""
    ops, L = _ops()
    wt = rnd((Cout, C0, 3, 3), 340, -0.1, 0.1)
    lo = rnd((N, C0, h, w), 341).double().requires_grad_(True)
    yr = _up2_ref(lo, None, wt.double(), None)
    g = rnd(tuple(yr.shape), 342)
    yr.backward(g.double())
    wd, gz = wt.cuda(), nhwc(g)
    wp3 = ops.pack_up2_weight_dgrad_bf3(wd, torch.empty(ops.up2_packed_weight_elems(C0, Cout) * 3 // 2, device="cuda"), 0, C0)
    ext3 = ops.conv_up2_phase_dgrad_bf3(gz, wp3, torch.full((N, h + 2, w + 2, C0), float("nan"), device="cuda"))
    wpu = ops.pack_up2_weight_dgrad(wd, torch.empty(ops.up2_packed_weight_elems(C0, Cout), device="cuda"), 0, C0)
    d = ops.make_desc(N, h + 2, w + 2, 2 * h, 2 * w, Cout, 0, C0, 4, 2, 3, L.GATHER_FWD_ZERO)
    ext = ops.conv_igemm(d, gz, None, wpu, torch.empty((N, h + 2, w + 2, C0), device="cuda"))
    assert relerr(ext3, ext.cpu()) <= 2e-5
    dlow = ops.up2_fold_bwd(ext3, torch.empty((N, h, w, C0), device="cuda"))
    check(nchw(dlow), lo.grad, "up2 phase dgrad bf3", 3e-6)


test_up2_phase_dgrad_bf3 = pytest.mark.parametrize("N,h,w,C0,Cout", [
    (2, 24, 80, 64, 64), (1, 96, 320, 64, 32), (2, 6, 20, 256, 256), (2, 1, 1, 16, 16), (1, 3, 2, 24, 48), (3, 9, 17, 20, 72)])(
        test_up2_phase_dgrad_bf3_case)


@pytest.mark.parametrize("N,h,w,C0,C1,Cout", [
    (2, 6, 20, 64, 64, 64), (2, 12, 40, 32, 32, 48), (1, 24, 80, 64, 0, 32), (2, 1, 1, 16, 0, 16), (1, 1, 5, 16, 16, 16),
    (2, 3, 2, 24, 40, 48), (1, 48, 160, 64, 64, 64)])
def test_up2_phase_dgrad(N, h, w, C0, C1, Cout):
    """d(low) through the 4x4 stride-2 conv + border fold (+ addend, ELU'), d(skip) through the sliced DGRAD_REFLECT."""
    ops, L = _ops()
    wt = rnd((Cout, C0 + C1, 3, 3), 310, -0.1, 0.1)
    pre = rnd((N, C0, h, w), 311, -2.0, 2.0).requires_grad_(True)
    skip = rnd((N, C1, 2 * h, 2 * w), 312).requires_grad_(True) if C1 else None
    lo = F.elu(pre)
    extra = rnd((N, C0, h, w), 313)
    yr = _up2_ref(lo, skip, wt, None)
    g = rnd(tuple(yr.shape), 314)
    ((yr * g).sum() + (lo * extra).sum()).backward()
    wd, gz = wt.cuda(), nhwc(g)
    wpu = ops.pack_up2_weight_dgrad(wd, torch.empty(ops.up2_packed_weight_elems(C0, Cout), device="cuda"), 0, C0)
    ext = torch.empty((N, h + 2, w + 2, C0), device="cuda")
    d = ops.make_desc(N, h + 2, w + 2, 2 * h, 2 * w, Cout, 0, C0, 4, 2, 3, L.GATHER_FWD_ZERO)
    ops.conv_igemm(d, gz, None, wpu, ext)
    dlow = ops.up2_fold_bwd(ext, torch.empty((N, h, w, C0), device="cuda"), addend=nhwc(extra), ylow=nhwc(lo.detach()))
    check(nchw(dlow), pre.grad, "up2 phase dgrad (low)")
    if C1:
        wps = ops.pack_conv_weight_dgrad_slice(wd, torch.empty(ops.packed_weight_elems(Cout, C1, 3, True), device="cuda"), C0, C1)
        ds = ops.make_desc(N, 2 * h, 2 * w, 2 * h, 2 * w, Cout, 0, C1, 3, 1, 1, L.GATHER_DGRAD_REFLECT)
        dsk = ops.conv_igemm(ds, gz, None, wps, torch.empty((N, 2 * h, 2 * w, C1), device="cuda"))
        check(nchw(dsk), skip.grad, "up2 phase dgrad (skip)")


@pytest.mark.parametrize("bf3", [False, True])
@pytest.mark.parametrize("N,h,w,C0,C1,Cout,acc", [
    (2, 24, 80, 64, 64, 64, False), (2, 12, 48, 32, 32, 32, True), (1, 48, 160, 64, 0, 32, False), (4, 7, 16, 32, 64, 96, True),
    (16, 2, 16, 32, 0, 32, False), (1, 96, 320, 64, 0, 32, True), (3, 6, 27, 64, 0, 64, False)])
def test_up2_phase_wgrad(N, h, w, C0, C1, Cout, acc, bf3):
    """weight gradient: upsampled half by phase (wgrad_up2_phase.hip), skip half as a slice of the same OIHW gradient"""
    ops, L = _ops()
    wt = rnd((Cout, C0 + C1, 3, 3), 320, -0.1, 0.1).requires_grad_(True)
    lo = rnd((N, C0, h, w), 321)
    skip = rnd((N, C1, 2 * h, 2 * w), 322) if C1 else None
    yr = _up2_ref(lo, skip, wt, None)
    g = rnd(tuple(yr.shape), 323)
    yr.backward(g)
    assert ops.up2_phase_wgrad_supported(N, h, w, C0, Cout)
    init = rnd((Cout, C0 + C1, 3, 3), 324) if acc else torch.full((Cout, C0 + C1, 3, 3), float("nan"))
    dw = init.clone().cuda()
    gz = nhwc(g)
    binit = rnd((Cout,), 325) if acc else torch.full((Cout,), float("nan"))
    db = binit.clone().cuda() if bf3 else None
    ops.conv_up2_phase_wgrad(nhwc(lo), gz, dw, 0, accumulate=acc, bf3=bf3, db=db)
    if bf3:                                                          # bias gradient from the same pass over dz
        check(db.cpu(), g.sum((0, 2, 3)) + (binit if acc else 0), "up2 phase wgrad bias", 1e-5)
    if C1:
        d = ops.make_desc(N, 2 * h, 2 * w, 2 * h, 2 * w, C1, 0, Cout, 3, 1, 1, L.GATHER_FWD_REFLECT)
        ops.conv_wgrad_slice(d, nhwc(skip), None, gz, dw, C0, accumulate=acc)
    ref = wt.grad + (init if acc else 0)
    check(dw.cpu(), ref, "up2 phase wgrad")


def test_up2_phase_wgrad_unsupported_shapes():
    ops, _ = _ops()
    assert not ops.up2_phase_wgrad_supported(2, 6, 20, 256, 256)      # 37.5 % padded chunks
    assert not ops.up2_phase_wgrad_supported(2, 24, 80, 16, 16)       # channels not multiples of 32


def test_pack_weights_batched_matches_single_launches():
    """one launch over a job table == the per-tensor packers, bit for bit (every layout, incl. channel slices)"""
    ops, L = _ops()
    ws = [rnd((64, 3, 7, 7), 330).cuda(), rnd((48, 40, 3, 3), 331).cuda(), rnd((32, 64, 1, 1), 332).cuda(), rnd((64, 128, 3, 3), 333).cuda()]
    single, jobs = [], []

    def add(kind, w, n, c_begin, c_count, fn):
        a, b = torch.full((n,), float("nan"), device="cuda"), torch.full((n,), float("nan"), device="cuda")
        fn(a)
        single.append(a)
        jobs.append((kind, w, b, c_begin, c_count))

    add(L.PACK_STEM, ws[0], ops.packed_weight_elems(64, 3, 7, False, True), 0, 3, lambda o: ops.pack_conv_weight(ws[0], o, True))
    for w in ws[1:]:
        Cout, Cin, K, _ = w.shape
        add(L.PACK_FWD, w, ops.packed_weight_elems(Cout, Cin, K), 0, Cin, lambda o, w=w: ops.pack_conv_weight(w, o))
        add(L.PACK_DGRAD, w, ops.packed_weight_elems(Cout, Cin, K, True), 0, Cin, lambda o, w=w: ops.pack_conv_weight_dgrad(w, o))
    w = ws[3]
    add(L.PACK_UP2_FWD, w, ops.up2_packed_weight_elems(64, 64), 0, 64, lambda o: ops.pack_up2_weight(w, o, 0, 64))
    add(L.PACK_UP2_DGRAD, w, ops.up2_packed_weight_elems(64, 64), 0, 64, lambda o: ops.pack_up2_weight_dgrad(w, o, 0, 64))
    add(L.PACK_FWD, w, ops.packed_weight_elems(64, 64, 3), 64, 64, lambda o: ops.pack_conv_weight_slice(w, o, 64, 64))
    add(L.PACK_DGRAD, w, ops.packed_weight_elems(64, 64, 3, True), 64, 64, lambda o: ops.pack_conv_weight_dgrad_slice(w, o, 64, 64))
    w2 = ws[1]
    add(L.PACK_UP2_FWD, w2, ops.up2_packed_weight_elems(48, 24), 8, 24, lambda o: ops.pack_up2_weight(w2, o, 8, 24))
    # split layouts (bf16 triples, scaled fp16 pairs): the tile form of the batched kernel against the element-wise single-tensor kernels
    for w in ws[1:]:
        Cout, Cin, K, _ = w.shape
        for dg in (False, True):
            add(L.PACK_DGRAD_BF3 if dg else L.PACK_FWD_BF3, w, ops.packed_weight_elems_bf3(Cout, Cin, K, dg), 0, Cin,
                lambda o, w=w, dg=dg: ops.pack_conv_weight_bf3(w, o, dg))
    add(L.PACK_UP2_FWD_BF3, w, ops.up2_packed_weight_elems(64, 64) * 3 // 2, 0, 64, lambda o: ops.pack_up2_weight_bf3(w, o, 0, 64))
    add(L.PACK_UP2_DGRAD_BF3, w, ops.up2_packed_weight_elems(64, 64) * 3 // 2, 0, 64, lambda o: ops.pack_up2_weight_dgrad_bf3(w, o, 0, 64))
    nplain = len(jobs)
    for w in ws[1:]:
        Cout, Cin, K, _ = w.shape
        for dg in (False, True):
            slot_a, slot_b = ops.new_slot(), ops.new_slot()
            a, b = (torch.full((ops.packed_weight_elems_hp(Cout, Cin, K, dg),), float("nan"), device="cuda") for _ in range(2))
            ops.pack_conv_weight_hp(w, a, slot_a, dg)
            single.append(a)
            jobs.append((L.PACK_DGRAD_HP if dg else L.PACK_FWD_HP, w, b, 0, Cin, slot_b))
    table = ops.build_pack_table(jobs, "cuda")
    for max_wgs in (0, 3):                                  # one workgroup per virtual block; three persistent workgroups
        for j in jobs:
            j[2].fill_(float("nan"))
        ops.pack_weights_amax(table)
        ops.pack_weights_batched(table, max_wgs=max_wgs)
        for i, (a, j) in enumerate(zip(single, jobs)):
            assert not torch.isnan(a).any() and torch.equal(a.view(torch.int32), j[2].view(torch.int32)), (max_wgs, i, j[0], tuple(j[1].shape), j[3:5])
    # the phase layouts as fp16 pairs have no single-tensor packer: (h + m) * 2^-k must reproduce the fp32 phase layout to 2^-21, with one power
    # of two per tensor, in the same [virtual tap][chunk][column][16] order with the two planes interleaved per (tap, chunk)
    for kind32, kind_hp, dg in ((L.PACK_UP2_FWD, L.PACK_UP2_FWD_HP, False), (L.PACK_UP2_DGRAD, L.PACK_UP2_DGRAD_HP, True)):
        n = ops.up2_packed_weight_elems(64, 64)
        ref, hp_, s = torch.empty(n, device="cuda"), torch.full((n,), float("nan"), device="cuda"), ops.new_slot()
        t32 = ops.build_pack_table([(kind32, w, ref, 0, 64)], "cuda")
        thp = ops.build_pack_table([(kind_hp, w, hp_, 0, 64, s)], "cuda")
        ops.pack_weights_batched(t32)
        ops.pack_weights_amax(thp)
        ops.pack_weights_batched(thp, max_wgs=2)
        ncols = 64
        pair = hp_.view(torch.float16).view(-1, 2, ncols, 16).float()
        got = pair[:, 0] + pair[:, 1]
        if kind32 == L.PACK_UP2_DGRAD:                      # fp32 layout: taps r*4+s; fp16-pair layout: [phase][tap] with r = (py+1)%2 + 2a, s = (px+1)%2 + 2b
            r32 = ref.view(16, -1, ncols, 16)
            order = [(((ph >> 1) + 1) % 2 + 2 * (tp >> 1)) * 4 + ((ph & 1) + 1) % 2 + 2 * (tp & 1) for ph in range(4) for tp in range(4)]
            want = r32[order].reshape(-1, ncols, 16)
        else:
            want = ref.view(-1, ncols, 16)
        big = want.abs() > 0.1 * want.abs().max()
        ratio = float((got[big] / want[big]).median())
        scale = 2.0 ** round(float(np.log2(ratio)))
        assert abs(ratio / scale - 1.0) < 1e-5, ratio
        assert ((got / scale - want).abs() <= want.abs() * 2.0 ** -21 + 1e-30).all(), kind_hp


# ----------------------------------------------------------------------------------------------------------
# exactly split bf16x3 operands (conv3x3_tile_bf3.hip): same operation as the fp32-MFMA tile kernel, tighter tolerance
# ----------------------------------------------------------------------------------------------------------
def pack_bf3(w, dgrad=False):
    ops, _ = _ops()
    Cout, Cin, K, _k = w.shape
    wp = torch.empty(ops.packed_weight_elems_bf3(Cout, Cin, K, dgrad), device="cuda")
    return ops.pack_conv_weight_bf3(w.contiguous().cuda(), wp, dgrad)


@pytest.mark.parametrize("mode,N,H,W,C0,Cout", [
    ("zero", 12, 48, 160, 64, 64), ("reflect", 4, 96, 320, 64, 64), ("reflect", 2, 192, 640, 32, 32), ("reflect", 12, 24, 80, 128, 128),
    ("zero", 6, 40, 176, 32, 96), ("zero", 20, 24, 80, 24, 40), ("dgrad", 12, 48, 160, 64, 64), ("dgrad", 12, 24, 80, 128, 128),
    ("dgrad_reflect", 12, 48, 160, 64, 64), ("dgrad_reflect", 2, 192, 640, 32, 32), ("dgrad_reflect", 6, 40, 144, 64, 128),
    ("dgrad_reflect", 12, 24, 80, 128, 64),
    # small grids: 6x20 tiles and / or split-K over the channel chunks
    ("zero", 12, 12, 40, 256, 256), ("zero", 12, 6, 20, 512, 512), ("reflect", 12, 12, 40, 256, 128), ("reflect", 12, 24, 80, 64, 64),
    ("dgrad", 12, 12, 40, 256, 256), ("dgrad", 12, 6, 20, 512, 512), ("dgrad_reflect", 12, 6, 20, 256, 512),
    ("dgrad_reflect", 12, 12, 40, 128, 256), ("zero", 8, 18, 60, 48, 80)])
def test_conv3x3_bf3_kernel(mode, N, H, W, C0, Cout):
    ops, L = _ops()
    tol = 2e-6        # split operands are exact; six products + fp32 accumulation: tighter than the fp32 MFMA itself
    if mode == "dgrad_reflect":
        pre = rnd((N, C0, H, W), 92, -2.0, 2.0).requires_grad_(True)
        w = rnd((Cout, C0, 3, 3), 93, -0.1, 0.1)
        x = F.elu(pre)
        yr = F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode="reflect"), w.double())
        g, extra = rnd(tuple(yr.shape), 94), rnd(tuple(x.shape), 95)
        ((yr * g.double()).sum() + (x.double() * extra.double()).sum()).backward()
        dz = torch.empty((N, H, W, C0), device="cuda")
        d = ops.make_desc(N, H, W, H, W, Cout, 0, C0, 3, 1, 1, L.GATHER_DGRAD_REFLECT, epi=L.EPI_ACTGRAD_ELU)
        assert ops.conv3x3_bf3_supported(d)
        ops.conv3x3_bf3(d, nhwc(g), pack_bf3(w, dgrad=True), dz, actsrc=nhwc(x.detach()), addend=nhwc(extra))
        check(nchw(dz), pre.grad, "bf3 dgrad_reflect (fold) + epilogue", tol)
        return
    if mode == "dgrad":
        x = rnd((N, C0, H, W), 80).double().requires_grad_(True)
        w = rnd((Cout, C0, 3, 3), 81, -0.1, 0.1)
        yr = F.conv2d(x, w.double(), None, 1, 1)
        g = rnd(tuple(yr.shape), 82)
        add, msk, act = rnd(tuple(x.shape), 83), rnd(tuple(x.shape), 84), rnd(tuple(x.shape), 85)
        yr.backward(g.double())
        ref = (x.grad + add * (msk > 0).float()) * (act > 0).float()
        dx = torch.empty((N, H, W, C0), device="cuda")
        d = ops.make_desc(N, H, W, H, W, Cout, 0, C0, 3, 1, 1, L.GATHER_DGRAD_ZERO, epi=L.EPI_ACTGRAD_RELU)
        ops.conv3x3_bf3(d, nhwc(g), pack_bf3(w, dgrad=True), dx, addend=nhwc(add), addend_mask=nhwc(msk), actsrc=nhwc(act))
        check(nchw(dx), ref, "bf3 dgrad_zero + epilogue", tol)
        return
    w, b = rnd((Cout, C0, 3, 3), 86, -0.1, 0.1), rnd((Cout,), 87)
    x = rnd((N, C0, H, W), 90)
    if mode == "reflect":
        ref = F.elu(F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode="reflect"), w.double(), b.double()))
        gather = L.GATHER_FWD_REFLECT
    else:
        ref = F.elu(F.conv2d(x.double(), w.double(), b.double(), 1, 1))
        gather = L.GATHER_FWD_ZERO
    y = torch.empty((N, H, W, Cout), device="cuda")
    d = ops.make_desc(N, H, W, H, W, C0, 0, Cout, 3, 1, 1, gather, act=L.ACT_ELU)
    assert ops.conv3x3_bf3_supported(d)
    ops.conv3x3_bf3(d, nhwc(x), pack_bf3(w), y, bias=b.cuda())
    check(nchw(y), ref, "bf3 forward " + mode, tol)


# ----------------------------------------------------------------------------------------------------------
# halo-tile 3x3 kernel (conv3x3_tile.hip): shapes large enough (>= 384 workgroups) to be dispatched to it
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode,N,H,W,C0,C1,Cout", [
    ("zero", 12, 48, 160, 64, 0, 64), ("reflect", 4, 96, 320, 64, 0, 64), ("reflect", 2, 192, 640, 32, 0, 32),
    ("up2", 4, 96, 320, 64, 64, 64), ("up2", 2, 192, 640, 64, 0, 32), ("reflect", 12, 24, 80, 128, 0, 128),
    ("zero", 6, 40, 176, 32, 0, 96), ("dgrad", 12, 48, 160, 64, 0, 64), ("dgrad", 6, 24, 80, 128, 0, 128),
    ("dgrad_reflect", 12, 48, 160, 64, 0, 64), ("dgrad_reflect", 2, 192, 640, 32, 0, 32), ("dgrad_reflect", 4, 40, 144, 64, 0, 128),
    ("dgrad_reflect", 12, 24, 80, 128, 0, 64)])
def test_conv3x3_tile_kernel(mode, N, H, W, C0, C1, Cout):
    ops, L = _ops()
    if mode == "dgrad_reflect":       # data-gradient of a reflection-padded conv + ELU' + second-consumer addend
        pre = rnd((N, C0, H, W), 92, -2.0, 2.0).requires_grad_(True)
        w = rnd((Cout, C0, 3, 3), 93, -0.1, 0.1)
        x = F.elu(pre)
        yr = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w)
        g, extra = rnd(tuple(yr.shape), 94), rnd(tuple(x.shape), 95)
        ((yr * g).sum() + (x * extra).sum()).backward()
        dz = torch.empty((N, H, W, C0), device="cuda")
        d = ops.make_desc(N, H, W, H, W, Cout, 0, C0, 3, 1, 1, L.GATHER_DGRAD_REFLECT, epi=L.EPI_ACTGRAD_ELU)
        ops.conv_igemm(d, nhwc(g), None, pack(w, dgrad=True), dz, actsrc=nhwc(x.detach()), addend=nhwc(extra))
        check(nchw(dz), pre.grad, "tile dgrad_reflect (fold) + epilogue")
        return
    if mode == "dgrad":
        x = rnd((N, C0, H, W), 80).requires_grad_(True)
        w = rnd((Cout, C0, 3, 3), 81, -0.1, 0.1)
        yr = F.conv2d(x, w, None, 1, 1)
        g = rnd(tuple(yr.shape), 82)
        add, msk, act = rnd(tuple(x.shape), 83), rnd(tuple(x.shape), 84), rnd(tuple(x.shape), 85)
        yr.backward(g)
        ref = (x.grad + add * (msk > 0).float()) * (act > 0).float()
        dx = torch.empty((N, H, W, C0), device="cuda")
        d = ops.make_desc(N, H, W, H, W, Cout, 0, C0, 3, 1, 1, L.GATHER_DGRAD_ZERO, epi=L.EPI_ACTGRAD_RELU)
        ops.conv_igemm(d, nhwc(g), None, pack(w, dgrad=True), dx, addend=nhwc(add), addend_mask=nhwc(msk), actsrc=nhwc(act))
        check(nchw(dx), ref, "tile dgrad_zero + epilogue")
        return
    w, b = rnd((Cout, C0 + C1, 3, 3), 86, -0.1, 0.1), rnd((Cout,), 87)
    if mode == "up2":
        lo = rnd((N, C0, H // 2, W // 2), 88)
        skip = rnd((N, C1, H, W), 89) if C1 else None
        up = F.interpolate(lo, scale_factor=2, mode="nearest")
        xin = torch.cat([up, skip], 1) if C1 else up
        ref = F.elu(F.conv2d(F.pad(xin, (1, 1, 1, 1), mode="reflect"), w, b))
        src0, src1, gather = nhwc(lo), (nhwc(skip) if C1 else None), L.GATHER_FWD_REFLECT_UP2
    elif mode == "reflect":
        x = rnd((N, C0, H, W), 90)
        ref = F.elu(F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w, b))
        src0, src1, gather = nhwc(x), None, L.GATHER_FWD_REFLECT
    else:
        x = rnd((N, C0, H, W), 91)
        ref = F.elu(F.conv2d(x, w, b, 1, 1))
        src0, src1, gather = nhwc(x), None, L.GATHER_FWD_ZERO
    y = torch.empty((N, H, W, Cout), device="cuda")
    d = ops.make_desc(N, H, W, H, W, C0, C1, Cout, 3, 1, 1, gather, act=L.ACT_ELU)
    ops.conv_igemm(d, src0, src1, pack(w), y, bias=b.cuda())
    check(nchw(y), ref, "tile conv " + mode)


# ----------------------------------------------------------------------------------------------------------
# data gradients
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,H,W,Cin,Cout,K,stride", [(2, 8, 12, 16, 8, 3, 1), (2, 9, 13, 16, 24, 3, 2), (2, 12, 40, 64, 128, 3, 2),
                                                     (2, 12, 40, 64, 128, 1, 2), (1, 24, 80, 128, 128, 3, 1), (2, 7, 9, 32, 16, 1, 2),
                                                     (12, 12, 40, 256, 512, 3, 2), (3, 6, 10, 32, 48, 3, 2), (1, 2, 2, 16, 16, 3, 2)])
def test_conv_dgrad_zero(N, H, W, Cin, Cout, K, stride):
    ops, L = _ops()
    pad = K // 2
    x = rnd((N, Cin, H, W), 12).requires_grad_(True)
    w = rnd((Cout, Cin, K, K), 13, -0.1, 0.1)
    yref = F.conv2d(x, w, None, stride, pad)
    g = rnd(tuple(yref.shape), 14)
    yref.backward(g)
    OH, OW = yref.shape[2:]
    dx = torch.empty((N, H, W, Cin), device="cuda")
    d = ops.make_desc(N, H, W, OH, OW, Cout, 0, Cin, K, stride, pad, L.GATHER_DGRAD_ZERO)
    ops.conv_igemm(d, nhwc(g), None, pack(w, dgrad=True), dx)
    check(nchw(dx), x.grad, "conv_dgrad_zero")


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 6, 10, 16, 8), (2, 2, 3, 64, 32), (1, 3, 5, 16, 16), (1, 24, 80, 128, 64),
                                            (1, 64, 96, 64, 32), (2, 2, 2, 16, 8)])
def test_conv_dgrad_reflect_with_elu_grad(N, H, W, Cin, Cout):
    ops, L = _ops()
    # x = elu(pre) is the saved activation feeding the conv; we want dL/dpre = dgrad * elu'(x)
    pre = rnd((N, Cin, H, W), 15, -2.0, 2.0).requires_grad_(True)
    w = rnd((Cout, Cin, 3, 3), 16, -0.1, 0.1)
    x = F.elu(pre)
    y = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w)
    g = rnd(tuple(y.shape), 17)
    extra = rnd(tuple(x.shape), 18)                 # a second consumer's gradient (addend)
    (y * g).sum().backward(retain_graph=True)
    ref = pre.grad.clone()
    pre.grad = None
    ((y * g).sum() + (x * extra).sum()).backward()
    ref_add = pre.grad
    dz = torch.empty((N, H, W, Cin), device="cuda")
    d = ops.make_desc(N, H, W, H, W, Cout, 0, Cin, 3, 1, 1, L.GATHER_DGRAD_REFLECT, epi=L.EPI_ACTGRAD_ELU)
    xs = nhwc(x.detach())
    ops.conv_igemm(d, nhwc(g), None, pack(w, dgrad=True), dz, actsrc=xs)
    check(nchw(dz), ref, "conv_dgrad_reflect")
    ops.conv_igemm(d, nhwc(g), None, pack(w, dgrad=True), dz, actsrc=xs, addend=nhwc(extra))
    check(nchw(dz), ref_add, "conv_dgrad_reflect+addend")


@pytest.mark.parametrize("N,H,W,C", [(2, 6, 20, 512), (2, 12, 40, 256), (12, 6, 20, 512), (2, 24, 80, 128)])
def test_conv_dgrad_zero_splitk_with_addend(N, H, W, C):
    """The encoder's block-input gradient: dgrad(conv1) + masked residual gradient, on small grids (split-K path)."""
    ops, L = _ops()
    x = rnd((N, C, H, W), 70).requires_grad_(True)
    w = rnd((C, C, 3, 3), 71, -0.05, 0.05)
    y = F.conv2d(x, w, None, 1, 1)
    g = rnd(tuple(y.shape), 72)
    add = rnd(tuple(x.shape), 73)
    y.backward(g)
    dx = torch.empty((N, H, W, C), device="cuda")
    d = ops.make_desc(N, H, W, H, W, C, 0, C, 3, 1, 1, L.GATHER_DGRAD_ZERO)
    wp = pack(w, dgrad=True)
    ops.conv_igemm(d, nhwc(g), None, wp, dx, addend=nhwc(add))
    check(nchw(dx), x.grad + add, "dgrad_zero split-K + addend")
    dx2 = nhwc(add)
    d.epi = L.EPI_ACCUM
    ops.conv_igemm(d, nhwc(g), None, wp, dx2)
    check(nchw(dx2), x.grad + add, "dgrad_zero split-K + accum")


def test_conv_epilogue_relu_mask_and_accum():
    ops, L = _ops()
    N, H, W, Cin, Cout = 2, 8, 12, 32, 16
    x, w = rnd((N, Cin, H, W), 19), rnd((Cout, Cin, 3, 3), 20, -0.1, 0.1)
    base = F.conv2d(x, w, None, 1, 1)
    add, mask, prev, act = rnd(tuple(base.shape), 21), rnd(tuple(base.shape), 22), rnd(tuple(base.shape), 23), rnd(tuple(base.shape), 24)
    ref = (base + add * (mask > 0).float()) * (act > 0).float() + prev
    y = nhwc(prev)
    d = ops.make_desc(N, H, W, H, W, Cin, 0, Cout, 3, 1, 1, L.GATHER_FWD_ZERO, epi=L.EPI_ACTGRAD_RELU | L.EPI_ACCUM)
    ops.conv_igemm(d, nhwc(x), None, pack(w), y, addend=nhwc(add), addend_mask=nhwc(mask), actsrc=nhwc(act))
    check(nchw(y), ref, "epilogue")
    d2 = ops.make_desc(N, H, W, H, W, Cin, 0, Cout, 3, 1, 1, L.GATHER_FWD_ZERO, act=L.ACT_RELU)
    y2 = torch.empty_like(y)
    ops.conv_igemm(d2, nhwc(x), None, pack(w), y2)
    check(nchw(y2), F.relu(base), "epilogue relu")


# ----------------------------------------------------------------------------------------------------------
# weight gradients
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,H,W,Cin,Cout,K,stride,mode", [
    (2, 8, 12, 16, 8, 3, 1, "zero"), (2, 9, 13, 16, 24, 3, 2, "zero"), (2, 12, 40, 64, 128, 1, 2, "zero"),
    (2, 24, 80, 128, 128, 3, 1, "zero"), (2, 6, 10, 16, 8, 3, 1, "reflect"), (1, 48, 160, 64, 32, 3, 1, "reflect"),
    (1, 12, 40, 256, 128, 3, 1, "reflect"), (3, 6, 20, 512, 256, 3, 1, "reflect")])
def test_conv_wgrad(N, H, W, Cin, Cout, K, stride, mode):
    ops, L = _ops()
    pad = K // 2
    x = rnd((N, Cin, H, W), 25)
    w = rnd((Cout, Cin, K, K), 26, -0.1, 0.1).requires_grad_(True)
    if mode == "zero":
        y = F.conv2d(x, w, None, stride, pad)
        gather = L.GATHER_FWD_ZERO
    else:
        y = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w)
        gather = L.GATHER_FWD_REFLECT
    g = rnd(tuple(y.shape), 27)
    y.backward(g)
    OH, OW = y.shape[2:]
    d = ops.make_desc(N, OH, OW, H, W, Cin, 0, Cout, K, stride, pad, gather)
    dw = torch.empty((Cout, Cin, K, K), device="cuda")
    ops.conv_wgrad(d, nhwc(x), None, nhwc(g), dw)
    check(dw, w.grad, "conv_wgrad")
    ops.conv_wgrad(d, nhwc(x), None, nhwc(g), dw, accumulate=True)
    check(dw, 2 * w.grad, "conv_wgrad accumulate")
    db = torch.empty((Cout,), device="cuda")
    ops.colsum(nhwc(g).view(-1, Cout), db)
    check(db, g.sum((0, 2, 3)), "colsum")


@pytest.mark.parametrize("N,h,w,C0,C1,Cout", [(2, 4, 6, 8, 8, 8), (1, 24, 80, 64, 64, 64), (1, 32, 48, 64, 0, 32)])
def test_conv_wgrad_up2cat_and_bwd(N, h, w, C0, C1, Cout):
    ops, L = _ops()
    pre = rnd((N, C0, h, w), 28, -2.0, 2.0).requires_grad_(True)
    lo = F.elu(pre)
    skip = rnd((N, C1, 2 * h, 2 * w), 29).requires_grad_(True) if C1 else None
    wt = rnd((Cout, C0 + C1, 3, 3), 30, -0.1, 0.1).requires_grad_(True)
    up = F.interpolate(lo, scale_factor=2, mode="nearest")
    xin = torch.cat([up, skip], 1) if C1 else up
    y = F.conv2d(F.pad(xin, (1, 1, 1, 1), mode="reflect"), wt)
    g = rnd(tuple(y.shape), 31)
    extra = rnd(tuple(lo.shape), 32)
    ((y * g).sum() + (lo * extra).sum()).backward()
    H, W = 2 * h, 2 * w
    d = ops.make_desc(N, H, W, H, W, C0, C1, Cout, 3, 1, 1, L.GATHER_FWD_REFLECT_UP2)
    dw = torch.empty((Cout, C0 + C1, 3, 3), device="cuda")
    gz = nhwc(g)
    ops.conv_wgrad(d, nhwc(lo.detach()), nhwc(skip.detach()) if C1 else None, gz, dw)
    check(dw, wt.grad, "wgrad up2cat")
    # dgrad at hi-res over all C0+C1 channels, then split/pool
    dd = ops.make_desc(N, H, W, H, W, Cout, 0, C0 + C1, 3, 1, 1, L.GATHER_DGRAD_REFLECT)
    dxv = torch.empty((N, H, W, C0 + C1), device="cuda")
    ops.conv_igemm(dd, gz, None, pack(wt.detach(), dgrad=True), dxv)
    dlow = torch.empty((N, h, w, C0), device="cuda")
    dskip = torch.ones((N, H, W, C1), device="cuda") if C1 else None
    ops.up2cat_bwd(dxv, N, h, w, C0, C1, dlow, addend=nhwc(extra), ylow=nhwc(lo.detach()), dskip=dskip, accumulate_skip=True)
    check(nchw(dlow), pre.grad, "up2cat_bwd dlow")
    if C1:
        check(nchw(dskip), skip.grad + 1.0, "up2cat_bwd dskip(accumulate)")


@pytest.mark.parametrize("mode,N,H,W,C0,C1,Cout", [
    ("zero", 12, 48, 160, 64, 0, 64), ("reflect", 4, 96, 320, 64, 0, 32), ("reflect", 2, 192, 640, 32, 0, 32),
    ("up2", 4, 96, 320, 64, 64, 64), ("up2", 2, 192, 640, 64, 0, 32), ("reflect", 12, 24, 80, 128, 0, 128),
    ("zero", 6, 12, 40, 256, 0, 256), ("reflect", 3, 10, 44, 32, 0, 96), ("up2", 2, 24, 80, 128, 128, 128)])
def test_wgrad3x3_tile_kernel(mode, N, H, W, C0, C1, Cout):
    """All-taps LDS-DMA weight-gradient kernel (wgrad3x3_tile.hip): 32-aligned channels, 3x3 stride 1."""
    ops, L = _ops()
    w = rnd((Cout, C0 + C1, 3, 3), 96, -0.1, 0.1).requires_grad_(True)
    if mode == "up2":
        lo = rnd((N, C0, H // 2, W // 2), 97)
        skip = rnd((N, C1, H, W), 98) if C1 else None
        up = F.interpolate(lo, scale_factor=2, mode="nearest")
        xin = torch.cat([up, skip], 1) if C1 else up
        y = F.conv2d(F.pad(xin, (1, 1, 1, 1), mode="reflect"), w)
        src0, src1, gather = nhwc(lo), (nhwc(skip) if C1 else None), L.GATHER_FWD_REFLECT_UP2
    elif mode == "reflect":
        x = rnd((N, C0, H, W), 99)
        y = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w)
        src0, src1, gather = nhwc(x), None, L.GATHER_FWD_REFLECT
    else:
        x = rnd((N, C0, H, W), 100)
        y = F.conv2d(x, w, None, 1, 1)
        src0, src1, gather = nhwc(x), None, L.GATHER_FWD_ZERO
    g = rnd(tuple(y.shape), 101)
    y.backward(g)
    d = ops.make_desc(N, H, W, H, W, C0, C1, Cout, 3, 1, 1, gather)
    dw = torch.empty((Cout, C0 + C1, 3, 3), device="cuda")
    ops.conv_wgrad(d, src0, src1, nhwc(g), dw)
    check(dw, w.grad, "wgrad tile " + mode)
    ops.conv_wgrad(d, src0, src1, nhwc(g), dw, accumulate=True)
    check(dw, 2 * w.grad, "wgrad tile accumulate " + mode)


@pytest.mark.parametrize("N,H,W", [(2, 64, 96), (2, 192, 640), (3, 72, 104), (1, 16, 32)])
def test_stem_wgrad(N, H, W):
    """7x7 / 2 stem weight gradient (stem_tile.hip: patch-in-LDS kernel; partial tiles at the right / bottom edge; accumulate)"""
    ops, L = _ops()
    img = rnd((N, 3, H, W), 33, 0.0, 1.0)
    w = rnd((64, 3, 7, 7), 34, -0.1, 0.1).requires_grad_(True)
    y = F.conv2d((img - 0.45) / 0.225, w, None, 2, 3)
    g = rnd(tuple(y.shape), 35)
    y.backward(g)
    d = ops.make_desc(N, H // 2, W // 2, H, W, 3, 0, 64, 7, 2, 3, L.GATHER_STEM)
    dw = torch.empty((64, 3, 7, 7), device="cuda")
    ops.conv_wgrad(d, img.cuda(), None, nhwc(g), dw)
    check(dw, w.grad, "stem wgrad")
    ops.conv_wgrad(d, img.cuda(), None, nhwc(g), dw, accumulate=True)
    check(dw, 2 * w.grad, "stem wgrad accumulate")


# ----------------------------------------------------------------------------------------------------------
# heads
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("Cin,scale,sig,hw", [(16, 1, False, (6, 10)), (16, 2, True, (6, 10)), (32, 1, True, (6, 10)),
                                              (64, 4, False, (6, 10)), (128, 8, True, (6, 10)), (64, 2, True, (6, 10)),
                                              (32, 1, True, (37, 70)), (64, 2, False, (21, 19)), (128, 8, True, (3, 5)),
                                              (4, 4, True, (2, 2)), (8, 2, False, (2, 3)), (32, 3, True, (5, 4))])
def test_head_fwd_bwd(Cin, scale, sig, hw):
    ops, L = _ops()
    N, (h, w) = 2, hw
    x = rnd((N, Cin, h, w), 36).requires_grad_(True)
    wt = rnd((2, Cin, 3, 3), 37, -0.2, 0.2).requires_grad_(True)
    b = rnd((2,), 38).requires_grad_(True)
    z = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), wt, b)
    s = torch.sigmoid(z) if sig else z
    o = F.interpolate(s, scale_factor=scale, mode="bilinear", align_corners=False) if scale != 1 else s
    g = rnd(tuple(o.shape), 39)
    o.backward(g)
    H, W = h * scale, w * scale
    xs = nhwc(x.detach())
    low = torch.empty((N, h, w, 2), device="cuda")
    ops.head_fwd(xs, wt.detach().cuda(), b.detach().cuda(), low, sig)
    check(nchw(low), s.detach(), "head_fwd")
    out = torch.zeros((N, 4, H, W), device="cuda")
    ops.head_upsample(low, out, scale, 2)
    check(out[:, 2:4], o.detach(), "head_upsample")
    assert float(out[:, :2].abs().max()) == 0.0
    gout = torch.zeros((N, 4, H, W), device="cuda")
    gout[:, 2:4] = g.cuda()
    dz = torch.empty((N, h, w, 2), device="cuda")
    ops.head_upsample_bwd(gout, low, dz, scale, 2, sig)
    dx = torch.empty((N, h, w, Cin), device="cuda")
    ops.head_dgrad(dz, wt.detach().cuda(), dx)
    check(nchw(dx), x.grad, "head_dgrad")
    ops.head_dgrad(dz, wt.detach().cuda(), dx, elu_src=xs)                 # fused ELU backward from the layer's OUTPUT
    xc = x.detach()
    check(nchw(dx), x.grad * torch.where(xc > 0, torch.ones_like(xc), xc + 1), "head_dgrad+elu")
    dw, db = torch.empty((2, Cin, 3, 3), device="cuda"), torch.empty((2,), device="cuda")
    ops.head_wgrad(xs, dz, dw, db)
    check(dw, wt.grad, "head_wgrad")
    check(db, b.grad, "head_bgrad")


# ----------------------------------------------------------------------------------------------------------
# batch norm, max-pool
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,H,W,Cn", [(2, 8, 12, 16), (3, 48, 160, 64), (2, 6, 20, 512), (12, 12, 40, 256)])
def test_bn_train_fwd_bwd(N, H, W, Cn):
    ops, L = _ops()
    z = (rnd((N, Cn, H, W), 40) * 2 + rnd((1, Cn, 1, 1), 41) * 3).requires_grad_(True)
    gamma, beta = rnd((Cn,), 42, 0.5, 1.5).requires_grad_(True), rnd((Cn,), 43).requires_grad_(True)
    rm, rv = rnd((Cn,), 44), rnd((Cn,), 45, 0.5, 2.0)
    res = rnd((N, Cn, H, W), 46)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = F.relu(F.batch_norm(z, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5) + res)
    g = rnd(tuple(y.shape), 47)
    y.backward(g)
    M = N * H * W
    zs = nhwc(z.detach())
    dev = lambda t: t.detach().clone().cuda()
    rmd, rvd, nbt = dev(rm), dev(rv), torch.zeros((), dtype=torch.int64, device="cuda")
    mean, invstd, scale, shift = (torch.empty(Cn, device="cuda") for _ in range(4))
    ops.bn_train_stats(zs.view(M, Cn), dev(gamma), dev(beta), rmd, rvd, nbt, mean, invstd, scale, shift)
    check(rmd, rm_ref, "running_mean", 1e-5)
    check(rvd, rv_ref, "running_var", 1e-5)
    assert int(nbt) == 1
    ys = torch.empty_like(zs)
    ops.bn_apply(zs.view(M, Cn), scale, shift, ys.view(M, Cn), residual=nhwc(res).view(M, Cn), relu=True)
    check(nchw(ys), y.detach(), "bn_apply")
    dz, gout = torch.empty_like(zs), torch.empty_like(zs)
    dgam, dbet = torch.empty(Cn, device="cuda"), torch.empty(Cn, device="cuda")
    ops.bn_bwd(nhwc(g).view(M, Cn), ys.view(M, Cn), zs.view(M, Cn), mean, invstd, dev(gamma), dz.view(M, Cn), dgam, dbet,
               g_out=gout.view(M, Cn))
    check(nchw(dz), z.grad, "bn_bwd dz")
    check(dgam, gamma.grad, "bn_bwd dgamma")
    check(dbet, beta.grad, "bn_bwd dbeta")
    check(nchw(gout), g * (y.detach() > 0).float(), "bn_bwd g_out")
    # eval coefficients
    ops.bn_eval_coeffs(dev(gamma), dev(beta), dev(rm), dev(rv), scale, shift)
    ops.bn_apply(zs.view(M, Cn), scale, shift, ys.view(M, Cn), relu=False)
    check(nchw(ys), F.batch_norm(z.detach(), rm, rv, gamma.detach(), beta.detach(), False, 0.1, 1e-5), "bn eval")


@pytest.mark.parametrize("N,H,W,Cn", [(2, 8, 12, 16), (2, 96, 320, 64), (1, 7, 9, 8)])
def test_maxpool(N, H, W, Cn):
    ops, L = _ops()
    x = F.relu(rnd((N, Cn, H, W), 48)).requires_grad_(True)       # many exact ties at 0, like the real input
    y = F.max_pool2d(x, 3, 2, 1)
    g = rnd(tuple(y.shape), 49)
    y.backward(g)
    OH, OW = y.shape[2:]
    ys = torch.empty((N, OH, OW, Cn), device="cuda")
    am = torch.empty((N, OH, OW, Cn), dtype=torch.uint8, device="cuda")
    ops.maxpool_fwd(nhwc(x.detach()), ys, am)
    assert torch.equal(nchw(ys), y.detach())
    dx = torch.empty((N, H, W, Cn), device="cuda")
    ops.maxpool_bwd(nhwc(g), am, dx)
    check(nchw(dx), x.grad, "maxpool_bwd", 1e-6)


# ----------------------------------------------------------------------------------------------------------
# loss, Adam, layout
# ----------------------------------------------------------------------------------------------------------
def test_loss_against_oracle_and_golden():
    ops, L = _ops()
    from oracle import restatement as R
    from tests.golden.digest import load
    from tests.test_oracle_golden import g4_inputs
    gold = load("g4_loss")
    preds, batch = g4_inputs()
    pd = [preds[k].detach().cuda() for k in R.SCALES]
    tg = {k: v.cuda() for k, v in batch.items()}
    out = torch.empty(21, device="cuda")
    dp = [torch.empty_like(p) for p in pd]
    ops.loss_fwd_bwd(pd, tg, out, dp)
    np.testing.assert_allclose(out.cpu().double().numpy(), gold["loss.values"], rtol=2e-5)
    for k, d in zip(R.SCALES, dp):
        check(d, torch.from_numpy(gold["loss.dpred" + k]), "dpred" + k, 1e-5)
    # a second, larger, case against the oracle (forward only + backward)
    B, H, W = 3, 40, 72
    batch = R.make_batch(B, H, W, tag="lossbig")
    pr = {k: rnd((B, 4, H, W), 50 + i, -4.0, 4.0) for i, k in enumerate(R.SCALES)}
    for k in pr:
        pr[k][:, 2:] = torch.sigmoid(pr[k][:, 2:])
        pr[k].requires_grad_(True)
    losses, _ = R.loss_manager(pr, batch)
    losses["loss"].backward()
    pd = [pr[k].detach().cuda() for k in R.SCALES]
    dp = [torch.empty_like(p) for p in pd]
    ops.loss_fwd_bwd(pd, {k: v.cuda() for k, v in batch.items()}, out, dp)
    ref = np.array([float(losses[k]) for k in R.LOSS_KEYS])
    np.testing.assert_allclose(out.cpu().double().numpy(), ref, rtol=2e-5)
    for k, d in zip(R.SCALES, dp):
        check(d, pr[k].grad, "dpred big " + k, 1e-5)
    out2 = torch.empty(21, device="cuda")
    ops.loss_fwd_bwd(pd, {k: v.cuda() for k, v in batch.items()}, out2, None)      # forward-only, bit-stable
    assert torch.equal(out, out2)


def test_adam_matches_torch():
    ops, L = _ops()
    n = 100003
    p0, g1, g2 = rnd((n,), 60), rnd((n,), 61, -0.01, 0.01), rnd((n,), 62, -0.01, 0.01)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-4)
    n_pad = (n + 3) // 4 * 4
    p, m, v = torch.zeros(n_pad, device="cuda"), torch.zeros(n_pad, device="cuda"), torch.zeros(n_pad, device="cuda")
    p[:n] = p0.cuda()
    for step, g in enumerate((g1, g2), start=1):
        pr.grad = g.clone()
        opt.step()
        gd = torch.zeros(n_pad, device="cuda")
        gd[:n] = g.cuda()
        ops.adam_step(p[:n], gd[:n], m[:n], v[:n], 1e-4, 0.9, 0.999, 1e-8, step)
        err = (p[:n].cpu() - pr.detach()).abs().max().item()
        assert err <= 1.3e-7, "adam step %d: %.3e" % (step, err)      # |p| <= 1: at most ~1 ulp (6e-8) per step
    st = opt.state[pr]
    check(m[:n], st["exp_avg"], "exp_avg", 1e-6)
    check(v[:n], st["exp_avg_sq"], "exp_avg_sq", 1e-6)


def test_layout_roundtrip():
    ops, L = _ops()
    x = rnd((2, 5, 7, 9), 63).cuda()
    y = ops.nchw_to_nhwc(x)
    assert torch.equal(y, x.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(ops.nhwc_to_nchw(y), x)
    z = torch.empty(1001, device="cuda")
    ops.fill(z, 2.5)
    assert float(z.min()) == 2.5 and float(z.max()) == 2.5


@pytest.mark.parametrize("N,h,w,C0,Cout", [(12, 6, 20, 256, 256), (4, 8, 24, 32, 64), (6, 6, 16, 64, 32)])
def test_wgrad3x3_bf3_up2_gather(N, h, w, C0, Cout):
    """weight gradient of the upsampled half of a concat conv with the nearest-x2 gather inside the bf16x3 kernel"""
    ops, L = _ops()
    wt = rnd((Cout, C0, 3, 3), 350, -0.1, 0.1).double().requires_grad_(True)
    lo = rnd((N, C0, h, w), 351)
    y = _up2_ref(lo.double(), None, wt, None)
    g = rnd(tuple(y.shape), 352)
    y.backward(g.double())
    d = ops.make_desc(N, 2 * h, 2 * w, 2 * h, 2 * w, C0, 0, Cout, 3, 1, 1, L.GATHER_FWD_REFLECT_UP2)
    assert ops.conv_wgrad_bf3_supported(d)
    dw = torch.empty((Cout, C0, 3, 3), device="cuda")
    db = torch.empty((Cout,), device="cuda")
    ops.conv_wgrad_bf3(d, nhwc(lo), nhwc(g), dw, 0, db=db)
    check(dw, wt.grad, "wgrad bf3 up2 gather", 3e-6)
    check(db, g.double().sum((0, 2, 3)), "wgrad bf3 up2 gather bias", 2e-6)


@pytest.mark.parametrize("N,h,w,C0,C1,Cout", [(12, 6, 20, 256, 256, 256), (8, 24, 32, 32, 16, 32), (16, 16, 32, 32, 0, 64), (12, 4, 24, 64, 64, 96)])
def test_conv3x3_bf3_up2_concat_gather(N, h, w, C0, C1, Cout):
    """cat[nearest_x2(low), skip] -> reflect pad -> 3x3 conv + bias + ELU inside the bf16x3 tile kernel (float64 reference)"""
    ops, L = _ops()
    wt, b = rnd((Cout, C0 + C1, 3, 3), 340, -0.1, 0.1), rnd((Cout,), 341)
    lo = rnd((N, C0, h, w), 342)
    skip = rnd((N, C1, 2 * h, 2 * w), 343) if C1 else None
    ref = F.elu(_up2_ref(lo.double(), skip.double() if C1 else None, wt.double(), b.double()))
    d = ops.make_desc(N, 2 * h, 2 * w, 2 * h, 2 * w, C0, C1, Cout, 3, 1, 1, L.GATHER_FWD_REFLECT_UP2, act=L.ACT_ELU)
    assert ops.conv3x3_bf3_supported(d)
    y = torch.empty((N, 2 * h, 2 * w, Cout), device="cuda")
    ops.conv3x3_bf3(d, nhwc(lo), pack_bf3(wt), y, bias=b.cuda(), src1=nhwc(skip) if C1 else None)
    # 3e-6: the 12x40 case accumulates K = 9 * 512 = 4608 products in ONE fp32 chain since grids of >= 160 tiles run without split-K
    # (measured 2.3e-6 of max|y|; four partial chains of 1152 gave 1.6e-6) -- the rounding of the fp32 accumulator, not of the operands
    check(nchw(y), ref, "bf3 up2 concat", 3e-6)


@pytest.mark.parametrize("mode,N,H,W,C0,Cout", [
    ("zero", 12, 48, 160, 64, 64), ("reflect", 4, 96, 320, 64, 32), ("reflect", 2, 192, 640, 32, 32), ("reflect", 12, 24, 80, 128, 128),
    ("zero", 6, 12, 40, 256, 256), ("reflect", 3, 10, 46, 32, 96), ("zero", 12, 8, 32, 64, 32), ("reflect", 5, 7, 32, 32, 32),
    ("zero", 12, 6, 20, 512, 256), ("reflect", 12, 6, 20, 256, 256), ("zero", 4, 6, 20, 64, 32), ("reflect", 2, 16, 20, 32, 64)])
def test_wgrad3x3_bf3_kernel(mode, N, H, W, C0, Cout):
    """weight gradient with exactly split bf16x3 operands (wgrad3x3_bf3.hip), float64 reference, slice destination + accumulate"""
    ops, L = _ops()
    w = rnd((Cout, C0, 3, 3), 96, -0.1, 0.1).double().requires_grad_(True)
    x = rnd((N, C0, H, W), 99)
    if mode == "reflect":
        y = F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode="reflect"), w)
        gather = L.GATHER_FWD_REFLECT
    else:
        y = F.conv2d(x.double(), w, None, 1, 1)
        gather = L.GATHER_FWD_ZERO
    g = rnd(tuple(y.shape), 101)
    y.backward(g.double())
    d = ops.make_desc(N, H, W, H, W, C0, 0, Cout, 3, 1, 1, gather)
    assert ops.conv_wgrad_bf3_supported(d)
    pad = 32                                                     # destination = channel slice [pad, pad + C0) of a wider gradient
    dw = torch.full((Cout, C0 + 2 * pad, 3, 3), 7.0, device="cuda")
    db = torch.full((Cout,), 3.0, device="cuda")
    ops.conv_wgrad_bf3(d, nhwc(x), nhwc(g), dw, pad, db=db)
    check(dw[:, pad:pad + C0], w.grad, "wgrad bf3 " + mode, 3e-6)
    assert bool((dw[:, :pad] == 7.0).all()) and bool((dw[:, pad + C0:] == 7.0).all())
    bref = g.double().sum((0, 2, 3))                               # bias gradient = column sums of dz, from the same pass
    check(db, bref, "wgrad bf3 bias " + mode, 2e-6)
    ops.conv_wgrad_bf3(d, nhwc(x), nhwc(g), dw, pad, accumulate=True, db=db)
    check(dw[:, pad:pad + C0], 2 * w.grad, "wgrad bf3 accumulate " + mode, 3e-6)
    check(db, 2 * bref, "wgrad bf3 bias accumulate " + mode, 2e-6)
