"""GPU: the fp16-pair ("hp") operand format (include/footprints_hip.h): amax slots and their publication by producers, HP weight
packing, and every hp kernel against a float64 reference with the bounds of the exact bf16x3 kernels it replaces (the operands
carry 22 significant bits after a per-tensor power-of-two scaling; products and accumulation are as exact as the bf16 split's)."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_kernels import _ops, _up2_ref, check, nchw, nhwc, relerr, rnd

pytestmark = pytest.mark.gpu


def slot_of(t):
    ops, _ = _ops()
    return ops.amax_f32(t, ops.new_slot())


def pack_hp(w, dgrad=False):
    ops, _ = _ops()
    Cout, Cin = w.shape[:2]
    s = ops.new_slot()
    wp = ops.pack_conv_weight_hp(w.contiguous().cuda(), torch.empty(ops.packed_weight_elems_hp(Cout, Cin, 3, dgrad), device="cuda"), s, dgrad)
    assert ops.amax_value(s) == float(w.abs().max())
    return wp, s


def pack_job_hp(kind, w, elems, c_begin, c_count):
    """the batched path (what the engine uses): amax pass + pack pass over a one-job table"""
    ops, _ = _ops()
    wd = w.contiguous().cuda()
    wp, s = torch.empty(elems, device="cuda"), ops.new_slot()
    table = ops.build_pack_table([(kind, wd, wp, c_begin, c_count, s)], "cuda")
    ops.pack_weights_amax(table)
    ops.pack_weights_batched(table)
    assert ops.amax_value(s) == float(w.abs().max())
    return wp, s


@pytest.mark.parametrize("N,H,W", [(2, 64, 96), (1, 192, 640), (3, 34, 50)])
def test_conv_stem_hp(N, H, W):
    """fp_conv_stem_hp: the 7x7 / 2 stem with fp16-pair operands (K re-indexed as 7 rows x 24, image scaled by 2^12) against float64, at the
    tolerance of the other fp16-pair kernels; statistics sink; bias + ReLU epilogue; amax of the output published"""
    ops, L = _ops()
    img, w = rnd((N, 3, H, W), 810, 0.0, 1.0), rnd((64, 3, 7, 7), 811, -0.1, 0.1)
    ref = F.conv2d((img.double() - 0.45) / 0.225, w.double(), None, 2, 3)
    OH, OW = ref.shape[2:]
    wp, sw = pack_job_hp(L.PACK_STEM_HP, w, 11 * 64 * 16, 0, 3)
    d = ops.make_desc(N, OH, OW, H, W, 3, 0, 64, 7, 2, 3, L.GATHER_STEM)
    assert ops.conv_stem_hp_supported(d)
    y = torch.full((N, OH, OW, 64), float("nan"), device="cuda")
    tiles = N * ((OH + 7) // 8) * ((OW + 15) // 16)
    part = torch.full((tiles * 64 * 3,), float("nan"), device="cuda")
    cell = ops.bn_stats_out(part)
    so = ops.new_slot()
    ops.conv_stem_hp(d, img.cuda(), wp, y, sw, amax_out=so, bn_out=cell)
    torch.cuda.synchronize()
    check(nchw(y), ref, "stem hp", 2e-6)
    assert cell.nblk == tiles and ops.amax_value(so) == float(y.abs().max())
    assert float(part.view(tiles, 64, 3)[:, :, 0].sum(0).min()) == float(N * OH * OW)
    g, b = rnd((64,), 812, 0.5, 1.5).cuda(), rnd((64,), 813).cuda()
    outs = [[torch.zeros(64, device="cuda") for _ in range(4)] for _ in range(2)]
    rms, rvs = [torch.zeros(64, device="cuda") for _ in range(2)], [torch.ones(64, device="cuda") for _ in range(2)]
    nbt = [torch.zeros((), dtype=torch.int64, device="cuda") for _ in range(2)]
    ops.bn_train_stats_partials(part, tiles, 64, g, b, rms[0], rvs[0], nbt[0], *outs[0])
    ops.bn_train_stats(y.view(-1, 64), g, b, rms[1], rvs[1], nbt[1], *outs[1])
    yd = y.double().cpu().view(-1, 64)
    mean, invstd = yd.mean(0), 1.0 / torch.sqrt(yd.var(0, unbiased=False) + 1e-5)
    for k, want in enumerate((mean, invstd)):
        for o in outs:
            assert relerr(o[k], want) < 2e-6, k
    y2 = torch.empty_like(y)                                       # bias + ReLU (the eval path's folded BatchNorm), no statistics
    cell = ops.bn_stats_out(part)
    d2 = ops.make_desc(N, OH, OW, H, W, 3, 0, 64, 7, 2, 3, L.GATHER_STEM, act=L.ACT_RELU)
    ops.conv_stem_hp(d2, img.cuda(), wp, y2, sw, bias=b, bn_out=cell)
    assert cell.nblk == 0
    check(nchw(y2), torch.relu(ref + b.double().cpu().view(1, 64, 1, 1)), "stem hp + bias + relu", 2e-6)


@pytest.mark.parametrize("N,H,W", [(2, 64, 96), (1, 192, 640), (3, 34, 50), (12, 192, 640)])
def test_conv_stem_wgrad_hp(N, H, W):
    """fp_conv_stem_wgrad_hp: the stem's weight gradient with fp16-pair operands (transposing LDS reads, K blocks of one kernel row) against
    float64; ragged tiles at the right / bottom edge; accumulate"""
    ops, L = _ops()
    img = rnd((N, 3, H, W), 820, 0.0, 1.0)
    w = rnd((64, 3, 7, 7), 821, -0.1, 0.1).double().requires_grad_(True)
    y = F.conv2d((img.double() - 0.45) / 0.225, w, None, 2, 3)
    g = rnd(tuple(y.shape), 822)
    y.backward(g.double())
    OH, OW = y.shape[2:]
    d = ops.make_desc(N, OH, OW, H, W, 3, 0, 64, 7, 2, 3, L.GATHER_STEM)
    gz = nhwc(g)
    dw = torch.full((64, 3, 7, 7), float("nan"), device="cuda")
    ops.conv_stem_wgrad_hp(d, img.cuda(), gz, dw, slot_of(gz))
    check(dw, w.grad, "stem wgrad hp", 2e-6)
    first = dw.clone()
    ops.conv_stem_wgrad_hp(d, img.cuda(), gz, dw, slot_of(gz), accumulate=True)
    assert torch.equal(dw, first + first)                            # bit-reproducible, and accumulate adds exactly
    dw32 = torch.empty_like(dw)                                      # the fp32-MFMA kernel on the same inputs: the fp16-pair form is at least as close
    ops.conv_wgrad(d, img.cuda(), None, gz, dw32)
    assert relerr(first, w.grad) <= max(2.0 * relerr(dw32, w.grad), 5e-7)


def test_amax_reduction_and_zeroing():
    ops, _ = _ops()
    for n, scale in ((1, 1.0), (3, 1e-30), (1000, 1e30), (12 * 96 * 320 * 64, 7.0), (1 << 20, 1e-3)):
        x = (torch.rand(n + 4, device="cuda")[:n] - 0.5) * scale
        s = slot_of(x)
        assert ops.amax_value(s) == float(x.abs().max()), (n, scale)
        ops.zero_u32(s)
        assert int(s.abs().sum()) == 0
    z = torch.zeros(4096, device="cuda")
    assert ops.amax_value(slot_of(z)) == 0.0
    x = torch.rand(4096, device="cuda")
    x[77] = float("inf")
    assert ops.amax_value(slot_of(x)) == float("inf")


def test_producers_publish_their_exact_amax():
    """fp_aux.amax_out: bn_apply, bn_bwd (dz), maxpool_fwd, up2_fold_bwd, head_dgrad and the split-K reduce publish max |output| from
    XCD-local atomics; repeated launches must never lose an update"""
    ops, L = _ops()
    torch.manual_seed(5)
    for rep in range(6):
        M, C = 12 * 48 * 160, 64
        z, res = torch.randn(M, C, device="cuda") * (10.0 ** (rep - 3)), torch.randn(M, C, device="cuda")
        sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
        y, s = torch.empty_like(z), ops.new_slot()
        ops.bn_apply(z, sc, sh, y, residual=res, relu=True, amax_out=s)
        assert ops.amax_value(s) == float(y.abs().max())
        x4 = z.view(12, 48, 160, C)
        pool, am, s = torch.empty((12, 24, 80, C), device="cuda"), torch.empty((12, 24, 80, C), dtype=torch.uint8, device="cuda"), ops.new_slot()
        ops.maxpool_fwd(x4, pool, am, amax_out=s)
        assert ops.amax_value(s) == float(pool.abs().max())
        ext, s = torch.randn(2, 26, 82, 64, device="cuda"), ops.new_slot()
        dlow = ops.up2_fold_bwd(ext, torch.empty((2, 24, 80, 64), device="cuda"), amax_out=s)
        assert ops.amax_value(s) == float(dlow.abs().max())
        dzl, hw = torch.randn(2, 48, 160, 2, device="cuda"), torch.randn(2, 64, 3, 3, device="cuda")
        dx, s = torch.empty((2, 48, 160, 64), device="cuda"), ops.new_slot()
        ops.head_dgrad(dzl, hw, dx, amax_out=s)
        assert ops.amax_value(s) == float(dx.abs().max())
        # bn_bwd: the apply stage publishes max |dz|
        dy, out = torch.randn(M, C, device="cuda"), torch.relu(torch.randn(M, C, device="cuda"))
        mean, invstd, gamma = torch.randn(C, device="cuda"), torch.rand(C, device="cuda") + 0.5, torch.rand(C, device="cuda") + 0.5
        dz, s = torch.empty(M, C, device="cuda"), ops.new_slot()
        ops.bn_bwd(dy, out, z, mean, invstd, gamma, dz, torch.empty(C, device="cuda"), torch.empty(C, device="cuda"), amax_out=s)
        assert ops.amax_value(s) == float(dz.abs().max())
    # a sink is consumed by exactly one launch
    s = ops.new_slot()
    ops.bn_apply(z, sc, sh, y, amax_out=s)
    before = ops.amax_value(s)
    ops.bn_apply(z * 100.0, sc, sh, y)
    assert ops.amax_value(s) == before


@pytest.mark.parametrize("mode,N,H,W,C0,Cout", [
    ("zero", 12, 48, 160, 64, 64), ("reflect", 4, 96, 320, 64, 64), ("reflect", 2, 192, 640, 32, 32), ("reflect", 12, 24, 80, 128, 128),
    ("zero", 6, 40, 176, 32, 96), ("zero", 20, 24, 80, 24, 40), ("dgrad", 12, 48, 160, 64, 64), ("dgrad", 12, 24, 80, 128, 128),
    ("dgrad_reflect", 12, 48, 160, 64, 64), ("dgrad_reflect", 2, 192, 640, 32, 32), ("dgrad_reflect", 6, 40, 144, 64, 128),
    ("zero", 12, 12, 40, 256, 256), ("zero", 12, 6, 20, 512, 512), ("reflect", 12, 12, 40, 256, 128),
    ("dgrad", 12, 6, 20, 512, 512), ("dgrad_reflect", 12, 6, 20, 256, 512), ("zero", 8, 18, 60, 48, 80)])
@pytest.mark.parametrize("scale", [1.0, 1e-7])
def test_conv3x3_hp_kernel(mode, N, H, W, C0, Cout, scale):
    """same cases and bound as test_conv3x3_bf3_kernel; scale 1e-7 = gradient-sized operands (the per-tensor scaling must carry them);
    the published amax of the output is exact, also on the split-K path"""
    ops, L = _ops()
    tol = 2e-6
    so = ops.new_slot()
    if mode == "dgrad_reflect":
        pre = rnd((N, C0, H, W), 92, -2.0, 2.0).requires_grad_(True)
        w = rnd((Cout, C0, 3, 3), 93, -0.1, 0.1)
        x = F.elu(pre)
        yr = F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode="reflect"), w.double())
        g, extra = rnd(tuple(yr.shape), 94) * scale, rnd(tuple(x.shape), 95) * scale
        ((yr * g.double()).sum() + (x.double() * extra.double()).sum()).backward()
        dz = torch.empty((N, H, W, C0), device="cuda")
        d = ops.make_desc(N, H, W, H, W, Cout, 0, C0, 3, 1, 1, L.GATHER_DGRAD_REFLECT, epi=L.EPI_ACTGRAD_ELU)
        wp, sw = pack_hp(w, dgrad=True)
        gz = nhwc(g)
        ops.conv3x3_hp(d, gz, wp, dz, slot_of(gz), sw, amax_out=so, actsrc=nhwc(x.detach()), addend=nhwc(extra))
        check(nchw(dz), pre.grad, "hp dgrad_reflect (fold) + epilogue", tol)
        assert ops.amax_value(so) == float(dz.abs().max())
        return
    if mode == "dgrad":
        x = rnd((N, C0, H, W), 80).double().requires_grad_(True)
        w = rnd((Cout, C0, 3, 3), 81, -0.1, 0.1)
        yr = F.conv2d(x, w.double(), None, 1, 1)
        g = rnd(tuple(yr.shape), 82) * scale
        add, msk, act = rnd(tuple(x.shape), 83) * scale, rnd(tuple(x.shape), 84), rnd(tuple(x.shape), 85)
        yr.backward(g.double())
        ref = (x.grad + add * (msk > 0).float()) * (act > 0).float()
        dx = torch.empty((N, H, W, C0), device="cuda")
        d = ops.make_desc(N, H, W, H, W, Cout, 0, C0, 3, 1, 1, L.GATHER_DGRAD_ZERO, epi=L.EPI_ACTGRAD_RELU)
        wp, sw = pack_hp(w, dgrad=True)
        gz = nhwc(g)
        ops.conv3x3_hp(d, gz, wp, dx, slot_of(gz), sw, amax_out=so, addend=nhwc(add), addend_mask=nhwc(msk), actsrc=nhwc(act))
        check(nchw(dx), ref, "hp dgrad_zero + epilogue", tol)
        assert ops.amax_value(so) == float(dx.abs().max())
        return
    w, b = rnd((Cout, C0, 3, 3), 86, -0.1, 0.1), rnd((Cout,), 87) * scale
    x = rnd((N, C0, H, W), 90) * scale
    if mode == "reflect":
        ref = F.elu(F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode="reflect"), w.double(), b.double()))
        gather = L.GATHER_FWD_REFLECT
    else:
        ref = F.elu(F.conv2d(x.double(), w.double(), b.double(), 1, 1))
        gather = L.GATHER_FWD_ZERO
    y = torch.empty((N, H, W, Cout), device="cuda")
    d = ops.make_desc(N, H, W, H, W, C0, 0, Cout, 3, 1, 1, gather, act=L.ACT_ELU)
    wp, sw = pack_hp(w)
    xs = nhwc(x)
    ops.conv3x3_hp(d, xs, wp, y, slot_of(xs), sw, amax_out=so, bias=b.cuda())
    check(nchw(y), ref, "hp forward " + mode, tol)
    assert ops.amax_value(so) == float(y.abs().max())


def test_conv3x3_hp_dynamic_range_and_specials():
    """heavy-tailed operands (|x| over 24 binades), an all-zero source, and inf / nan propagation like fp32's"""
    ops, L = _ops()
    N, H, W, C0, Cout = 4, 48, 160, 64, 64
    g = torch.Generator().manual_seed(3)
    x = torch.randn((N, C0, H, W), generator=g) * torch.exp2(-torch.rand((N, C0, H, W), generator=g) * 24.0)
    w = torch.randn((Cout, C0, 3, 3), generator=g) * 0.05 * torch.exp2(-torch.rand((Cout, C0, 3, 3), generator=g) * 12.0)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    d = ops.make_desc(N, H, W, H, W, C0, 0, Cout, 3, 1, 1, L.GATHER_FWD_ZERO)
    wp, sw = pack_hp(w)
    xs, y = nhwc(x), torch.empty((N, H, W, Cout), device="cuda")
    ops.conv3x3_hp(d, xs, wp, y, slot_of(xs), sw)
    e_hp = ((nchw(y).double() - ref).norm() / ref.norm()).item()
    e_32 = ((F.conv2d(x, w, None, 1, 1).double() - ref).norm() / ref.norm()).item()
    assert e_hp <= 2.0 * e_32 + 1e-7, (e_hp, e_32)          # as good as an fp32 convolution on the CPU
    z = torch.zeros_like(xs)
    ops.conv3x3_hp(d, z, wp, y, slot_of(z), sw)
    assert float(y.abs().max()) == 0.0
    xs2 = xs.clone()
    xs2[1, 7, 9, 3] = float("inf")
    ops.conv3x3_hp(d, xs2, wp, y, slot_of(xs2), sw)
    assert not bool(torch.isfinite(y[1, 6:9, 8:11]).all()) and bool(torch.isfinite(y[0]).all())


@pytest.mark.parametrize("N,H,W,C0,Cout,emits", [
    (12, 48, 160, 64, 64, True),         # encoder layer 1: 8 x 16 tiles, interior
    (12, 24, 80, 128, 128, True),        # layer 2 (small-grid WPF variant, two channel tiles)
    (12, 12, 40, 256, 256, True),        # layer 3: 6 x 20 tiles (120 of 128 MFMA rows valid)
    (6, 44, 72, 32, 32, True),           # ragged: partial tiles on both borders, 32-channel variant (four M waves); 180 workgroups (under 128 the shape goes to the flattened kernel)
    (10, 21, 45, 64, 128, True),         # ragged, two channel tiles
    (12, 6, 20, 512, 512, "reduce"),     # split-K grid: the statistics come out of the reduce launch, one triple per block of the reduce launch
    (12, 6, 20, 256, 512, "reduce"),
])
def test_conv3x3_hp_emits_batchnorm_partials(N, H, W, C0, Cout, emits):
    """fp_aux.bn_part: the forward tile convolution in front of a train-mode BatchNorm writes (count, mean, M2) per pixel tile and
    channel of what it stores; fp_bn_train_stats_partials turns them into the same coefficients as fp_bn_train_stats on the tensor"""
    ops, L = _ops()
    w = rnd((Cout, C0, 3, 3), 601, -0.1, 0.1)
    x = rnd((N, C0, H, W), 602) * 2.0 + 0.75                      # a mean of the order of the spread, like post-ReLU activations
    d = ops.make_desc(N, H, W, H, W, C0, 0, Cout, 3, 1, 1, L.GATHER_FWD_ZERO)
    y = torch.empty((N, H, W, Cout), device="cuda")
    wp, sw = pack_hp(w)
    xs = nhwc(x)
    cap = N * ((H + 5) // 6) * ((W + 15) // 16) * Cout * 3
    rows = 256 // (Cout // 4) * 4
    nred = min(512, (N * H * W + rows - 1) // rows)
    if emits == "reduce":
        cap = max(cap, nred * Cout * 3)
    part = torch.full((cap,), float("nan"), device="cuda")
    cell = ops.bn_stats_out(part)
    ops.conv3x3_hp(d, xs, wp, y, slot_of(xs), sw, bn_out=cell)
    torch.cuda.synchronize()
    expect_tiles = cell.nblk
    if emits == "reduce":
        assert expect_tiles == nred
    else:
        assert expect_tiles in ((N * ((H + 7) // 8) * ((W + 15) // 16), N * ((H + 5) // 6) * ((W + 19) // 20)) if emits else (0,))
    check(nchw(y), F.conv2d(x.double(), w.double(), None, 1, 1), "hp forward with statistics sink", 2e-6)
    y2 = torch.empty_like(y)                                       # the sink is one-shot: the next launch emits nothing
    ops.conv3x3_hp(d, xs, wp, y2, slot_of(xs), sw)
    assert torch.equal(y, y2)
    if expect_tiles == 0:
        assert bool(torch.isnan(part).all())
        return
    used = expect_tiles * Cout * 3
    assert not bool(torch.isnan(part[:used]).any()) and bool(torch.isnan(part[used:]).all())
    assert float(part[:used].view(expect_tiles, Cout, 3)[:, :, 0].sum(0).min()) == float(N * H * W)      # every pixel counted once per channel
    g, b = rnd((Cout,), 603, 0.5, 1.5).cuda(), rnd((Cout,), 604).cuda()
    outs = [[torch.zeros(Cout, device="cuda") for _ in range(4)] for _ in range(2)]
    rms = [torch.zeros(Cout, device="cuda") for _ in range(2)]
    rvs = [torch.ones(Cout, device="cuda") for _ in range(2)]
    nbt = [torch.zeros((), dtype=torch.int64, device="cuda") for _ in range(2)]
    ops.bn_train_stats_partials(part, expect_tiles, Cout, g, b, rms[0], rvs[0], nbt[0], *outs[0])
    ops.bn_train_stats(y.view(-1, Cout), g, b, rms[1], rvs[1], nbt[1], *outs[1])
    yd, gc, bc = y.double().cpu().view(-1, Cout), g.double().cpu(), b.double().cpu()
    mean, var = yd.mean(0), yd.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    for k, ref in enumerate((mean, invstd, gc * invstd, bc - mean * gc * invstd)):
        for o in outs:
            assert relerr(o[k], ref) < 2e-6, (k, relerr(o[k], ref))
    assert relerr(rms[0], rms[1].cpu()) < 1e-6 and relerr(rvs[0], rvs[1].cpu()) < 1e-6 and int(nbt[0]) == int(nbt[1]) == 1


@pytest.mark.parametrize("N,H,W,C,emits", [
    (12, 48, 160, 64, True),             # layer 1: 8 x 16 tiles, 720 workgroups
    (12, 24, 80, 128, True),             # layer 2 (small-grid WPF variant, two channel tiles)
    (12, 12, 40, 256, True),             # layer 3: 6 x 20 tiles
    (10, 21, 45, 128, True),             # ragged tiles on both borders, two channel tiles (180 workgroups: unsplit)
    (12, 6, 20, 512, "reduce"),          # split-K grid: the sums come out of the reduce launch, one pair per block of the reduce launch
])
@pytest.mark.parametrize("fmt", ["fp16_pair", "exact"])
def test_conv3x3_hp_data_gradient_emits_batchnorm_backward_sums(N, H, W, C, emits, fmt):
    """fp_aux.bn_part + bnb_*: a tile data gradient whose epilogue applies the ReLU mask stores g = (dgrad + residual) * (out > 0) of the
    BatchNorm below it AND that BatchNorm's backward sums (sum g, sum g * xhat) per pixel tile and channel; fp_bn_bwd_partials turns them into
    the same dz / dgamma / dbeta as fp_bn_bwd on (dy, relu_out, z) -- against float64 (torchvision BatchNorm2d backward)."""
    ops, L = _ops()
    w = rnd((C, C, 3, 3), 701, -0.1, 0.1)                          # the forward conv C -> C whose data gradient is taken
    gz = rnd((N, C, H, W), 702)                                    # gradient wrt its output
    resid = rnd((N, C, H, W), 703)                                 # residual-branch gradient added on top (addend)
    out = rnd((N, C, H, W), 704)                                   # the BatchNorm block's output after ReLU (sign = mask)
    z = rnd((N, C, H, W), 705) * 1.5 + 0.3                         # the BatchNorm's input
    gamma = rnd((C,), 706, 0.5, 1.5)
    # float64 reference: dy = conv_transpose(gz) + resid; g = dy * (out > 0); batch-norm backward on (g, z)
    xin = torch.zeros((N, C, H, W), dtype=torch.float64, requires_grad=True)
    F.conv2d(xin, w.double(), None, 1, 1).backward(gz.double())
    g64 = (xin.grad + resid.double()) * (out > 0).double()
    zd = z.double()
    mean, var = zd.mean((0, 2, 3)), zd.var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    xhat = (zd - mean.view(1, C, 1, 1)) * invstd.view(1, C, 1, 1)
    s1, s2 = g64.sum((0, 2, 3)), (g64 * xhat).sum((0, 2, 3))
    M = N * H * W
    dz64 = gamma.double().view(1, C, 1, 1) * invstd.view(1, C, 1, 1) * (g64 - (s1 / M).view(1, C, 1, 1) - xhat * (s2 / M).view(1, C, 1, 1))
    # device
    d = ops.make_desc(N, H, W, H, W, C, 0, C, 3, 1, 1, L.GATHER_DGRAD_ZERO, epi=L.EPI_ACTGRAD_RELU)
    hp = fmt == "fp16_pair"             # "exact": the same sink behind fp_conv3x3_bf3 (bf16x3 operands; the engine's default since round 5)
    if hp:
        wp, sw = pack_hp(w, dgrad=True)
    else:
        from tests.test_gpu_kernels import pack_bf3
        wp, sw = pack_bf3(w, dgrad=True), None
    gzs, zs = nhwc(gz), nhwc(z)

    def conv(dst, amax_out=None, bn_out=None):
        if hp:
            ops.conv3x3_hp(d, gzs, wp, dst, slot_of(gzs), sw, amax_out=amax_out, addend=nhwc(resid), actsrc=nhwc(out), bn_out=bn_out)
        else:
            ops.conv3x3_bf3(d, gzs, wp, dst, addend=nhwc(resid), actsrc=nhwc(out), bn_out=bn_out)
    mean_d, invstd_d = mean.float().cuda(), invstd.float().cuda()
    cap = N * ((H + 5) // 6) * ((W + 15) // 16) * C * 2
    rows = 256 // (C // 4) * 4
    nred = min(512, (N * H * W + rows - 1) // rows)
    if emits == "reduce":
        cap = max(cap, nred * C * 2)
    part = torch.full((cap,), float("nan"), device="cuda")
    gout = torch.empty((N, H, W, C), device="cuda")
    cell = ops.bn_bwd_out(part, zs.view(-1, C), mean_d, invstd_d)
    so = ops.new_slot()
    conv(gout, so, cell)
    torch.cuda.synchronize()
    tiles = cell.nblk
    if emits == "reduce":
        assert tiles == nred
    else:
        assert tiles in ((N * ((H + 7) // 8) * ((W + 15) // 16), N * ((H + 5) // 6) * ((W + 19) // 20)) if emits else (0,))
    check(nchw(gout), g64, "masked data gradient", 2e-6)
    if hp:
        assert ops.amax_value(so) == float(gout.abs().max())
    g2 = torch.empty_like(gout)                                    # the sink is one-shot
    part_before = part.clone()
    conv(g2)
    torch.cuda.synchronize()
    assert torch.equal(gout, g2) and torch.equal(torch.nan_to_num(part, nan=7.0), torch.nan_to_num(part_before, nan=7.0))
    dz_ref, dg_ref, db_ref = (torch.empty((N, H, W, C), device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"))
    ops.bn_bwd(gout.view(-1, C), None, zs.view(-1, C), mean_d, invstd_d, gamma.cuda(), dz_ref.view(-1, C), dg_ref, db_ref)
    if tiles == 0:
        assert bool(torch.isnan(part).all())
        check(nchw(dz_ref), dz64, "bn backward after the masked data gradient", 3e-6)
        return
    used = tiles * C * 2
    assert not bool(torch.isnan(part[:used]).any()) and bool(torch.isnan(part[used:]).all())
    sums = part[:used].view(tiles, C, 2).double().sum(0).cpu()
    # (six MFMA products per multiply-add instead of three: twice the accumulation steps into the fp32 accumulator, and the per-channel sum over
    # ~1e5 pixels shows their common sign -- measured 2.2e-6 of the largest channel sum with the exact operands where the fp16 pairs sit at ~1e-6)
    tol_s = 2e-6 if hp else 3e-6
    assert relerr(sums[:, 0], s1) < tol_s and relerr(sums[:, 1], s2) < tol_s
    dz, dg, db = torch.empty((N, H, W, C), device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    sz = ops.new_slot()
    ops.bn_bwd_partials(gout.view(-1, C), zs.view(-1, C), mean_d, invstd_d, gamma.cuda(), dz.view(-1, C), dg, db, part, tiles, amax_out=sz)
    check(nchw(dz), dz64, "bn backward from the epilogue's sums", 3e-6)
    check(nchw(dz_ref), dz64, "bn backward, own reduction", 3e-6)
    assert relerr(dg, s2) < tol_s and relerr(db, s1) < tol_s and relerr(dg_ref, s2) < tol_s and relerr(db_ref, s1) < tol_s
    assert ops.amax_value(sz) == float(dz.abs().max())
    dg2, db2 = dg.clone(), db.clone()                               # accumulate = True adds on top
    ops.bn_bwd_partials(gout.view(-1, C), zs.view(-1, C), mean_d, invstd_d, gamma.cuda(), dz.view(-1, C), dg2, db2, part, tiles, accumulate=True)
    assert relerr(dg2, 2 * s2) < tol_s and relerr(db2, 2 * s1) < tol_s


@pytest.mark.parametrize("N,h,w,C0,C1,Cout", [(12, 6, 20, 256, 256, 256), (8, 24, 32, 32, 16, 32), (16, 16, 32, 32, 0, 64), (12, 4, 24, 64, 64, 96)])
def test_conv3x3_hp_up2_concat_gather(N, h, w, C0, C1, Cout):
    """cat[nearest_x2(low), skip] inside the tile kernel: the two sources share one scale (the larger amax); sources of very different size"""
    ops, L = _ops()
    wt, b = rnd((Cout, C0 + C1, 3, 3), 340, -0.1, 0.1), rnd((Cout,), 341)
    lo = rnd((N, C0, h, w), 342) * 1e-3
    skip = rnd((N, C1, 2 * h, 2 * w), 343) * 30.0 if C1 else None
    ref = F.elu(_up2_ref(lo.double(), skip.double() if C1 else None, wt.double(), b.double()))
    d = ops.make_desc(N, 2 * h, 2 * w, 2 * h, 2 * w, C0, C1, Cout, 3, 1, 1, L.GATHER_FWD_REFLECT_UP2, act=L.ACT_ELU)
    y = torch.empty((N, 2 * h, 2 * w, Cout), device="cuda")
    wp, sw = pack_hp(wt)
    los, sks = nhwc(lo), nhwc(skip) if C1 else None
    ops.conv3x3_hp(d, los, wp, y, slot_of(los), sw, bias=b.cuda(), src1=sks, amax_src1=slot_of(sks) if C1 else None)
    check(nchw(y), ref, "hp up2 concat", 2e-6)


@pytest.mark.parametrize("mode,N,H,W,C0,Cout", [
    ("zero", 12, 48, 160, 64, 64), ("reflect", 4, 96, 320, 64, 32), ("reflect", 2, 192, 640, 32, 32), ("reflect", 12, 24, 80, 128, 128),
    ("zero", 6, 12, 40, 256, 256), ("reflect", 3, 10, 46, 32, 96), ("zero", 12, 6, 20, 512, 256), ("reflect", 2, 16, 20, 32, 64)])
@pytest.mark.parametrize("gscale", [1.0, 1e-8])
def test_wgrad3x3_hp_kernel(mode, N, H, W, C0, Cout, gscale):
    ops, L = _ops()
    w = rnd((Cout, C0, 3, 3), 96, -0.1, 0.1).double().requires_grad_(True)
    x = rnd((N, C0, H, W), 99) * 20.0
    if mode == "reflect":
        y = F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode="reflect"), w)
        gather = L.GATHER_FWD_REFLECT
    else:
        y = F.conv2d(x.double(), w, None, 1, 1)
        gather = L.GATHER_FWD_ZERO
    g = rnd(tuple(y.shape), 101) * gscale
    y.backward(g.double())
    d = ops.make_desc(N, H, W, H, W, C0, 0, Cout, 3, 1, 1, gather)
    assert ops.conv_wgrad_bf3_supported(d)
    pad = 32
    dw = torch.full((Cout, C0 + 2 * pad, 3, 3), 7.0, device="cuda")
    db = torch.full((Cout,), 3.0, device="cuda")
    xs, gz = nhwc(x), nhwc(g)
    am = (slot_of(xs), slot_of(gz))
    ops.conv_wgrad_bf3(d, xs, gz, dw, pad, db=db, amax=am)
    check(dw[:, pad:pad + C0], w.grad, "wgrad hp " + mode, 3e-6)
    assert bool((dw[:, :pad] == 7.0).all()) and bool((dw[:, pad + C0:] == 7.0).all())
    bref = g.double().sum((0, 2, 3))
    check(db, bref, "wgrad hp bias " + mode, 2e-6)
    ops.conv_wgrad_bf3(d, xs, gz, dw, pad, accumulate=True, db=db, amax=am)
    check(dw[:, pad:pad + C0], 2 * w.grad, "wgrad hp accumulate " + mode, 3e-6)


@pytest.mark.parametrize("N,h,w,C0,Cout", [(12, 6, 20, 256, 256), (4, 8, 24, 32, 64), (6, 6, 16, 64, 32)])
def test_wgrad3x3_hp_up2_gather(N, h, w, C0, Cout):
    ops, L = _ops()
    wt = rnd((Cout, C0, 3, 3), 350, -0.1, 0.1).double().requires_grad_(True)
    lo = rnd((N, C0, h, w), 351)
    y = _up2_ref(lo.double(), None, wt, None)
    g = rnd(tuple(y.shape), 352) * 1e-5
    y.backward(g.double())
    d = ops.make_desc(N, 2 * h, 2 * w, 2 * h, 2 * w, C0, 0, Cout, 3, 1, 1, L.GATHER_FWD_REFLECT_UP2)
    dw, db = torch.empty((Cout, C0, 3, 3), device="cuda"), torch.empty((Cout,), device="cuda")
    los, gz = nhwc(lo), nhwc(g)
    ops.conv_wgrad_bf3(d, los, gz, dw, 0, db=db, amax=(slot_of(los), slot_of(gz)))
    check(dw, wt.grad, "wgrad hp up2 gather", 3e-6)
    check(db, g.double().sum((0, 2, 3)), "wgrad hp up2 gather bias", 2e-6)


@pytest.mark.parametrize("N,h,w,C0,Cout", [(2, 24, 80, 64, 64), (2, 96, 320, 64, 32), (1, 5, 7, 32, 16), (2, 1, 1, 16, 16), (2, 9, 17, 20, 72)])
def test_up2_phase_fwd_hp(N, h, w, C0, Cout):
    ops, L = _ops()
    wt, b = rnd((Cout, C0, 3, 3), 304, -0.1, 0.1), rnd((Cout,), 305)
    lo, prev = rnd((N, C0, h, w), 306) * 50.0, rnd((N, Cout, 2 * h, 2 * w), 307)
    ref = F.elu(_up2_ref(lo.double(), None, wt.double(), b.double()) + prev.double())
    wph, sw = pack_job_hp(L.PACK_UP2_FWD_HP, wt, ops.up2_packed_weight_elems(Cout, C0), 0, C0)
    y, los, so = nhwc(prev), nhwc(lo), ops.new_slot()
    ops.conv_up2_phase_fwd_hp(los, wph, b.cuda(), y, slot_of(los), sw, amax_out=so, act=L.ACT_ELU, addend=y)
    check(nchw(y), ref, "up2 phase fwd hp", 3e-6)
    assert ops.amax_value(so) == float(y.abs().max())


@pytest.mark.parametrize("N,h,w,C0,Cout", [(2, 24, 80, 64, 64), (1, 96, 320, 64, 32), (2, 6, 20, 256, 256), (2, 1, 1, 16, 16), (1, 3, 2, 24, 48),
                                           (3, 9, 17, 20, 72)])
def test_up2_phase_dgrad_hp(N, h, w, C0, Cout):
    ops, L = _ops()
    wt = rnd((Cout, C0, 3, 3), 340, -0.1, 0.1)
    lo = rnd((N, C0, h, w), 341).double().requires_grad_(True)
    yr = _up2_ref(lo, None, wt.double(), None)
    g = rnd(tuple(yr.shape), 342) * 1e-6
    yr.backward(g.double())
    gz = nhwc(g)
    wp, sw = pack_job_hp(L.PACK_UP2_DGRAD_HP, wt, ops.up2_packed_weight_elems(C0, Cout), 0, C0)
    ext = ops.conv_up2_phase_dgrad_hp(gz, wp, torch.full((N, h + 2, w + 2, C0), float("nan"), device="cuda"), slot_of(gz), sw)
    dlow = ops.up2_fold_bwd(ext, torch.empty((N, h, w, C0), device="cuda"))
    check(nchw(dlow), lo.grad, "up2 phase dgrad hp", 3e-6)


@pytest.mark.parametrize("N,h,w,C0,Cout,acc", [(2, 24, 80, 64, 64, False), (2, 12, 48, 32, 32, True), (1, 48, 160, 64, 32, False),
                                               (16, 2, 16, 32, 32, False), (1, 96, 320, 64, 32, True), (3, 6, 27, 64, 64, False)])
def test_up2_phase_wgrad_hp(N, h, w, C0, Cout, acc):
    ops, L = _ops()
    wt = rnd((Cout, C0, 3, 3), 320, -0.1, 0.1).requires_grad_(True)
    lo = rnd((N, C0, h, w), 321)
    yr = _up2_ref(lo, None, wt, None)
    g = rnd(tuple(yr.shape), 323)
    yr.backward(g)
    init = rnd((Cout, C0, 3, 3), 324) if acc else torch.full((Cout, C0, 3, 3), float("nan"))
    binit = rnd((Cout,), 325) if acc else torch.full((Cout,), float("nan"))
    dw, db = init.clone().cuda(), binit.clone().cuda()
    los, gz = nhwc(lo), nhwc(g)
    ops.conv_up2_phase_wgrad_hp(los, gz, dw, slot_of(los), slot_of(gz), 0, accumulate=acc, db=db)
    check(db.cpu(), g.sum((0, 2, 3)) + (binit if acc else 0), "up2 phase wgrad hp bias", 1e-5)
    check(dw.cpu(), wt.grad + (init if acc else 0), "up2 phase wgrad hp")


# ---- fp_conv_igemm_hp (round 3): the flattened kernel with fp16-pair operands -- 3x3 stride 2, 1x1 (stride 1 / 2), their data gradients ----
@pytest.mark.parametrize("N,H,W,Cin,Cout,K,stride", [
    (12, 48, 160, 64, 128, 3, 2), (12, 24, 80, 128, 256, 3, 2), (12, 12, 40, 256, 512, 3, 2),      # layer2/3/4 block 0 conv1 (KITTI bs=12)
    (12, 48, 160, 64, 128, 1, 2), (12, 12, 40, 256, 512, 1, 2), (4, 128, 160, 64, 128, 1, 2),      # 1x1 downsample convs (+ Matterport layer2)
    (2, 9, 13, 16, 24, 3, 2), (3, 7, 5, 4, 8, 1, 1), (1, 6, 20, 512, 128, 1, 1), (2, 10, 14, 36, 40, 3, 2)])     # ragged, odd sizes, C % 16 != 0
@pytest.mark.parametrize("mode", ["fwd", "dgrad"])
@pytest.mark.parametrize("fmt", ["fp16_pair", "exact"])
def test_igemm_hp_against_float64(N, H, W, Cin, Cout, K, stride, mode, fmt):
    """against a float64 convolution / its data gradient of the same fp32 operands: relative L2 <= 1e-6 (measured ~3e-7, like the tile
    kernel's), split-K and parity-major plans included; the accumulate epilogue (the form the backward pass uses) on top"""
    import torch.nn.functional as F
    from footprints_amd import _lib as L
    from footprints_amd import ops
    g = torch.Generator().manual_seed(K * 1000 + Cin + H)
    x = ((torch.rand(N, Cin, H, W, generator=g) * 2 - 1) * (torch.rand(N, Cin, H, W, generator=g) > 0.3)).cuda()
    w = ((torch.rand(Cout, Cin, K, K, generator=g) * 2 - 1) * 0.1).cuda()
    pad = K // 2
    OH, OW = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    slot_w = torch.zeros(ops.amax_elems(), dtype=torch.int32, device="cuda")
    exact = fmt == "exact"              # fp_conv_igemm_bf3 (round 5): the same kernel template with exactly split bf16x3 operands, no amax slots

    def packw(dgrad):
        if exact:
            return ops.pack_conv_weight_bf3(w, torch.empty(ops.packed_weight_elems_bf3(Cout, Cin, K, dgrad), device="cuda"), dgrad)
        return ops.pack_conv_weight_hp(w, torch.empty(ops.packed_weight_elems_hp(Cout, Cin, K, dgrad), device="cuda"), slot_w, dgrad)
    if mode == "fwd":
        wp = packw(False)
        src = x.permute(0, 2, 3, 1).contiguous()
        d = ops.make_desc(N, OH, OW, H, W, Cin, 0, Cout, K, stride, pad, L.GATHER_FWD_ZERO)
        y = torch.empty(N, OH, OW, Cout, device="cuda")
        ref = F.conv2d(x.double(), w.double(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    else:
        wp = packw(True)
        dz = ((torch.rand(N, Cout, OH, OW, generator=g) * 2 - 1) * 1e-6).cuda()              # gradients are small numbers: the scale has to find them
        src = dz.permute(0, 2, 3, 1).contiguous()
        d = ops.make_desc(N, H, W, OH, OW, Cout, 0, Cin, K, stride, pad, L.GATHER_DGRAD_ZERO)
        y = torch.empty(N, H, W, Cin, device="cuda")
        xin = torch.zeros(N, Cin, H, W, dtype=torch.float64, device="cuda", requires_grad=True)
        F.conv2d(xin, w.double(), stride=stride, padding=pad).backward(dz.double())
        ref = xin.grad.permute(0, 2, 3, 1)
    assert ops.conv_igemm_hp_supported(d)
    slot_s = ops.amax_f32(src, torch.zeros(ops.amax_elems(), dtype=torch.int32, device="cuda"))

    def run(dst):
        if exact:
            ops.conv_igemm_bf3(d, src, wp, dst)
        else:
            ops.conv_igemm_hp(d, src, wp, dst, slot_s, slot_w)
    run(y)
    err = ((y.double() - ref).norm() / ref.norm()).item()
    assert err <= 1e-6, err
    # the accumulating form, on top of an existing tensor
    base = torch.rand(y.shape, generator=g).cuda() * ref.abs().max().float()
    y2 = base.clone()
    d.epi = L.EPI_ACCUM
    run(y2)
    err2 = ((y2.double() - (ref + base.double())).norm() / (ref + base.double()).norm()).item()
    assert err2 <= 1e-6, err2
    # bit-reproducible
    y3 = base.clone()
    run(y3)
    assert torch.equal(y2, y3)
