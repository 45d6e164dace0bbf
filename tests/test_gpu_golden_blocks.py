"""GPU: the reference-authored G1 block vectors (tests/golden/g1_blocks.npz -- ConvBlock, ConvUpsampleAndConcatBlock, OutConvBlock x 4 scales x
sigmoid on / off, written by tests/golden/make_golden.py from the reference's own modules, network.py:104-183) fed through the HIP kernels
over the C ABI: outputs, dX, dW, db at the fixture's tolerance (1e-4 of the tensor's max).  Until round 4 these vectors only pinned the
oracle; their channel counts (16 -> 8) are exactly what the split-operand tile kernels reject, so this is the path through the fp32-MFMA
flattened kernels, the fused up2 + concat gather and the head kernels that the real network's shapes exercise least."""
import pytest
import torch

from tests.golden.digest import compare, fill, fill_value, load

pytestmark = pytest.mark.gpu


def _ops():
    from footprints_amd import _lib, ops
    return ops, _lib


def nhwc(t):
    return t.detach().permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def _pack(w, dgrad=False):
    ops, _ = _ops()
    Cout, Cin, K, _k = w.shape
    wp = torch.empty(ops.packed_weight_elems(Cout, Cin, K, dgrad, False), device="cuda")
    wd = w.detach().contiguous().cuda()
    return ops.pack_conv_weight_dgrad(wd, wp) if dgrad else ops.pack_conv_weight(wd, wp, False)


def _convblock_weights(tag, prefix, cin, cout):
    P = {}
    for key, shape in ((".conv1.weight", (cout, cin, 3, 3)), (".conv1.bias", (cout,)), (".conv2.weight", (cout, cout, 3, 3)), (".conv2.bias", (cout,))):
        P[key] = fill_value(tag, (prefix + key).lstrip("."), shape)
    return P


def _elu_grad(g, y):
    """d / d pre of ELU from its OUTPUT (network.py:129,135 run ELU in place): 1 where y > 0, y + 1 elsewhere"""
    return g * torch.where(y > 0, torch.ones_like(y), y + 1.0)


class _ConvBlock:
    """ConvBlock.forward / backward (network.py:124-138, use_bn=False) as the engine issues it: reflection padding, bias and ELU inside the
    convolution launches, ELU' of the first layer inside the second layer's data gradient"""

    def __init__(self, P, N, H, W, cin, cout, c1_split=None):
        ops, L = _ops()
        self.P, self.dims, self.cin, self.cout, self.c1_split = P, (N, H, W), cin, cout, c1_split

    def forward(self, x, skip=None):
        ops, L = _ops()
        N, H, W = self.dims
        P = self.P
        if self.c1_split is None:
            d1 = ops.make_desc(N, H, W, H, W, self.cin, 0, self.cout, 3, 1, 1, L.GATHER_FWD_REFLECT, act=L.ACT_ELU)
        else:                                   # x is the LOW-resolution tensor: nearest x2 + concat [up, skip] inside the gather
            c0, c1 = self.c1_split
            d1 = ops.make_desc(N, H, W, H, W, c0, c1, self.cout, 3, 1, 1, L.GATHER_FWD_REFLECT_UP2, act=L.ACT_ELU)
        self.x, self.skip = x, skip
        self.a = torch.empty((N, H, W, self.cout), device="cuda")
        ops.conv_igemm(d1, x, skip, _pack(P[".conv1.weight"]), self.a, bias=P[".conv1.bias"].cuda())
        d2 = ops.make_desc(N, H, W, H, W, self.cout, 0, self.cout, 3, 1, 1, L.GATHER_FWD_REFLECT, act=L.ACT_ELU)
        self.y = torch.empty((N, H, W, self.cout), device="cuda")
        ops.conv_igemm(d2, self.a, None, _pack(P[".conv2.weight"]), self.y, bias=P[".conv2.bias"].cuda())
        return self.y

    def backward(self, gy):
        """gy = d loss / d y (after ELU).  Returns d loss / d (conv1's gathered input) at this block's resolution and the four parameter gradients"""
        ops, L = _ops()
        N, H, W = self.dims
        P, co = self.P, self.cout
        g2 = _elu_grad(gy, self.y)
        d2 = ops.make_desc(N, H, W, H, W, co, 0, co, 3, 1, 1, L.GATHER_FWD_REFLECT)
        dw2, db2 = torch.empty((co, co, 3, 3), device="cuda"), torch.empty((co,), device="cuda")
        ops.conv_wgrad(d2, self.a, None, g2, dw2)
        ops.colsum(g2.view(-1, co), db2)
        g1 = torch.empty((N, H, W, co), device="cuda")
        dd2 = ops.make_desc(N, H, W, H, W, co, 0, co, 3, 1, 1, L.GATHER_DGRAD_REFLECT, epi=L.EPI_ACTGRAD_ELU)
        ops.conv_igemm(dd2, g2, None, _pack(P[".conv2.weight"], dgrad=True), g1, actsrc=self.a)
        cin = self.cin if self.c1_split is None else sum(self.c1_split)
        dw1, db1 = torch.empty((co, cin, 3, 3), device="cuda"), torch.empty((co,), device="cuda")
        if self.c1_split is None:
            d1 = ops.make_desc(N, H, W, H, W, cin, 0, co, 3, 1, 1, L.GATHER_FWD_REFLECT)
        else:
            d1 = ops.make_desc(N, H, W, H, W, self.c1_split[0], self.c1_split[1], co, 3, 1, 1, L.GATHER_FWD_REFLECT_UP2)
        ops.conv_wgrad(d1, self.x, self.skip, g1, dw1)
        ops.colsum(g1.view(-1, co), db1)
        dx = torch.empty((N, H, W, cin), device="cuda")
        dd1 = ops.make_desc(N, H, W, H, W, co, 0, cin, 3, 1, 1, L.GATHER_DGRAD_REFLECT)
        ops.conv_igemm(dd1, g1, None, _pack(P[".conv1.weight"], dgrad=True), dx)
        return dx, dw1, db1, dw2, db2


def test_g1_convblock_through_the_hip_kernels():
    gold = load("g1_blocks")
    P = _convblock_weights("g1.convblock", "", 16, 8)
    x = fill("g1.convblock.x", (2, 16, 6, 10))
    blk = _ConvBlock(P, 2, 6, 10, 16, 8)
    y = blk.forward(nhwc(x))
    compare(gold, "convblock.y", nchw(y))
    dx, dw1, _, _, db2 = blk.backward(nhwc(fill("g1.convblock.g", (2, 8, 6, 10))))
    compare(gold, "convblock.dx", nchw(dx))
    compare(gold, "convblock.dw1", dw1)
    compare(gold, "convblock.db2", db2)


def test_g1_upsample_and_concat_block_through_the_hip_kernels():
    """ConvUpsampleAndConcatBlock (network.py:151-158): pre ConvBlock at 4 x 6, nearest x2 + cat[x, skip] inside the post block's first gather,
    and on the way back the split of the high-resolution data gradient into the skip's gradient and the 2 x 2-pooled low-resolution one with
    the pre block's ELU' (fp_up2cat_bwd)"""
    ops, L = _ops()
    gold = load("g1_blocks")
    pre = _ConvBlock(_convblock_weights("g1.upcat", ".pre_concat_conv", 16, 8), 2, 4, 6, 16, 8)
    post = _ConvBlock(_convblock_weights("g1.upcat", ".post_concat_conv", 16, 8), 2, 8, 12, 16, 8, c1_split=(8, 8))
    x, skip = fill("g1.upcat.x", (2, 16, 4, 6)), fill("g1.upcat.skip", (2, 8, 8, 12))
    low = pre.forward(nhwc(x))
    y = post.forward(low, nhwc(skip))
    compare(gold, "upcat.y", nchw(y))
    dcat, dw1_post, _, _, _ = post.backward(nhwc(fill("g1.upcat.g", (2, 8, 8, 12))))
    compare(gold, "upcat.post.dw1", dw1_post)
    dlow, dskip = torch.empty((2, 4, 6, 8), device="cuda"), torch.empty((2, 8, 12, 8), device="cuda")
    # d loss / d low after the pre block's ELU: pool the upsampled half 2 x 2; the pre block's backward applies ELU' itself
    ops.up2cat_bwd(dcat, 2, 4, 6, 8, 8, dlow, dskip=dskip)
    compare(gold, "upcat.dskip", nchw(dskip))
    dx, _, _, dw2_pre, _ = pre.backward(dlow)
    compare(gold, "upcat.pre.dw2", dw2_pre)
    compare(gold, "upcat.dx", nchw(dx))


@pytest.mark.parametrize("sig", [False, True])
@pytest.mark.parametrize("scale", [1, 2, 4, 8])
def test_g1_outconv_through_the_hip_kernels(scale, sig):
    """OutConvBlock (network.py:174-183): reflect 3x3 conv Cin -> 2 (+ sigmoid BEFORE the bilinear upsample), and its backward"""
    ops, L = _ops()
    gold = load("g1_blocks")
    wt, b = fill_value("g1.outconv", "conv1.weight", (2, 16, 3, 3)).cuda(), fill_value("g1.outconv", "conv1.bias", (2,)).cuda()
    x = nhwc(fill("g1.outconv.x", (2, 16, 6, 10)))
    N, h, w = 2, 6, 10
    H, W = h * scale, w * scale
    tag = "outconv.s%d.%s" % (scale, "sig" if sig else "lin")
    low = torch.empty((N, h, w, 2), device="cuda")
    ops.head_fwd(x, wt, b, low, sig)
    out = torch.zeros((N, 4, H, W), device="cuda")
    c0 = 2 if sig else 0                       # the depth decoder (sigmoid) owns output channels 2, 3; the mask decoder 0, 1 (network.py:28)
    ops.head_upsample(low, out, scale, c0)
    compare(gold, tag + ".y", out[:, c0:c0 + 2])
    gout = torch.zeros((N, 4, H, W), device="cuda")
    gout[:, c0:c0 + 2] = fill("g1.outconv.g%d" % scale, (N, 2, H, W)).cuda()
    dz = torch.empty((N, h, w, 2), device="cuda")
    ops.head_upsample_bwd(gout, low, dz, scale, c0, sig)
    dx = torch.empty((N, h, w, 16), device="cuda")
    ops.head_dgrad(dz, wt, dx)
    compare(gold, tag + ".dx", nchw(dx))
    dw, db = torch.empty((2, 16, 3, 3), device="cuda"), torch.empty((2,), device="cuda")
    ops.head_wgrad(x, dz, dw, db)
    compare(gold, tag + ".dw", dw)
    compare(gold, tag + ".db", db)
