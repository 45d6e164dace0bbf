"""Segmentor -- drop-in nn.Module surface of the reference's ground-segmentation network
(footprints/preprocessing/segmentation/network.py:13-207), executed by the same HIP engine as
FootprintNetwork: the ResNet-34 encoder and the four ConvUpsampleAndConcatBlocks are the SAME kernels and schedules; new are
the pyramid-pooling module in front of block1 (csrc/psp.hip) and the 1-channel heads at their own resolution, which run through
the Cin -> 2 head kernels with a zero second filter.

The module tree only owns parameters under the reference's state_dict keys (`encoder.layerK...`, `decoder.blockK...`,
`decoder.outconvK.conv1`, `decoder.outconv4.{0,1}...`, `decoder.PSP.blockK.reduce.weight`); `forward` returns the reference's
list of four logit maps [B,1,H/8,W/8], [B,1,H/4,W/4], [B,1,H/2,W/2], [B,1,H,W] (network.py:84-99), autograd-capable.  The
bilinear up-sizing of the predictions and the masked BCE of the segmentation trainer (segmentation/train.py:184-193,
evaluation.py:39-58) stay plain torch ops on those tensors: they are a few elementwise kernels on 1-channel maps, not a hot spot.
No CPU compute path: a non-CUDA input raises.
"""
import torch
import torch.nn as nn

from ...network import ConvBlock, ConvUpsampleAndConcatBlock, ResnetEncoder, _NoForward, is_dead_param


class OutConvBlock(_NoForward):
    """segmentation/network.py:161-171: reflection-padded 3x3 conv to `out_ch` logits, no activation, no up-sampling"""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        assert out_ch == 1
        self.conv1 = nn.Conv2d(in_ch, out_ch, 3)


class PSPBlock(_NoForward):
    """segmentation/network.py:174-190"""

    def __init__(self, pool_size, feats, reduce_factor=4):
        super().__init__()
        self.pool_size = pool_size
        self.reduce = nn.Conv2d(feats, feats // reduce_factor, kernel_size=1, bias=False)


class PSP(_NoForward):
    """segmentation/network.py:193-207"""

    def __init__(self):
        super().__init__()
        self.block1 = PSPBlock(1, 512)
        self.block2 = PSPBlock(2, 512)
        self.block3 = PSPBlock(4, 512)
        self.block4 = PSPBlock(6, 512)


class SkipDecoder(_NoForward):
    """segmentation/network.py:54-99"""

    def __init__(self, use_PSP=False):
        super().__init__()
        self.use_PSP = use_PSP
        inp_channels = 1024 if use_PSP else 512
        if use_PSP:
            self.PSP = PSP()
        self.block1 = ConvUpsampleAndConcatBlock(inp_channels, 256)
        self.block2 = ConvUpsampleAndConcatBlock(256, 128)
        self.block3 = ConvUpsampleAndConcatBlock(128, 64)
        self.block4 = ConvUpsampleAndConcatBlock(64, 64)
        self.outconv1 = OutConvBlock(128, 1)
        self.outconv2 = OutConvBlock(64, 1)
        self.outconv3 = OutConvBlock(64, 1)
        self.outconv4 = nn.Sequential(ConvBlock(64, 32), OutConvBlock(32, 1))


class Segmentor(nn.Module):
    """forward(x: float32 [B,3,H,W] in [0,1], H, W % 32 == 0) -> [logits 1/8, 1/4, 1/2, 1/1], each [B,1,h,w] (network.py:20-25)"""

    def __init__(self, pretrained=True, use_PSP=False):
        super().__init__()
        self.encoder = ResnetEncoder(pretrained=pretrained)
        self.decoder = SkipDecoder(use_PSP=use_PSP)
        self._engine = None

    def engine(self):
        from ...engine import Engine
        dev = next(self.parameters()).device
        if self._engine is None or self._engine.device != dev:
            self._engine = Engine(self)
        return self._engine

    def live_named_parameters(self):
        return [(n, p) for n, p in self.named_parameters() if not is_dead_param(n)]

    def forward(self, input_image):
        if not input_image.is_cuda:
            raise RuntimeError("footprints_amd Segmentor has no CPU compute path: move the model and the input to a MI355X (`.cuda()`)")
        eng = self.engine()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if need_grad:
            from ...engine import NetFunction
            outs = NetFunction.apply(eng, input_image, *eng.live_params)
        else:
            outs = eng.forward(input_image, training=self.training, save_for_backward=False)
        return [o[:, 0:1] for o in outs]          # channel 1 of every buffer is the padding filter's zero output
