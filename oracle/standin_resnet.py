"""Module-form ResNet-34 used ONLY to import the reference in this container.

Test infrastructure (see oracle/__init__.py).  The reference does
``from torchvision.models import resnet34`` (network.py:10) and re-wraps the
attributes ``conv1, bn1, relu, maxpool, layer1..layer4`` (network.py:40-44).
torchvision is not installed and not vendored under /root/reference, so this
file provides an object with exactly those attributes, built from stock
``torch.nn`` layers following the published ResNet-34 architecture
(BasicBlock x [3,4,6,3], 7x7/2 stem, 3x3/2 maxpool, 1x1/2 downsample, bias-free
convs, BN eps 1e-5 momentum 0.1).  Consequence: the ENCODER golden
vectors are produced by this stand-in, not by reference-authored arithmetic --
"parity unpinned" by the reference for the encoder; decoder + loss vectors come
from the reference's own code.
"""
import torch.nn as nn


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class ResNet34(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for i, (n, c) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512)), start=1):
            blocks = []
            for b in range(n):
                blocks.append(BasicBlock(cin, c, 2 if (b == 0 and i > 1) else 1))
                cin = c
            setattr(self, "layer%d" % i, nn.Sequential(*blocks))


def resnet34(pretrained=False, **_):
    # no network in this image: ImageNet weights cannot be fetched; weights are
    # always overwritten by load_state_dict in the fixtures anyway.
    return ResNet34()
