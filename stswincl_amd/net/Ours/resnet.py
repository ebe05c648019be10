"""Drop-in for ``net.Ours.resnet`` (seg18/net/Ours/resnet.py): ResNet18 with output stride 8.

``self.resnet`` reproduces torchvision's ``resnet18`` children[:-4] (conv1, bn1, relu, maxpool, layer1, layer2;
resnet.py:98-102) without depending on torchvision, so state-dict keys are ``resnet.{0,1,4.b,5.b}.*``;
``layer4`` / ``layer5`` are the reference's own dilated BasicBlocks (resnet.py:22-53, :117-119).
Random init (no ImageNet file is read: BASELINE.json asks for random-init weights).

Round-1 status: convolutions / BatchNorm of this feeder run on the ROCm library path (MIOpen) in
channels-last bf16; SURVEY.md section 8(f) row f1 schedules their HIP implicit-GEMM replacement.
"""
from __future__ import annotations

import torch.nn as nn
import torch.nn.functional as F


class _TVBasicBlock(nn.Module):
    """torchvision-style BasicBlock (child names conv1, bn1, relu, conv2, bn2, downsample)."""

    def __init__(self, cin, cout, stride=1):
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
        idn = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idn)


def make_layer(block, in_channels, channels, num_blocks, stride=1, dilation=1):
    blocks = []
    for s in [stride] + [1] * (num_blocks - 1):
        blocks.append(block(in_channels=in_channels, channels=channels, stride=s, dilation=dilation))
        in_channels = block.expansion * channels
    return nn.Sequential(*blocks)


class BasicBlock(nn.Module):
    """resnet.py:22-53."""
    expansion = 1

    def __init__(self, in_channels, channels, stride=1, dilation=1):
        super().__init__()
        out_channels = self.expansion * channels
        self.conv1 = nn.Conv2d(in_channels, channels, 3, stride, dilation, dilation, bias=False)
        self.bn1 = nn.BatchNorm2d(channels)
        self.conv2 = nn.Conv2d(channels, channels, 3, 1, dilation, dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(channels)
        if stride != 1 or in_channels != out_channels:
            self.downsample = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1, stride, bias=False),
                                            nn.BatchNorm2d(out_channels))
        else:
            self.downsample = nn.Sequential()

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return F.relu(out + self.downsample(x))


class ResNet_BasicBlock_OS8(nn.Module):
    """(B,3,H,W) -> (B,512,H/8,W/8); resnet.py:93-133."""

    def __init__(self, num_layers=18):
        super().__init__()
        if num_layers != 18:
            raise Exception("num_layers must be 18 (the only depth the hot path uses)")
        self.resnet = nn.Sequential(
            nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1),
            nn.Sequential(_TVBasicBlock(64, 64), _TVBasicBlock(64, 64)),
            nn.Sequential(_TVBasicBlock(64, 128, 2), _TVBasicBlock(128, 128)))
        self.layer4 = make_layer(BasicBlock, in_channels=128, channels=256, num_blocks=2, stride=1, dilation=2)
        self.layer5 = make_layer(BasicBlock, in_channels=256, channels=512, num_blocks=2, stride=1, dilation=4)

    def forward(self, x):
        c3 = self.resnet(x)
        return self.layer5(self.layer4(c3))


def ResNet18_OS8():
    return ResNet_BasicBlock_OS8(num_layers=18)
