"""Drop-in for ``net.Ours.ASPP`` (seg18/net/Ours/ASPP.py:7-52; ``ASPPv5`` of the contrastive package is identical).

Round-1 status: library convolutions (MIOpen, channels-last bf16); the segmented gather GEMM of
libstswin_hip already implements dilated 3x3 as implicit GEMM (tests/test_hip_head.py) and replaces these next.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class ASPP(nn.Module):
    def __init__(self, num_classes):
        super().__init__()
        nc = 1024
        self.conv_1x1_1 = nn.Conv2d(nc, nc // 2, kernel_size=1)
        self.bn_conv_1x1_1 = nn.BatchNorm2d(nc // 2)
        self.conv_3x3_1 = nn.Conv2d(nc, nc // 2, kernel_size=3, stride=1, padding=6, dilation=6)
        self.bn_conv_3x3_1 = nn.BatchNorm2d(nc // 2)
        self.conv_3x3_2 = nn.Conv2d(nc, nc // 2, kernel_size=3, stride=1, padding=12, dilation=12)
        self.bn_conv_3x3_2 = nn.BatchNorm2d(nc // 2)
        self.conv_3x3_3 = nn.Conv2d(nc, nc // 2, kernel_size=3, stride=1, padding=18, dilation=18)
        self.bn_conv_3x3_3 = nn.BatchNorm2d(nc // 2)
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.conv_1x1_2 = nn.Conv2d(nc, nc // 2, kernel_size=1)
        self.bn_conv_1x1_2 = nn.BatchNorm2d(nc // 2)
        self.conv_1x1_3 = nn.Conv2d(nc // 2 * 5, nc // 2, kernel_size=1)
        self.bn_conv_1x1_3 = nn.BatchNorm2d(nc // 2)
        self.conv_1x1_4 = nn.Conv2d(nc // 2, num_classes, kernel_size=1)

    def forward(self, feature_map):
        h, w = feature_map.shape[2:]
        o1 = F.relu(self.bn_conv_1x1_1(self.conv_1x1_1(feature_map)))
        o2 = F.relu(self.bn_conv_3x3_1(self.conv_3x3_1(feature_map)))
        o3 = F.relu(self.bn_conv_3x3_2(self.conv_3x3_2(feature_map)))
        o4 = F.relu(self.bn_conv_3x3_3(self.conv_3x3_3(feature_map)))
        img = self.avg_pool(feature_map)
        img = F.relu(self.bn_conv_1x1_2(self.conv_1x1_2(img)))
        img = F.interpolate(img, size=(h, w), mode="bilinear", align_corners=False)
        out = torch.cat([o1, o2, o3, o4, img], 1)
        out = F.relu(self.bn_conv_1x1_3(self.conv_1x1_3(out)))
        return self.conv_1x1_4(out)


ASPPv5 = ASPP
