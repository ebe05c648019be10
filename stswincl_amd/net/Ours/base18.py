"""Drop-in for ``net.Ours.base18`` (seg18/net/Ours/base18.py:52-108): the TswinPlus segmentation model.

``TswinPlus(num_classes)`` keeps the reference signature, attribute names (.swin .resnet .aspp .project1-3
.classifier) and state-dict keys; ``input_resolution`` (feature-map size H/8 x W/8) is an optional extra because
the reference hard-codes 64x80.  forward: (B,4,3,H,W) -> logits (B,num_classes,H,W).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .ASPP import ASPP
from .resnet import ResNet18_OS8
from .swin_512 import SwinTransformerLayerv5


class TswinPlus(nn.Module):
    def __init__(self, num_classes, input_resolution=(64, 80)):
        super().__init__()
        self.swin = SwinTransformerLayerv5(dim=512, input_resolution=input_resolution, num_heads=4)
        self.resnet = ResNet18_OS8()
        self.aspp = ASPP(num_classes=256)
        self.project1 = nn.Sequential(nn.Conv2d(512, 48, 1, bias=False), nn.BatchNorm2d(48), nn.ReLU(inplace=True))
        self.project2 = nn.Sequential(nn.Conv2d(512, 48, 1, bias=False), nn.BatchNorm2d(48), nn.ReLU(inplace=True))
        self.project3 = nn.Sequential(nn.Conv2d(1024, 48, 1, bias=False), nn.BatchNorm2d(48), nn.ReLU(inplace=True))
        self.classifier = nn.Sequential(nn.Conv2d(400, 256, 3, padding=1, bias=False), nn.BatchNorm2d(256),
                                        nn.ReLU(inplace=True), nn.Conv2d(256, num_classes, 1))

    def features(self, x):
        """Frames -> the 400-channel decode feature at (H/8, W/8) (base18.py:80-105 up to the concat).
        The ResNet runs once per frame so that train-mode BatchNorm statistics stay per frame (:86-89)."""
        b, t = x.shape[:2]
        x = x.contiguous(memory_format=torch.channels_last_3d) if False else x
        seq = [self.resnet(x[:, i].contiguous(memory_format=torch.channels_last)) for i in range(t)]
        tem = torch.stack(seq, dim=1)
        res_output = seq[-1]
        tem1, tem2 = self.swin(tem)
        t1, t2 = tem1[:, -1], tem2[:, -1]
        aspp_output = self.aspp(t2)
        p1 = self.project1(res_output)
        p2 = self.project2(t1)
        p3 = self.project3(t2)
        p3 = F.interpolate(p3, size=p1.shape[2:], mode="bilinear", align_corners=False)
        aspp_output = F.interpolate(aspp_output, size=p1.shape[2:], mode="bilinear", align_corners=False)
        return torch.cat([p1, p2.to(p1.dtype), p3.to(p1.dtype), aspp_output.to(p1.dtype)], dim=1)

    def forward(self, x):
        h, w = x.shape[3:]
        out = self.classifier(self.features(x))
        return F.interpolate(out, (h, w), mode="bilinear", align_corners=False)
