"""MI355X-native drop-in for ``net.Ours.base18`` (seg18/net/Ours/base18.py:52-108): the TswinPlus model.

``TswinPlus(num_classes)`` keeps the reference signature, attribute names (.swin .resnet .aspp .project1-3
.classifier) and state-dict keys; ``input_resolution`` (feature-map size H/8 x W/8) is an optional extra because the
reference hard-codes 64x80.  forward: (B,4,3,H,W) -> logits (B,num_classes,H,W).

Everything after the ResNet feeder runs on NHWC tokens through libstswin_hip: temporal Swin, ASPP, the three 1x1
projections (+BN+ReLU), bilinear x2, the 400-channel concat (kept as a 448-wide padded token matrix), the 3x3
classifier (implicit GEMM) and the final bilinear-to-NCHW upsample.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import headops as H
from ... import hip
from .ASPP import ASPP
from .resnet import ResNet18_OS8
from .swin_512 import SwinTransformerLayerv5

L48 = H.Layout.dense(48)
LCAT = H.Layout.concat([L48, L48, L48, H.Layout.dense(256)])   # logical 400 channels inside 448 columns


def decode_tokens(resnet, swin, aspp, proj1, proj2, proj3, x):
    """Frames (B,4,3,Hi,Wi) -> 400-channel decode feature as a padded token matrix [B*h*w][448] (base18.py:80-105 up to
    the concat; PixPro_swin_v5.py:302-327 is the same pipeline).  The ResNet runs once per frame so that train-mode
    BatchNorm statistics stay per frame (base18.py:86-89)."""
    b, t = x.shape[:2]
    tem, h, w = resnet.forward_frames(x)                                     # (B, 4, L, 512) tokens
    c = tem.shape[-1]
    res_last = tem[:, -1].reshape(b * h * w, c)
    t1_all, t2_all = swin.forward_tokens(tem)
    h2, w2 = h // 2, w // 2
    t1 = t1_all[:, -1].reshape(b * h * w, c)
    t2 = t2_all[:, -1].reshape(b * h2 * w2, 2 * c)
    a = aspp.forward_tokens(t2, (b, h2, w2))
    # the four parts are written straight into their column slices of the 448-wide padded concat (base18.py:104): no cat kernel
    widths = (L48.width, L48.width, L48.width, a.shape[1])
    buf, (v1, v2, v3, v4) = H.concat_buffer(b * h * w, widths, a.dtype, a.device)
    with H.syncbn_group() as g:                  # three independent projections: one SyncBatchNorm exchange
        g.conv_bn_relu(res_last, proj1[0], proj1[1], (b, h, w), lout=L48, out=v1)
        g.conv_bn_relu(t1, proj2[0], proj2[1], (b, h, w), lout=L48, out=v2)
        g.conv_bn_relu(t2, proj3[0], proj3[1], (b, h2, w2), lout=L48)
    p1, p2, p3 = g.results()
    p3 = H.BilinearTokFn.apply(p3, (b, h2, w2, h, w), v3)
    a = H.BilinearTokFn.apply(a, (b, h2, w2, h, w), v4)
    return H.ConcatColsFn.apply(buf, widths, p1, p2, p3, a), (b, h, w)


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
        self.num_classes = num_classes

    def forward(self, x):
        hi, wi = x.shape[3:]
        H.refuse_replica(self)
        hip.arena_reset(x.device)                           # one zero-fill block per step (forward + backward accumulators)
        with H.deferred_bn_counters():
            cat, (b, h, w) = decode_tokens(self.resnet, self.swin, self.aspp, self.project1, self.project2, self.project3, x)
            y = H.conv_bn_relu(cat, self.classifier[0], self.classifier[1], (b, h, w), lin=LCAT)
            y = H.conv1x1_tokens(y, self.classifier[3], b, h, w)
        return H.LogitsUpFn.apply(y, (b, h, w, hi, wi, self.num_classes))
