"""MI355X-native drop-in for ``net.Ours.ASPP`` (seg18/net/Ours/ASPP.py:7-52; ``ASPPv5`` of the contrastive package is
the same module).  Same parameter names / shapes; forward runs on NHWC tokens with libstswin_hip kernels:
every convolution (1x1 and the three dilated 3x3) is the segmented gather GEMM, BatchNorm+ReLU, the image-pool
average / broadcast are HBM-bound HIP kernels.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import headops as H


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
        self.num_classes = num_classes

    def forward_tokens(self, x, geom):
        """x [F*h*w][1024] -> [F*h*w][pad64(num_classes)]  (ASPP.py:33-52)."""
        f, h, w = geom
        if self.training and f == 1:   # the reference's BatchNorm2d refuses a (1, C, 1, 1) map in train mode (ASPP.py:44-45)
            raise ValueError(f"Expected more than 1 value per channel when training, got input size torch.Size([1, 512, 1, 1])")
        # the five branches write straight into their column slices of the 2560-wide concat (ASPP.py:48): no cat kernel, and the
        # backward hands every branch a strided view of the concat's gradient instead of a split copy
        nb = self.conv_1x1_1.out_channels
        buf, (v1, v2, v3, v4, v5) = H.concat_buffer(x.shape[0], [nb] * 5, H.compute_dtype(x), x.device)
        # (the five branches are independent: under SyncBatchNorm their statistics travel in ONE all-gather - H.syncbn_group)
        with H.syncbn_group() as g:
            g.conv_bn_relu(x, self.conv_1x1_1, self.bn_conv_1x1_1, geom, out=v1)
            g.conv_bn_relu(x, self.conv_3x3_1, self.bn_conv_3x3_1, geom, out=v2)
            g.conv_bn_relu(x, self.conv_3x3_2, self.bn_conv_3x3_2, geom, out=v3)
            g.conv_bn_relu(x, self.conv_3x3_3, self.bn_conv_3x3_3, geom, out=v4)
            g.conv_bn_relu(H.AvgPoolTokFn.apply(x, f), self.conv_1x1_2, self.bn_conv_1x1_2, (f, 1, 1))
        o1, o2, o3, o4, img = g.results()
        img = H.BroadcastTokFn.apply(img, h * w, v5)
        cat = H.ConcatColsFn.apply(buf, (nb,) * 5, o1, o2, o3, o4, img)
        out = H.conv_bn_relu(cat, self.conv_1x1_3, self.bn_conv_1x1_3, geom)
        return H.conv1x1_tokens(out, self.conv_1x1_4, f, h, w)

    def forward(self, feature_map):
        f, c, h, w = feature_map.shape
        out = self.forward_tokens(H.to_tokens(feature_map), (f, h, w))
        return H.from_tokens(out, f, h, w)[:, :self.num_classes]


ASPPv5 = ASPP
