"""MI355X-native drop-in for ``net.Ours.resnet`` (seg18/net/Ours/resnet.py): ResNet18 with output stride 8.

``self.resnet`` reproduces torchvision's ``resnet18`` children[:-4] (conv1, bn1, relu, maxpool, layer1, layer2;
resnet.py:98-102) without depending on torchvision, so the state-dict keys are ``resnet.{0,1,4.b,5.b}.*``;
``layer4`` / ``layer5`` are the reference's own dilated BasicBlocks (resnet.py:22-53, :117-119).  Random init (no
ImageNet file is read: BASELINE.json asks for random-init weights).

Everything runs on NHWC tokens through libstswin_hip: the stem is im2col + GEMM, every 3x3 (stride 1/2, dilation
1/2/4) and 1x1 downsample convolution is the segmented gather GEMM, BatchNorm(+residual)+ReLU and the max-pool are
HBM-bound kernels.  ``forward_frames`` runs the T frame passes of base18.py:86-89 as ONE batch with T BatchNorm
statistic groups, which is numerically the per-frame computation (same batch statistics, same sequential
running-stat updates) without 4x the launches.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import headops as H
from ...ops import compute_dtype


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

    def forward_tokens(self, x, frames, h, w, groups, il=False):
        return _block_tokens(self, x, frames, h, w, groups, il)


def make_layer(block, in_channels, channels, num_blocks, stride=1, dilation=1):
    blocks = []
    for s in [stride] + [1] * (num_blocks - 1):
        blocks.append(block(in_channels=in_channels, channels=channels, stride=s, dilation=dilation))
        in_channels = block.expansion * channels
    return nn.Sequential(*blocks)


import os
_FRAME_MAJOR = os.environ.get("STSWIN_RESNET_FRAME_MAJOR") == "1"


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

    def forward_tokens(self, x, frames, h, w, groups, il=False):
        return _block_tokens(self, x, frames, h, w, groups, il)


def _block_tokens(blk, x, frames, h, w, groups, il=False):
    """out = relu(bn2(conv2(relu(bn1(conv1(x))))) + downsample(x))   (resnet.py:42-51)."""
    # (train mode: the BatchNorm statistics of each convolution output come from that convolution's GEMM epilogue)
    tr = blk.bn1.training and H._sync_world(blk.bn1) == 1      # (SyncBatchNorm across ranks gathers pivot-shifted colstats instead)
    ds = blk.downsample
    has_ds = ds is not None and len(ds) > 0
    # x has two consumers (conv1 and the shortcut): their gradients are joined in a GEMM epilogue, not by an autograd add
    link = H.GradLink(2 if has_ds else 1) if (H._RESID_GRAD_LINK and torch.is_grad_enabled() and x.requires_grad) else None
    y, ho, wo, tab = H.conv_tokens(x, blk.conv1, frames, h, w, stats=tr, link=link)
    ilf = frames if il else 0
    if has_ds:
        # conv1 and the shortcut's 1x1 convolution both read x: their two BatchNorms go through ONE exchange under SyncBatchNorm
        # (H.syncbn_group; without a process group both calls run at once, exactly as before)
        idn, _, _, tabd = H.conv_tokens(x, ds[0], frames, h, w, stats=tr, link=link)
        with H.syncbn_group() as g:
            g.bn(y, blk.bn1, relu=True, groups=groups, il_frames=ilf, stats=tab)
            g.bn(idn, ds[1], relu=False, groups=groups, il_frames=ilf, stats=tabd)
        y, idn = g.results()
        y, _, _, tab2 = H.conv_tokens(y, blk.conv2, frames, ho, wo, stats=tr)
        return H.batchnorm_tokens(y, blk.bn2, relu=True, resid=idn, groups=groups, il_frames=ilf, stats=tab2), ho, wo
    y = H.batchnorm_tokens(y, blk.bn1, relu=True, groups=groups, il_frames=ilf, stats=tab)
    y, _, _, tab2 = H.conv_tokens(y, blk.conv2, frames, ho, wo, stats=tr)
    return H.batchnorm_tokens(y, blk.bn2, relu=True, resid=x, groups=groups, il_frames=ilf, stats=tab2, resid_link=link), ho, wo


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

    def forward_tokens(self, img, groups=1, il=False):
        """img (F,3,H,W) with F = groups * (frames per statistic group) -> (tokens [F*h*w][512], h, w).  il = False: the
        frames are group-major (group g = frames [g*F/groups, (g+1)*F/groups)); il = True: group g = frames g, g + groups,
        ... (clip-major clips of `groups` frames: no reordering of the batch)."""
        f, _, hh, ww = img.shape
        dt = compute_dtype(img)
        x, tab = H.stem_conv_tokens(img, self.resnet[0].weight, dt, stats=self.resnet[1].training and H._sync_world(self.resnet[1]) == 1)
        h, w = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
        x = H.batchnorm_relu_maxpool_tokens(x, self.resnet[1], (f, h, w), groups=groups, il_frames=f if il else 0, stats=tab)
        h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        for layer in (self.resnet[4], self.resnet[5], self.layer4, self.layer5):
            for blk in layer:
                x, h, w = blk.forward_tokens(x, f, h, w, groups, il)
        return x, h, w

    def forward_frames(self, x):
        """(B,T,3,H,W) -> tokens (B, T, h*w, 512): the T sequential per-frame calls of base18.py:86-89 in one batch."""
        b, t = x.shape[:2]
        # clip-major as stored; statistic group = frame index t, i.e. the frames t, t + T, t + 2T, ... of the flattened batch
        # (interleaved groups of the BatchNorm kernels): no frame-major copy of the input, none of the tokens, none of their
        # gradients
        # (SyncBatchNorm across ranks gathers the per-group statistics of interleaved groups just the same)
        if _FRAME_MAJOR:                                                            # (A/B switch)
            tok, h, w = self.forward_tokens(x.transpose(0, 1).reshape(t * b, *x.shape[2:]), groups=t)
            return tok.view(t, b, h * w, tok.shape[1]).transpose(0, 1).contiguous(), h, w
        tok, h, w = self.forward_tokens(x.reshape(b * t, *x.shape[2:]), groups=t, il=True)
        return tok.view(b, t, h * w, tok.shape[1]), h, w

    def forward(self, x):
        tok, h, w = self.forward_tokens(x, groups=1)
        return H.from_tokens(tok, x.shape[0], h, w)


def ResNet18_OS8():
    return ResNet_BasicBlock_OS8(num_layers=18)
