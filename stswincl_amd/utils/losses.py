"""MI355X-native drop-in for ``utils.losses.OhemCELoss2D`` (seg18/utils/losses.py:16-40).

The reference sorts all B*H*W per-pixel losses and then branches on ``loss[n_min] > thresh`` (a host sync).  Here a HIP
kernel computes the per-pixel CE together with count / sum of the losses above the threshold, the branch is resolved
on the device (no sync), and only the rarely-taken top-n_min branch uses a selection.  Same value; ties at the
selection boundary have equal losses so the mean is identical.
"""
from __future__ import annotations

import math

import torch.nn as nn

from ..headops import OhemCEFn


class OhemCELoss2D(nn.CrossEntropyLoss):
    def __init__(self, n_min, thresh=0.7, ignore_index=-1):
        super().__init__(None, None, ignore_index, reduction="none")
        self.thresh = -math.log(thresh)
        self.n_min = n_min
        self.ignore_index = ignore_index

    def forward(self, pred, target):
        return self.OhemCELoss(pred, target)

    def OhemCELoss(self, logits, labels):
        return OhemCEFn.apply(logits, labels, self.n_min, self.thresh, self.ignore_index)
