"""Drop-in for ``utils.losses.OhemCELoss2D`` (seg18/utils/losses.py:16-40).

The reference sorts all B*H*W per-pixel losses and then branches on ``loss[n_min] > thresh`` (a host sync).
Here the same selected set is found without the full sort: count(loss > thresh) decides the branch; the
top-n_min branch uses torch.topk.  Ties at the boundary have equal values, so the mean is identical.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class OhemCELoss2D(nn.CrossEntropyLoss):
    def __init__(self, n_min, thresh=0.7, ignore_index=-1):
        super().__init__(None, None, ignore_index, reduction="none")
        self.thresh = -math.log(thresh)
        self.n_min = n_min
        self.ignore_index = ignore_index

    def forward(self, pred, target):
        return self.OhemCELoss(pred, target)

    def OhemCELoss(self, logits, labels):
        loss = F.cross_entropy(logits.float(), labels, ignore_index=self.ignore_index, reduction="none").view(-1)
        hard = loss > self.thresh
        n_hard = hard.sum()
        # loss_sorted[n_min] > thresh  <=>  more than n_min elements exceed thresh
        if int(n_hard) > self.n_min:
            return (loss * hard).sum() / n_hard
        return torch.topk(loss, self.n_min, sorted=False)[0].mean()
