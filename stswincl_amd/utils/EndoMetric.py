"""Drop-in for ``utils.EndoMetric`` (seg18/utils/EndoMetric.py) plus the fused device-side evaluation step of
seg18/test.py:153-175: ``predict_and_score`` resizes the logits (bilinear, align_corners=True), takes the arg-max and
accumulates the per-class counts in one HIP kernel; Dice / IoU then follow the reference's formulas per frame over the
classes present in the ground truth (class 0 = background is skipped)."""
from __future__ import annotations

import numpy as np
import torch

from .. import hip


def jaccard(y_true, y_pred):
    inter = (y_true * y_pred).sum()
    union = y_true.sum() + y_pred.sum() - inter
    return (inter + 1e-15) / (union + 1e-15)


def dice(y_true, y_pred):
    return (2 * (y_true * y_pred).sum() + 1e-15) / (y_true.sum() + y_pred.sum() + 1e-15)


def general_dice(y_true, y_pred):
    return [[c, dice(y_true == c, y_pred == c)] for c in set(np.asarray(y_true).flatten()) if c != 0]


def general_jaccard(y_true, y_pred):
    return [[c, jaccard(y_true == c, y_pred == c)] for c in set(np.asarray(y_true).flatten()) if c != 0]


def predict_and_score(logits: torch.Tensor, size, gt: torch.Tensor = None):
    """logits (F,nc,h,w) on the GPU -> labels (F,H,W) uint8 and, with gt (F,H,W) int64, per-frame lists
    [[class, dice], ...], [[class, iou], ...] exactly as general_dice / general_jaccard would give them."""
    H, W = size
    labels, counts = hip.upsample_argmax(logits, H, W, gt)
    if gt is None:
        return labels
    cnt = counts.cpu().numpy().astype(np.float64)          # [F][3][nc]: |gt|, |pred|, |gt & pred|
    dices, ious = [], []
    for f in range(cnt.shape[0]):
        g, p, i = cnt[f]
        present = [c for c in range(1, cnt.shape[2]) if g[c] > 0]
        dices.append([[c, (2 * i[c] + 1e-15) / (g[c] + p[c] + 1e-15)] for c in present])
        ious.append([[c, (i[c] + 1e-15) / (g[c] + p[c] - i[c] + 1e-15)] for c in present])
    return labels, dices, ious
