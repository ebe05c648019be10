"""f4: fused upsample(align_corners=True) + argmax + Dice/IoU counts vs the reference's formulation (seg18/test.py:153-175)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from stswincl_amd.utils import EndoMetric as E

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_predict_and_score_matches_reference_formulation(dtype):
    torch.manual_seed(0)
    f, nc, h, w, H, W = 2, 12, 64, 80, 128, 160
    logits = (torch.randn(f, nc, h, w) * 3).to(dtype)
    gt = torch.randint(0, nc, (f, H, W))
    gt[0, :, :40] = 0
    ref = torch.argmax(F.softmax(F.interpolate(logits.float(), (H, W), mode="bilinear", align_corners=True), dim=1), dim=1)
    labels, dices, ious = E.predict_and_score(logits.cuda(), (H, W), gt.cuda())
    lab = labels.cpu().long()
    mism = (lab != ref)
    assert float(mism.float().mean()) < 2e-4          # only exact near-ties may differ (interpolation rounding)
    for i in range(f):
        rd = E.general_dice(gt[i].numpy(), lab[i].numpy())
        rj = E.general_jaccard(gt[i].numpy(), lab[i].numpy())
        assert [c for c, _ in rd] == [c for c, _ in dices[i]]
        assert np.allclose([v for _, v in rd], [v for _, v in dices[i]], rtol=1e-12)
        assert np.allclose([v for _, v in rj], [v for _, v in ious[i]], rtol=1e-12)
