"""f2: multi-tensor Adam / SGD-momentum / EMA kernel vs torch.optim on the same parameters."""
import pytest
import torch

from stswincl_amd.optim import FusedAdam, FusedSGD, ema_update

pytestmark = pytest.mark.gpu


def _params(seed):
    torch.manual_seed(seed)
    shapes = [(513, 67), (1024,), (3,), (64, 3, 7, 7), (1,), (2048, 512)] + [(17, 5)] * 60     # > 48 tensors, odd sizes
    return [torch.randn(s, device="cuda") for s in shapes]


@pytest.mark.parametrize("wd", [0.0, 1e-2])
def test_fused_adam_matches_torch(wd):
    a = [p.clone().requires_grad_(True) for p in _params(0)]
    b = [p.clone().requires_grad_(True) for p in _params(0)]
    oa = torch.optim.Adam(a, 1e-3, weight_decay=wd)
    ob = FusedAdam(b, 1e-3, weight_decay=wd)
    for step in range(4):
        for i, (x, y) in enumerate(zip(a, b)):
            g = torch.randn_like(x) * (1 + i % 3)
            x.grad, y.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    for x, y in zip(a, b):
        assert torch.allclose(x, y, atol=1e-6, rtol=1e-5)


def test_fused_sgd_matches_torch():
    a = [p.clone().requires_grad_(True) for p in _params(1)]
    b = [p.clone().requires_grad_(True) for p in _params(1)]
    oa = torch.optim.SGD([{"params": a[:3], "lr": 0.1}, {"params": a[3:]}], lr=0.01, momentum=0.9, weight_decay=1e-4)
    ob = FusedSGD([{"params": b[:3], "lr": 0.1}, {"params": b[3:]}], lr=0.01, momentum=0.9, weight_decay=1e-4)
    for step in range(3):
        for x, y in zip(a, b):
            g = torch.randn_like(x)
            x.grad, y.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    for x, y in zip(a, b):
        assert torch.allclose(x, y, atol=1e-6, rtol=1e-5)


def test_ema_update():
    k, q = _params(2), _params(3)
    ref = [kk * 0.99 + qq * (1 - 0.99) for kk, qq in zip(k, q)]
    ema_update(k, q, 0.99)
    for x, y in zip(k, ref):
        assert torch.allclose(x, y, atol=1e-7, rtol=1e-6)
