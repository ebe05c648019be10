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


def test_fused_adam_keeps_a_step_count_per_parameter():
    """A parameter that gets no gradient on the first step (an unused branch, a parameter unfrozen later) has its own bias
    correction in torch.optim.Adam."""
    a = [p.clone().requires_grad_(True) for p in _params(4)[:6]]
    b = [p.clone().requires_grad_(True) for p in _params(4)[:6]]
    oa, ob = torch.optim.Adam(a, 1e-2), FusedAdam(b, 1e-2)
    for step in range(3):
        for i, (x, y) in enumerate(zip(a, b)):
            if step == 0 and i % 2:
                x.grad = y.grad = None
                continue
            g = torch.randn_like(x)
            x.grad, y.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    for x, y in zip(a, b):
        assert torch.allclose(x, y, atol=1e-6, rtol=1e-5)


def test_fused_sgd_accepts_a_loaded_state_without_buffers():
    b = [p.clone().requires_grad_(True) for p in _params(5)[:3]]
    ob = FusedSGD(b, 0.1, momentum=0.9)
    for p in b:
        ob.state[p]["momentum_buffer"] = None        # what torch.optim.SGD.state_dict() holds before the first step
        p.grad = torch.ones_like(p)
    before = [p.detach().clone() for p in b]
    ob.step()
    for p, q in zip(b, before):
        assert torch.allclose(p, q - 0.1, atol=1e-6)


def test_lars_matches_reference_golden():
    """contrast/lars.py (add_weight_decay + LARS around SGD momentum, main_pretrain_swinv5.py:37-47) against three steps of the
    reference itself (tests/golden/lars.npz): the zero-norm branch, a zero gradient (norm from the decay term only), 1-D
    parameters without trust ratio, momentum-buffer state of the wrapped optimizer."""
    import golden_util as gu
    from stswincl_amd.contrast.lars import LARS, add_weight_decay
    g = gu.load("lars.npz")
    names = [str(n) for n in g["names"]]

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc = torch.nn.Linear(24, 16)
            self.bn = torch.nn.BatchNorm1d(16)
            self.conv = torch.nn.Conv2d(4, 8, 3)
            self.zero = torch.nn.Linear(16, 8, bias=False)

    for inner in ("fused", "torch"):
        net = Net().cuda()
        params = dict(net.named_parameters())
        assert list(params) == names
        with torch.no_grad():
            for n in names:
                params[n].copy_(torch.from_numpy(g[f"p0/{n}"]))
        groups = add_weight_decay(net, float(g["wd"]))
        base = (FusedSGD if inner == "fused" else torch.optim.SGD)(groups, lr=float(g["lr"]), momentum=float(g["momentum"]))
        opt = LARS(base)
        assert opt.eps == float(g["eps"]) and opt.trust_coef == float(g["trust_coef"])
        for step in range(3):
            for n in names:
                params[n].grad = torch.from_numpy(g[f"g{step}/{n}"]).cuda()
            v0 = params["fc.weight"]._version
            opt.step()
            assert params["fc.weight"]._version > v0              # GEMM weight caches must see the update
            for n in names:
                ref = torch.from_numpy(g[f"p{step + 1}/{n}"])
                assert torch.allclose(params[n].detach().cpu(), ref, rtol=2e-5, atol=1e-7), (inner, step, n)
        for n in names:
            assert torch.allclose(opt.state[params[n]]["momentum_buffer"].cpu(), torch.from_numpy(g[f"buf/{n}"]), rtol=2e-5, atol=1e-7)
        sd = opt.state_dict()
        assert len(sd["param_groups"]) == 2 and sd["param_groups"][1]["ignore"] is False


def test_batched_repack_after_optimizer_step_matches_lazy_packing():
    """ops.repack (two launches for every cached GEMM operand of the stepped parameters) must leave exactly what the lazy
    per-weight packing produces: W / W^T of Linear weights, tap-major forward / dgrad matrices of convolutions (incl. a padded
    channel layout), and stamps that keep the cache from re-packing again."""
    from stswincl_amd import headops as H, hip, ops
    torch.manual_seed(0)
    lin = [torch.nn.Parameter(torch.randn(n, k, device="cuda")) for n, k in ((512, 512), (1536, 512), (2048, 512), (64, 192), (12, 256))]
    convs = [torch.nn.Parameter(torch.randn(co, ci, k, k, device="cuda")) for co, ci, k in ((64, 64, 3), (512, 256, 3), (48, 512, 1), (256, 400, 3))]
    lays = [(H.Layout.dense(64), H.Layout.dense(64)), (H.Layout.dense(256), H.Layout.dense(512)), (H.Layout.dense(512), H.Layout.dense(48)),
            (H.Layout.concat([H.Layout.dense(48)] * 3 + [H.Layout.dense(256)]), H.Layout.dense(256))]
    ops.clear_caches()
    for dt in (torch.bfloat16, torch.float32):
        for p in lin:
            ops.wcast(p, dt), ops.wcast(p, dt, True)
        for p, (li, lo) in zip(convs, lays):
            H._conv_mats(p, dt, li, lo, False)
    opt = FusedSGD(lin + convs, 0.5)
    for p in lin + convs:
        p.grad = torch.randn_like(p)
    opt.step()                                             # -> _mark_updated -> ops.repack
    calls = {"lin": 0, "conv": 0}
    real_lin, real_conv = hip.linear_pack, hip.conv_pack
    hip.linear_pack = lambda *a, **k: (calls.__setitem__("lin", calls["lin"] + 1), real_lin(*a, **k))[1]
    hip.conv_pack = lambda *a, **k: (calls.__setitem__("conv", calls["conv"] + 1), real_conv(*a, **k))[1]
    try:
        for dt in (torch.bfloat16, torch.float32):
            for p in lin:
                w, wt = ops.wcast(p, dt), ops.wcast(p, dt, True)
                assert torch.equal(w, p.detach().to(dt)) and torch.equal(wt, p.detach().t().to(dt))
            for p, (li, lo) in zip(convs, lays):
                fwd, dg = H._conv_mats(p, dt, li, lo, False), H._conv_mats(p, dt, li, lo, True)
                rf, rd = real_conv(p, dt, lo.index_map(p.device), li.index_map(p.device))
                assert torch.equal(fwd, rf) and torch.equal(dg, rd)
    finally:
        hip.linear_pack, hip.conv_pack = real_lin, real_conv
    # the (12, 256) weight has n % 4 == 0 too; nothing may have been re-packed lazily after the batched pass
    assert calls == {"lin": 0, "conv": 0}, calls


def test_fused_adam_parameters_that_sit_steps_out_leave_their_shared_device_clock():
    """Parameters that stepped together share a device clock (stswincl_amd.optim._Clock); one that later sits a step out (grad None)
    must continue with ITS count, not the clock's (round 6: found by tests/fuzz/fuzz_ops.py::fuzz_optim_groups)."""
    a = [p.clone().requires_grad_(True) for p in _params(6)[:8]]
    b = [p.clone().requires_grad_(True) for p in _params(6)[:8]]
    oa = torch.optim.Adam([{"params": a[:3], "lr": 1e-2}, {"params": a[3:]}], 3e-3)
    ob = FusedAdam([{"params": b[:3], "lr": 1e-2}, {"params": b[3:]}], 3e-3)
    torch.manual_seed(7)
    for step in range(6):
        for i, (x, y) in enumerate(zip(a, b)):
            if (step + i) % 4 == 1 or (step == 2 and i < 5):       # different parameters sit out different (later) steps
                x.grad = y.grad = None
                continue
            g = torch.randn_like(x)
            x.grad, y.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    for x, y in zip(a, b):
        assert torch.allclose(x, y, atol=1e-6, rtol=1e-5)
    steps_a = [int(oa.state[p]["step"]) for p in a]
    steps_b = [ob.state_dict()["state"][i]["step"] for i in range(len(b))]
    assert steps_a == steps_b, (steps_a, steps_b)
