"""a16-a20: pixel-contrastive similarity kernel, regression_loss, PixPro / ConsistencyLoss vs the reference goldens."""
import types

import pytest
import torch

import golden_util as gu
from oracle import stswin_oracle as O
from stswincl_amd.contrast.models import PixPro_swin_v5 as P

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("mode,tol", [("fp32", 1e-4), ("bf16", 2e-2)])
def test_regression_loss_matches_reference(mode, tol):
    g = gu.load("regression_loss.npz")
    n, c, h, w = [int(v) for v in g["shape"]]
    feats = [torch.nn.functional.normalize(gu.det_tensor(f"regression/f{i}", (n, c, h, w)), dim=1).cuda() for i in range(6)]
    labs = [torch.from_numpy(g[f"l{i}"]).cuda() for i in range(6)]
    q = feats[0].clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(mode == "bf16")):
        loss = P.regression_loss(q, *feats[1:], *labs, 12)
    assert abs(float(loss) - float(g["loss"])) < tol * abs(float(g["loss"])) + 1e-7
    loss.backward()
    assert rel(q.grad, g["dq"]) < 50 * tol


@pytest.mark.parametrize("n,c,h,w", [(2, 256, 16, 16), (1, 256, 7, 9), (3, 64, 12, 20)])
def test_regression_loss_vs_oracle_shapes(n, c, h, w):
    torch.manual_seed(h * w)
    feats = [torch.nn.functional.normalize(torch.randn(n, c, h, w), dim=1) for _ in range(6)]
    labs = [torch.randint(0, 12, (n, 1, h, w)).float() for _ in range(6)]
    q = feats[0].clone().requires_grad_(True)
    lo = O.regression_loss(q, *feats[1:], *labs, 12)
    lo.backward()
    qg = feats[0].clone().cuda().requires_grad_(True)
    lg = P.regression_loss(qg, *[f.cuda() for f in feats[1:]], *[l.cuda() for l in labs], 12)
    lg.backward()
    assert abs(float(lg) - float(lo)) < 1e-5 * abs(float(lo))
    assert rel(qg.grad, q.grad) < 1e-3


def _args():
    return types.SimpleNamespace(pixpro_p=1.0, pixpro_momentum=0.99, pixpro_clamp_value=0.0, pixpro_transform_layer=1,
                                 pixpro_ins_loss_weight=0.0, pixpro_pos_ratio=0.7, data="endo18", tag="1",
                                 pretrainpth="none", num_instances=2235, batch_size=2, epochs=150, start_epoch=1)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("views", ["batched", "sequential"])
def test_consistency_loss_step_matches_reference(views, mode, monkeypatch):
    # batched: the 2 query / 6 key passes as one batch each with per-view BatchNorm statistics; sequential: one pass per view
    monkeypatch.setenv("STSWIN_SEQUENTIAL_VIEWS", "1" if views == "sequential" else "0")
    g = gu.load("consistency.npz")
    hh, ww = [int(v) for v in g["hw"]]
    net = P.ConsistencyLoss(_args(), input_resolution=(hh // 8, ww // 8))
    assert net.pixpro.K == int(g["big_k"]) == 167625
    sd = gu.det_fill(gu.skeleton_sd(g["keys"], g["shapes"], g["dtypes"]))
    r = net.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys and all(k.endswith(("attn_mask", "relative_position_index")) for k in r.missing_keys)
    assert [k for k, _ in net.pixpro.named_parameters()] == [str(k) for k in g["param_keys"]]
    net = net.cuda().train()
    ims = [gu.det_tensor(f"consistency/im{i}", (2, 4, 3, hh, ww)).cuda() for i in range(6)]
    masks = [torch.floor(gu.det_tensor(f"consistency/mask{i}", (2, 1, hh // 8, ww // 8), "uniform", 12.0))
             .clamp(0, 11).repeat_interleave(8, 2).repeat_interleave(8, 3).cuda() for i in range(6)]
    # bf16 (what bench.py --workload contrast runs): the loss is a mean over 2 x 2 x 256 pixel terms of O(1) log-ratios of
    # normalised-embedding similarities, so the bf16 storage error of the embeddings (2^-9 relative per element, 256-d dot
    # products) averages down: 1e-2 on the loss; the projector gradient sees the whole bf16 encoder: 0.15 rel-L2.
    tol_loss, tol_grad = (1e-3, 1e-2) if mode == "fp32" else (1e-2, 0.15)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(mode == "bf16")):
        loss = net(*ims, *masks)
    assert net.pixpro.k == int(g["k1"])
    print(f"consistency {views} {mode}: loss {float(loss):.6f} vs {float(g['loss']):.6f}")
    assert abs(float(loss) - float(g["loss"])) < tol_loss * abs(float(g["loss"]))
    loss.backward()
    r_g = rel(net.pixpro.projector.linear2.weight.grad, g["d_projector_linear2"])
    print(f"  d projector.linear2 rel-L2 {r_g:.4f}")
    assert r_g < tol_grad
    sd_after = net.state_dict()
    for key in [f for f in g.files if f.startswith("probe/")]:      # EMA'd key weights / running statistics (fp32 master copies)
        t = sd_after[key[len("probe/"):]].double()
        assert abs(float(t.abs().sum()) - float(g[key][1])) < (1e-3 if mode == "fp32" or "running" not in key else 2e-2) * float(g[key][1]) + 1e-9, key
    assert all(p.grad is None for p in net.pixpro.encoder_k_2.parameters())


def test_reference_pretrain_loop_idiom_ddp_lars_gradscaler():
    """build_model + train of pixcontrast_18/main_pretrain_swinv5.py:28-54,156-171 verbatim on one rank: LARS(torch.optim.SGD(
    add_weight_decay(model.pixpro, wd))) from contrast/lars.py, torch's own DistributedDataParallel(find_unused_parameters=True,
    broadcast_buffers=False) over RCCL, `with amp.autocast():` + GradScaler.  Two steps: finite losses, every step taken, the
    query encoder moves, the momentum key encoder follows, the step counter of PixPro advances."""
    import os
    import torch.distributed as dist
    from torch.cuda import amp
    from torch.nn.parallel import DistributedDataParallel
    from stswincl_amd.contrast.lars import LARS, add_weight_decay
    created = False
    if not dist.is_initialized():
        import socket
        with socket.socket() as sock:                       # a free port: nothing else may own 29xxx on the box
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        created = True
    try:
        g = gu.load("consistency.npz")
        hh, ww = [int(v) for v in g["hw"]]
        torch.manual_seed(0)
        model = P.ConsistencyLoss(_args(), input_resolution=(hh // 8, ww // 8)).cuda()
        params = add_weight_decay(model.pixpro, 1e-5)
        optimizer = LARS(torch.optim.SGD(params, lr=2 * 1 / 256 * 1.0, momentum=0.9))
        model = DistributedDataParallel(model, device_ids=[0], broadcast_buffers=False, find_unused_parameters=True)
        scaler = amp.GradScaler()
        model.train()
        q0 = model.module.pixpro.encoder_1.resnet[0].weight.detach().clone()
        k0 = model.module.pixpro.encoder_k_1.resnet[0].weight.detach().clone()
        kk = model.module.pixpro.k
        ims = [gu.det_tensor(f"consistency/im{i}", (2, 4, 3, hh, ww)).cuda() for i in range(6)]
        masks = [torch.floor(gu.det_tensor(f"consistency/mask{i}", (2, 1, hh // 8, ww // 8), "uniform", 12.0))
                 .clamp(0, 11).repeat_interleave(8, 2).repeat_interleave(8, 3).cuda() for i in range(6)]
        losses = []
        for _ in range(2):
            optimizer.zero_grad()
            with amp.autocast():
                loss = model(*ims, *masks)
            scaler.scale(loss).backward()
            scaler.step(optimizer)
            scaler.update()
            losses.append(float(loss.detach()))
        assert all(l == l and abs(l) < 1e4 for l in losses), losses
        assert scaler.get_scale() >= 65536.0
        assert model.module.pixpro.k == kk + 2
        assert not torch.equal(model.module.pixpro.encoder_1.resnet[0].weight, q0)
        assert not torch.equal(model.module.pixpro.encoder_k_1.resnet[0].weight, k0)
    finally:
        if created:
            dist.destroy_process_group()


def test_consistency_loss_non_square_256x448():
    """The resolution the reference's pre-training actually ran at in SURVEY section 8c's import check (B = 2, 256 x 448: Swin
    resolution (32, 56), stage 2 (16, 28)): forward + backward in bf16, finite loss and gradients, step counter."""
    torch.manual_seed(0)
    net = P.ConsistencyLoss(_args(), input_resolution=(32, 56)).cuda().train()
    ims = [torch.randn(2, 4, 3, 256, 448, device="cuda") for _ in range(6)]
    masks = [torch.randint(0, 12, (2, 1, 32, 56), device="cuda").float().repeat_interleave(8, 2).repeat_interleave(8, 3) for _ in range(6)]
    k0 = net.pixpro.k
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = net(*ims, *masks)
    assert torch.isfinite(loss) and net.pixpro.k == k0 + 1
    loss.backward()
    gs = [p.grad for p in net.pixpro.parameters() if p.requires_grad and p.grad is not None]
    assert len(gs) > 100 and all(torch.isfinite(g).all() for g in gs)
