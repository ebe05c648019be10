"""a16-a20: pixel-contrastive similarity kernel, regression_loss, PixPro / ConsistencyLoss vs the reference goldens."""
import types

import pytest
import torch
import torch.nn.functional as F

import golden_util as gu
from oracle import stswin_oracle as O
from stswincl_amd import hip
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


# ---------------------------------------------------------------------------------------------------------------------------
# Round 5: the glue between the encoders and the similarity kernel as HIP launches (rownorm_scatter, labels_resize, label_counts,
# pair_loss) - each against the torch formulation it replaces, then the whole token path against the torch-glue path.
@pytest.mark.parametrize("dtype,views,b,hw,c", [(torch.float32, 2, 3, 40, 256), (torch.bfloat16, 6, 2, 64, 256), (torch.bfloat16, 1, 5, 17, 128)])
def test_rownorm_scatter_is_normalize_plus_view_deinterleave(dtype, views, b, hw, c):
    torch.manual_seed(views + b)
    R = views * b * hw
    x = (torch.randn(R, c, device="cuda") * 3).to(dtype)
    x[5] = 0                                                 # a zero row: x / max(0, 1e-12) = 0, no NaN
    y = torch.empty(R, c, dtype=dtype, device="cuda")
    inv = hip.rownorm_scatter(x, y, views, hw, b, want_inv=True)
    xr = x.float().requires_grad_(True)
    ref = F.normalize(xr, dim=1)                             # [clip = sample * views + view][px]
    ref_v = ref.view(b, views, hw, c).permute(1, 0, 2, 3).reshape(R, c)
    assert torch.equal(y, ref_v.to(dtype)), float((y.float() - ref_v).abs().max())
    dy = torch.randn(R, c, device="cuda")
    ref_v.backward(dy)
    dx = hip.rownorm_scatter_bwd(x, inv, dy, views, hw, b)
    assert dx.dtype == dtype
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert float((dx.float() - xr.grad).abs().max()) <= tol * float(xr.grad.abs().max())
    assert float(dx[5].abs().max()) < 1e30 and bool(torch.isfinite(dx).all())


def test_labels_resize_and_counts_match_the_torch_formulation():
    torch.manual_seed(3)
    N, S, h = 3, 64, 8
    masks = [torch.randint(0, 12, (N, 1, S, S + 16), device="cuda").float() for _ in range(6)]
    masks[2][0, 0, :8] = 13.0                                # a label beyond class_num - 1: clamped in the counts like one_hot's operand
    lb = hip.labels_resize(masks, h, h + 2)
    ref = torch.stack([F.interpolate(m, size=[h, h + 2], mode="nearest").reshape(-1).to(torch.int32) for m in masks], 0)
    assert torch.equal(lb, ref)
    HW = h * (h + 2)
    gmap = [[1, 2, 3, 4, 5], [0, 2, 3, 4, 5]]
    lq = torch.cat([lb[0], lb[1]], 0).contiguous()
    for q_block, bank_block in ((HW, HW), (N * HW, N * HW)):
        cnt = hip.label_counts(lq, lb, q_sets=2, q_block=q_block, bank_block=bank_block, ncls=12, gmap=gmap)
        ref_cnt = P._label_counts(lq, lb, gmap, 2, q_block, bank_block, 12)
        assert torch.equal(cnt, ref_cnt)


@pytest.mark.parametrize("M,groups,visible", [(2 * 512, 5, 300), (2 * 4096, 5, 1024), (2 * 7, 3, 9)])
def test_pair_loss_kernel_forward_and_backward(M, groups, visible):
    torch.manual_seed(M)
    cnt = torch.randint(0, visible + 1, (M, groups), device="cuda").float()
    cnt[0] = 0.0                                             # empty positive sets
    cnt[1] = float(visible)                                  # empty negative sets
    pos = (torch.randn(M, groups, device="cuda") * cnt * 0.3).requires_grad_(True)
    neg = (torch.randn(M, groups, device="cuda") * (visible - cnt) * 0.3).requires_grad_(True)
    Pp = pos.sum(-1) / (cnt.sum(-1) + 1e-6)
    Nn = (neg / ((visible - cnt) + 1e-6)).sum(-1)
    pe, ne = torch.exp(Pp), torch.exp(Nn)
    ref = (-torch.log(pe / (pe + ne) + 1e-6)).view(2, -1).mean(1).sum()
    ref.backward()
    tot = (pos + neg).detach()
    loss = hip.pair_loss(pos.detach(), tot, cnt, 2, visible)
    assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref)) + 1e-7
    assert torch.equal(loss, hip.pair_loss(pos.detach(), tot, cnt, 2, visible))          # fixed-order sums
    dpos, dneg = hip.pair_loss_bwd(pos.detach(), tot, cnt, torch.ones(1, device="cuda"), 2, visible)
    assert float((dpos - pos.grad).abs().max()) <= 1e-5 * float(pos.grad.abs().max()) + 1e-12
    assert float((dneg - neg.grad).abs().max()) <= 1e-5 * float(neg.grad.abs().max()) + 1e-12


@pytest.mark.parametrize("bank", ["sample", "batch"])
def test_token_path_of_the_consistency_loss_equals_the_torch_glue_path(bank, monkeypatch):
    """ConsistencyLoss.forward through the token path (one normalise-and-scatter launch per encoder pass, labels / counts / loss
    kernels: PairLossFn) against the same module with the torch formulation of the glue (STSWIN_CONTRAST_TORCH_GLUE=1): same loss to
    fp32 rounding, same gradients to bf16 rounding of the projector output's gradient, and far fewer torch launches."""
    a = _args()
    a.pixpro_bank = bank
    a.pixpro_bank_stats = True
    g = gu.load("consistency.npz")
    hh, ww = [int(v) for v in g["hw"]]
    ims = [gu.det_tensor(f"consistency/im{i}", (2, 4, 3, hh, ww)).cuda() for i in range(6)]
    masks = [torch.floor(gu.det_tensor(f"consistency/mask{i}", (2, 1, hh // 8, ww // 8), "uniform", 12.0))
             .clamp(0, 11).repeat_interleave(8, 2).repeat_interleave(8, 3).cuda() for i in range(6)]
    res = {}
    for glue in ("1", "0"):
        monkeypatch.setenv("STSWIN_CONTRAST_TORCH_GLUE", glue)
        torch.manual_seed(0)
        net = P.ConsistencyLoss(a, input_resolution=(hh // 8, ww // 8)).cuda().train()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = net(*ims, *masks)
        loss.backward()
        res[glue] = (float(loss), net.pixpro.projector.linear2.weight.grad.clone(), None, net.last_lse.clone(), net.last_rowmax.clone())
    assert abs(res["0"][0] - res["1"][0]) <= 2e-6 * abs(res["1"][0]), (res["0"][0], res["1"][0])
    assert rel(res["0"][1], res["1"][1]) < 2e-3
    # the (query, map) rows of the monitoring outputs are ordered the same way in both paths
    assert torch.allclose(res["0"][3], res["1"][3], rtol=1e-5, atol=1e-5) and torch.allclose(res["0"][4], res["1"][4], rtol=1e-5, atol=1e-5)
