"""SyncBatchNorm semantics of the grouped-BN kernels: two processes (gloo) sharing the one GPU, each with half the batch,
must reproduce the full-batch BatchNorm forward, running stats and input gradient."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stswincl_amd import headops as H
        torch.manual_seed(0)
        f, c, h, w = 4, 64, 6, 5
        x = torch.randn(f, c, h, w) * 2 + 1
        g = torch.randn(f, c, h, w)
        bn = torch.nn.BatchNorm2d(c)
        bn.weight.data = 1 + 0.2 * torch.randn(c)
        bn.bias.data = 0.1 * torch.randn(c)
        ref_bn = torch.nn.BatchNorm2d(c)
        ref_bn.load_state_dict(bn.state_dict())
        xr = x.clone().requires_grad_(True)
        yr = torch.relu(ref_bn(xr))
        (yr * g).sum().backward()
        sbn = torch.nn.SyncBatchNorm.convert_sync_batchnorm(torch.nn.Sequential(bn))[0].cuda().train()
        sl = slice(rank * 2, rank * 2 + 2)
        xt = H.to_tokens(x[sl].cuda()).contiguous().requires_grad_(True)
        yt = H.batchnorm_tokens(xt, sbn, relu=True)
        (yt * H.to_tokens(g[sl].cuda())).sum().backward()
        e_y = float((H.from_tokens(yt, 2, h, w).cpu() - yr[sl].detach()).abs().max())
        e_dx = float((H.from_tokens(xt.grad, 2, h, w).cpu() - xr.grad[sl]).abs().max())
        e_rm = float((sbn.running_mean.cpu() - ref_bn.running_mean).abs().max())
        e_rv = float((sbn.running_var.cpu() - ref_bn.running_var).abs().max())
        dw = sbn.weight.grad.cpu().clone()
        dist.all_reduce(dw)                      # local sums add up to the full-batch weight gradient
        e_dw = float((dw - ref_bn.weight.grad).abs().max())
        # interleaved statistic groups (the batched views / per-frame groups of a clip-major batch): every rank holds 4 frames,
        # group g = frames g, g + 2; SyncBN statistics of group g = those frames of BOTH ranks; running statistics updated group
        # by group like sequential calls
        torch.manual_seed(1)
        xs = [torch.randn(4, c, h, w) * (1 + r) + r for r in range(world)]
        gs = [torch.randn(4, c, h, w) for _ in range(world)]
        bn2 = torch.nn.BatchNorm2d(c)
        bn2.weight.data = 1 + 0.2 * torch.randn(c)
        bn2.bias.data = 0.1 * torch.randn(c)
        ref2 = torch.nn.BatchNorm2d(c)
        ref2.load_state_dict(bn2.state_dict())
        outs, xrs = {}, {}
        for grp in range(2):
            xr2 = torch.cat([xs[r][grp::2] for r in range(world)]).clone().requires_grad_(True)      # rank-major
            yr2 = torch.relu(ref2(xr2))
            (yr2 * torch.cat([gs[r][grp::2] for r in range(world)])).sum().backward()
            outs[grp], xrs[grp] = yr2.detach(), xr2.grad
        sbn2 = torch.nn.SyncBatchNorm.convert_sync_batchnorm(torch.nn.Sequential(bn2))[0].cuda().train()
        xt2 = H.to_tokens(xs[rank].cuda()).contiguous().requires_grad_(True)
        yt2 = H.batchnorm_tokens(xt2, sbn2, relu=True, groups=2, il_frames=4)
        (yt2 * H.to_tokens(gs[rank].cuda())).sum().backward()
        y2, dx2 = H.from_tokens(yt2, 4, h, w).cpu(), H.from_tokens(xt2.grad, 4, h, w).cpu()
        e_il = 0.0
        for grp in range(2):
            mine = slice(rank * 2, rank * 2 + 2)                                                   # this rank's two frames of the group
            e_il = max(e_il, float((y2[grp::2] - outs[grp][mine]).abs().max()), float((dx2[grp::2] - xrs[grp][mine]).abs().max()))
        e_il = max(e_il, float((sbn2.running_mean.cpu() - ref2.running_mean).abs().max()),
                   float((sbn2.running_var.cpu() - ref2.running_var).abs().max()))
        # uneven shards (an unpadded last batch): rank 0 holds 3 frames, rank 1 one frame; statistics, running statistics and the
        # input gradient must still be those of the 4-frame batch (counts ride in the all-gather payload, nn.SyncBatchNorm does the same)
        bn3 = torch.nn.BatchNorm2d(c)
        bn3.load_state_dict(bn.state_dict())
        ref3 = torch.nn.BatchNorm2d(c)
        ref3.load_state_dict(bn.state_dict())
        xr3 = x.clone().requires_grad_(True)
        yr3 = torch.relu(ref3(xr3))
        (yr3 * g).sum().backward()
        sbn3 = torch.nn.SyncBatchNorm.convert_sync_batchnorm(torch.nn.Sequential(bn3))[0].cuda().train()
        sl3 = slice(0, 3) if rank == 0 else slice(3, 4)
        nf = 3 if rank == 0 else 1
        c0 = dict(H.COLLECTIVES)
        xt3 = H.to_tokens(x[sl3].cuda()).contiguous().requires_grad_(True)
        yt3 = H.batchnorm_tokens(xt3, sbn3, relu=True)
        (yt3 * H.to_tokens(g[sl3].cuda())).sum().backward()
        e_un = max(float((H.from_tokens(yt3, nf, h, w).cpu() - yr3[sl3].detach()).abs().max()),
                   float((H.from_tokens(xt3.grad, nf, h, w).cpu() - xr3.grad[sl3]).abs().max()),
                   float((sbn3.running_mean.cpu() - ref3.running_mean).abs().max()),
                   float((sbn3.running_var.cpu() - ref3.running_var).abs().max()))
        n_coll = (H.COLLECTIVES["syncbn_all_gather"] - c0["syncbn_all_gather"], H.COLLECTIVES["syncbn_all_reduce"] - c0["syncbn_all_reduce"])
        q.put((rank, e_y, e_dx, e_rm, e_rv, e_dw, e_il, int(sbn2.num_batches_tracked), e_un, n_coll))
    finally:
        dist.destroy_process_group()


def test_sync_batchnorm_two_ranks_one_gpu():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, e_y, e_dx, e_rm, e_rv, e_dw, e_il, nbt, e_un, n_coll in res:
        assert e_un < 2e-4, (rank, e_un)
        assert n_coll == (1, 1), n_coll          # one all-gather per BatchNorm forward, one all-reduce per backward
        assert e_y < 1e-4 and e_dx < 1e-4 and e_rm < 1e-5 and e_rv < 1e-4 and e_dw < 1e-3, (rank, e_y, e_dx, e_rm, e_rv, e_dw)
        assert e_il < 2e-4 and nbt == 2, (rank, e_il, nbt)


def _worker_group(rank, world, port, q):
    """Three independent SyncBatchNorm layers (different widths, one with interleaved per-frame groups, one without ReLU) through
    H.syncbn_group (ONE exchange) and one by one: outputs, input gradients, parameter gradients, running statistics and batch
    counters must be IDENTICAL (same kernels, same values on the wire, only packed differently)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stswincl_amd import headops as H
        torch.manual_seed(3)
        specs = [(64, 4 * 30, True, 1, 0), (128, 4 * 30, False, 2, 4), (256, 4 * 9, True, 1, 0)]    # (C, rows, relu, groups, il_frames)
        res = {}
        for mode in ("each", "group"):
            torch.manual_seed(7)
            bns, xs, gs = [], [], []
            for C, rows, relu, groups, ilf in specs:
                bn = torch.nn.BatchNorm2d(C)
                bn.weight.data = 1 + 0.2 * torch.randn(C)
                bn.bias.data = 0.1 * torch.randn(C)
                bns.append(torch.nn.SyncBatchNorm.convert_sync_batchnorm(torch.nn.Sequential(bn))[0].cuda().train())
                xa = torch.randn(world, rows, C) * 2 + 1
                ga = torch.randn(world, rows, C)
                xs.append(xa[rank].cuda().requires_grad_(True))
                gs.append(ga[rank].cuda())
            c0 = dict(H.COLLECTIVES)
            if mode == "each":
                ys = [H.batchnorm_tokens(x, bn, relu=sp[2], groups=sp[3], il_frames=sp[4]) for x, bn, sp in zip(xs, bns, specs)]
            else:
                with H.syncbn_group() as grp:
                    for x, bn, sp in zip(xs, bns, specs):
                        grp.bn(x, bn, relu=sp[2], groups=sp[3], il_frames=sp[4])
                ys = grp.results()
            sum((y * g).sum() for y, g in zip(ys, gs)).backward()
            torch.cuda.synchronize()
            n_coll = (H.COLLECTIVES.get("syncbn_all_gather", 0) - c0.get("syncbn_all_gather", 0),
                      H.COLLECTIVES.get("syncbn_all_reduce", 0) - c0.get("syncbn_all_reduce", 0))
            res[mode] = ([y.detach().cpu() for y in ys], [x.grad.cpu() for x in xs], [bn.weight.grad.cpu() for bn in bns],
                         [bn.bias.grad.cpu() for bn in bns], [bn.running_mean.cpu() for bn in bns], [bn.running_var.cpu() for bn in bns],
                         [int(bn.num_batches_tracked) for bn in bns], n_coll)
        same = all(torch.equal(a, b) for k in range(6) for a, b in zip(res["each"][k], res["group"][k]))
        q.put((rank, same, res["each"][6] == res["group"][6], res["each"][7], res["group"][7]))
    finally:
        dist.destroy_process_group()


def test_syncbn_group_one_exchange_equals_per_layer_exchanges():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_group, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, same, same_nbt, c_each, c_group in res:
        assert same and same_nbt, rank
        assert c_each == (3, 3) and c_group == (1, 1), (c_each, c_group)


def _worker_budget(rank, world, port, q):
    """One data-parallel ConsistencyLoss step (2 ranks sharing the GPU over gloo, GradBucketReducer, SyncBatchNorm): counts the
    collectives the step issues."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import types
        from stswincl_amd import headops as H
        from stswincl_amd.contrast.models import PixPro_swin_v5 as P
        from stswincl_amd.dp import GradBucketReducer
        args = types.SimpleNamespace(pixpro_p=1.0, pixpro_momentum=0.99, pixpro_clamp_value=0.0, pixpro_transform_layer=1,
                                     pixpro_ins_loss_weight=0.0, pixpro_pos_ratio=0.7, data="endo18", tag="1", pretrainpth="none",
                                     num_instances=2235, batch_size=2, epochs=150, start_epoch=1)
        torch.manual_seed(0)
        net = P.ConsistencyLoss(args, input_resolution=(8, 8)).cuda().train()
        px = net.pixpro
        q_mods = (px.encoder_1, px.encoder_2, px.encoder_3, px.proj1, px.proj2, px.proj3, px.projector)
        k_mods = (px.encoder_k_1, px.encoder_k_2, px.encoder_k_3, px.proj_k_1, px.proj_k_2, px.proj_k_3, px.projector_k)
        n_q = sum(isinstance(m, torch.nn.SyncBatchNorm) for mod in q_mods for m in mod.modules())
        n_k = sum(isinstance(m, torch.nn.SyncBatchNorm) for mod in k_mods for m in mod.modules())
        red = GradBucketReducer([p for p in net.parameters() if p.requires_grad], bucket_mb=32.0)
        torch.manual_seed(10 + rank)
        ims = [torch.randn(2, 4, 3, 64, 64, device="cuda") for _ in range(6)]
        masks = [torch.randint(0, 12, (2, 1, 8, 8), device="cuda").float().repeat_interleave(8, 2).repeat_interleave(8, 3) for _ in range(6)]
        counts = []
        for _ in range(2):
            c0, r0, b0 = dict(H.COLLECTIVES), red.collectives, red.copied_bytes
            for p in net.parameters():
                p.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = net(*ims, *masks)
            loss.backward()
            red.finish()
            torch.cuda.synchronize()
            counts.append((H.COLLECTIVES.get("syncbn_all_gather", 0) - c0.get("syncbn_all_gather", 0),
                           H.COLLECTIVES.get("syncbn_all_reduce", 0) - c0.get("syncbn_all_reduce", 0),
                           red.collectives - r0, red.copied_bytes - b0))
        gsum = float(sum(p.grad.double().abs().sum() for p in net.parameters() if p.grad is not None))
        q.put((rank, n_q, n_k, len(red.buckets), red.bytes_per_step(), counts, float(loss.detach()), gsum))
    finally:
        dist.destroy_process_group()


def test_collective_budget_of_a_data_parallel_contrastive_step():
    """What a step may cost on the wire (checked without multi-GPU hardware): SyncBatchNorm = ONE all-gather per BatchNorm layer
    per batched pass (the 2 query views are one pass, the 6 key views another - not one exchange per view) + ONE all-reduce per
    query-side layer in backward - and layers whose inputs are independent (ASPP branches, decode-head projections, conv1 /
    downsample of a ResNet block) share one exchange (round 4: 9 exchanges fewer per pass and direction); gradients = one all-reduce per 32 MB bucket; and the large weight gradients are born inside
    their bucket slices (most bytes are never copied).  Both ranks end with identical averaged gradients."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_budget, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # exchanges saved per encoder pass by H.syncbn_group: the five ASPP branches share one (-4), the three decode-head projections
    # share one (-2), conv1 / downsample of the three ResNet blocks that have a shortcut convolution share one each (-3)
    merged = 4 + 2 + 3
    for rank, n_q, n_k, nb, nbytes, counts, loss, gsum in res:
        assert n_q == n_k and n_q >= 30
        for ag, ar, coll, copied in counts:
            assert ag == (n_q - merged) + (n_k - merged), (ag, n_q, n_k)
            assert ar == n_q - merged, (ar, n_q)
            assert coll == nb, (coll, nb)
            assert copied < 0.35 * nbytes, (copied, nbytes)         # the Swin / 1x1 weight gradients (most of the bytes) are written in place
        assert loss == loss
    assert abs(res[0][7] - res[1][7]) <= 1e-6 * abs(res[0][7])
