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
        q.put((rank, e_y, e_dx, e_rm, e_rv, e_dw, e_il, int(sbn2.num_batches_tracked)))
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
    for rank, e_y, e_dx, e_rm, e_rv, e_dw, e_il, nbt in res:
        assert e_y < 1e-4 and e_dx < 1e-4 and e_rm < 1e-5 and e_rv < 1e-4 and e_dw < 1e-3, (rank, e_y, e_dx, e_rm, e_rv, e_dw)
        assert e_il < 2e-4 and nbt == 2, (rank, e_il, nbt)
