"""N > 1 data-parallel path on CPU: world_size 2, gloo.  Covers clip sharding, the bucketed overlapped gradient
all-reduce (what RCCL does over xGMI on the GPU box) and the embedding all-gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stswincl_amd.dp import GradBucketReducer, all_gather_embeddings, grad_dest, shard_indices


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, overlap, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)                       # same weights on every rank
        model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32), torch.nn.ReLU(),
                                    torch.nn.Linear(32, 4))
        frozen = torch.nn.Linear(4, 4)             # an unused / frozen parameter like PixPro.value_transform
        for p in frozen.parameters():
            p.requires_grad = False
        params = list(model.parameters()) + list(frozen.parameters())
        red = GradBucketReducer(params, bucket_mb=0.0005, overlap=overlap, hold_tn_fused=True)  # tiny buckets -> several of them
        assert len(red.buckets) >= 3, len(red.buckets)
        # hold_tn_fused=True: an overlapped reducer of a world > 1 holds the library's refcounted switch - weight-gradient GEMMs use the
        # separate split-K combine pass while it lives (dp.py; the default leaves the fused combine on: it no longer needs the whole grid
        # resident); a second holder and any order of release must work, and the environment is not touched
        from stswincl_amd import hip
        assert hip.tn_fused_holds() == (1 if overlap else 0) and "STSWIN_TN_FUSED" not in os.environ
        extra = hip.TnFusedHold()
        assert hip.tn_fused_holds() == (2 if overlap else 1)
        data = torch.arange(8 * 16, dtype=torch.float32).reshape(8, 16) / 100.0
        mine = shard_indices(8, rank, world)
        for step in range(2):                      # two steps: hooks / bucket state must reset
            model.zero_grad()
            loss = model(data[mine]).pow(2).mean()
            loss.backward()
            red.finish()
        g = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
        # single-process reference on the union of the shards (mean over ranks of per-rank means)
        ref_model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32), torch.nn.ReLU(),
                                        torch.nn.Linear(32, 4))
        ref_model.load_state_dict(model.state_dict())
        tot = sum(ref_model(data[shard_indices(8, r, world)]).pow(2).mean() for r in range(world)) / world
        tot.backward()
        gr = torch.cat([p.grad.reshape(-1) for p in ref_model.parameters()])
        emb = all_gather_embeddings(torch.full((3, 2), float(rank)))
        red.close()
        assert hip.tn_fused_holds() == 1                     # (the reducer's hold is back, the second holder's is not)
        red.close()                                          # idempotent
        assert hip.tn_fused_holds() == 1
        del extra                                            # a holder dropped without release() gives its hold back
        import gc
        gc.collect()
        assert hip.tn_fused_holds() == 0
        q.put((rank, float((g - gr).abs().max()), emb.shape, emb[:, 0].tolist(), mine))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_bucketed_allreduce_world2(overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, shape, col, mine in res:
        assert err < 1e-6, (rank, err)
        assert tuple(shape) == (6, 2) and col == [0.0, 0.0, 0.0, 1.0, 1.0, 1.0]
        assert mine == [rank, rank + 2, rank + 4, rank + 6]


def _worker_accum(rank, world, port, q):
    """Gradient accumulation (two micro-batches, the first under no_sync), a parameter whose gradient is None on ONE rank
    only (bucket sizes must still agree), and the loud failure of a second backward() without no_sync."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        a, b = torch.nn.Linear(8, 8), torch.nn.Linear(8, 8)        # b is used by rank 0 only
        params = list(a.parameters()) + list(b.parameters())
        red = GradBucketReducer(params, bucket_mb=1.0)
        x = torch.arange(4 * 8, dtype=torch.float32).reshape(4, 8) / 10 + rank

        def loss_of(xx):
            y = a(xx)
            return (b(y) if rank == 0 else y).pow(2).mean()

        with red.no_sync():
            loss_of(x[:2]).backward()
        loss_of(x[2:]).backward()
        red.finish()
        ga = a.weight.grad.clone()
        gb = b.weight.grad.clone()
        # reference: mean over ranks of (sum of the two micro-batch gradients); rank 1 contributes zeros to b
        refs_a, refs_b = [], []
        for r in range(world):
            a2, b2 = torch.nn.Linear(8, 8), torch.nn.Linear(8, 8)
            a2.load_state_dict(a.state_dict()), b2.load_state_dict(b.state_dict())
            xr = torch.arange(4 * 8, dtype=torch.float32).reshape(4, 8) / 10 + r
            for part in (xr[:2], xr[2:]):
                y = a2(part)
                (b2(y) if r == 0 else y).pow(2).mean().backward()
            refs_a.append(a2.weight.grad)
            refs_b.append(b2.weight.grad if b2.weight.grad is not None else torch.zeros(8, 8))
        err = max(float((ga - sum(refs_a) / world).abs().max()), float((gb - sum(refs_b) / world).abs().max()))
        raised = False
        c = torch.nn.Linear(8, 8)
        red2 = GradBucketReducer(list(c.parameters()), bucket_mb=1.0)
        c(x[:2]).pow(2).mean().backward()
        try:
            c(x[2:]).pow(2).mean().backward()
        except RuntimeError as e:
            raised = "no_sync" in str(e)
        red2.finish()                                   # (drains the collective the first backward launched)
        q.put((rank, err, raised))
    finally:
        dist.destroy_process_group()


def test_accumulation_none_grads_and_double_backward_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_accum, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, err, raised in res:
        assert err < 1e-6, (rank, err)
        assert raised, "a second backward() before finish() must raise"


class _DirectLinear(torch.autograd.Function):
    """A weight-gradient producer that follows the protocol of stswincl_amd.ops.wgrad_buffer: it asks for the parameter's
    bucket slice, OVERWRITES it and returns it as the gradient."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        buf = grad_dest(w, w.shape)
        if buf is None:
            buf = torch.empty_like(w)
        torch.mm(g.t(), x, out=buf)
        return g @ w, buf


def _worker_direct(rank, world, port, comm_dtype, q):
    """Bucket slices as gradient destinations: gradients written in place are not copied (copied_bytes counts only the
    torch-produced bias gradients), p.grad IS the bucket slice after finish(), one collective per bucket per step, a
    parameter used twice in one graph and zero_grad(set_to_none=False) both still give the right sums; and the
    compressed-wire variant (comm_dtype=bf16: the conversion back runs behind the collective)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        lins = [torch.nn.Linear(16, 16) for _ in range(4)]
        params = [p for l in lins for p in l.parameters()]
        red = GradBucketReducer(params, bucket_mb=0.001, comm_dtype=comm_dtype)
        nb = len(red.buckets)
        x = torch.arange(6 * 16, dtype=torch.float32).reshape(6, 16) / 50 + rank

        def fwd(xx, twice):
            h = xx
            for l in lins:
                h = torch.tanh(_DirectLinear.apply(h, l.weight) + l.bias)
            if twice:                                      # lins[0].weight a second time in the same graph
                h = h + _DirectLinear.apply(xx, lins[0].weight)
            return h.pow(2).mean()

        out = {}
        for step, (twice, to_none) in enumerate([(False, True), (True, True), (False, False)]):
            for p in params:
                if to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()
            c0, b0 = red.collectives, red.copied_bytes
            fwd(x, twice).backward()
            red.finish()
            refs = []
            for r in range(world):
                ls = [torch.nn.Linear(16, 16) for _ in range(4)]
                for a, b in zip(ls, lins):
                    a.load_state_dict(b.state_dict())
                xr = torch.arange(6 * 16, dtype=torch.float32).reshape(6, 16) / 50 + r
                h = xr
                for l in ls:
                    h = torch.tanh(h @ l.weight.t() + l.bias)
                if twice:
                    h = h + xr @ ls[0].weight.t()
                h.pow(2).mean().backward()
                refs.append([p.grad for l in ls for p in l.parameters()])
            err = max(float((p.grad - sum(rf[i] for rf in refs) / world).abs().max() / (refs[0][i].abs().max() + 1e-12))
                      for i, p in enumerate(params))
            views = {id(p): v for b, vs in zip(red.buckets, red._views) for p, v in zip(b, vs)}
            in_bucket = all(p.grad.data_ptr() == views[id(p)].data_ptr() for p in params)
            out[step] = (err, in_bucket, red.collectives - c0, red.copied_bytes - b0)
        q.put((rank, nb, sum(l.bias.numel() * 4 for l in lins), out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("comm_dtype", [None, torch.bfloat16])
def test_bucket_slices_as_gradient_destinations_world2(comm_dtype):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_direct, args=(r, 2, port, comm_dtype, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    tol = 1e-5 if comm_dtype is None else 2e-2
    for rank, nb, bias_bytes, out in res:
        assert nb >= 3
        for step, (err, in_bucket, ncoll, copied) in out.items():
            assert err < tol, (rank, step, err)
            assert in_bucket, (rank, step)
            assert ncoll == nb, (rank, step, ncoll, nb)             # one all-reduce per bucket per step, never per tensor
        # step 0: only the biases (torch-produced) were copied; step 1: the twice-used weight's two gradients are summed by the
        # engine's input buffer before they reach the parameter (the slice was handed out once; the sum may live elsewhere and
        # is then copied: at most one weight more); step 2 (zero_grad in place): everything accumulates into the slices
        assert out[0][3] == bias_bytes, out[0]
        assert out[1][3] in (bias_bytes, bias_bytes + 16 * 16 * 4), out[1]
        assert out[2][3] == 0, out[2]


def _worker_bank(rank, world, port, q):
    """Inter-video key bank: gather_bank's layout ([maps][world * rows][C], rank-major inside a map) and the bank loss of one
    rank's queries against the gathered bank == the dense oracle on the concatenated keys."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import stswin_oracle as O
        from stswincl_amd.contrast.models.PixPro_swin_v5 import gather_bank
        maps, rows, c = 6, 40, 32
        gen = [torch.Generator().manual_seed(100 + r) for r in range(world)]
        banks = [torch.nn.functional.normalize(torch.randn(maps, rows, c, generator=g), dim=2) for g in gen]
        labels = [torch.randint(0, 12, (maps, rows), generator=g, dtype=torch.int32) for g in gen]
        bank, lb = gather_bank(banks[rank], labels[rank])
        ref_bank, ref_lb = torch.cat(banks, 1), torch.cat(labels, 1)
        ok_layout = torch.equal(bank, ref_bank) and torch.equal(lb, ref_lb)
        qg = torch.Generator().manual_seed(7 + rank)
        qq = torch.nn.functional.normalize(torch.randn(2 * rows, c, generator=qg), dim=1)
        lq = torch.cat([labels[rank][0], labels[rank][1]]).long()
        gmap = [[1, 2, 3, 4, 5], [0, 2, 3, 4, 5]]
        a = O.bank_contrast_loss(qq, lq, bank, lb.long(), gmap, rows, world * rows)
        b = O.bank_contrast_loss(qq, lq, ref_bank, ref_lb.long(), gmap, rows, world * rows)
        q.put((rank, ok_layout, float(a), float(b), tuple(bank.shape)))
    finally:
        dist.destroy_process_group()


def test_gather_bank_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok_layout, a, b, shape in res:
        assert ok_layout and shape == (6, 80, 32), (rank, shape)
        assert a == b


def test_shard_indices_padding_and_shuffle():
    assert shard_indices(5, 0, 2) == [0, 2, 4] and shard_indices(5, 1, 2) == [1, 3, 0]
    a, b = shard_indices(10, 0, 2, epoch=3, shuffle=True), shard_indices(10, 1, 2, epoch=3, shuffle=True)
    assert sorted(a + b) == list(range(10))
    assert a != shard_indices(10, 0, 2, epoch=4, shuffle=True)


def test_compat_aliases_expose_reference_paths():
    from stswincl_amd import compat
    compat.install()
    import importlib
    m = importlib.import_module("net.Ours.swin_512")
    assert hasattr(m, "SwinTransformerLayerv5") and hasattr(m, "WindowAttention")
    assert hasattr(importlib.import_module("net.Ours.base18"), "TswinPlus")
    assert hasattr(importlib.import_module("utils.losses"), "OhemCELoss2D")
    assert hasattr(importlib.import_module("contrast.models.PixPro_swin_v5"), "ConsistencyLoss")


def test_sync_bn_stat_combination_matches_full_batch():
    """The cross-rank statistic merge used for SyncBatchNorm (headops.combine_bn_stats) on a split batch."""
    from stswincl_amd.headops import combine_bn_stats
    torch.manual_seed(0)
    x = torch.randn(3, 50, 7) * 3 + 5                 # [groups][rows][C]
    parts = [x[:, :20], x[:, 20:]]                    # two "ranks" with different row counts
    mean_r = torch.stack([p.mean(1) for p in parts])
    m2_r = torch.stack([((p - p.mean(1, keepdim=True)) ** 2).sum(1) for p in parts])
    n_r = torch.tensor([20.0, 30.0])
    mean, var, n = combine_bn_stats(mean_r, m2_r, n_r)
    assert torch.allclose(mean, x.mean(1), atol=1e-5) and torch.allclose(var, x.var(1, unbiased=False), atol=1e-4)
    assert int(n) == 50
