"""N > 1 data-parallel path on CPU: world_size 2, gloo.  Covers clip sharding, the bucketed overlapped gradient
all-reduce (what RCCL does over xGMI on the GPU box) and the embedding all-gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stswincl_amd.dp import GradBucketReducer, all_gather_embeddings, shard_indices


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
        red = GradBucketReducer(params, bucket_mb=0.0005, overlap=overlap)  # tiny buckets -> several of them
        assert len(red.buckets) >= 3, len(red.buckets)
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
