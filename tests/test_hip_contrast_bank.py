"""N1: bank mode of the pixel-contrastive similarity (stswin_contrast_bank_fwd / class_sums / bank_dq) against the dense CPU
oracle (oracle.bank_scores: logits + one-hot style masks like PixPro_swin_v5.py:82-113) and the reference goldens."""
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


def _case(seed, M, C, maps, seg, ncls=12):
    torch.manual_seed(seed)
    q = F.normalize(torch.randn(M, C), dim=1)
    bank = F.normalize(torch.randn(maps, seg, C), dim=2)
    lq = torch.randint(0, ncls, (M,), dtype=torch.int32)
    lb = torch.randint(0, ncls, (maps, seg), dtype=torch.int32)
    return q, lq, bank, lb


# (M, C, maps, seg, gmap, q_block, bank_block): per-sample blocks with ragged HW (63 = 7x9, 240 = 12x20), both directions in one
# launch, one block seeing a long segment (bank splits > 1), C = 64 / 128 / 256
CASES = [
    (2 * 64, 256, 5, 2 * 64, [[0, 1, 2, 3, 4]], 64, 64),
    (2 * 3 * 63, 256, 6, 3 * 63, [[1, 2, 3, 4, 5], [0, 2, 3, 4, 5]], 63, 63),
    (3 * 240, 64, 5, 3 * 240, [[0, 1, 2, 3, 4]], 240, 240),
    (2 * 256, 128, 6, 4096, [[1, 2, 3, 4, 5], [0, 2, 3, 4, 5]], 256, 4096),
    (300, 256, 2, 5000, [[0, 1]], 300, 5000),
]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_bank_fwd_matches_dense_oracle(case, dtype, tol):
    M, C, maps, seg, gmap, qb, bb = CASES[case]
    q, lq, bank, lb = _case(case, M, C, maps, seg)
    lb[0, :bb] = lq[0]                                # a query whose negative set in group 0 may be small / empty
    qd, bd = q.to(dtype), bank.to(dtype)
    pos, tot, rmax, lse = hip.contrast_bank_fwd(qd.cuda(), lq.cuda(), bd.cuda(), lb.cuda(), q_sets=len(gmap), q_block=qb,
                                                bank_block=bb, gmap=gmap, inv_tau=5.0, want_lse=True)
    rp, rn, rm, rl = O.bank_scores(qd.float(), lq.long(), bd.float(), lb.long(), gmap, qb, bb, inv_tau=5.0)
    scale = float((rp + rn).abs().max()) + 1e-6
    assert float((pos.cpu() - rp).abs().max()) <= tol * scale, "pos"
    assert float(((tot - pos).cpu() - rn).abs().max()) <= tol * scale, "neg = all - pos"
    assert float((rmax.cpu() - rm).abs().max()) <= tol * 5.0, "row max"
    assert float((lse.cpu() - rl).abs().max()) <= tol * 5.0 + 1e-4, "log-sum-exp"


def test_empty_negative_set_is_exactly_zero():
    """All visible keys share the query's label: neg = all - pos must be bitwise 0 (its denominator is 0 + 1e-6) and its
    gradient exactly zero, as with the reference's masked products (PixPro_swin_v5.py:103-113)."""
    M, C, maps, seg, gmap, qb, bb = 128, 256, 5, 128, [[0, 1, 2, 3, 4]], 64, 64
    q, lq, bank, lb = _case(7, M, C, maps, seg)
    lq[:64] = 3
    lb[2, :64] = 3                                   # group 2 of block 0: no negatives at all
    for dtype in (torch.float32, torch.bfloat16):
        pos, tot, _, _ = hip.contrast_bank_fwd(q.to(dtype).cuda(), lq.cuda(), bank.to(dtype).cuda(), lb.cuda(), q_sets=1, q_block=qb,
                                               bank_block=bb, gmap=gmap)
        assert torch.equal(pos[:64, 2], tot[:64, 2])
    qg = q.clone().cuda().requires_grad_(True)
    loss, _, _ = P.bank_contrast_loss(qg, lq.cuda(), bank.cuda(), lb.cuda(), gmap, qb, bb, 12)
    loss.backward()
    qo = q.clone().requires_grad_(True)
    lo = O.bank_contrast_loss(qo, lq.long(), bank, lb.long(), gmap, qb, bb)
    lo.backward()
    assert abs(float(loss) - float(lo)) < 1e-5 * abs(float(lo))
    assert rel(qg.grad, qo.grad) < 1e-4


@pytest.mark.parametrize("mode", ["sample", "batch"])
@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-4), ("bf16", 2e-2)])
def test_pair_loss_and_gradient_vs_oracle(mode, dtype, tol):
    """consistency_pair_loss (both directions, one launch) in the reference's per-sample mode and with the rank's whole batch as
    the bank, value and d/d(pred_1), d/d(pred_2) against the dense oracle."""
    torch.manual_seed(11)
    n, c, h, w = 3, 256, 8, 12
    feats = [F.normalize(torch.randn(n, c, h, w), dim=1) for _ in range(8)]
    m = [torch.randint(0, 12, (n, 1, h, w)).float() for _ in range(6)]
    p1, p2 = feats[0].clone().cuda().requires_grad_(True), feats[1].clone().cuda().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(dtype == "bf16")):
        loss, _, _ = P.consistency_pair_loss(p1, p2, *[f.cuda() for f in feats[2:]], [x.cuda() for x in m], 12, bank_mode=mode)
    loss.backward()
    tok = lambda t: t.permute(0, 2, 3, 1).reshape(n * h * w, c)      # noqa: E731
    q = torch.cat([tok(feats[0]), tok(feats[1])], 0).clone().requires_grad_(True)
    bank = torch.stack([tok(f) for f in feats[2:]], 0)
    lb = torch.stack([x.reshape(-1).long() for x in m], 0)
    lq = torch.cat([lb[0], lb[1]])
    qb, bb = (h * w, h * w) if mode == "sample" else (n * h * w, n * h * w)
    lo = O.bank_contrast_loss(q, lq, bank, lb, [[1, 2, 3, 4, 5], [0, 2, 3, 4, 5]], qb, bb)
    lo.backward()
    assert abs(float(loss) - float(lo)) < tol * abs(float(lo))
    g = q.grad.view(2, n, h, w, c).permute(0, 1, 4, 2, 3)
    assert rel(p1.grad, g[0]) < 50 * tol and rel(p2.grad, g[1]) < 50 * tol


def test_legacy_per_map_kernel_agrees_with_bank_kernel():
    """stswin_contrast_fwd (round-1 entry point, one launch per direction) and the bank kernel on the same problem."""
    torch.manual_seed(3)
    n, hw, c = 2, 240, 256
    q = F.normalize(torch.randn(n * hw, c), dim=1).cuda()
    keys = [F.normalize(torch.randn(n * hw, c), dim=1).cuda() for _ in range(5)]
    lq = torch.randint(0, 12, (n, hw), dtype=torch.int32).cuda()
    lks = [torch.randint(0, 12, (n, hw), dtype=torch.int32).cuda() for _ in range(5)]
    p0, t0 = hip.contrast_fwd(q, keys, lq, lks, n, hw)
    p1, t1, _, _ = hip.contrast_bank_fwd(q, lq.reshape(-1), torch.stack(keys, 0), torch.stack([x.reshape(-1) for x in lks], 0),
                                         q_sets=1, q_block=hw, bank_block=hw, gmap=[[0, 1, 2, 3, 4]])
    assert torch.allclose(p0.view(n * hw, 5), p1, atol=2e-4) and torch.allclose(t0.view(n * hw, 5), t1, atol=2e-4)


def test_production_size_bank_of_65k_entries():
    """BASELINE configs[3] bank size: 8192 query pixels x 2 directions against a 65536-entry bank per key map (8 ranks x 8 clips x
    32x32 pixels), bf16.  Dense CPU reference on a sample of the rows (the whole problem is 1.4 TFLOP)."""
    M, C, maps, seg = 2 * 8192, 256, 6, 65536
    q, lq, bank, lb = _case(5, M, C, maps, seg)
    gmap = [[1, 2, 3, 4, 5], [0, 2, 3, 4, 5]]
    qd, bd = q.bfloat16(), bank.bfloat16()
    pos, tot, rmax, lse = hip.contrast_bank_fwd(qd.cuda(), lq.cuda(), bd.cuda(), lb.cuda(), q_sets=2, q_block=8192, bank_block=seg,
                                                gmap=gmap, inv_tau=10.0, want_lse=True)
    rows = torch.cat([torch.arange(0, 8192, 257), torch.arange(8192, 16384, 263)])
    bf = bd.float()
    for r in rows.tolist():
        s = r // 8192
        logits = torch.stack([bf[mp] @ qd[r].float() for mp in gmap[s]], 0)                  # [5][65536]
        match = torch.stack([lb[mp] == lq[r] for mp in gmap[s]], 0)
        rp = (logits * match).sum(1)
        ra = logits.sum(1)
        assert float((pos[r].cpu() - rp).abs().max()) < 2e-2 * float(logits.abs().sum(1).max()) ** 0.5 + 0.05, r
        assert float((tot[r].cpu() - ra).abs().max()) < 2e-2 * float(logits.abs().sum(1).max()) ** 0.5 + 0.05, r
        assert abs(float(rmax[r]) - 10.0 * float(logits.max())) < 2e-2
        assert abs(float(lse[r]) - float(torch.logsumexp(10.0 * logits.reshape(-1), 0))) < 2e-2
    # the fixed-reference form for unit-norm rows (what the model's token path calls): same outputs to fp32 rounding
    pos_u, tot_u, rmax_u, lse_u = hip.contrast_bank_fwd(qd.cuda(), lq.cuda(), bd.cuda(), lb.cuda(), q_sets=2, q_block=8192, bank_block=seg,
                                                        gmap=gmap, inv_tau=10.0, want_lse=True, unit_rows=True)
    assert torch.equal(pos_u, pos) and torch.equal(tot_u, tot)
    assert float((rmax_u - rmax).abs().max()) <= 1e-5 * float(rmax.abs().max()) + 1e-6
    assert float((lse_u - lse).abs().max()) <= 1e-4


def test_unit_rows_lse_at_cold_temperatures_takes_the_online_form():
    """The fixed-reference log-sum-exp of unit-norm rows sums exp(inv_tau (s - 1)): at inv_tau > 40 scores near -1 flush to zero in fp32
    (round-5 advisor), so the launcher falls back to the online (running-maximum) form there - same numbers as unit_rows=False."""
    M, C, maps, seg = 256, 64, 2, 512
    torch.manual_seed(11)
    q = F.normalize(torch.randn(M, C), dim=1)
    bank = -q[torch.randint(0, M, (maps, seg))] + 1e-3 * torch.randn(maps, seg, C)          # every key is (almost) the negative of a query
    bank = F.normalize(bank, dim=2)
    lq = torch.randint(0, 12, (M,), dtype=torch.int32)
    lb = torch.randint(0, 12, (maps, seg), dtype=torch.int32)
    out = {}
    for unit in (False, True):
        out[unit] = hip.contrast_bank_fwd(q.cuda(), lq.cuda(), bank.cuda(), lb.cuda(), q_sets=1, q_block=M, bank_block=seg, gmap=[[0, 1]],
                                          inv_tau=200.0, want_lse=True, unit_rows=unit)
    assert torch.isfinite(out[True][3]).all()
    assert torch.equal(out[True][3], out[False][3]) and torch.equal(out[True][2], out[False][2])
    ref = torch.logsumexp(200.0 * torch.cat([bank[0] @ q.T, bank[1] @ q.T], 0).T, 1)
    assert float((out[True][3].cpu() - ref).abs().max()) < 1e-2


# ---------------------------------------------------------------------------------------------------------------------------------
# bank_mode 'world' through the TOKEN path (PairLossFn) on two ranks sharing the GPU (gloo): the query labels are this rank's, the
# bank and its labels are every rank's.  (Round-5 advisor finding: the query labels were taken from the GATHERED label matrix.)
def _world_token_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import stswin_oracle as O
        from stswincl_amd.contrast.models import PixPro_swin_v5 as P
        from stswincl_amd import headops as H
        b, h, w, C, ncls = 2, 4, 5, 64, 12          # (C: a multiple of 64, rownorm_scatter's row layout)
        HW = h * w

        def rank_data(r):
            g = torch.Generator().manual_seed(100 + r)
            proj = torch.randn(2 * b * HW, C, generator=g)                       # clip-major rows: clip = sample * 2 + view
            keys = torch.nn.functional.normalize(torch.randn(6, b * HW, C, generator=g), dim=2)
            lab = torch.randint(0, ncls, (6, b * HW), generator=g, dtype=torch.int32)
            return proj, keys, lab

        data = [rank_data(r) for r in range(world)]
        proj, keys, lab = data[rank]
        pq = proj.cuda().requires_grad_(True)
        loss, _, _ = P.pair_loss_tokens(pq, keys.cuda().contiguous(), lab.cuda().contiguous(), b, HW, ncls, bank_mode="world")
        loss.backward()
        # expectation without any collective: this rank's (normalised, view-major) queries against the rank-major concatenation
        qn = torch.nn.functional.normalize(proj, dim=1).view(b, 2, HW, C).permute(1, 0, 2, 3).reshape(2 * b * HW, C)
        qn = qn.clone().requires_grad_(True)
        bank = torch.cat([d[1] for d in data], 1)
        lb = torch.cat([d[2] for d in data], 1).long()
        lq = torch.cat([lab[0], lab[1]]).long()
        gmap = [[1, 2, 3, 4, 5], [0, 2, 3, 4, 5]]
        ref = O.bank_contrast_loss(qn, lq, bank, lb, gmap, b * HW, world * b * HW)
        # the torch-glue formulation of the same mode (consistency_pair_loss) on NCHW embeddings
        qv = torch.nn.functional.normalize(proj, dim=1).view(b, 2, HW, C)
        nchw = lambda t: t.reshape(b, h, w, C).permute(0, 3, 1, 2).contiguous().cuda()
        ks = [nchw(keys[i]) for i in range(6)]
        ms = [lab[i].view(b, 1, h, w).float().cuda() for i in range(6)]
        glue, _, _ = P.consistency_pair_loss(nchw(qv[:, 0]), nchw(qv[:, 1]), *ks, ms, ncls, bank_mode="world")
        # gradient reference: oracle autograd through the normalisation
        pr = proj.clone().requires_grad_(True)
        qr = torch.nn.functional.normalize(pr, dim=1).view(b, 2, HW, C).permute(1, 0, 2, 3).reshape(2 * b * HW, C)
        O.bank_contrast_loss(qr, lq, bank, lb, gmap, b * HW, world * b * HW).backward()
        e_g = float((pq.grad.cpu() - pr.grad).norm() / pr.grad.norm())
        q.put((rank, float(loss), float(ref), float(glue), e_g))
    finally:
        dist.destroy_process_group()


def test_token_path_world_bank_on_two_ranks():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_world_token_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0, "a rank failed (its traceback is in the captured stderr)"
    res = [q.get(timeout=10) for _ in procs]
    losses = {}
    for rank, loss, ref, glue, e_g in res:
        assert abs(loss - ref) <= 2e-5 * abs(ref), (rank, loss, ref)
        assert abs(glue - ref) <= 2e-5 * abs(ref), (rank, glue, ref)
        assert e_g < 2e-4, (rank, e_g)
        losses[rank] = loss
    assert losses[0] != losses[1]          # different queries per rank: a path that read rank 0's labels everywhere would not differ only here


def test_query_label_count_is_checked():
    Q = torch.nn.functional.normalize(torch.randn(64, 64, device="cuda"), dim=1)
    bank = torch.nn.functional.normalize(torch.randn(6, 64, 64, device="cuda"), dim=2)
    lb = torch.randint(0, 12, (6, 64), device="cuda", dtype=torch.int32)
    with pytest.raises(hip.StswinHipError):
        hip.contrast_bank_fwd(Q, lb[:2].reshape(-1).repeat(2), bank, lb, q_sets=2, q_block=32, bank_block=64,
                              gmap=((1, 2, 3, 4, 5), (0, 2, 3, 4, 5)))
