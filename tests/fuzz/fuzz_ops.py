#!/usr/bin/env python3
"""Randomised parity sweep of the module-level HIP paths against torch / the CPU oracle: convolutions on tokens (random
geometry), BatchNorm on tokens (groups, interleaved units, residual, train / eval), Swin blocks and PatchMerging (random
resolution / window / shift / batch), max-pool, bilinear, OHEM, regression_loss, fused optimizers.  fp32 path unless noted.
Prints every mismatch; exit code 1 if any.   usage: fuzz_ops.py [cases per family] [seed]   (FUZZ_ONLY / FUZZ_SKIP: family name substrings, FUZZ_BF16=1)
Test infrastructure: lives under tests/ because it checks against oracle/."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn as nn
import torch.nn.functional as F
from oracle import stswin_oracle as O
from stswincl_amd import headops as H
from stswincl_amd.net.Ours import swin_512 as S
from stswincl_amd.utils.losses import OhemCELoss2D
from stswincl_amd.contrast.models import PixPro_swin_v5 as P
from stswincl_amd.optim import FusedAdam, FusedSGD

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0


def rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


BF16 = os.environ.get("FUZZ_BF16") == "1"          # conv / BatchNorm / Swin families under bf16 autocast, tolerances x BF16_TOL
BF16_TOL = 40.0


def amp():
    return torch.autocast("cuda", dtype=torch.bfloat16, enabled=BF16)


def check(name, r, tol, info):
    global bad
    if BF16 and name.split("/")[0] in ("conv", "bn", "swin", "merge"):
        tol = min(tol * BF16_TOL, 0.08)
        if name in ("bn/dbeta", "bn/dgamma"):      # sums over a few hundred masked elements: one flipped ReLU is several per cent
            tol = 0.25
    if not (r < tol):
        bad += 1
        print(f"MISMATCH {name}: rel {r:.3e} (tol {tol}) {info}", flush=True)


def fuzz_conv():
    cin, cout = 8 * rng.randint(1, 40), 4 * rng.randint(1, 70)
    k = rng.choice([1, 3]); stride = rng.choice([1, 1, 2]); dil = rng.choice([1, 1, 2, 4, 6]) if k == 3 else 1
    f, h, w = rng.randint(1, 5), rng.randint(3, 20), rng.randint(3, 20)
    bias = rng.random() < 0.5
    info = f"conv cin={cin} cout={cout} k={k} s={stride} d={dil} f={f} h={h} w={w} bias={bias}"
    globals()["LAST"] = info
    conv = nn.Conv2d(cin, cout, k, stride=stride, padding=dil if k == 3 else 0, dilation=dil, bias=bias)
    x = torch.randn(f, cin, h, w, requires_grad=True)
    y = conv(x); g = torch.randn_like(y); (y * g).sum().backward()
    convg = nn.Conv2d(cin, cout, k, stride=stride, padding=dil if k == 3 else 0, dilation=dil, bias=bias).cuda()
    convg.load_state_dict(conv.state_dict())
    lin, lout = H.Layout.dense(cin), H.Layout.dense(cout)
    xt = H.pad_cols(H.to_tokens(x.detach().cuda()), lin.width).requires_grad_(True)
    with amp():
        yt, ho, wo = H.conv_tokens(xt, convg, f, h, w)
    check("conv/y", rel(H.from_tokens(yt, f, ho, wo)[:, :cout], y), 1e-3, info)
    (yt.float() * H.pad_cols(H.to_tokens(g.cuda()), lout.width)).sum().backward()
    check("conv/dx", rel(H.from_tokens(xt.grad, f, h, w)[:, :cin], x.grad), 2e-3, info)
    check("conv/dw", rel(convg.weight.grad, conv.weight.grad), 2e-3, info)
    if bias:
        check("conv/db", rel(convg.bias.grad, conv.bias.grad), 2e-3, info)


def fuzz_bn():
    c = 8 * rng.randint(1, 64); groups = rng.choice([1, 1, 2, 4]); clips = rng.randint(1, 3)
    f = groups * clips; h, w = rng.randint(2, 12), rng.randint(2, 12)
    relu, res, training, il = rng.random() < 0.6, rng.random() < 0.4, rng.random() < 0.8, rng.random() < 0.5 and groups > 1
    info = f"bn c={c} groups={groups} f={f} h={h} w={w} relu={relu} res={res} train={training} il={il}"
    bn = nn.BatchNorm2d(c); bn.weight.data = 1 + 0.2 * torch.randn(c); bn.bias.data = 0.2 * torch.randn(c)
    bn.running_mean.data = 0.1 * torch.randn(c); bn.running_var.data = 0.5 + torch.rand(c)
    bng = nn.BatchNorm2d(c).cuda(); bng.load_state_dict(bn.state_dict()); bn.train(training); bng.train(training)
    x = (torch.randn(f, c, h, w) * 2 + 1).requires_grad_(True); r = torch.randn(f, c, h, w, requires_grad=True); g = torch.randn(f, c, h, w)
    # group of frame i: contiguous -> i // clips ; interleaved (clip-major clips of `groups` frames) -> i % groups
    gid = [(i % groups) if il else (i // clips) for i in range(f)]
    ys = [None] * f
    for gi in range(groups):                       # sequential per-group calls, group 0 first (base18.py:86-89)
        idx = [i for i in range(f) if gid[i] == gi]
        out = bn(x[idx])
        for j, i in enumerate(idx):
            ys[i] = out[j:j + 1]
    y = torch.cat(ys, 0)
    if res: y = y + r
    if relu: y = F.relu(y)
    (y * g).sum().backward()
    lay = H.Layout.dense(c)
    xt = H.pad_cols(H.to_tokens(x.detach().cuda()).contiguous(), lay.width).requires_grad_(True)
    rt = H.pad_cols(H.to_tokens(r.detach().cuda()).contiguous(), lay.width).requires_grad_(True) if res else None
    with amp():
        yt = H.batchnorm_tokens(xt, bng, relu=relu, resid=rt, groups=groups, il_frames=f if il else 0)
    check("bn/y", rel(H.from_tokens(yt, f, h, w)[:, :c], y), 1e-3, info)
    (yt.float() * H.pad_cols(H.to_tokens(g.cuda()), lay.width)).sum().backward()
    check("bn/dx", rel(H.from_tokens(xt.grad, f, h, w)[:, :c], x.grad), 3e-3, info)
    check("bn/dgamma", rel(bng.weight.grad, bn.weight.grad), 3e-3, info)
    check("bn/dbeta", rel(bng.bias.grad, bn.bias.grad), 3e-3, info)
    if res: check("bn/dres", rel(H.from_tokens(rt.grad, f, h, w)[:, :c], r.grad), 3e-3, info)
    if training:
        check("bn/rm", rel(bng.running_mean, bn.running_mean), 1e-3, info); check("bn/rv", rel(bng.running_var, bn.running_var), 1e-3, info)


def fuzz_swin_block():
    ws = rng.choice([4, 8]); heads = 4; dim = heads * rng.choice([32, 64, 128] if ws == 8 else [32, 64, 128])   # (head dims the attention kernels are built for)
    res = (ws * rng.randint(1, 3), ws * rng.randint(1, 4)); shift = rng.choice([0, ws // 2]); B = rng.randint(1, 3)
    info = f"swin dim={dim} res={res} ws={ws} shift={shift} B={B}"
    torch.manual_seed(rng.randint(0, 1 << 30))
    blk = S.SwinTransformerBlock(dim, res, heads, window_size=ws, shift_size=shift)
    sd = {k: v.clone() for k, v in blk.state_dict().items()}
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith("attn_mask"):
            sd[k] = v + 0.05 * torch.randn_like(v)
    blk.load_state_dict(sd)
    L = res[0] * res[1]
    x = torch.randn(B, 2, L, dim); g = torch.randn(B, 2, L, dim)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and not k.endswith("attn_mask")}
    sdo = dict(sd); sdo.update(params)
    xo = x.clone().requires_grad_(True)
    yo = O.swin_block(xo, sdo, "", res, heads, ws, shift); (yo * g).sum().backward()
    blk = blk.cuda(); xg = x.cuda().requires_grad_(True)
    with amp():
        y = blk(xg)
    (y.float() * g.cuda()).sum().backward()
    check("swin/y", rel(y, yo), 1e-3, info); check("swin/dx", rel(xg.grad, xo.grad), 2e-3, info)
    for k, p in blk.named_parameters():
        check("swin/d" + k, rel(p.grad, params[k].grad), 3e-3, info)


def fuzz_patch_merge():
    dim = 32 * rng.randint(1, 4); res = (2 * rng.randint(1, 8), 2 * rng.randint(1, 8)); B = rng.randint(1, 3)      # (2 dim / 4 dim: whole 64-wide K tiles)
    info = f"merge dim={dim} res={res} B={B}"
    pm = S.PatchMerging(res, dim)
    sd = {k: v.clone() + 0.05 * torch.randn_like(v) for k, v in pm.state_dict().items()}
    pm.load_state_dict(sd)
    x = torch.randn(B, 4, res[0] * res[1], dim)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xo = x.clone().requires_grad_(True)
    yo = O.patch_merging(xo, params, "", res); g = torch.randn_like(yo); (yo * g).sum().backward()
    pm = pm.cuda(); xg = x.cuda().requires_grad_(True)
    with amp():
        y = pm(xg)
    (y.float() * g.cuda()).sum().backward()
    check("merge/y", rel(y, yo), 1e-3, info); check("merge/dx", rel(xg.grad, xo.grad), 2e-3, info)
    for k, p in pm.named_parameters():
        check("merge/d" + k, rel(p.grad, params[k].grad), 3e-3, info)


def fuzz_pool_bilinear():
    f, c, h, w = rng.randint(1, 4), 8 * rng.randint(1, 16), rng.randint(3, 30), rng.randint(3, 30)
    info = f"pool f={f} c={c} h={h} w={w}"
    x = torch.randn(f, c, h, w, requires_grad=True)
    y = F.max_pool2d(x, 3, 2, 1); g = torch.randn_like(y); (y * g).sum().backward()
    xt = H.to_tokens(x.detach().cuda()).contiguous().requires_grad_(True)
    yt = H.MaxPoolTokFn.apply(xt, (f, h, w)); ho, wo = y.shape[2:]
    check("maxpool/y", rel(H.from_tokens(yt, f, ho, wo), y), 1e-6, info)
    (yt.float() * H.to_tokens(g.cuda())).sum().backward()
    check("maxpool/dx", rel(H.from_tokens(xt.grad, f, h, w), x.grad), 1e-6, info)
    H2, W2 = rng.randint(2, 40), rng.randint(2, 40)
    x2 = torch.randn(f, c, h, w, requires_grad=True)
    y2 = F.interpolate(x2, (H2, W2), mode="bilinear", align_corners=False); g2 = torch.randn_like(y2); (y2 * g2).sum().backward()
    xt2 = H.to_tokens(x2.detach().cuda()).contiguous().requires_grad_(True)
    yt2 = H.BilinearTokFn.apply(xt2, (f, h, w, H2, W2))
    check("bilinear/y", rel(H.from_tokens(yt2, f, H2, W2), y2), 1e-5, info + f" -> {H2}x{W2}")
    (yt2.float() * H.to_tokens(g2.cuda())).sum().backward()
    check("bilinear/dx", rel(H.from_tokens(xt2.grad, f, h, w), x2.grad), 1e-5, info + f" -> {H2}x{W2}")


def fuzz_ohem():
    b, nc, h, w = rng.randint(1, 3), rng.choice([2, 5, 12, 21]), rng.randint(4, 64), rng.randint(4, 64)
    n_min = max(1, (b * h * w) // rng.choice([2, 16, 64])); thresh = rng.choice([0.3, 0.7, 0.9])
    info = f"ohem b={b} nc={nc} h={h} w={w} n_min={n_min} thresh={thresh}"
    logits = (torch.randn(b, nc, h, w) * rng.choice([0.5, 2.0, 6.0])).requires_grad_(True)
    labels = torch.randint(0, nc, (b, h, w))
    if rng.random() < 0.5:
        labels[torch.rand(b, h, w) < 0.2] = -1
    lo = O.ohem_ce(logits, labels, n_min, thresh); lo.backward()
    lg = logits.detach().cuda().requires_grad_(True)
    out = OhemCELoss2D(n_min, thresh)(lg, labels.cuda()); out.backward()
    check("ohem/loss", abs(float(out) - float(lo)) / (abs(float(lo)) + 1e-9), 1e-4, info)
    check("ohem/dlogits", rel(lg.grad, logits.grad), 1e-3, info)


def fuzz_regression_loss():
    nb, c, h, w = rng.randint(1, 3), rng.choice([32, 64, 128, 256]), rng.randint(2, 20), rng.randint(2, 20)
    ncls = rng.choice([2, 5, 12])
    info = f"regloss n={nb} c={c} h={h} w={w} classes={ncls}"
    feats = [F.normalize(torch.randn(nb, c, h, w), dim=1) for _ in range(6)]
    labs = [torch.randint(0, ncls, (nb, 1, h, w)).float() for _ in range(6)]
    q = feats[0].clone().requires_grad_(True)
    lo = O.regression_loss(q, *feats[1:], *labs, ncls); lo.backward()
    qg = feats[0].clone().cuda().requires_grad_(True)
    lg = P.regression_loss(qg, *[f_.cuda() for f_ in feats[1:]], *[l.cuda() for l in labs], ncls); lg.backward()
    check("regloss/loss", abs(float(lg) - float(lo)) / (abs(float(lo)) + 1e-9), 1e-4, info)
    check("regloss/dq", rel(qg.grad, q.grad), 2e-3, info)


def fuzz_optim():
    shapes = [tuple(rng.randint(1, 70) for _ in range(rng.randint(1, 3))) for _ in range(rng.randint(1, 60))]
    kind = rng.choice(["adam", "sgd"]); wd = rng.choice([0.0, 1e-4]); steps = rng.randint(1, 3)
    info = f"optim {kind} tensors={len(shapes)} wd={wd} steps={steps}"
    ref = [torch.randn(s) for s in shapes]
    pa = [nn.Parameter(t.clone().cuda()) for t in ref]; pb = [nn.Parameter(t.clone().cuda()) for t in ref]
    if kind == "adam":
        oa, ob = FusedAdam(pa, 1e-2, weight_decay=wd), torch.optim.Adam(pb, 1e-2, weight_decay=wd)
    else:
        oa, ob = FusedSGD(pa, 1e-2, momentum=0.9, weight_decay=wd), torch.optim.SGD(pb, 1e-2, momentum=0.9, weight_decay=wd)
    for _ in range(steps):
        gs = [torch.randn(s).cuda() for s in shapes]
        for p, q_, g in zip(pa, pb, gs):
            p.grad, q_.grad = g.clone(), g.clone()
        oa.step(); ob.step()
    worst = max(rel(p, q_) for p, q_ in zip(pa, pb))
    check("optim", worst, 1e-5, info)


def fuzz_bank():
    from stswincl_amd import hip
    C = rng.choice([64, 128, 256]); maps = rng.randint(2, 6); qb = rng.choice([rng.randint(20, 300), 63, 64, 256])
    nblk = rng.randint(1, 3)
    if rng.random() < 0.6:
        seg, bb = nblk * qb, qb                    # per-sample blocks: query block i sees bank block i of every map
    else:
        nblk = 1; seg = rng.randint(qb, 3000); bb = seg      # one query block sees the whole segment (the library's other legal form)
    M = nblk * qb
    sets = rng.choice([1, 2]) if maps >= 3 else 1
    gmap = [[m for m in range(maps) if m != s_] for s_ in range(sets)] if sets == 2 else [list(range(maps))]
    Mq = M * sets
    info = f"bank C={C} maps={maps} q_block={qb} blocks={nblk} seg={seg} bank_block={bb} sets={sets}"
    globals()["LAST"] = info
    q = F.normalize(torch.randn(Mq, C), dim=1); bank = F.normalize(torch.randn(maps, seg, C), dim=2)
    lq = torch.randint(0, 12, (Mq,), dtype=torch.int32); lb = torch.randint(0, 12, (maps, seg), dtype=torch.int32)
    pos, tot, rmax, lse = hip.contrast_bank_fwd(q.cuda(), lq.cuda(), bank.cuda(), lb.cuda(), q_sets=len(gmap), q_block=qb, bank_block=bb,
                                                gmap=gmap, inv_tau=5.0, want_lse=True)
    rp, rn, rm, rl = O.bank_scores(q, lq.long(), bank, lb.long(), gmap, qb, bb, inv_tau=5.0)
    scale = float((rp + rn).abs().max()) + 1e-6
    check("bank/pos", float((pos.cpu() - rp).abs().max()) / scale, 5e-5, info)
    check("bank/neg", float(((tot - pos).cpu() - rn).abs().max()) / scale, 5e-5, info)
    check("bank/max", float((rmax.cpu() - rm).abs().max()), 1e-4, info)
    check("bank/lse", float((lse.cpu() - rl).abs().max()), 2e-4, info)


def fuzz_argmax():
    from stswincl_amd.utils import EndoMetric as E
    f, nc, h, w = rng.randint(1, 3), rng.choice([2, 8, 12]), rng.randint(4, 70), rng.randint(4, 90)
    Hh, Ww = rng.randint(h, 4 * h), rng.randint(w, 4 * w)
    info = f"argmax f={f} nc={nc} {h}x{w} -> {Hh}x{Ww}"
    globals()["LAST"] = info
    logits = torch.randn(f, nc, h, w) * 3
    gt = torch.randint(0, nc, (f, Hh, Ww))
    ref = torch.argmax(F.interpolate(logits, (Hh, Ww), mode="bilinear", align_corners=True), dim=1)
    labels, dices, ious = E.predict_and_score(logits.cuda(), (Hh, Ww), gt.cuda())
    check("argmax/labels", float((labels.cpu().long() != ref).float().mean()), 5e-4, info)


def fuzz_conv_bn_stats():
    """bf16, >= 8192 output rows: statistics from the GEMM epilogue against the colstats pass (both HIP)."""
    cin, cout = 64 * rng.randint(1, 8), 64 * rng.randint(1, 8); k = rng.choice([1, 3]); dil = rng.choice([1, 2, 4]) if k == 3 else 1
    G = rng.choice([1, 2, 4]); side = rng.choice([16, 32, 48, 64]); f = G * rng.randint(1, 4)
    while f * side * side < 8192: f += G
    il = rng.random() < 0.5 and G > 1
    info = f"conv+bn stats cin={cin} cout={cout} k={k} d={dil} G={G} f={f} side={side} il={il}"
    globals()["LAST"] = info
    conv = nn.Conv2d(cin, cout, k, padding=dil if k == 3 else 0, dilation=dil, bias=rng.random() < 0.3).cuda()
    bns = [nn.BatchNorm2d(cout).cuda() for _ in range(2)]
    x = torch.randn(f * side * side, cin, device="cuda").bfloat16()
    outs = []
    for bn, fused in zip(bns, (True, False)):
        H._FUSED_BN_STATS = fused
        with torch.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():
            y, ho, wo, tab = H.conv_tokens(x, conv, f, side, side, stats=True)
            if fused and (f * side * side) % 256 == 0 and tab is None:
                check("stats/eligible", 1.0, 0.5, info)
            outs.append(H.batchnorm_tokens(y, bn, relu=True, groups=G, il_frames=f if il else 0, stats=tab))
    H._FUSED_BN_STATS = True
    check("stats/y", rel(outs[0], outs[1]), 8e-3, info)
    check("stats/rv", rel(bns[0].running_var, bns[1].running_var), 3e-3, info)
    check("stats/rm", float((bns[0].running_mean - bns[1].running_mean).abs().max()) / float(bns[1].running_var.sqrt().max()), 2e-3, info)


def fuzz_tswinplus():
    from stswincl_amd.net.Ours.base18 import TswinPlus
    B = rng.randint(2, 4); h, w = 64 * rng.randint(1, 2), 64 * rng.randint(1, 3)
    info = f"tswinplus B={B} {h}x{w}"
    globals()["LAST"] = info
    m = TswinPlus(12, (h // 8, w // 8))
    sd = {k_: v.clone() for k_, v in m.state_dict().items()}
    x = torch.randn(B, 4, 3, h, w)
    with torch.no_grad():
        ref = O.tswin_plus(x, sd, training=True)
    y = m.cuda().train()(x.cuda())
    check("tswinplus/y", rel(y, ref), 1e-3, info)


def fuzz_window_attention():
    """WindowAttention at the production head shapes (ws 8 / C 512 and ws 4 / C 1024, 4 heads) and the reduced ones, random window
    counts, with and without an SW-MSA style mask, against the oracle; fp32 or bf16 (FUZZ_BF16)."""
    ws, C = rng.choice([(8, 512), (4, 1024), (8, 128), (4, 256), (8, 256), (4, 512), (4, 128)])
    nW = rng.choice([1, 2, 4, 6]); groups = rng.randint(1, 3); B_ = nW * groups
    masked = rng.random() < 0.5 and nW > 1
    info = f"winattn ws={ws} C={C} nW={nW} B_={B_} masked={masked}"
    globals()["LAST"] = info
    att = S.WindowAttention(C, (ws, ws), 4)
    sd = {k_: (v + 0.02 * torch.randn_like(v) if v.is_floating_point() else v) for k_, v in att.state_dict().items()}
    att.load_state_dict(sd)
    N = ws * ws
    x = torch.randn(B_, 2, N, C) * 0.5; g = torch.randn(B_, 2, N, C)
    mask = None
    if masked:
        mask = torch.zeros(nW, N, N)
        for wi in range(nW):
            ids = torch.randint(0, 3, (N,))
            mask[wi] = (ids[:, None] != ids[None, :]).float() * -100.0
    params = {k_: v.clone().requires_grad_(True) for k_, v in sd.items() if v.is_floating_point()}
    sdo = dict(sd); sdo.update(params)
    xo = x.clone().requires_grad_(True)
    yo = O.window_attention(xo, sdo, "", 4, ws, mask); (yo * g).sum().backward()
    att = att.cuda(); xg = x.cuda().requires_grad_(True)
    with amp():
        y = att(xg, mask.cuda() if mask is not None else None)
    (y.float() * g.cuda()).sum().backward()
    tol = 0.04 if BF16 else 1e-3
    check("winattn/y", rel(y, yo), tol, info); check("winattn/dx", rel(xg.grad, xo.grad), 2 * tol, info)
    for k_, p_ in att.named_parameters():
        check("winattn/d" + k_, rel(p_.grad, params[k_].grad), 3 * tol, info)


def fuzz_ohem_edges():
    b, nc, h, w = 2, 12, 16, 16
    kind = rng.choice(["all_ignored", "one_valid", "all_below", "all_above", "n_min_ge_count"])
    info = f"ohem edge {kind}"
    globals()["LAST"] = info
    logits = torch.randn(b, nc, h, w).requires_grad_(True)
    labels = torch.randint(0, nc, (b, h, w)); n_min, thresh = b * h * w // 16, 0.7
    if kind == "all_ignored": labels[:] = -1
    if kind == "one_valid": labels[:] = -1; labels[0, 3, 4] = 5
    if kind == "all_below":
        logits = (F.one_hot(labels, nc).permute(0, 3, 1, 2).float() * 30).requires_grad_(True)
    if kind == "all_above": thresh = 0.01
    if kind == "n_min_ge_count": n_min = b * h * w - 1
    lo = O.ohem_ce(logits, labels, n_min, thresh); lo.backward()
    lg = logits.detach().cuda().requires_grad_(True)
    out = OhemCELoss2D(n_min, thresh)(lg, labels.cuda()); out.backward()
    same_nan = (float(lo) != float(lo)) and (float(out) != float(out))
    if not same_nan:
        check("ohemedge/loss", abs(float(out) - float(lo)) / (abs(float(lo)) + 1e-6), 1e-4, info + f" ref={float(lo)} got={float(out)}")
        check("ohemedge/dlogits", float((lg.grad.cpu() - logits.grad).abs().max()), 1e-5, info)


def fuzz_optim_groups():
    """Parameter groups with their own lr / weight decay, frozen parameters (grad None) and repeated steps."""
    kind = rng.choice(["adam", "sgd"])
    shapes = [tuple(rng.randint(1, 40) for _ in range(rng.randint(1, 3))) for _ in range(rng.randint(2, 30))]
    ref = [torch.randn(s_) for s_ in shapes]
    pa = [nn.Parameter(t.clone().cuda()) for t in ref]; pb = [nn.Parameter(t.clone().cuda()) for t in ref]
    cut = rng.randint(1, len(shapes) - 1)
    ga = [{"params": pa[:cut], "lr": 1e-2, "weight_decay": 1e-4}, {"params": pa[cut:], "lr": 3e-3, "weight_decay": 0.0}]
    gb = [{"params": pb[:cut], "lr": 1e-2, "weight_decay": 1e-4}, {"params": pb[cut:], "lr": 3e-3, "weight_decay": 0.0}]
    info = f"optim groups {kind} tensors={len(shapes)} cut={cut}"
    globals()["LAST"] = info
    if kind == "adam":
        oa, ob = FusedAdam(ga, 1e-3), torch.optim.Adam(gb, 1e-3)
    else:
        oa, ob = FusedSGD(ga, 1e-3, momentum=0.9), torch.optim.SGD(gb, 1e-3, momentum=0.9)
    for step in range(3):
        for i, (p_, q_) in enumerate(zip(pa, pb)):
            if rng.random() < 0.2:
                p_.grad = q_.grad = None
            else:
                gten = torch.randn(shapes[i]).cuda(); p_.grad, q_.grad = gten.clone(), gten.clone()
        oa.step(); ob.step()
    check("optimgroups", max(rel(p_, q_) for p_, q_ in zip(pa, pb)), 1e-5, info)


def fuzz_consistency():
    """Whole ConsistencyLoss step (2 query + 6 key passes, EMA, both regression_loss terms) against the oracle in fp32: random
    clip counts and (non-square) frame sizes, batched and sequential views."""
    import types
    B = rng.randint(2, 3); h, w = 64 * rng.randint(1, 2), 64 * rng.randint(1, 2)
    seq = rng.random() < 0.3
    os.environ["STSWIN_SEQUENTIAL_VIEWS"] = "1" if seq else "0"
    info = f"consistency B={B} {h}x{w} sequential={seq}"
    globals()["LAST"] = info
    args = types.SimpleNamespace(pixpro_p=1.0, pixpro_momentum=0.99, pixpro_clamp_value=0.0, pixpro_transform_layer=1,
                                 pixpro_ins_loss_weight=0.0, pixpro_pos_ratio=0.7, data="endo18", tag="1", pretrainpth="none",
                                 num_instances=2235, batch_size=B, epochs=150, start_epoch=1)
    net = P.ConsistencyLoss(args, input_resolution=(h // 8, w // 8))
    sd = {"pixpro." + k_: v.clone() for k_, v in net.pixpro.state_dict().items()}
    pkeys = [k_ for k_, _ in net.pixpro.named_parameters()]
    ims = [torch.randn(B, 4, 3, h, w) for _ in range(6)]
    masks = [torch.randint(0, 12, (B, 1, h // 8, w // 8)).float().repeat_interleave(8, 2).repeat_interleave(8, 3) for _ in range(6)]
    with torch.no_grad():
        lo, _ = O.consistency_loss(ims, masks, {k_: v.clone() for k_, v in sd.items()}, pkeys, net.pixpro.k, net.pixpro.K)
    net = net.cuda().train()
    with torch.no_grad():
        lg = net(*[i.cuda() for i in ims], *[m_.cuda() for m_ in masks])
    check("consistency/loss", abs(float(lg) - float(lo)) / abs(float(lo)), 2e-3, info + f" ref={float(lo):.5f} got={float(lg):.5f}")


def fuzz_resnet_feeder():
    """The dedicated kernels of the ResNet18 feeder (conv_halo.hip, stem_s2d, bn_relu_pool; bf16 only) through the module-level
    entry points, on random geometries that the dispatch accepts - and neighbours it must turn away - against torch fp32 on the CPU."""
    which = rng.choice(["c64", "c64", "stem", "tail", "tail32"])
    tol = 0.03
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=which != "tail32"):
        if which == "c64":
            w = rng.choice([16, 32, 64, 128, 24, 48])
            h = rng.choice([q for q in range(2, 70) if (q * w) % 128 == 0] or [8])
            f = rng.randint(1, 6)
            info = f"feeder c64 f={f} h={h} w={w}"
            globals()["LAST"] = info
            conv = nn.Conv2d(64, 64, 3, padding=1, bias=False)
            x = torch.randn(f, 64, h, w, requires_grad=True)
            y = conv(x); g = torch.randn_like(y); (y * g).sum().backward()
            convg = nn.Conv2d(64, 64, 3, padding=1, bias=False).cuda(); convg.load_state_dict(conv.state_dict())
            xt = H.to_tokens(x.detach().cuda()).requires_grad_(True)
            yt, ho, wo = H.conv_tokens(xt, convg, f, h, w)
            check("feeder/c64 y", rel(H.from_tokens(yt, f, ho, wo), y), tol, info)
            (yt.float() * H.to_tokens(g.cuda())).sum().backward()
            check("feeder/c64 dx", rel(H.from_tokens(xt.grad, f, h, w), x.grad), tol, info)
            check("feeder/c64 dw", rel(convg.weight.grad, conv.weight.grad), tol, info)
        elif which == "stem":
            hh, ww = rng.choice([(32, 256), (64, 255), (40, 512), (33, 100), (18, 258), (64, 64)])
            f = rng.randint(1, 4)
            info = f"feeder stem f={f} h={hh} w={ww}"
            globals()["LAST"] = info
            conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
            img = torch.randn(f, 3, hh, ww)
            y = conv(img); g = torch.randn_like(y); (y * g).sum().backward()
            convg = nn.Conv2d(3, 64, 7, 2, 3, bias=False).cuda(); convg.load_state_dict(conv.state_dict())
            yt = H.StemConvFn.apply(img.cuda(), convg.weight, torch.bfloat16)
            check("feeder/stem y", rel(H.from_tokens(yt, f, *y.shape[2:]), y), tol, info)
            (yt.float() * H.to_tokens(g.cuda())).sum().backward()
            check("feeder/stem dw", rel(convg.weight.grad, conv.weight.grad), tol, info)
        else:
            f, hh, ww = rng.randint(1, 6), rng.randint(3, 40), rng.randint(3, 40)
            groups = rng.choice([1, 1, f]) if f > 1 else 1
            info = f"feeder tail f={f} h={hh} w={ww} groups={groups}"
            globals()["LAST"] = info
            bn = nn.BatchNorm2d(64)
            with torch.no_grad():
                bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
            x = torch.randn(f, 64, hh, ww, requires_grad=True)
            if groups == 1:
                yb = bn(x)
            else:                                   # one statistic group per frame
                yb = torch.cat([F.batch_norm(x[i:i + 1], None, None, bn.weight, bn.bias, True, 0.1, bn.eps) for i in range(f)])
            y = F.max_pool2d(F.relu(yb), 3, 2, 1)
            g = torch.randn_like(y); (y * g).sum().backward()
            bng = nn.BatchNorm2d(64).cuda(); bng.load_state_dict(bn.state_dict())
            xt = H.to_tokens(x.detach().cuda()).requires_grad_(True)
            yt = H.batchnorm_relu_maxpool_tokens(xt, bng, (f, hh, ww), groups=groups)
            # bf16: rounding creates ties the fp32 reference does not have, and a flipped arg-max moves a whole gradient value (the fused
            # path is bitwise the two-operator path, tests/test_hip_head.py); fp32: the same taps as torch
            t_y, t_g = (tol, 0.15) if which == "tail" else (1e-5, 2e-4)
            check("feeder/tail y", rel(H.from_tokens(yt, f, *y.shape[2:]), y), t_y, info + " " + which)
            (yt.float() * H.to_tokens(g.cuda())).sum().backward()
            check("feeder/tail dx", rel(H.from_tokens(xt.grad, f, hh, ww), x.grad), t_g, info + " " + which)
            check("feeder/tail dgamma", rel(bng.weight.grad, bn.weight.grad), t_g, info + " " + which)


FAMILIES = [fuzz_consistency, fuzz_window_attention, fuzz_ohem_edges, fuzz_optim_groups, fuzz_conv, fuzz_bn, fuzz_swin_block, fuzz_patch_merge, fuzz_pool_bilinear, fuzz_ohem, fuzz_regression_loss, fuzz_optim,
            fuzz_bank, fuzz_argmax, fuzz_conv_bn_stats, fuzz_resnet_feeder, fuzz_tswinplus]
only = os.environ.get("FUZZ_ONLY")
skip = [t for t in os.environ.get("FUZZ_SKIP", "").split(",") if t]
for fam in FAMILIES:
    if (only and only not in fam.__name__) or any(t in fam.__name__ for t in skip):
        continue
    for i in range(n):
        torch.manual_seed(rng.randint(0, 1 << 30))
        try:
            fam()
        except Exception as e:                      # noqa: BLE001  (a crash is a finding too)
            bad += 1
            import traceback
            tb = traceback.extract_tb(e.__traceback__)
            where = "; ".join(f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line}" for fr in tb[-2:])
            print(f"EXCEPTION in {fam.__name__}: {type(e).__name__}: {str(e)[:200]} @ {where} [{globals().get('LAST', '')}]", flush=True)
    print(f"{fam.__name__}: {n} cases done, {bad} findings so far", flush=True)
sys.exit(1 if bad else 0)
