"""Decode-head kernels (conv as gather GEMM, grouped BatchNorm, bilinear, pooling, logits upsample, OHEM-CE) and the
ASPP module: HIP vs torch fp32 on the CPU / the reference golden.  fp32 path 1e-3 (typically 1e-5), bf16 path 3e-2."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import golden_util as gu
from stswincl_amd import headops as H
from stswincl_amd.net.Ours.ASPP import ASPP

pytestmark = pytest.mark.gpu
MODES = [("fp32", 1e-3), ("bf16", 3e-2)]


def rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def ac(mode):
    return torch.autocast("cuda", dtype=torch.bfloat16, enabled=(mode == "bf16"))


@pytest.mark.parametrize("mode,tol", MODES)
@pytest.mark.parametrize("cin,cout,k,dil,bias,stride", [(128, 64, 3, 1, False, 1), (64, 128, 3, 6, True, 1),
                                                        (128, 48, 1, 1, False, 1), (400, 64, 3, 1, False, 1),
                                                        (64, 12, 1, 1, True, 1), (64, 128, 3, 1, False, 2),
                                                        (64, 128, 1, 1, False, 2), (128, 128, 3, 4, False, 1)])
def test_conv_tokens_fwd_bwd(mode, tol, cin, cout, k, dil, bias, stride):
    torch.manual_seed(cin + cout)
    f, h, w = 2, 12, 10
    conv = nn.Conv2d(cin, cout, k, stride=stride, padding=dil if k == 3 else 0, dilation=dil, bias=bias)
    x = torch.randn(f, cin, h, w, requires_grad=True)
    y = conv(x)
    g = torch.randn_like(y)
    (y * g).sum().backward()
    lin, lout = H.Layout.dense(cin), H.Layout.dense(cout)
    convg = nn.Conv2d(cin, cout, k, stride=stride, padding=dil if k == 3 else 0, dilation=dil, bias=bias).cuda()
    convg.load_state_dict(conv.state_dict())
    xt = H.pad_cols(H.to_tokens(x.detach().cuda()), lin.width).requires_grad_(True)
    with ac(mode):
        yt, ho, wo = H.conv_tokens(xt, convg, f, h, w)
    assert (ho, wo) == tuple(y.shape[2:])
    yl = H.from_tokens(yt, f, ho, wo)[:, :cout]
    assert rel(yl, y) < tol
    if lout.width > cout:
        assert float(yt[:, cout:].abs().max()) == 0.0
    gt = H.pad_cols(H.to_tokens(g.cuda()), lout.width)
    (yt.float() * gt).sum().backward()
    h, w = x.shape[2:]
    assert rel(H.from_tokens(xt.grad, f, h, w)[:, :cin], x.grad) < 2 * tol
    assert rel(convg.weight.grad, conv.weight.grad) < 2 * tol
    if bias:
        assert rel(convg.bias.grad, conv.bias.grad) < 2 * tol


@pytest.mark.parametrize("mode,tol", MODES)
@pytest.mark.parametrize("clips,T,h,w,relu,res", [(3, 4, 8, 8, True, True), (2, 4, 6, 5, True, False), (4, 2, 16, 4, False, False)])
def test_batchnorm_interleaved_groups(mode, tol, clips, T, h, w, relu, res):
    """Statistic group = frame index t of clip-major clips (group g owns frames g, g + T, ...: stswin_* unit_rows) against the
    same data reordered frame-major with contiguous groups - outputs, all gradients and the running statistics."""
    torch.manual_seed(clips * 10 + T)
    c, f = 64, clips * T
    mk = lambda: nn.BatchNorm2d(c).cuda()                                             # noqa: E731
    bn_a, bn_b = mk(), mk()
    bn_a.weight.data = 1 + 0.2 * torch.randn(c, device="cuda"); bn_a.bias.data = 0.2 * torch.randn(c, device="cuda")
    bn_b.load_state_dict(bn_a.state_dict())
    x = (torch.randn(clips, T, h * w, c, device="cuda") * 2 + 1)
    r = torch.randn(clips, T, h * w, c, device="cuda")
    g = torch.randn(clips, T, h * w, c, device="cuda")
    fm = lambda t: t.transpose(0, 1).reshape(f * h * w, c).contiguous()               # noqa: E731  frame-major rows
    xa, ra = x.reshape(-1, c).clone().requires_grad_(True), r.reshape(-1, c).clone().requires_grad_(True)
    xb, rb = fm(x).requires_grad_(True), fm(r).requires_grad_(True)
    with ac(mode):
        ya = H.batchnorm_tokens(xa, bn_a, relu=relu, resid=ra if res else None, groups=T, il_frames=f)
        yb = H.batchnorm_tokens(xb, bn_b, relu=relu, resid=rb if res else None, groups=T)
    back = lambda t: t.view(T, clips, h * w, c).transpose(0, 1).reshape(-1, c)        # noqa: E731  frame-major -> clip-major
    etol = 1e-5 if mode == "fp32" else tol          # same arithmetic, other summation order (chunking follows the unit size)
    assert rel(ya, back(yb)) < etol
    (ya.float() * g.reshape(-1, c)).sum().backward()
    (yb.float() * fm(g)).sum().backward()
    assert rel(xa.grad, back(xb.grad)) < 3 * etol and rel(bn_a.weight.grad, bn_b.weight.grad) < 3 * etol
    assert rel(bn_a.bias.grad, bn_b.bias.grad) < 3 * etol
    if res:
        assert rel(ra.grad, back(rb.grad)) < 3 * etol
    assert rel(bn_a.running_mean, bn_b.running_mean) < 1e-5 and rel(bn_a.running_var, bn_b.running_var) < 1e-5


@pytest.mark.parametrize("mode,tol", MODES)
@pytest.mark.parametrize("groups,relu,res,training", [(1, True, False, True), (4, True, True, True), (1, False, False, True),
                                                      (1, True, False, False)])
def test_batchnorm_tokens(mode, tol, groups, relu, res, training):
    torch.manual_seed(groups)
    f, c, h, w = 4, 64, 6, 5
    bn = nn.BatchNorm2d(c)
    bn.weight.data = 1 + 0.2 * torch.randn(c)
    bn.bias.data = 0.2 * torch.randn(c)
    bn.running_mean.data = 0.1 * torch.randn(c)
    bn.running_var.data = 0.5 + torch.rand(c)
    bng = nn.BatchNorm2d(c).cuda()
    bng.load_state_dict(bn.state_dict())
    bn.train(training)
    bng.train(training)
    x = (torch.randn(f, c, h, w) * 2 + 3).requires_grad_(True)
    r = torch.randn(f, c, h, w, requires_grad=True)
    g = torch.randn(f, c, h, w)
    if groups == 1:
        y = bn(x)
    else:   # one BN call per frame, sequentially (base18.py:86-89)
        y = torch.cat([bn(x[i:i + 1]) for i in range(f)], 0)
    if res:
        y = y + r
    if relu:
        y = F.relu(y)
    (y * g).sum().backward()
    xt = H.to_tokens(x.detach().cuda()).contiguous().requires_grad_(True)
    rt = H.to_tokens(r.detach().cuda()).contiguous().requires_grad_(True) if res else None
    with ac(mode):
        yt = H.batchnorm_tokens(xt, bng, relu=relu, resid=rt, groups=groups)
    assert rel(H.from_tokens(yt, f, h, w), y) < tol
    (yt.float() * H.to_tokens(g.cuda())).sum().backward()
    assert rel(H.from_tokens(xt.grad, f, h, w), x.grad) < 3 * tol
    assert rel(bng.weight.grad, bn.weight.grad) < 3 * tol and rel(bng.bias.grad, bn.bias.grad) < 3 * tol
    if res:
        assert rel(H.from_tokens(rt.grad, f, h, w), r.grad) < 3 * tol
    if training:
        assert rel(bng.running_mean, bn.running_mean) < tol and rel(bng.running_var, bn.running_var) < tol
        assert int(bng.num_batches_tracked) == int(bn.num_batches_tracked)


@pytest.mark.parametrize("cin,cout,k,dil,stride,side,il", [(64, 64, 3, 1, 1, 32, True), (64, 128, 3, 1, 2, 64, True),
                                                           (64, 128, 1, 1, 2, 64, False), (128, 256, 3, 2, 1, 32, True),
                                                           (256, 512, 3, 4, 1, 32, False), (512, 256, 3, 1, 1, 32, False)])
def test_conv_bn_statistics_from_gemm_epilogue(cin, cout, k, dil, stride, side, il, monkeypatch):
    """conv -> BatchNorm(train) with the statistics taken in the convolution's GEMM epilogue (production path for >= 8192
    output rows, bf16) against the same modules with the colstats pass: outputs, every gradient, running statistics; 4
    statistic groups (contiguous / interleaved).  The fused path must actually have run."""
    from stswincl_amd import hip
    torch.manual_seed(cin + cout + k)
    f, G = 16, 4
    conv = nn.Conv2d(cin, cout, k, stride=stride, padding=dil if k == 3 else 0, dilation=dil, bias=False).cuda()
    bn_a, bn_b = nn.BatchNorm2d(cout).cuda(), nn.BatchNorm2d(cout).cuda()
    bn_a.weight.data = 1 + 0.2 * torch.randn(cout, device="cuda"); bn_a.bias.data = 0.2 * torch.randn(cout, device="cuda")
    bn_b.load_state_dict(bn_a.state_dict())
    x = torch.randn(f * side * side, cin, device="cuda").bfloat16()
    calls = []
    real = hip.bn_table_finalize
    monkeypatch.setattr(hip, "bn_table_finalize", lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])

    def run(bn, fused):
        monkeypatch.setattr(H, "_FUSED_BN_STATS", fused)
        conv.weight.grad = None
        xt = x.clone().requires_grad_(True)
        with ac("bf16"):
            y, ho, wo, tab = H.conv_tokens(xt, conv, f, side, side, stats=True)
            assert (tab is not None) == fused
            z = H.batchnorm_tokens(y, bn, relu=True, groups=G, il_frames=f if il else 0, stats=tab)
        torch.manual_seed(1)
        (z.float() * torch.randn(z.shape, device="cuda")).sum().backward()
        return z, xt.grad, conv.weight.grad.clone(), bn.weight.grad, bn.bias.grad

    outs_a, outs_b = run(bn_a, True), run(bn_b, False)
    assert len(calls) == 1
    for name, a, b in zip(("y", "dx", "dW", "dgamma", "dbeta"), outs_a, outs_b):
        # statistics of the fp32 accumulators vs of the bf16-rounded outputs: differences of rounding size only
        assert rel(a, b) < 1.2e-2, name
    # (means are ~0 against a standard deviation of ~1: an absolute bound in units of the deviation)
    mclose = lambda a, b, v: float((a - b).abs().max()) < 1e-3 * float(v.sqrt().max())          # noqa: E731
    assert mclose(bn_a.running_mean, bn_b.running_mean, bn_b.running_var) and rel(bn_a.running_var, bn_b.running_var) < 2e-3
    # and against fp32 torch on the same bf16 inputs
    xi = H.from_tokens(x.float(), f, side, side)
    yr = F.conv2d(xi, conv.weight.float(), None, stride, dil if k == 3 else 0, dil)
    fr = (lambda t: t.view(f // G, G, *t.shape[1:]).transpose(0, 1).reshape(t.shape)) if il else (lambda t: t)
    yg = fr(yr).view(G, f // G, *yr.shape[1:])
    mean, var = yg.mean((1, 3, 4)), yg.var((1, 3, 4), unbiased=False)
    rm = torch.zeros(cout, device="cuda"); rv = torch.ones(cout, device="cuda")
    for g in range(G):
        n = yg[g].numel() // cout
        rm = 0.9 * rm + 0.1 * mean[g]; rv = 0.9 * rv + 0.1 * var[g] * n / (n - 1)
    assert mclose(bn_a.running_mean, rm, rv) and rel(bn_a.running_var, rv) < 2e-3


@pytest.mark.parametrize("mode,tol", MODES)
@pytest.mark.parametrize("h,w,H_,W_", [(8, 8, 16, 16), (6, 10, 12, 20), (4, 4, 32, 32), (5, 7, 13, 9)])
def test_bilinear_tokens(mode, tol, h, w, H_, W_):
    torch.manual_seed(h * w)
    f, c = 2, 64
    x = torch.randn(f, c, h, w, requires_grad=True)
    g = torch.randn(f, c, H_, W_)
    y = F.interpolate(x, size=(H_, W_), mode="bilinear", align_corners=False)
    (y * g).sum().backward()
    xt = H.to_tokens(x.detach().cuda()).contiguous().requires_grad_(True)
    with ac(mode):
        yt = H.BilinearTokFn.apply(xt, (f, h, w, H_, W_))
    assert rel(H.from_tokens(yt, f, H_, W_), y) < tol
    (yt.float() * H.to_tokens(g.cuda())).sum().backward()
    assert rel(H.from_tokens(xt.grad, f, h, w), x.grad) < 2 * tol


@pytest.mark.parametrize("mode,tol", MODES)
def test_pool_broadcast_logits_up(mode, tol):
    torch.manual_seed(9)
    f, c, h, w = 3, 64, 5, 6
    x = torch.randn(f, c, h, w, requires_grad=True)
    p = F.adaptive_avg_pool2d(x, 1)
    up = F.interpolate(p, size=(h, w), mode="bilinear", align_corners=False)
    g = torch.randn(f, c, h, w)
    (up * g).sum().backward()
    xt = H.to_tokens(x.detach().cuda()).contiguous().requires_grad_(True)
    with ac(mode):
        pt = H.AvgPoolTokFn.apply(xt, f)
        ut = H.BroadcastTokFn.apply(pt, h * w)
    assert rel(pt, p.reshape(f, c)) < tol and rel(H.from_tokens(ut, f, h, w), up) < tol
    (ut.float() * H.to_tokens(g.cuda())).sum().backward()
    assert rel(H.from_tokens(xt.grad, f, h, w), x.grad) < 2 * tol
    # logits upsample x8 to NCHW
    nc = 12
    lt = torch.randn(f, nc, h, w, requires_grad=True)
    gl = torch.randn(f, nc, 8 * h, 8 * w)
    yl = F.interpolate(lt, (8 * h, 8 * w), mode="bilinear", align_corners=False)
    (yl * gl).sum().backward()
    tok = H.pad_cols(H.to_tokens(lt.detach().cuda()), 64).requires_grad_(True)
    with ac(mode):
        out = H.LogitsUpFn.apply(tok, (f, h, w, 8 * h, 8 * w, nc))
    assert out.shape == yl.shape and rel(out, yl) < tol
    (out.float() * gl.cuda()).sum().backward()
    assert rel(H.from_tokens(tok.grad, f, h, w)[:, :nc], lt.grad) < 2 * tol


@pytest.mark.parametrize("mode,tol", MODES)
def test_stem_and_maxpool(mode, tol):
    torch.manual_seed(4)
    f, hh, ww = 3, 36, 28
    conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
    img = torch.randn(f, 3, hh, ww)
    y = F.relu(conv(img))
    p = F.max_pool2d(y, 3, 2, 1)
    g = torch.randn_like(p)
    (p * g).sum().backward()
    convg = nn.Conv2d(3, 64, 7, 2, 3, bias=False).cuda()
    convg.load_state_dict(conv.state_dict())
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    yt = torch.relu(H.StemConvFn.apply(img.cuda(), convg.weight, dt))
    ho, wo = y.shape[2:]
    assert rel(H.from_tokens(yt, f, ho, wo), y) < tol
    pt = H.MaxPoolTokFn.apply(yt, (f, ho, wo))
    assert rel(H.from_tokens(pt, f, *p.shape[2:]), p) < tol
    (pt.float() * H.to_tokens(g.cuda())).sum().backward()
    assert rel(convg.weight.grad, conv.weight.grad) < 3 * tol


@pytest.mark.parametrize("mode,tol", [("fp32", 1e-3), ("bf16", 4e-2)])
def test_aspp_matches_reference_golden(mode, tol):
    g = gu.load("aspp.npz")
    net = ASPP(num_classes=256)
    net.load_state_dict(gu.det_fill(gu.skeleton_sd(g["keys"], g["shapes"], g["dtypes"])))
    net = net.cuda()
    x = torch.from_numpy(g["x"]).cuda()
    net.train()
    with ac(mode):
        y = net(x)
    assert y.shape == (2, 256, 8, 8)
    assert rel(y, g["y_train"]) < tol
    assert rel(net.bn_conv_3x3_2.running_mean, g["rm_after"]) < tol
    net.eval()
    with ac(mode), torch.no_grad():
        assert rel(net(x), g["y_eval"]) < tol
    net.train()
    with pytest.raises(ValueError):
        net(x[:1])


@pytest.mark.parametrize("n,n_min,kind", [(1 << 20, 1 << 16, "random"), (1 << 20, 1 << 16, "ties"), (1 << 20, 700000, "zeros"),
                                          (4099, 256, "random"), (4099, 4099, "random"), (1 << 18, 1, "random")])
def test_ohem_select_matches_sorted_reference(n, n_min, kind):
    """stswin_ohem_select (radix select over the loss bits) against the reference's sort (losses.py:35-39): both branches,
    ties at the cut, a cut inside the ignored (zero) losses, n_min = 1 and n_min = n."""
    from stswincl_amd import hip
    g = torch.Generator().manual_seed(n_min + len(kind))
    loss = torch.rand(n, generator=g) * 3
    if kind == "ties":
        loss = (loss * 8).round() / 8                      # ~24 distinct values: the k-th value is shared by thousands
    if kind == "zeros":
        loss[torch.rand(n, generator=g) < 0.5] = 0.0       # ignore_index pixels: more than n - n_min zeros
    loss = loss.cuda()
    srt = torch.sort(loss.double(), descending=True)[0]
    for thresh in (0.5, 2.9, 10.0):
        n_hard = (loss > thresh).sum().float()
        # stats layout of stswin_ce_fwd: [0] count, [2..3] the sum as one unsigned 64-bit integer in 2^-32 fixed point
        stats = torch.zeros(4, device="cuda")
        stats[0] = n_hard
        stats[2:4].view(torch.int64)[0] = int(round(float(loss[loss > thresh].double().sum()) * 4294967296.0))
        value, sel = hip.ohem_select(loss, stats, n_min, thresh)
        if int(n_hard) > n_min:                            # loss[n_min] > thresh
            ref = srt[srt > thresh].mean()
            assert float(sel[0]) == pytest.approx(thresh) and float(sel[2]) == 0.0
            assert float(sel[1]) == pytest.approx(1.0 / int(n_hard), rel=1e-6)
        else:
            ref = srt[:n_min].mean()
            assert float(sel[0]) == float(srt[n_min - 1]), "k-th largest loss (bit exact)"
            assert float(sel[2]) == 1.0 and float(sel[1]) == pytest.approx(1.0 / n_min, rel=1e-6)
        assert float(value) == pytest.approx(float(ref), rel=2e-6, abs=1e-7)


def test_resnet_statistics_from_epilogues_no_worse_than_colstats_path(monkeypatch):
    """ResNet18-OS8 in train mode, bf16: every conv -> BatchNorm pair (stem included) takes its statistics from the GEMM epilogue.
    Two bf16 runs whose statistics differ in the last bit drift apart through 17 layers, so the yardstick is the fp32 path of the
    same weights: the epilogue-statistics run must be as close to it as the colstats run (outputs, running statistics, weight
    gradients)."""
    from stswincl_amd.net.Ours.resnet import ResNet18_OS8
    torch.manual_seed(0)
    nets = [ResNet18_OS8().cuda() for _ in range(3)]
    for n in nets[1:]:
        n.load_state_dict(nets[0].state_dict())
    x = torch.randn(2, 4, 3, 256, 256, device="cuda")

    def run(net, mode, fused):
        monkeypatch.setattr(H, "_FUSED_BN_STATS", fused)
        with ac(mode):
            tok, h, w = net.forward_frames(x)
        torch.manual_seed(1)
        (tok.float() * torch.randn(tok.shape, device="cuda")).sum().backward()
        return tok

    calls = []
    from stswincl_amd import hip
    real = hip.bn_table_finalize
    monkeypatch.setattr(hip, "bn_table_finalize", lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
    ta = run(nets[0], "bf16", True)
    n_fused = len(calls)
    tb, tc = run(nets[1], "bf16", False), run(nets[2], "fp32", False)
    assert n_fused == 20 and len(calls) == n_fused          # stem + 16 block + 3 downsample convolutions (>= 8192 rows each)
    ea, eb = rel(ta, tc), rel(tb, tc)
    assert ea < 1.25 * eb + 2e-3 and ea < 6e-2, (ea, eb)
    sd = [n.state_dict() for n in nets]
    for k in sd[0]:
        if k.endswith("running_var") or k.endswith("running_mean"):
            scale = float(sd[2][k.replace("mean", "var")].sqrt().max()) if k.endswith("mean") else float(sd[2][k].abs().max())
            da, db = float((sd[0][k] - sd[2][k]).abs().max()) / scale, float((sd[1][k] - sd[2][k]).abs().max()) / scale
            assert da < 1.5 * db + 2e-3, (k, da, db)
    g = [dict(n.named_parameters()) for n in nets]
    for k in ("resnet.0.weight", "resnet.4.0.conv1.weight", "layer5.1.conv2.weight"):
        ga, gb = rel(g[0][k].grad, g[2][k].grad), rel(g[1][k].grad, g[2][k].grad)
        assert ga < 1.5 * gb + 1e-2, (k, ga, gb)


@pytest.mark.parametrize("mode,train", [("fp32", False), ("bf16", False), ("bf16", True)])
def test_resnet_residual_gradient_link_equals_autograd_sum(mode, train, monkeypatch):
    """The two consumers of a block input (conv1 + shortcut: identity residual of bn2 in 6 blocks, downsample convolution in 2) join
    their gradients in the epilogue of an input-gradient GEMM (GradLink) instead of an autograd add: every parameter gradient
    against the plain autograd graph of the same weights.  With eval-mode BatchNorm the chain is deterministic and benign: 2e-5 in
    fp32, 2e-2 in bf16 (sum rounded once instead of twice).  In train mode 17 untrained BatchNorm layers on 128-sample groups
    amplify any last-bit difference (2-3e-3 run to run even in fp32, 0.2 for a bf16 rounding): only a sanity bound there - a lost
    contribution would be an error of order 1 on every layer below it."""
    from stswincl_amd.net.Ours.resnet import ResNet18_OS8
    torch.manual_seed(0)
    net = ResNet18_OS8().cuda().train(train)
    x0 = torch.randn(4, 3, 64, 64, device="cuda")
    gout = None

    def run(link):
        nonlocal gout
        monkeypatch.setattr(H, "_RESID_GRAD_LINK", link)
        net.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        img = x * 1.0                                       # (the stem has no input gradient: give the first block one through a scale)
        with ac(mode):
            tok, h, w = net.forward_tokens(img, groups=2)
        if gout is None:
            gout = torch.randn(tok.shape, device="cuda")
        (tok.float() * gout).sum().backward()
        return {k: p.grad.clone() for k, p in net.named_parameters()}

    ga, gb, gc = run(True), run(False), run(False)
    noise = max(rel(gc[k], gb[k]) for k in ga)
    worst = max(rel(ga[k], gb[k]) for k in ga)
    if not train:
        assert noise < 1e-5 and worst < (2e-5 if mode == "fp32" else 2e-2), (worst, noise)
    else:
        assert sorted(rel(ga[k], gb[k]) for k in ga)[len(ga) // 2] < 0.5, worst


@pytest.mark.parametrize("il", [True, False])
def test_conv_bn_epilogue_statistics_with_24_groups(il, monkeypatch):
    """24 statistic groups (6 batched key views x 4 frames of the contrastive step): the one-workgroup-per-group form of
    stswin_bn_table_finalize (+ the running-statistic kernel) against the colstats path - outputs and the sequentially updated
    running statistics."""
    from stswincl_amd import hip
    torch.manual_seed(5)
    f, G, side, c = 48, 24, 32, 64
    conv = nn.Conv2d(c, c, 3, 1, 1, bias=False).cuda()
    bn_a, bn_b = nn.BatchNorm2d(c).cuda(), nn.BatchNorm2d(c).cuda()
    x = torch.randn(f * side * side, c, device="cuda").bfloat16()
    calls = []
    real = hip.bn_table_finalize
    monkeypatch.setattr(hip, "bn_table_finalize", lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
    outs = []
    for bn, fused in ((bn_a, True), (bn_b, False)):
        monkeypatch.setattr(H, "_FUSED_BN_STATS", fused)
        with ac("bf16"), torch.no_grad():
            y, ho, wo, tab = H.conv_tokens(x, conv, f, side, side, stats=True)
            outs.append(H.batchnorm_tokens(y, bn, relu=True, groups=G, il_frames=f if il else 0, stats=tab))
    assert len(calls) == 1
    assert rel(outs[0], outs[1]) < 6e-3
    assert float((bn_a.running_mean - bn_b.running_mean).abs().max()) < 1e-3 * float(bn_b.running_var.sqrt().max())
    assert rel(bn_a.running_var, bn_b.running_var) < 2e-3


def test_single_frame_nchw_input_to_tokens():
    """One NCHW frame with a 64-multiple channel count: permute + reshape is a strided VIEW there, not a copy - to_tokens must still
    hand the kernels row-major tokens (ASPP on a single feature map, found by tests/fuzz/fuzz_ops.py)."""
    torch.manual_seed(0)
    x = torch.randn(1, 128, 5, 7, device="cuda")
    t = H.to_tokens(x)
    assert t.is_contiguous() and torch.equal(t, x[0].permute(1, 2, 0).reshape(35, 128))
    conv = nn.Conv2d(128, 64, 3, padding=2, dilation=2).cuda()
    y, ho, wo = H.conv_tokens(t, conv, 1, 5, 7)
    assert rel(H.from_tokens(y, 1, ho, wo), conv(x)) < 1e-3
    net = ASPP(num_classes=256).cuda().eval()
    fm = torch.randn(1, 1024, 6, 6, device="cuda")
    with torch.no_grad():
        out = net(fm)
    assert out.shape == (1, 256, 6, 6) and torch.isfinite(out).all()


@pytest.mark.parametrize("dil,hw", [(18, 32), (12, 16), (6, 8)])
def test_dilated_conv_tap_skipping_is_bitwise_neutral(dil, hw):
    """STSWIN_GF_TAPSKIP (set by ConvTokFn for 2 * dilation >= H + 3): the tiled gather GEMM skips the taps that are padding for every
    row of a tile - a tap it skips only ever multiplied zeros, so forward and input gradient are bitwise those of the all-taps run."""
    from stswincl_amd import hip
    torch.manual_seed(dil)
    f, cin, cout = 4, 128, 64
    M = f * hw * hw
    x = torch.randn(M, cin, device="cuda").to(torch.bfloat16)
    wm = (torch.randn(cout, 9 * cin, device="cuda") / (9 * cin) ** 0.5).to(torch.bfloat16)
    rmap = hip.conv_rowmap(f, hw, hw, hw, hw, 3, 1, dil, dil, False, "cuda")
    assert float((rmap >= 0).float().mean()) < 0.75
    ya, yb = torch.empty(M, cout, dtype=torch.bfloat16, device="cuda"), torch.empty(M, cout, dtype=torch.bfloat16, device="cuda")
    hip.gemm_nt(x, wm, ya, M=M, a_rows=rmap, S=9)
    hip.gemm_nt(x, wm, yb, M=M, a_rows=rmap, S=9, flags=hip.GF_TAPSKIP)
    assert torch.equal(ya, yb)
    ref = torch.zeros(M, cout)
    xf, wf = x.float().cpu(), wm.float().cpu().view(cout, 9, cin)
    rm = rmap.cpu().long()
    for t in range(9):
        ok = rm[t] >= 0
        ref[ok] += xf[rm[t][ok]] @ wf[:, t].t()
    assert float((yb.float().cpu() - ref).abs().max()) <= 1.5e-2 * float(ref.abs().max())


@pytest.mark.parametrize("f,h,w", [(2, 128, 128), (3, 64, 64), (5, 32, 32), (16, 16, 16), (1, 24, 32), (300, 16, 16)])
@pytest.mark.parametrize("sign", [1, -1])
def test_halo_conv3x3_c64_matches_gather_gemm_and_fp64(f, h, w, sign):
    """stswin_conv3x3_c64 (resnet18.layer1 convolutions, resnet.py:31-51 via :104-105) against (a) an fp64 convolution of the same
    bf16 operands - within bf16 rounding of the result - and (b) the gather GEMM it replaces, forward and input gradient, with the
    residual operand and the BatchNorm statistics table."""
    from stswincl_amd import hip
    torch.manual_seed(f * 1000 + h + w + sign)
    M = f * h * w
    x = torch.randn(M, 64, device="cuda").to(torch.bfloat16)
    wt = torch.randn(64, 64, 3, 3, device="cuda") / 24.0
    ident = torch.arange(64, dtype=torch.int32, device="cuda")
    fwd, dg = hip.conv_pack(wt, torch.bfloat16, ident, ident)
    res = torch.randn(M, 64, device="cuda").to(torch.bfloat16)
    mat = fwd if sign > 0 else dg
    rmap = hip.conv_rowmap(f, h, w, h, w, 3, 1, 1, 1, sign < 0, "cuda")
    tab_a, tab_b = hip.stats_table(M, 64, "cuda"), hip.stats_table(M, 64, "cuda")
    ya, yb = torch.empty_like(x), torch.empty_like(x)
    hip.gemm_nt(x, mat, ya, M=M, a_rows=rmap, S=9, resid=res, flags=hip.GF_RESID, stats_out=tab_a)
    hip.conv3x3_c64(x, mat, yb, f, h, w, sign, resid=res, stats_out=tab_b)
    torch.cuda.synchronize()
    # fp64 reference from the bf16-rounded operands
    xi = x.double().view(f, h, w, 64).permute(0, 3, 1, 2)
    wb = wt.to(torch.bfloat16).double()
    if sign > 0:
        ref = F.conv2d(xi, wb, padding=1)
    else:
        ref = F.conv_transpose2d(xi, wb, padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(M, 64) + res.double()
    scale = float(ref.abs().max())
    assert float((yb.double() - ref).abs().max()) <= 2.0 ** -8 * scale * 1.01 + 1e-6
    assert float((ya.double() - ref).abs().max()) <= 2.0 ** -8 * scale * 1.01 + 1e-6
    # the two kernels: same fp32 sums up to summation order -> at most one bf16 step apart, almost everywhere equal
    d = (ya.float() - yb.float()).abs()
    assert float(d.max()) <= 2.0 ** -7 * scale
    assert float((d > 0).float().mean()) < 0.02
    # statistics tables: sums over 128-row blocks of the STORED values
    nb = (M + 127) // 128
    ta, tb = tab_a.view(2, -1, 64)[:, :nb], tab_b.view(2, -1, 64)[:, :nb]
    want = torch.stack([yb.float().view(nb, -1, 64).sum(1), (yb.float() ** 2).view(nb, -1, 64).sum(1)]) if M % 128 == 0 else None
    if want is not None:
        assert float((tb - want).abs().max()) <= 2e-2 * float(want.abs().max())   # (sums of the fp32 values before the bf16 store)
    if nb % 2 == 0:      # (the GEMM's tiles may leave the sum of a 256-row pair in its first row and zero in the second)
        pa, pb = ta.reshape(2, nb // 2, 2, 64).sum(2), tb.reshape(2, nb // 2, 2, 64).sum(2)
        assert float((pa - pb).abs().max()) <= 1e-3 * float(pa.abs().max())
    # plain call: no residual, no table; bitwise reproducible
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    hip.conv3x3_c64(x, mat, y1, f, h, w, sign)
    hip.conv3x3_c64(x, mat, y2, f, h, w, sign)
    assert torch.equal(y1, y2)
    assert float((y1.double() - (ref - res.double())).abs().max()) <= 2.0 ** -8 * scale * 1.01 + 1e-6


def test_halo_conv_rejects_other_geometries_and_is_what_layer1_runs():
    from stswincl_amd import hip
    x = torch.zeros(2 * 20 * 24, 64, dtype=torch.bfloat16, device="cuda")
    wm = torch.zeros(64, 576, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(hip.StswinHipError):
        hip.conv3x3_c64(x, wm, torch.empty_like(x), 2, 20, 24)
    assert not hip.conv3x3_c64_ok(2, 20, 24, 64, 64, 3, 1, 1, 1, torch.bfloat16)
    assert hip.conv3x3_c64_ok(16, 128, 128, 64, 64, 3, 1, 1, 1, torch.bfloat16)
    assert not hip.conv3x3_c64_ok(16, 128, 128, 64, 64, 3, 1, 1, 1, torch.float32)
    # through the module path: forward + backward of a 64 -> 64 convolution under autocast equals the gather path within bf16 rounding
    torch.manual_seed(3)
    conv = nn.Conv2d(64, 64, 3, padding=1, bias=False).cuda()
    xt = torch.randn(2 * 32 * 32, 64, device="cuda", requires_grad=True)
    outs = []
    for flag in (True, False):
        H._HALO_CONV = H._HALO_WGRAD = flag
        try:
            conv.weight.grad = None
            xt.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y, _, _ = H.conv_tokens(xt, conv, 2, 32, 32)
            (y.float() * torch.linspace(-1, 1, 64, device="cuda")).sum().backward()
            outs.append((y.detach().float().clone(), xt.grad.clone(), conv.weight.grad.clone()))
        finally:
            H._HALO_CONV = H._HALO_WGRAD = True
    for a, b in list(zip(*outs))[:2]:
        assert float((a - b).abs().max()) <= 2.0 ** -7 * float(b.abs().max())
    # (weight gradients: fp32 slabs here, bf16 split-K slabs on the gemm_tn path)
    assert float((outs[0][2] - outs[1][2]).abs().max()) <= 5e-3 * float(outs[1][2].abs().max())


@pytest.mark.parametrize("f,h,w", [(2, 128, 128), (16, 128, 128), (3, 64, 64), (5, 32, 32), (1, 12, 32), (40, 8, 32), (7, 64, 128)])
def test_halo_wgrad_c64_matches_fp64_and_gemm_tn(f, h, w):
    """stswin_conv3x3_c64_wgrad against an fp64 weight gradient of the same bf16 operands and against the gemm_tn path it replaces;
    both output layouts, accumulate, and bitwise reproducibility (fixed-order fold of the workgroup slabs)."""
    from stswincl_amd import hip
    torch.manual_seed(f * 100 + h + w)
    M = f * h * w
    x = torch.randn(M, 64, device="cuda").to(torch.bfloat16)
    dy = (torch.randn(M, 64, device="cuda") / 8).to(torch.bfloat16)
    # fp64 reference: dW[co][ci][ky][kx]
    xi = x.double().view(f, h, w, 64).permute(0, 3, 1, 2)
    gi = dy.double().view(f, h, w, 64).permute(0, 3, 1, 2)
    ref = torch.nn.grad.conv2d_weight(xi, (64, 64, 3, 3), gi, padding=1)
    scale = float(ref.abs().max())
    dw_t = torch.empty(64, 64, 3, 3, dtype=torch.float32, device="cuda")
    hip.conv3x3_c64_wgrad(dy, x, dw_t, f, h, w, tapminor=True)
    assert float((dw_t.double() - ref).abs().max()) <= 2e-5 * scale + 1e-6 * M ** 0.5      # fp32 accumulation of exact bf16 products
    dw_g = torch.empty(64, 576, dtype=torch.float32, device="cuda")
    hip.conv3x3_c64_wgrad(dy, x, dw_g, f, h, w, tapminor=False)
    assert torch.equal(dw_g.view(64, 9, 64).permute(0, 2, 1).reshape(64, 64, 3, 3), dw_t)
    # the path it replaces
    fmap = hip.conv_rowmap(f, h, w, h, w, 3, 1, 1, 1, False, "cuda")
    old = torch.empty(64, 576, dtype=torch.float32, device="cuda")
    hip.gemm_tn(dy, x, old, Mk=M, bt_rows=fmap, bseg=64, overwrite=True)
    hip.tn_join()
    torch.cuda.synchronize()
    assert float((old - dw_g).abs().max()) <= 5e-3 * scale           # (its split-K slabs are bf16)
    # accumulate and reproducibility
    again = dw_t.clone()
    hip.conv3x3_c64_wgrad(dy, x, again, f, h, w, tapminor=True, accumulate=True)
    assert torch.equal(again, dw_t + dw_t)
    rep = torch.empty_like(dw_t)
    hip.conv3x3_c64_wgrad(dy, x, rep, f, h, w, tapminor=True)
    assert torch.equal(rep, dw_t)


@pytest.mark.parametrize("f,hh,ww", [(2, 64, 256), (1, 37, 255), (3, 16, 512), (16, 512, 512)])
def test_stem_wgrad_ring_kernel_matches_fp64_and_gemm_tn(f, hh, ww):
    """stswin_stem_wgrad against the fp64 weight gradient of the 7x7 / 2 / 3 convolution on the same bf16 operands, and against the
    gemm_tn path over the row map; bitwise reproducible; accumulate."""
    from stswincl_amd import hip
    torch.manual_seed(f + hh + ww)
    img = torch.randn(f, 3, hh, ww, device="cuda")
    ho, wo = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
    M = f * ho * wo
    dy = (torch.randn(M, 64, device="cuda") / 8).to(torch.bfloat16)
    A, hs, ws = hip.stem_s2d(img, torch.bfloat16)
    assert hip.stem_wgrad_ok(hh, ww, torch.bfloat16)
    dw = torch.empty(64, 256, dtype=torch.float32, device="cuda")
    hip.stem_wgrad(dy, A, dw, f, hh, ww)
    got = H._stem_unpack(dw).double()
    gi = dy.double().view(f, ho, wo, 64).permute(0, 3, 1, 2)
    ref = torch.nn.grad.conv2d_weight(img.to(torch.bfloat16).double(), (64, 3, 7, 7), gi, stride=2, padding=3)
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2e-5 * scale + 1e-6 * M ** 0.5
    # the structural zeros of the packed layout (taps outside the 7x7 window, positions 12..15) stay exactly zero
    full = dw.view(64, 4, 4, 16)
    assert float(full[..., 12:].abs().max()) == 0.0
    rmap = H._stem_rowmap(f, ho, wo, hs, ws, "cuda")
    old = torch.empty_like(dw)
    hip.gemm_tn(dy, A, old, Mk=M, bt_rows=rmap, bseg=64, overwrite=True)
    hip.tn_join()
    assert float((old - dw).abs().max()) <= 5e-3 * scale
    rep = torch.empty_like(dw)
    hip.stem_wgrad(dy, A, rep, f, hh, ww)
    assert torch.equal(rep, dw)
    hip.stem_wgrad(dy, A, rep, f, hh, ww, accumulate=True)
    assert torch.equal(rep, dw + dw)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("f,hh,ww,groups,il,train", [(4, 16, 16, 1, False, True), (8, 32, 16, 4, True, True), (4, 17, 13, 1, False, True),
                                                     (4, 32, 32, 2, False, True), (3, 16, 24, 1, False, False)])
def test_fused_bn_relu_maxpool_equals_the_two_operators(mode, f, hh, ww, groups, il, train):
    """BNReluPoolFn (stswin_bn_relu_pool; the stem tail conv1 -> bn1 -> relu -> maxpool of resnet.py:98-102) against
    BNTokFn + MaxPoolTokFn: pooled values, running statistics and every gradient are bitwise the same (same expressions, same
    rounding points, same summation order); and against torch on the CPU."""
    torch.manual_seed(f * hh + ww)
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    x0 = torch.randn(f * hh * ww, 64, device="cuda")
    res = []
    for fused in (True, False):
        H._FUSED_STEM_TAIL = fused
        try:
            bn = nn.BatchNorm2d(64).cuda()
            with torch.no_grad():
                bn.weight.copy_(torch.linspace(0.5, 1.5, 64))
                bn.bias.copy_(torch.linspace(-0.3, 0.3, 64))
                bn.running_mean.copy_(torch.linspace(-0.1, 0.1, 64))
                bn.running_var.copy_(torch.linspace(0.8, 1.2, 64))
            bn.train(train)
            x = x0.clone().to(dt).requires_grad_(True)
            y = H.batchnorm_relu_maxpool_tokens(x, bn, (f, hh, ww), groups=groups, il_frames=f if il else 0)
            w = torch.linspace(-1, 1, y.numel(), device="cuda").view_as(y)
            (y.float() * w).sum().backward()
            res.append((y.detach().clone(), x.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(), bn.running_mean.clone(),
                        bn.running_var.clone(), int(bn.num_batches_tracked)))
        finally:
            H._FUSED_STEM_TAIL = True
    for a, b in zip(*res):
        assert (a == b) if isinstance(a, int) else torch.equal(a, b)
    if groups == 1 and mode == "fp32":
        bn = nn.BatchNorm2d(64)
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, 64)); bn.bias.copy_(torch.linspace(-0.3, 0.3, 64))
            bn.running_mean.copy_(torch.linspace(-0.1, 0.1, 64)); bn.running_var.copy_(torch.linspace(0.8, 1.2, 64))
        bn.train(train)
        xc = x0.cpu().view(f, hh, ww, 64).permute(0, 3, 1, 2).clone().requires_grad_(True)
        yc = F.max_pool2d(F.relu(bn(xc)), 3, 2, 1)
        wc = torch.linspace(-1, 1, yc.numel()).view(f, yc.shape[2], yc.shape[3], 64).permute(0, 3, 1, 2)
        (yc * wc).sum().backward()
        assert rel(H.from_tokens(res[0][0], f, yc.shape[2], yc.shape[3]), yc) < 1e-5
        assert rel(H.from_tokens(res[0][1], f, hh, ww), xc.grad) < 1e-4
        assert rel(res[0][2], bn.weight.grad) < 1e-4 and rel(res[0][3], bn.bias.grad) < 1e-4


@pytest.mark.parametrize("f,hh,ww", [(2, 64, 256), (1, 37, 255), (3, 16, 512), (16, 512, 512)])
def test_stem_conv_ring_kernel_matches_fp64_and_gemm_nt(f, hh, ww):
    """stswin_stem_conv against the fp64 7x7 / 2 / 3 convolution of the same bf16 operands and against gemm_nt over the row map, with
    the BatchNorm statistics table; bitwise reproducible."""
    from stswincl_amd import hip
    torch.manual_seed(f + hh + ww)
    img = torch.randn(f, 3, hh, ww, device="cuda")
    wt = torch.randn(64, 3, 7, 7, device="cuda") / 12
    ho, wo = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
    M = f * ho * wo
    A, hs, ws = hip.stem_s2d(img, torch.bfloat16)
    wm = H._stem_pack(wt, torch.bfloat16)
    y, y2, y3 = (torch.empty(M, 64, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    tab, tab2 = hip.stats_table(M, 64, "cuda"), hip.stats_table(M, 64, "cuda")
    hip.stem_conv(A, wm, y, f, hh, ww, stats_out=tab)
    ref = F.conv2d(img.to(torch.bfloat16).double(), wt.to(torch.bfloat16).double(), stride=2, padding=3).permute(0, 2, 3, 1).reshape(M, 64)
    scale = float(ref.abs().max())
    assert float((y.double() - ref).abs().max()) <= 2.0 ** -8 * scale * 1.01 + 1e-6
    rmap = H._stem_rowmap(f, ho, wo, hs, ws, "cuda")
    hip.gemm_nt(A, wm, y2, M=M, a_rows=rmap, S=4, stats_out=tab2)
    d = (y.float() - y2.float()).abs()
    assert float(d.max()) <= 2.0 ** -7 * scale and float((d > 0).float().mean()) < 0.02
    # the statistics table is read per FRAME (whole statistic groups): the ring kernel keeps the sums of a run of units in the row of the
    # run's first block and zero rows for the others, the GEMM may keep a 256-row pair in its first row - frame sums agree
    nb = M // 128
    bpf = nb // f
    want = torch.stack([y.float().view(f, -1, 64).sum(1), (y.float() ** 2).view(f, -1, 64).sum(1)])
    got = tab.view(2, -1, 64)[:, :nb].reshape(2, f, bpf, 64).sum(2)
    assert float((got - want).abs().max()) <= 2e-3 * float(want.abs().max())
    ref_t = tab2.view(2, -1, 64)[:, :nb].reshape(2, f, bpf, 64).sum(2)
    assert float((got - ref_t).abs().max()) <= 1e-4 * float(ref_t.abs().max())
    hip.stem_conv(A, wm, y3, f, hh, ww)
    assert torch.equal(y3, y)

