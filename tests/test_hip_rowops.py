"""a1-a4 indexing (bit-exact), LayerNorm, column sums: HIP vs the CPU oracle / golden vectors."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_util as gu
from oracle import stswin_oracle as O
from stswincl_amd import hip

pytestmark = pytest.mark.gpu
DT = [torch.float32, torch.bfloat16]


@pytest.mark.parametrize("tag", ["s1_64x64", "s2_32x32", "s1_64x80", "s1_32x56", "s1_64x64_noshift", "s2_16x16"])
def test_win_rowmap_matches_reference_golden(tag):
    g = gu.load("index_maps.npz")
    b, t, h, w, ws, shift = [int(v) for v in g[tag + "_cfg"]]
    m = hip.win_rowmap(b, t, h, w, ws, shift).cpu().numpy()
    assert np.array_equal(m, g[tag])


def test_win_rowmap_frame_window_of_a_4_frame_clip():
    b, h, w, ws, shift = 2, 16, 16, 4, 2
    m = hip.win_rowmap(b, 2, h, w, ws, shift, f0=1, frames_total=4).cpu().numpy()
    ids = torch.arange(b * 4 * h * w).reshape(b, 4, h * w, 1)
    exp = O.pair_window_gather(ids[:, 1:3], h, w, ws, shift).reshape(-1).numpy()
    assert np.array_equal(m, exp)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cfg", [(2, 2, 16, 16, 4, 2, 64), (1, 2, 64, 64, 8, 4, 32), (2, 2, 32, 56, 8, 4, 16),
                                 (1, 2, 8, 8, 8, 0, 64), (4, 2, 64, 64, 8, 4, 512)])
def test_win_gather_scatter_bit_exact(dtype, cfg):
    b, t, h, w, ws, shift, c = cfg
    torch.manual_seed(0)
    x = torch.randn(b, t, h * w, c).to(dtype)
    exp = O.pair_window_gather(x.float(), h, w, ws, shift).reshape(-1, c)
    got = hip.win_move(x.cuda(), b, t, h, w, ws, shift, 0)
    assert torch.equal(got.float().cpu(), exp)
    back = hip.win_move(got, b, t, h, w, ws, shift, 1)
    assert torch.equal(back.cpu().reshape(b, t, h * w, c), x)
    exp_back = O.pair_window_scatter(exp.reshape(-1, t, ws * ws, c), h, w, ws, shift)
    assert torch.equal(back.float().cpu().reshape(b, t, h * w, c), exp_back)


def test_merge_and_conv_rowmaps():
    frames, h, w = 3, 8, 12
    m = hip.merge_rowmap(frames, h, w).cpu()
    ids = torch.arange(frames * h * w, dtype=torch.float32).reshape(1, frames, h * w, 1)
    exp = O.patch_merge_gather(ids, h, w).reshape(-1, 4).t().contiguous().to(torch.int32)
    assert torch.equal(m, exp)
    for dil in (1, 2, 6):
        cm = hip.conv3x3_rowmap(frames, h, w, dil).cpu()
        img = (torch.arange(frames * h * w, dtype=torch.float32) + 1).reshape(frames, 1, h, w)
        cols = F.unfold(img, 3, dilation=dil, padding=dil).reshape(frames, 9, h * w)  # 0 where padded
        exp = (cols.permute(1, 0, 2).reshape(9, -1) - 1).to(torch.int32)
        assert torch.equal(cm, exp), dil


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("m,c", [(37, 512), (64, 1024), (9, 2048), (130, 128), (5, 64), (4100, 512)])   # M >= 4096: replica workspace + fold
def test_layernorm_fwd_bwd(dtype, m, c):
    torch.manual_seed(1)
    x = (torch.randn(m, c) * 2 + 0.5).to(dtype)
    g = 1 + 0.1 * torch.randn(c)
    b = 0.1 * torch.randn(c)
    dy = torch.randn(m, c).to(dtype)
    xr = x.float().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.layer_norm(xr, (c,), gr, br)
    (yr * dy.float()).sum().backward()
    y, mean, rstd = hip.layernorm_fwd(x.cuda(), g.cuda(), b.cuda(), M=m)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert torch.allclose(y.float().cpu(), yr.detach(), atol=tol * 4, rtol=tol)
    dg = torch.zeros(c, device="cuda")
    db = torch.zeros(c, device="cuda")
    dx = hip.layernorm_bwd(dy.cuda(), x.cuda(), g.cuda(), mean, rstd, dg, db, M=m)
    assert torch.allclose(dx.float().cpu(), xr.grad, atol=tol * 8, rtol=tol * 2)
    assert torch.allclose(dg.cpu(), gr.grad, atol=2e-4 * m if dtype == torch.float32 else 0.05 * m ** 0.5, rtol=1e-3)
    assert torch.allclose(db.cpu(), br.grad, atol=2e-4 * m if dtype == torch.float32 else 0.05 * m ** 0.5, rtol=1e-3)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("m,c", [(16384, 1024), (4100, 1536), (8192, 512), (300, 2048), (16390, 256)])
@pytest.mark.parametrize("acc,dxs", [(False, True), (True, True), (True, False)])
def test_layernorm_bwd_accumulate_and_column_sums(dtype, m, c, acc, dxs):
    """The template switches of the backward (dx += / column sums of the written dx = the bias gradient of the Linear that produced x)
    in both workgroup geometries (8 waves where four would leave <= 512 workgroups, e.g. 16384 x 1024 and 8192 x 512) and with a
    ragged last piece (C = 1536: three of four 16-byte pieces per lane) - against torch on the GPU in fp32."""
    torch.manual_seed(m + c)
    x = (torch.randn(m, c, device="cuda") * 2 + 0.5).to(dtype)
    g = (1 + 0.1 * torch.randn(c, device="cuda"))
    b = 0.1 * torch.randn(c, device="cuda")
    dy = torch.randn(m, c, device="cuda").to(dtype)
    old = torch.randn(m, c, device="cuda").to(dtype)
    xr = x.float().requires_grad_(True)
    gr = g.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    (F.layer_norm(xr, (c,), gr, br) * dy.float()).sum().backward()
    _, mean, rstd = hip.layernorm_fwd(x, g, b, M=m)
    dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
    cs = torch.zeros(c, device="cuda") if dxs else None
    dx = old.clone() if acc else torch.empty_like(x)
    hip.layernorm_bwd(dy, x, g, mean, rstd, dg, db, M=m, dx=dx, accumulate=acc, dxsum=cs)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    want = xr.grad + (old.float() if acc else 0)
    assert torch.allclose(dx.float(), want, atol=tol * 8, rtol=tol * 2)
    lim = 2e-4 * m if dtype == torch.float32 else 0.05 * m ** 0.5
    assert torch.allclose(dg, gr.grad, atol=lim, rtol=1e-3) and torch.allclose(db, br.grad, atol=lim, rtol=1e-3)
    if dxs:
        assert torch.allclose(cs, dx.float().sum(0), atol=lim, rtol=2e-3)      # sums of the values as written (rounded to the output type)
    if acc:      # the same sum with the accumulated-into tensor as a separate read-only operand (stswin_layernorm_bwd_add): same bits
        dg2, db2 = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
        cs2 = torch.zeros(c, device="cuda") if dxs else None
        keep = old.clone()
        dx2 = hip.layernorm_bwd(dy, x, g, mean, rstd, dg2, db2, M=m, add=keep, dxsum=cs2)
        assert torch.equal(dx2, dx) and torch.equal(keep, old) and torch.equal(dg2, dg) and torch.equal(db2, db)
        assert cs2 is None or torch.equal(cs2, cs)


@pytest.mark.parametrize("dtype", DT)
def test_layernorm_with_patch_merge_gather(dtype):
    torch.manual_seed(2)
    frames, h, w, c = 4, 8, 8, 64
    x = torch.randn(1, frames, h * w, c).to(dtype)
    g, b = 1 + 0.1 * torch.randn(4 * c), 0.1 * torch.randn(4 * c)
    rows = hip.merge_rowmap(frames, h, w)
    mm = frames * h * w // 4
    y, mean, rstd = hip.layernorm_fwd(x.reshape(-1, c).cuda(), g.cuda(), b.cuda(), M=mm, rows=rows, S=4, Cseg=c)
    xr = x.float().requires_grad_(True)
    exp = F.layer_norm(O.patch_merge_gather(xr, h, w).reshape(mm, 4 * c), (4 * c,), g, b)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert torch.allclose(y.float().cpu(), exp.detach(), atol=4 * tol, rtol=tol)
    dy = torch.randn(mm, 4 * c).to(dtype)
    (exp * dy.float()).sum().backward()
    dg, db = torch.zeros(4 * c, device="cuda"), torch.zeros(4 * c, device="cuda")
    dx = torch.zeros(frames * h * w, c, dtype=dtype, device="cuda")
    hip.layernorm_bwd(dy.cuda(), x.reshape(-1, c).cuda(), g.cuda(), mean, rstd, dg, db, M=mm, rows=rows, S=4, Cseg=c, dx=dx)
    assert torch.allclose(dx.float().cpu().reshape(xr.shape), xr.grad, atol=8 * tol, rtol=2 * tol)


@pytest.mark.parametrize("dtype", DT)
def test_colsum(dtype):
    torch.manual_seed(3)
    y = torch.randn(1000, 264).to(dtype)
    out = torch.zeros(264, device="cuda")
    hip.colsum(y.cuda(), out)
    assert torch.allclose(out.cpu(), y.float().sum(0), atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("N,heads,nW", [(64, 4, 1), (64, 16, 5), (16, 4, 3)])
def test_relative_position_bias_expand_and_scatter(N, heads, nW):
    """stswin_bias_expand / stswin_bias_scatter against the indexing of swin_512.py:122-131 and its autograd backward."""
    torch.manual_seed(N + heads)
    ws = int(N ** 0.5)
    tsz = (2 * ws - 1) ** 2
    table = torch.randn(tsz, heads, requires_grad=True)
    index = torch.randint(0, tsz, (N * N,))
    mask = torch.where(torch.rand(nW, N, N) < 0.3, torch.full((nW, N, N), -100.0), torch.zeros(nW, N, N)) if nW > 1 else None
    ref = table[index].reshape(N, N, heads).permute(2, 0, 1)                  # [h][query i][key j]
    ref_full = ref.unsqueeze(0) + mask.unsqueeze(1) if mask is not None else ref.unsqueeze(0)
    out = hip.bias_expand(table.detach().cuda(), index.cuda(), mask.cuda() if mask is not None else None, N, heads)
    out = out.reshape(nW, heads, N, N)
    assert torch.equal(out.cpu(), ref_full.detach().transpose(2, 3).contiguous()), "[slot][h][key][query]"
    g = torch.randn(heads, N, N)                                              # d(loss)/d(bias)[h][key j][query i]
    (ref * g.transpose(1, 2)).sum().backward()
    dtable = torch.zeros(tsz, heads, device="cuda")
    hip.bias_scatter(g.cuda(), index.cuda(), dtable, N, heads)
    assert torch.allclose(dtable.cpu(), table.grad, atol=1e-4, rtol=1e-5)


def test_vec_gather_pads_and_unpads_channel_vectors():
    """stswin_vec_gather through Layout.pad_vec / unpad_vec: the 400-channel decode concat inside 448 columns."""
    from stswincl_amd import headops as H
    lay = H.Layout.concat([H.Layout.dense(48), H.Layout.dense(48), H.Layout.dense(48), H.Layout.dense(256)])
    v = torch.randn(lay.logical, device="cuda")
    p = lay.pad_vec(v, 1.5)
    assert p.shape == (lay.width,)
    ref = torch.full((lay.width,), 1.5)
    for a, n, b in lay.segs:
        ref[b:b + n] = v.cpu()[a:a + n]
    assert torch.equal(p.cpu(), ref)
    assert torch.equal(lay.unpad_vec(p).cpu(), v.cpu())


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("n,k", [(512, 2048), (132, 68), (64, 64), (4, 260)])
def test_linear_pack_cast_and_transpose(dt, n, k):
    """stswin_linear_pack: W and W^T of an nn.Linear weight in the compute dtype (tile edges, both dtypes)."""
    w = torch.randn(n, k, device="cuda")
    fwd, tr = hip.linear_pack(w, dt)
    assert torch.equal(fwd, w.to(dt)) and torch.equal(tr, w.t().to(dt).contiguous())
    fwd2, none = hip.linear_pack(w, dt, want_tr=False)
    assert none is None and torch.equal(fwd2, fwd)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("frames,H,W", [(2, 37, 53), (3, 64, 96), (1, 128, 258)])
def test_stem_im2col_matches_unfold(dtype, frames, H, W):
    """torchvision conv1 (7x7 / stride 2 / pad 3, resnet.py:98-102) as im2col: patch column (ky*7 + kx)*3 + c, zero padded to 192,
    against F.unfold; ragged widths (several 40-pixel tiles per row, a partial last one) and odd sizes."""
    import torch.nn.functional as F
    torch.manual_seed(H + W)
    img = torch.randn(frames, 3, H, W)
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    got = hip.stem_im2col(img.cuda(), dtype, Ho, Wo).float().cpu()
    cols = F.unfold(img, 7, padding=3, stride=2)                                  # [F][3*49][Ho*Wo], row index c*49 + tap
    ref = cols.view(frames, 3, 49, Ho * Wo).permute(0, 3, 2, 1).reshape(frames * Ho * Wo, 147)   # column tap*3 + c
    ref = ref.to(dtype).float()
    assert got.shape == (frames * Ho * Wo, 192)
    assert torch.equal(got[:, :147], ref) and float(got[:, 147:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("frames,H,W", [(2, 36, 28), (1, 35, 29), (3, 16, 64)])
def test_stem_space_to_depth_image_and_segments(dtype, frames, H, W):
    """hip.stem_s2d: every record of the padded 2 x 2 space-to-depth image, and the 64-value segments the GEMM row map reads, against a
    direct restatement; the convolution they define equals F.conv2d(7, 2, 3) (resnet.py:98-102) - odd image sizes included."""
    import torch.nn.functional as F
    from stswincl_amd import headops as Hd
    torch.manual_seed(H + W)
    img = torch.randn(frames, 3, H, W)
    A, Hs, Ws = hip.stem_s2d(img.cuda(), dtype)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    assert (Hs, Ws) == (Ho + 3, Wo + 3) and A.shape == (frames * Hs * Ws, 64) and A.stride() == (16, 1)
    rec = A[:, :16].float().cpu().view(frames, Hs, Ws, 16)
    want = torch.zeros(frames, Hs, Ws, 16)
    pad = F.pad(img.to(dtype).float(), (0, 2 * Wo - W, 0, 2 * Ho - H))
    want[:, 2:2 + Ho, 2:2 + Wo, :12] = pad.view(frames, 3, Ho, 2, Wo, 2).permute(0, 2, 4, 3, 5, 1).reshape(frames, Ho, Wo, 12)
    assert torch.equal(rec, want)
    assert torch.equal(A[5, 16:32], A[6, :16])                      # a row is the 4-record segment starting at its record
    w = torch.randn(64, 3, 7, 7) / 10
    rmap = Hd._stem_rowmap(frames, Ho, Wo, Hs, Ws, "cuda")
    wm = Hd._stem_pack(w.cuda(), torch.float32).cpu()
    seg = torch.stack([A.float().cpu()[rmap[s].cpu().long()] for s in range(4)], 1).reshape(frames * Ho * Wo, 256)
    y = seg @ wm.t()
    ref = F.conv2d(img.to(dtype).float(), w, stride=2, padding=3).permute(0, 2, 3, 1).reshape(-1, 64)
    assert float((y - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    g = torch.randn(64, 256)
    assert torch.equal(Hd._stem_unpack(Hd._stem_pack(w, torch.float32)), w)
    assert Hd._stem_unpack(g).shape == (64, 3, 7, 7)
