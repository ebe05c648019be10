"""Segmented gather GEMMs (NT with fused epilogues, TN for weight gradients): HIP vs fp32 torch on the CPU.

Tolerances: fp32 path (exact-f32 MFMA) 1e-5 relative; bf16 path compared with an fp32 reference evaluated on
the SAME bf16-rounded operands: 3e-3 relative to the output scale before the final bf16 rounding (+ 1 bf16 ulp)."""
import pytest
import torch
import torch.nn.functional as F

from stswincl_amd import hip

pytestmark = pytest.mark.gpu
DT = [torch.float32, torch.bfloat16]


def _close(got, exp, dtype, what=""):
    got, exp = got.float().cpu(), exp.float()
    scale = float(exp.abs().max()) + 1e-6
    err = float((got - exp).abs().max())
    tol = 2e-5 if dtype == torch.float32 else 1.2e-2
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} at {int((got - exp).abs().argmax())}"


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (256, 384, 512), (200, 136, 128), (37, 48, 192), (130, 12, 256),
                                   (1024, 1536, 512), (700, 64, 192), (512, 40, 64)])
def test_gemm_nt_plain_and_bias(dtype, m, n, k):
    torch.manual_seed(m + n + k)
    a = torch.randn(m, k).to(dtype)
    w = (torch.randn(n, k) / k ** 0.5).to(dtype)
    bias = torch.randn(n)
    out = torch.full((m, n), float("nan"), dtype=dtype, device="cuda")
    hip.gemm_nt(a.cuda(), w.cuda(), out, M=m, bias=bias.cuda())
    _close(out, F.linear(a.float(), w.float(), bias), dtype, "bias")
    out8 = torch.full((m, n), float("nan"), dtype=dtype, device="cuda")
    hip.gemm_nt(a.cuda(), w.cuda(), out8, M=m, bias=bias.cuda(), flags=hip.GF_WAVES4)      # 4-wave variant, same result
    assert torch.equal(out8, out)
    if dtype == torch.bfloat16:
        outb = torch.full((m, n), float("nan"), dtype=dtype, device="cuda")
        hip.gemm_nt(a.cuda(), w.cuda(), outb, M=m, bias=bias.cuda(), flags=hip.GF_BIG)     # 256x256 4-stage-ring kernel
        _close(outb, F.linear(a.float(), w.float(), bias), dtype, "big kernel")
        outb.fill_(float("nan"))
        hip.gemm_nt(a.cuda(), w.cuda(), outb, M=m, bias=bias.cuda(), flags=hip.GF_MID)     # 256x128 3-stage-ring kernel
        _close(outb, F.linear(a.float(), w.float(), bias), dtype, "mid kernel")
        outb.fill_(float("nan"))
        hip.gemm_nt(a.cuda(), w.cuda(), outb, M=m, bias=bias.cuda(), flags=hip.GF_HALF)    # 256x128 ping-pong ring kernel
        _close(outb, F.linear(a.float(), w.float(), bias), dtype, "half ping-pong kernel")
        outb.fill_(float("nan"))
        hip.gemm_nt(a.cuda(), w.cuda(), outb, M=m, bias=bias.cuda(), flags=hip.GF_BIG | hip.GF_NOPIPE)
        _close(outb, F.linear(a.float(), w.float(), bias), dtype, "big kernel, no ping-pong")
    out32 = torch.zeros(m, n, device="cuda")
    hip.gemm_nt(a.cuda(), w.cuda(), out32, M=m, flags=hip.GF_OUT_F32)
    hip.gemm_nt(a.cuda(), w.cuda(), out32, M=m, flags=hip.GF_OUT_F32 | hip.GF_ACCUM)
    _close(out32, 2 * F.linear(a.float(), w.float()), dtype, "f32 accumulate")


@pytest.mark.parametrize("dtype", DT)
def test_gemm_nt_epilogues(dtype):
    torch.manual_seed(5)
    m, n, k = 300, 256, 128
    a = torch.randn(m, k).to(dtype)
    w = (torch.randn(n, k) / k ** 0.5).to(dtype)
    bias = torch.randn(n)
    r = torch.randn(m, n).to(dtype)
    ac, wc, rc, bc = a.cuda(), w.cuda(), r.cuda(), bias.cuda()
    lin = F.linear(a.float(), w.float(), bias)
    # GELU with the pre-activation as second output (Mlp.fc1 + act, swin_512.py:18-19)
    out = torch.empty(m, n, dtype=dtype, device="cuda")
    pre = torch.empty(m, n, dtype=dtype, device="cuda")
    hip.gemm_nt(ac, wc, out, M=m, bias=bc, out2=pre, flags=hip.GF_GELU)
    _close(pre, lin, dtype, "pre-activation")
    _close(out, F.gelu(lin), dtype, "gelu")
    # residual (fc2 + shortcut)
    hip.gemm_nt(ac, wc, out, M=m, bias=bc, resid=rc, flags=hip.GF_RESID)
    _close(out, lin + r.float(), dtype, "resid")
    # gelu' multiply (backward of fc1's activation)
    hip.gemm_nt(ac, wc, out, M=m, resid=rc, flags=hip.GF_MUL_DGELU)
    x = r.float().requires_grad_(True)
    F.gelu(x).sum().backward()
    _close(out, F.linear(a.float(), w.float()) * x.grad, dtype, "dgelu")
    # q scaling of the first columns (swin_512.py:118) + relu
    hip.gemm_nt(ac, wc, out, M=m, bias=bc, scale=0.25, scale_cols=96)
    exp = lin.clone()
    exp[:, :96] *= 0.25
    _close(out, exp, dtype, "scale_cols")
    hip.gemm_nt(ac, wc, out, M=m, bias=bc, flags=hip.GF_RELU)
    _close(out, F.relu(lin), dtype, "relu")


@pytest.mark.parametrize("m,n,k,flags", [(8292, 512, 128, 0), (8292, 512, 128, hip.GF_NOBIG), (8292, 64, 64, 0),
                                         (8192 + 256 + 40, 768, 64, hip.GF_BIG), (8192, 256, 128, hip.GF_MID),
                                         (8320, 512, 64, hip.GF_BIG | hip.GF_STREAM)])
def test_gemm_nt_colsum_partial_table(m, n, k, flags, monkeypatch):
    """Bias-gradient column sums through the per-128-row-block table + stswin_cs_reduce (M >= 8192: default) against the
    fp32 atomics of the same kernel and the sums of the stored values; ragged M (odd / even block counts), every tile
    family (256x256 ring, 128x128, 256x64, 256x128, streaming)."""
    torch.manual_seed(5)
    a = torch.randn(m, k).bfloat16().cuda()
    w = (torch.randn(n, k) / k ** 0.5).bfloat16().cuda()
    out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    seed = torch.randn(n, device="cuda")
    cs_tab = seed.clone()
    assert m >= hip._CS_PARTIAL_MIN_M
    hip._cs_table(a.device, 1).fill_(float("nan"))          # stale table contents must not leak into the sums
    hip.gemm_nt(a, w, out, M=m, flags=flags, colsum_out=cs_tab)
    # the sums are taken in fp32 before the bf16 rounding of the stored values
    ref = seed + F.linear(a.float(), w.float()).sum(0)
    tol = 1e-4 * float(ref.abs().max()) + 1e-2
    assert float((cs_tab - ref).abs().max()) <= tol, "table + reduce"
    monkeypatch.setattr(hip, "_CS_PARTIAL_MIN_M", 1 << 30)
    cs_at = seed.clone()
    hip.gemm_nt(a, w, out, M=m, flags=flags, colsum_out=cs_at)
    assert float((cs_at - ref).abs().max()) <= tol, "atomics"


@pytest.mark.parametrize("m,n,k,flags,bias", [(8192, 512, 128, 0, False), (8192 + 512, 512, 128, hip.GF_NOBIG, False),
                                              (16384, 64, 64, 0, False), (8192 + 256, 768, 64, hip.GF_BIG, True),
                                              (8192, 128, 576, 0, True), (8192, 256, 128, hip.GF_NOBIG, True)])
def test_gemm_nt_statistics_table(m, n, k, flags, bias):
    """GF_CS_SQ: per-128-row-block column sums AND sums of squares of the output (the BatchNorm statistics of a convolution
    output) + stswin_cs_group_reduce over contiguous and interleaved statistic groups, against fp32 torch."""
    torch.manual_seed(7)
    a = torch.randn(m, k).bfloat16().cuda()
    w = (torch.randn(n, k) / k ** 0.5).bfloat16().cuda()
    b = torch.randn(n, device="cuda") if bias else None
    out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    tab = hip.stats_table(m, n, "cuda").fill_(float("nan"))
    hip.gemm_nt(a, w, out, M=m, bias=b, flags=flags, stats_out=tab)
    lin = F.linear(a.float(), w.float(), b)
    assert float((out.float() - lin).abs().max()) < 0.05
    s, ss = hip.cs_group_reduce(tab, m, 1)
    assert float((s[0] - lin.sum(0)).abs().max()) <= 1e-4 * float(lin.sum(0).abs().max()) + 2e-2
    assert float((ss[0] / (lin * lin).sum(0) - 1).abs().max()) < 1e-4
    if m % 1024 == 0:
        G = 4
        s, ss = hip.cs_group_reduce(tab, m, G)                              # contiguous groups
        ref = lin.view(G, m // G, n)
        assert float((s - ref.sum(1)).abs().max()) <= 2e-2 and float((ss / (ref * ref).sum(1) - 1).abs().max()) < 1e-4
        unit = 256
        s, ss = hip.cs_group_reduce(tab, m, G, unit)                        # group g = units g, g + G, ...
        ref = lin.view(m // (G * unit), G, unit, n).transpose(0, 1).reshape(G, -1, n)
        assert float((s - ref.sum(1)).abs().max()) <= 2e-2 and float((ss / (ref * ref).sum(1) - 1).abs().max()) < 1e-4
        rm, rv = torch.zeros(n, device="cuda"), torch.ones(n, device="cuda")
        mean, rstd = hip.bn_table_finalize(tab, m, rm, rv, G, 1e-5, 0.1, unit=unit)        # one launch: + running statistics
        var = ref.var(1, unbiased=False)
        assert float((mean - ref.mean(1)).abs().max()) < 1e-4 and float((rstd * torch.sqrt(var + 1e-5) - 1).abs().max()) < 1e-4
        erm, erv = torch.zeros(n, device="cuda"), torch.ones(n, device="cuda")
        for g in range(G):
            erm = 0.9 * erm + 0.1 * ref[g].mean(0)
            erv = 0.9 * erv + 0.1 * ref[g].var(0, unbiased=True)
        assert float((rm - erm).abs().max()) < 1e-4 and float((rv / erv - 1).abs().max()) < 1e-4
    if m == 8192:                                                           # the cap: 32 groups (24 = 6 key views x 4 frames in the
        G = 32                                                              # contrastive step), one 256-row unit each
        mean, rstd = hip.bn_table_finalize(tab, m, None, None, G, 1e-5, 0.1, unit=256)
        ref = lin.view(G, 256, n)
        assert float((mean - ref.mean(1)).abs().max()) < 1e-4
        assert float((rstd * torch.sqrt(ref.var(1, unbiased=False) + 1e-5) - 1).abs().max()) < 1e-4
    with pytest.raises(RuntimeError):
        hip.cs_group_reduce(tab, m, 3 if m % 3 else 5)
    with pytest.raises(RuntimeError):
        hip.bn_table_finalize(tab, m, None, None, 3 if m % 3 else 5)


@pytest.mark.parametrize("variant", ["ring", "stream", "duo"])
def test_gemm_nt_ring_register_epilogues(variant):
    """256x256 ring kernel (forced; also its persistent streaming variant): every mode-specialised register epilogue
    against fp32 torch, ragged M, row maps."""
    torch.manual_seed(11)
    dtype = torch.bfloat16
    m, n, k = 600, 512, 128
    a = torch.randn(m, k).to(dtype)
    w = (torch.randn(n, k) / k ** 0.5).to(dtype)
    bias = torch.randn(n)
    r = torch.randn(m, n).to(dtype)
    ac, wc, rc, bc = a.cuda(), w.cuda(), r.cuda(), bias.cuda()
    lin0 = F.linear(a.float(), w.float())
    lin = lin0 + bias
    if variant in ("stream", "duo") and not hip.load().stswin_tuning_build():
        pytest.skip("the persistent streaming / two-workgroups-per-CU variants exist in STSWIN_TUNING builds only (the product library ignores their flags)")
    BIG = hip.GF_BIG | (hip.GF_STREAM if variant == "stream" else 0) | (hip.GF_DUO if variant == "duo" else 0)
    out = torch.empty(m, n, dtype=dtype, device="cuda")
    pre = torch.empty(m, n, dtype=dtype, device="cuda")
    hip.gemm_nt(ac, wc, out, M=m, flags=BIG)
    if variant == "duo":
        assert hip.load().stswin_last_variant(0) == hip.VAR_NT_DUO
    _close(out, lin0, dtype, "plain")
    hip.gemm_nt(ac, wc, out, M=m, bias=bc, flags=BIG)
    _close(out, lin, dtype, "bias")
    hip.gemm_nt(ac, wc, out, M=m, bias=bc, out2=pre, flags=BIG | hip.GF_GELU)
    _close(pre, lin, dtype, "pre-activation")
    _close(out, F.gelu(lin), dtype, "gelu")
    x = lin.clone().requires_grad_(True)
    F.gelu(x).sum().backward()
    hip.gemm_nt(ac, wc, out, M=m, bias=bc, out2=pre, flags=BIG | hip.GF_GELU | hip.GF_C2_DGELU)
    _close(pre, x.grad, dtype, "gelu' as second output")
    _close(out, F.gelu(lin), dtype, "gelu (with gelu' output)")
    hip.gemm_nt(ac, wc, out, M=m, bias=bc, resid=rc, flags=BIG | hip.GF_RESID)
    _close(out, lin + r.float(), dtype, "bias + resid")
    hip.gemm_nt(ac, wc, out, M=m, resid=rc, flags=BIG | hip.GF_RESID)
    _close(out, lin0 + r.float(), dtype, "resid")
    xr = r.float().requires_grad_(True)
    F.gelu(xr).sum().backward()
    cs = torch.zeros(n, device="cuda")
    hip.gemm_nt(ac, wc, out, M=m, resid=rc, flags=BIG | hip.GF_MUL_DGELU, colsum_out=cs)
    _close(out, lin0 * xr.grad, dtype, "dgelu")
    ref_cs = out.float().sum(0)
    assert float((cs - ref_cs).abs().max()) <= 2e-3 * float(ref_cs.abs().max()) + 1e-2, "dgelu colsum"
    cs.zero_()
    hip.gemm_nt(ac, wc, out, M=m, resid=rc, flags=BIG | hip.GF_MUL_R, colsum_out=cs)
    _close(out, lin0 * r.float(), dtype, "mul_r")
    ref_cs = out.float().sum(0)
    assert float((cs - ref_cs).abs().max()) <= 2e-3 * float(ref_cs.abs().max()) + 1e-2, "mul_r colsum"
    hip.gemm_nt(ac, wc, out, M=m, bias=bc, scale=0.25, scale_cols=96, flags=BIG)
    exp = lin.clone()
    exp[:, :96] *= 0.25
    _close(out, exp, dtype, "scale_cols")
    hip.gemm_nt(ac, wc, out, M=m, bias=bc, flags=BIG | hip.GF_RELU)      # generic (runtime-flag) body
    _close(out, F.relu(lin), dtype, "relu")
    # scatter / gather row maps with the residual (proj + window_reverse + shortcut)
    cmap = torch.randperm(700)[:m].to(torch.int32).cuda()
    r2 = torch.randn(700, n).to(dtype).cuda()
    out2 = torch.zeros(700, n, dtype=dtype, device="cuda")
    hip.gemm_nt(ac, wc, out2, M=m, bias=bc, c_rows=cmap, resid=r2, r_rows=cmap, flags=BIG | hip.GF_RESID)
    ref = torch.zeros(700, n)
    ref[cmap.cpu().long()] = lin + r2.float().cpu()[cmap.cpu().long()]
    _close(out2, ref, dtype, "scatter + gathered resid")


@pytest.mark.parametrize("dtype", DT)
def test_gemm_nt_gather_scatter_segments(dtype):
    torch.manual_seed(6)
    rows_src, k, n, m = 500, 64, 136, 333
    a = torch.randn(rows_src, k).to(dtype)
    s = 3
    w = (torch.randn(n, s * k) / (s * k) ** 0.5).to(dtype)
    amap = torch.randint(-1, rows_src, (s, m), dtype=torch.int32)
    amap[:, 5] = -1
    cmap = torch.randperm(400)[:m].to(torch.int32)
    r = torch.randn(400, n).to(dtype)
    out = torch.zeros(400, n, dtype=dtype, device="cuda")
    hip.gemm_nt(a.cuda(), w.cuda(), out, M=m, a_rows=amap.cuda(), c_rows=cmap.cuda(), resid=r.cuda(),
                r_rows=cmap.cuda(), S=s, flags=hip.GF_RESID)
    if dtype == torch.bfloat16:
        outb = torch.zeros(400, n, dtype=dtype, device="cuda")
        cs = torch.zeros(n, device="cuda")
        hip.gemm_nt(a.cuda(), w.cuda(), outb, M=m, a_rows=amap.cuda(), c_rows=cmap.cuda(), resid=r.cuda(),
                    r_rows=cmap.cuda(), S=s, flags=hip.GF_RESID | hip.GF_MID, colsum_out=cs)
        assert torch.allclose(outb.float(), out.float(), atol=2e-2, rtol=2e-2)
        assert torch.allclose(cs.cpu(), outb.float().cpu()[cmap.long()].sum(0), atol=0.5, rtol=2e-2)
    af = a.float()
    gathered = torch.cat([torch.where((amap[i] >= 0)[:, None], af[amap[i].clamp(min=0).long()], torch.zeros(m, k))
                          for i in range(s)], 1)
    exp = torch.zeros(400, n)
    exp[cmap.long()] = gathered @ w.float().t() + r.float()[cmap.long()]
    _close(out, exp, dtype, "gather/scatter/segments")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("mk,ni,nj,splits", [(64, 128, 128, 1), (1000, 136, 264, 0), (4096, 512, 1536, 0), (33, 48, 64, 4)])
def test_gemm_tn(dtype, mk, ni, nj, splits):
    torch.manual_seed(mk)
    at = torch.randn(mk, ni).to(dtype)
    bt = (torch.randn(mk, nj) / mk ** 0.5).to(dtype)
    out = torch.ones(ni, nj, device="cuda")
    hip.gemm_tn(at.cuda(), bt.cuda(), out, Mk=mk, splits=splits)
    _close(out, 1 + at.float().t() @ bt.float(), torch.float32 if dtype == torch.float32 else dtype, "tn")
    out2 = torch.ones(ni, nj, device="cuda")
    hip.gemm_tn(at.cuda(), bt.cuda(), out2, Mk=mk, splits=splits, atomics=True)      # atomic split-K combine
    _close(out2, 1 + at.float().t() @ bt.float(), torch.float32 if dtype == torch.float32 else dtype, "tn atomics")


@pytest.mark.parametrize("dtype", DT)
def test_gemm_tn_row_maps(dtype):
    torch.manual_seed(8)
    mk, ni, nj = 777, 64, 128
    at = torch.randn(900, ni).to(dtype)
    bt = torch.randn(900, nj).to(dtype) / 30
    am = torch.randint(0, 900, (mk,), dtype=torch.int32)
    bm = torch.randint(-1, 900, (mk,), dtype=torch.int32)
    out = torch.zeros(ni, nj, device="cuda")
    hip.gemm_tn(at.cuda(), bt.cuda(), out, Mk=mk, at_rows=am.cuda(), bt_rows=bm.cuda())
    bsel = torch.where((bm >= 0)[:, None], bt.float()[bm.clamp(min=0).long()], torch.zeros(mk, nj))
    _close(out, at.float()[am.long()].t() @ bsel, dtype, "tn maps")


def test_gemm_tn_deferred_combine_equals_inline_combine(monkeypatch):
    """hip.tn_deferred(): the split-K combine on a side stream (stswin_gemm_tn with STSWIN_TN_NO_COMBINE + stswin_tn_combine) gives
    bitwise the result of the inline combine, for a chain of GEMMs that share the workspace (each launch waits for the previous
    combine), with overwrite and with accumulation, and the calling stream sees the results after the block."""
    monkeypatch.setattr(hip, "_TN_DEFER_ON", True)         # (off by default: slower end to end, see hip.py)
    torch.manual_seed(3)
    shapes = [(65536, 512, 512, True), (16384, 1024, 256, False), (65536, 256, 2048, True), (8192, 128, 384, True)]
    ops = []
    for mk, ni, nj, ow in shapes:
        at = torch.randn(mk, ni, device="cuda").bfloat16()
        bt = torch.randn(mk, nj, device="cuda").bfloat16()
        ops.append((at, bt, mk, ni, nj, ow))

    def run(deferred):
        outs = []
        ctx = hip.tn_deferred() if deferred else None
        if ctx:
            ctx.__enter__()
        for at, bt, mk, ni, nj, ow in ops:
            out = torch.empty(ni, nj, device="cuda") if ow else torch.ones(ni, nj, device="cuda")
            hip.gemm_tn(at, bt, out, Mk=mk, overwrite=ow)
            outs.append(out)
        if ctx:
            assert hip._TN_PENDING is not None            # a combine is in flight on the side stream
            ctx.__exit__(None, None, None)
            assert hip._TN_PENDING is None
        return [o.clone() for o in outs]                  # (clone on the calling stream: ordered behind the join)

    a, b = run(True), run(False)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    ref = ops[1][0].float().t() @ ops[1][1].float() + 1.0
    assert float((a[1] - ref).abs().max()) < 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("mk,ni,nj,ow", [(16544, 200, 896, True), (41504, 88, 728, False), (61824, 88, 192, False), (33088, 448, 136, True),
                                         (2112, 512, 512, True), (8224, 256, 2048, True)])
def test_gemm_tn_ragged_contraction_no_empty_split(mk, ni, nj, ow):
    """A contraction length that does not divide into the automatic split count (259 K tiles over 64 splits = 5 each: 12 splits
    would be empty) must not leave splits without work: their workgroups return before writing a slab and the combine pass would
    add whatever the shared workspace held (found by tests/fuzz/fuzz_gemm.py).  The workspace is poisoned with NaN first."""
    torch.manual_seed(mk)
    at = torch.randn(mk, ni, device="cuda").bfloat16()
    bt = torch.randn(mk, nj, device="cuda").bfloat16()
    hip._tn_workspace(at.device).fill_(float("nan"))
    c = torch.empty(ni, nj, device="cuda") if ow else torch.ones(ni, nj, device="cuda")
    hip.gemm_tn(at, bt, c, Mk=mk, overwrite=ow)
    ref = at.float().t() @ bt.float() + (0.0 if ow else 1.0)
    assert torch.isfinite(c).all()
    assert float((c - ref).abs().max()) < 1e-2 * float(ref.abs().max())


@pytest.mark.parametrize("M,N,K,relu,bias", [(4, 512, 1024, False, False), (4, 1024, 512, True, True), (1, 64, 64, False, True), (8, 260, 1088, True, False)])
def test_few_row_products_take_the_row_kernels(M, N, K, relu, bias):
    """M <= 8 rows (the 1x1 convolution behind AdaptiveAvgPool2d(1), ASPP.py:43-46, forward / input gradient / weight gradient):
    one wave per output column, and a sum of <= 8 outer products - against fp64 products of the same bf16 operands."""
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    bv = torch.randn(N, device="cuda") if bias else None
    c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    hip.gemm_nt(a, b, c, M=M, bias=bv, flags=hip.GF_RELU if relu else 0)
    assert hip.last_variant(0)["kernel"] == hip.VAR_NT_ROWS
    ref = a.double() @ b.double().t() + (bv.double() if bias else 0)
    ref = ref.clamp(min=0) if relu else ref
    assert float((c.double() - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max()) + 1e-6
    # weight gradient of the same layer: dW[n][k] = sum_m dy[m][n] * x[m][k]
    N8 = N // 8 * 8
    dy = torch.randn(M, N8, device="cuda").to(torch.bfloat16)
    dw = torch.full((N8, K), 0.5, dtype=torch.float32, device="cuda")
    hip.gemm_tn(dy, a, dw, Mk=M)
    assert hip.last_variant(1)["kernel"] == hip.VAR_TN_ROWS
    want = 0.5 + dy.double().t() @ a.double()
    assert float((dw.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
    hip.gemm_tn(dy, a, dw, Mk=M, overwrite=True)
    assert float((dw.double() - (want - 0.5)).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("M,N,Kseg,S,relu,bias,gather", [(4096, 512, 1024, 9, False, False, True), (2048, 1024, 512, 9, True, True, True),
                                                         (8192, 256, 448, 9, False, False, True), (2048, 512, 4096, 1, True, False, False),
                                                         (3000, 264, 1024, 3, False, True, True)])
def test_split_k_gemm_nt_matches_unsplit_and_fp64(M, N, Kseg, S, relu, bias, gather):
    """stswin_gemm_nt_splitk (few 256x256 tiles, long K: ASPP.py:13-20 dilated convolutions) against fp64 products of the same bf16
    operands and against the unsplit kernels; bitwise reproducible."""
    torch.manual_seed(M + N + S)
    rows = M + 37
    a = torch.randn(rows, Kseg, device="cuda").to(torch.bfloat16)
    b = (torch.randn(N, S * Kseg, device="cuda") / (S * Kseg) ** 0.5).to(torch.bfloat16)
    bv = torch.randn(N, device="cuda") if bias else None
    rmap = None
    if gather:
        rmap = torch.randint(-1, rows, (S, M), device="cuda", dtype=torch.int32)
    assert hip.load().stswin_gemm_nt_splitk_scratch(M, N, Kseg, S) > 0
    c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    hip.gemm_nt(a, b, c, M=M, a_rows=rmap, S=S, bias=bv, flags=hip.GF_RELU if relu else 0)
    v = hip.last_variant(0)
    assert v["kernel"] == hip.VAR_NT_SPLITK and v["splits"] >= 2
    ref = torch.zeros(M, N, dtype=torch.float64, device="cuda")
    for s_ in range(S):
        if gather:
            idx = rmap[s_].long()
            seg = torch.where((idx >= 0)[:, None], a.double()[idx.clamp(min=0)], torch.zeros((), dtype=torch.float64, device="cuda"))
        else:
            seg = a.double()[:M]
        ref += seg @ b.double()[:, s_ * Kseg:(s_ + 1) * Kseg].t()
    if bias:
        ref += bv.double()
    if relu:
        ref = ref.clamp(min=0)
    scale = float(ref.abs().max())
    assert float((c.double() - ref).abs().max()) <= 2.0 ** -8 * scale * 1.01 + 1e-6
    c2 = torch.empty_like(c)
    hip._NT_SPLITK = False
    try:
        hip.gemm_nt(a, b, c2, M=M, a_rows=rmap, S=S, bias=bv, flags=hip.GF_RELU if relu else 0)
    finally:
        hip._NT_SPLITK = True
    assert hip.last_variant(0)["kernel"] != hip.VAR_NT_SPLITK
    d = (c.float() - c2.float()).abs()
    assert float(d.max()) <= 2.0 ** -7 * scale and float((d > 0).float().mean()) < 0.02
    c3 = torch.empty_like(c)
    hip.gemm_nt(a, b, c3, M=M, a_rows=rmap, S=S, bias=bv, flags=hip.GF_RELU if relu else 0)
    assert torch.equal(c3, c)


@pytest.mark.parametrize("col0,wide", [(4, 1032), (8, 1028), (16, 1040)])
def test_split_k_gemm_nt_unaligned_out_slice_falls_back_or_runs_correctly(col0, wide):
    """`out` as a column slice of a wider buffer (the ASPP concat path hands gemm_nt such slices): the split-K combine stores 16-byte
    row pieces, so a slice that starts off an 8-column boundary or sits in a buffer whose pitch is not a multiple of 8 must be
    refused by stswin_gemm_nt_splitk (-1008) and computed by the tiled kernels; an aligned slice takes the split-K path.  Either way
    the values are right and nothing outside the slice is touched."""
    M, N, Kseg, S = 2048, 512, 1024, 4
    torch.manual_seed(col0)
    a = torch.randn(M, Kseg, device="cuda").to(torch.bfloat16)
    b = (torch.randn(N, S * Kseg, device="cuda") / (S * Kseg) ** 0.5).to(torch.bfloat16)
    rmap = torch.randint(-1, M, (S, M), device="cuda", dtype=torch.int32)
    buf = torch.full((M, wide), 7.0, dtype=torch.bfloat16, device="cuda")
    out = buf[:, col0:col0 + N]
    hip.gemm_nt(a, b, out, M=M, a_rows=rmap, S=S)
    aligned = col0 % 8 == 0 and wide % 8 == 0
    assert (hip.last_variant(0)["kernel"] == hip.VAR_NT_SPLITK) == aligned
    ref = torch.zeros(M, N, dtype=torch.float64, device="cuda")
    for s_ in range(S):
        idx = rmap[s_].long()
        ref += torch.where((idx >= 0)[:, None], a.double()[idx.clamp(min=0)], torch.zeros((), dtype=torch.float64, device="cuda")) \
            @ b.double()[:, s_ * Kseg:(s_ + 1) * Kseg].t()
    assert float((out.double() - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max()) * 1.01 + 1e-6
    assert bool((buf[:, :col0] == 7.0).all()) and bool((buf[:, col0 + N:] == 7.0).all())


@pytest.mark.parametrize("mk,ni,nj,mode,ow", [(65536, 512, 2048, "plain", True), (16384, 1024, 1024, "plain", False), (65536, 520, 1000, "plain", True),
                                              (65536, 512, 512, "at_rows", True), (65536, 1536, 512, "bt_rows", True),
                                              (65536, 256, 2304, "taps", True), (65536, 512, 4608, "taps_tapminor", True)])
def test_gemm_tn_fused_split_k_combine_equals_the_separate_reduce(mk, ni, nj, mode, ow, monkeypatch):
    """The split-K combine inside the weight-gradient GEMM launch (gemm_tn_ring_kernel<MODE, true>: arrival / departure counters, partials handed
    over through write-through stores) gives bit for bit what the separate tn_reduce pass gives (STSWIN_TN_FUSED=0, read per call): plain,
    gathered A rows, gathered B rows, per-tap maps (convolution weight gradients, also with the tap-minor output order), ragged tile edges,
    overwrite and accumulate - and again on the second launch (the counters re-arm themselves) with another shape in between."""
    torch.manual_seed(mk + ni)
    dev = "cuda"
    rows = mk
    at = torch.randn(rows, ni, device=dev).bfloat16()
    kw = {}
    if mode in ("taps", "taps_tapminor"):
        S, bseg = 9, nj // 9
        bt = (torch.randn(rows, bseg, device=dev) / 8).bfloat16()
        kw = {"bt_rows": torch.randint(-1, rows, (S, mk), device=dev, dtype=torch.int32), "bseg": bseg, "tapminor": mode == "taps_tapminor"}
    else:
        bt = (torch.randn(rows, nj, device=dev) / 8).bfloat16()
        if mode == "at_rows":
            kw = {"at_rows": torch.randperm(rows, device=dev).to(torch.int32)}
        elif mode == "bt_rows":
            kw = {"bt_rows": torch.randperm(rows, device=dev).to(torch.int32)}

    def run(fused):
        monkeypatch.setenv("STSWIN_TN_FUSED", "1" if fused else "0")
        out = torch.empty(ni, nj, device=dev) if ow else torch.full((ni, nj), 0.5, device=dev)
        hip.gemm_tn(at, bt, out, Mk=mk, overwrite=ow, **kw)
        v = hip.load().stswin_last_variant(1)
        return out, bool(v & hip.VAR_TN_FUSED), hip.last_tn_tapminor() if mode == "taps_tapminor" else None

    a, fa, ta = run(True)
    b, fb, tb = run(False)
    assert fa and not fb, "the fused combine did not run where the training step relies on it"
    assert ta == tb
    assert torch.equal(a, b)
    other = torch.empty(512, 512, device=dev)          # another tile geometry between two launches of this one
    hip.gemm_tn(at[:, :512].contiguous() if ni >= 512 else at, at[:, :512].contiguous() if ni >= 512 else at, other[:min(ni, 512), :min(ni, 512)].contiguous(), Mk=mk,
                overwrite=True)
    a2, fa2, _ = run(True)
    assert fa2 and torch.equal(a2, a)
    if mode == "plain":
        ref = at.float().t() @ bt.float() + (0.0 if ow else 0.5)
        assert float((a - ref).abs().max()) < 1e-2 * float(ref.abs().max())


@pytest.mark.parametrize("mk,ni,nj", [(65536, 512, 2048), (16384, 1024, 4096)])
def test_gemm_tn_fused_combine_beside_a_busy_second_stream(mk, ni, nj, monkeypatch):
    """Round 5 (round-4 advisor, medium): the fused split-K combine no longer relies on the whole grid being resident.  A second stream
    of the process keeps the compute units busy with long kernels (an RCCL all-reduce of a gradient bucket looks like this to the main
    stream) while the weight-gradient GEMMs run: nothing traps, nothing hangs, and every launch gives bit for bit the result of the
    separate tn_reduce pass.  The counters live in a per-stream region, so fused launches issued on TWO streams at once do not disturb
    each other either."""
    torch.manual_seed(mk + nj)
    dev = "cuda"
    at = torch.randn(mk, ni, device=dev).bfloat16()
    bt = (torch.randn(mk, nj, device=dev) / 8).bfloat16()
    monkeypatch.setenv("STSWIN_TN_FUSED", "0")
    ref = torch.empty(ni, nj, device=dev)
    hip.gemm_tn(at, bt, ref, Mk=mk, overwrite=True)
    assert not (hip.load().stswin_last_variant(1) & hip.VAR_TN_FUSED)
    monkeypatch.setenv("STSWIN_TN_FUSED", "1")
    torch.cuda.synchronize()
    side, side2 = torch.cuda.Stream(), torch.cuda.Stream()
    hog = torch.randn(1 << 28, device=dev)                 # 1 GB: every pass of the side stream is a ~0.4 ms all-CU kernel
    outs = [torch.empty(ni, nj, device=dev) for _ in range(12)]
    outs2 = [torch.empty(ni, nj, device=dev) for _ in range(4)]
    side.wait_stream(torch.cuda.current_stream())
    side2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(40):
            hog = hog * 1.0001 + 0.5
    for o in outs:
        hip.gemm_tn(at, bt, o, Mk=mk, overwrite=True)
        assert hip.load().stswin_last_variant(1) & hip.VAR_TN_FUSED
    with torch.cuda.stream(side2):                         # a second stream issuing fused launches of its own, concurrently
        for o in outs2:
            hip.gemm_tn(at, bt, o, Mk=mk, overwrite=True)
            assert hip.load().stswin_last_variant(1) & hip.VAR_TN_FUSED
    torch.cuda.synchronize()
    for o in outs + outs2:
        assert torch.equal(o, ref)


def test_gemm_tn_fused_hold_switch(monkeypatch):
    """stswin_tn_fused_hold: refcounted process-wide switch (what GradBucketReducer holds while its collectives overlap backward);
    the environment variable overrides it in both directions."""
    monkeypatch.delenv("STSWIN_TN_FUSED", raising=False)
    dev = "cuda"
    at = torch.randn(65536, 512, device=dev).bfloat16()
    bt = torch.randn(65536, 512, device=dev).bfloat16()
    out = torch.empty(512, 512, device=dev)

    def fused():
        hip.gemm_tn(at, bt, out, Mk=65536, overwrite=True)
        return bool(hip.load().stswin_last_variant(1) & hip.VAR_TN_FUSED)

    assert hip.tn_fused_holds() == 0 and fused()
    h1, h2 = hip.TnFusedHold(), hip.TnFusedHold()
    assert hip.tn_fused_holds() == 2 and not fused()
    monkeypatch.setenv("STSWIN_TN_FUSED", "1")
    assert fused()
    monkeypatch.delenv("STSWIN_TN_FUSED")
    h1.release(); h1.release()
    assert hip.tn_fused_holds() == 1 and not fused()
    h2.release()
    assert hip.tn_fused_holds() == 0 and fused()
    monkeypatch.setenv("STSWIN_TN_FUSED", "0")
    assert not fused()


@pytest.mark.parametrize("m,n,k,s_", [(4096, 512, 512, 1), (2048, 768, 1024, 1), (1000, 256, 192, 1), (1024, 256, 64, 1), (4096, 512, 64, 9),
                                      (2048, 512, 2048, 1), (512, 256, 448, 1), (768, 512, 128, 1), (600, 256, 320, 1), (512, 256, 576, 1)])
def test_gemm_nt_duo_kernel_is_bitwise_the_ring_kernel(m, n, k, s_):
    """Round 5: the 128x256 self-pipelined 4-wave kernel (two workgroups per CU, GF_DUO) adds the same products in the same order as
    the 256x256 ping-pong ring kernel and shares its register epilogues: same bits on every epilogue the Swin MLP uses, for stage
    counts that exercise the steady loop (period 6), every tail length, tap-segmented gathered A operands and ragged M."""
    if not hip.load().stswin_tuning_build():
        pytest.skip("GF_DUO is a STSWIN_TUNING build variant (measured slower inside the training step: profiles/r05_duo_in_step_ab.txt)")
    torch.manual_seed(m + k)
    dev, dt = "cuda", torch.bfloat16
    rows = m + 37
    a = torch.randn(rows, k, device=dev).to(dt)
    w = (torch.randn(n, s_ * k, device=dev) / (s_ * k) ** 0.5).to(dt)
    bias = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev).to(dt)
    amap = torch.randint(-1, rows, (s_, m), device=dev, dtype=torch.int32) if s_ > 1 else None
    cases = [dict(), dict(bias=bias), dict(bias=bias, flags=hip.GF_GELU), dict(bias=bias, out2=True, flags=hip.GF_GELU | hip.GF_C2_DGELU),
             dict(bias=bias, out2=True, flags=hip.GF_GELU), dict(bias=bias, resid=r, flags=hip.GF_RESID), dict(resid=r, flags=hip.GF_MUL_R, cs=True),
             dict(resid=r, flags=hip.GF_MUL_DGELU, cs=True), dict(bias=bias, flags=hip.GF_RELU)]
    for kw in cases:
        res = []
        for var in (hip.GF_BIG, hip.GF_BIG | hip.GF_DUO):
            k2 = dict(kw)
            out = torch.zeros(m, n, device=dev, dtype=dt)
            o2 = torch.zeros(m, n, device=dev, dtype=dt) if k2.pop("out2", False) else None
            cs = torch.zeros(n, device=dev) if k2.pop("cs", False) else None
            fl = k2.pop("flags", 0) | var
            hip.gemm_nt(a, w, out, M=m, a_rows=amap, S=s_, out2=o2, colsum_out=cs, flags=fl, **k2)
            want = hip.VAR_NT_DUO if var & hip.GF_DUO else hip.VAR_NT_RING256_REGEPI
            assert hip.load().stswin_last_variant(0) == want, (kw.keys(), hex(hip.load().stswin_last_variant(0)))
            res.append((out, o2, cs))
        assert torch.equal(res[0][0], res[1][0]), f"C differs: {sorted(kw)}"
        if res[0][1] is not None:
            assert torch.equal(res[0][1], res[1][1]), f"C2 differs: {sorted(kw)}"
        if res[0][2] is not None:                      # column sums: per-128-row blocks in both kernels, same fold order
            assert torch.allclose(res[0][2], res[1][2], rtol=1e-5, atol=1e-3), f"colsum differs: {sorted(kw)}"


@pytest.mark.parametrize("mk,c,hid", [(65536, 512, 2048), (16384, 1024, 4096), (32768, 512, 2048), (8192, 1024, 4096)])
def test_gemm_tn_group_is_bitwise_the_single_launches_at_the_same_split_counts(mk, c, hid):
    """stswin_gemm_tn_group: the three late weight gradients of a Swin block (fc1: plain, proj: gathered A rows, qkv: gathered B rows) in ONE
    launch.  A problem's result is a function of its operands and its split count alone: bit for bit what stswin_gemm_tn gives when it is
    forced to the same split count (ring kernel, fused combine) - for overwrite and accumulate, twice (the counters re-arm), and within
    bf16-partial rounding of the fp32 product."""
    torch.manual_seed(mk // 64 + c)
    dev = "cuda"
    dh = torch.randn(mk, hid, device=dev).bfloat16()
    n2 = (torch.randn(mk, c, device=dev) / 8).bfloat16()
    dx1 = torch.randn(mk, c, device=dev).bfloat16()
    o = (torch.randn(mk, c, device=dev) / 8).bfloat16()
    dqkv = torch.randn(mk, 3 * c, device=dev).bfloat16()
    x2 = (torch.randn(mk, c, device=dev) / 8).bfloat16()
    rmap = torch.randperm(mk, device=dev).to(torch.int32)

    def outs(ow):
        mk_ = (lambda *s: torch.empty(*s, device=dev)) if ow else (lambda *s: torch.full(s, 0.25, device=dev))
        return mk_(hid, c), mk_(c, c), mk_(3 * c, c)

    def problems(bufs, ow):
        return [dict(At=dh, Bt=n2, out=bufs[0], Mk=mk, overwrite=ow), dict(At=dx1, Bt=o, out=bufs[1], Mk=mk, at_rows=rmap, overwrite=ow),
                dict(At=dqkv, Bt=x2, out=bufs[2], Mk=mk, bt_rows=rmap, overwrite=ow)]
    for ow in (True, False):
        g = outs(ow)
        assert hip.gemm_tn_group(problems(g, ow)), "the library declined a set the training step relies on"
        sp = list(hip.LAST_TN_GROUP_SPLITS)
        assert len(sp) == 3 and min(sp) >= 1
        tiles = [(hid // 256) * (c // 256), (c // 256) ** 2, (3 * c // 256) * (c // 256)]
        wgs = sum(t * s_ for t, s_ in zip(tiles, sp))
        assert 224 <= wgs <= 256, f"{wgs} workgroups for {tiles} tiles x {sp} splits"
        g2 = outs(ow)
        assert hip.gemm_tn_group(problems(g2, ow))
        single = outs(ow)
        for q, s_, dst in zip(problems(single, ow), sp, single):
            hip.gemm_tn(q["At"], q["Bt"], dst, Mk=mk, at_rows=q.get("at_rows"), bt_rows=q.get("bt_rows"), overwrite=ow, splits=s_ | (1 << 29))
            assert hip.load().stswin_last_variant(1) & hip.VAR_TN_FUSED
        for a, a2, b in zip(g, g2, single):
            assert torch.equal(a, a2), "two grouped launches differ"
            assert torch.equal(a, b), "grouped launch and single launch at the same split count differ"
        ref = dh.float().t() @ n2.float() + (0.0 if ow else 0.25)
        assert float((g[0] - ref).abs().max()) < 1e-2 * float(ref.abs().max())
        refp = dx1.float()[rmap.long()].t() @ o.float() + (0.0 if ow else 0.25)
        assert float((g[1] - refp).abs().max()) < 1e-2 * float(refp.abs().max())
        refq = dqkv.float().t() @ x2.float()[rmap.long()] + (0.0 if ow else 0.25)
        assert float((g[2] - refq).abs().max()) < 1e-2 * float(refq.abs().max())


def test_gemm_tn_group_declines_what_the_ring_kernel_does_not_take(monkeypatch):
    """Sets the grouped launch must hand back untouched (the caller then launches one by one): a narrow output, a fused-combine hold, the
    A/B switch, too little work to fill 7/8 of the device."""
    dev = "cuda"
    a = torch.randn(8192, 512, device=dev).bfloat16()
    b = torch.randn(8192, 128, device=dev).bfloat16()
    sentinel = torch.full((512, 128), 7.0, device=dev)
    assert not hip.gemm_tn_group([dict(At=a, Bt=b, out=sentinel, Mk=8192)])
    assert bool((sentinel == 7.0).all())
    big = torch.randn(65536, 512, device=dev).bfloat16()
    out = torch.full((512, 512), 7.0, device=dev)
    ok = [dict(At=big, Bt=big, out=out, Mk=65536)]
    hold = hip.TnFusedHold()
    assert not hip.gemm_tn_group(ok)
    hold.release()
    monkeypatch.setenv("STSWIN_TN_GROUP", "0")
    assert not hip.gemm_tn_group(ok)
    monkeypatch.delenv("STSWIN_TN_GROUP")
    assert bool((out == 7.0).all())
    assert hip.gemm_tn_group(ok) and list(hip.LAST_TN_GROUP_SPLITS) == [64]        # 4 tiles x 64 splits: what the single launch picks, too
    small = torch.randn(1024, 512, device=dev).bfloat16()
    assert not hip.gemm_tn_group([dict(At=small, Bt=small, out=torch.empty(512, 512, device=dev), Mk=1024)])     # 4 tiles x 2 splits of 16 stages
