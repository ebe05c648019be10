"""Pin the CPU oracle (oracle/stswin_oracle.py) against outputs of the reference itself.

Every fixture under tests/golden/ was produced by tools/gen_golden.py importing
/root/reference in the build container.  Tolerances: indexing/tables bit-exact; fp32 math
1e-5 relative (same torch ops in a different composition order), whole model 1e-4.
"""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import stswin_oracle as O

torch.set_num_threads(8)


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float(((a - b).norm() / (b.norm() + 1e-30)).detach())


# ---------------------------------------------------------------- a1-a4
@pytest.mark.parametrize("tag", ["s1_64x64", "s2_32x32", "s1_64x80", "s1_32x56", "s1_64x64_noshift", "s2_16x16"])
def test_pair_window_index_bit_exact(tag):
    g = gu.load("index_maps.npz")
    b, t, h, w, ws, shift = [int(v) for v in g[tag + "_cfg"]]
    mine = O.pair_window_index(b, t, h, w, ws, shift)
    assert np.array_equal(mine, g[tag])
    # round trip through the scatter is the identity
    ids = torch.arange(b * t * h * w).reshape(b, t, h * w, 1)
    back = O.pair_window_scatter(O.pair_window_gather(ids, h, w, ws, shift), h, w, ws, shift)
    assert torch.equal(back, ids)


def test_window_partition_kat():
    g = gu.load("index_maps.npz")
    x = torch.arange(2 * 8 * 8, dtype=torch.float32).reshape(2, 8, 8, 1)
    xw = O.window_partition(x, 4)
    assert np.array_equal(xw.reshape(-1).numpy().astype(np.int32), g["kat_partition"])
    # SURVEY.md 8(a) a4 known answer, in the pair-regrouped layout (B=1, T=2): window 1, frame 1, first row
    pw = O.pair_window_gather(x.reshape(1, 2, 64, 1), 8, 8, 4, 0)
    assert pw[1, 1, :4, 0].tolist() == [68.0, 69.0, 70.0, 71.0]


# ---------------------------------------------------------------- a5 / a7
@pytest.mark.parametrize("ws,total,c00,c0l,cl0", [(8, 458752, 112, 0, 224), (4, 6144, 24, 0, 48)])
def test_relative_position_index(ws, total, c00, c0l, cl0):
    g = gu.load("tables.npz")
    idx = O.relative_position_index(ws)
    assert np.array_equal(idx.numpy().astype(np.int32), g[f"rel_index_ws{ws}"])
    assert int(idx.sum()) == total and int(idx[0, 0]) == c00 and int(idx[0, -1]) == c0l and int(idx[-1, 0]) == cl0


@pytest.mark.parametrize("tag", ["64x64", "32x32", "64x80", "32x56", "16x28"])
def test_shift_mask_exact(tag):
    g = gu.load("tables.npz")
    h, w, ws, shift = [int(v) for v in g[f"mask_{tag}_cfg"]]
    m = O.shift_attn_mask(h, w, ws, shift)
    assert set(torch.unique(m).tolist()) <= {0.0, -100.0}
    assert np.array_equal((m != 0).numpy().astype(np.uint8), g[f"mask_{tag}"])
    if tag == "64x64":
        assert int((m != 0).sum()) == 31744 and int((m.abs().sum((1, 2)) > 0).sum()) == 15


# ---------------------------------------------------------------- a6
@pytest.mark.parametrize("tag", ["nomask", "mask"])
def test_window_attention(tag):
    g = gu.load("window_attention.npz")
    sd = gu.to_sd(g)
    c, heads, ws, t, n_w, b = [int(v) for v in g["cfg"]]
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    sd2 = dict(sd)
    sd2.update(params)
    mask = torch.from_numpy(g["mask"]) if tag == "mask" else None
    y = O.window_attention(x, sd2, "", heads, ws, mask)
    assert rel(y, g[f"y_{tag}"]) < 1e-5
    (y * torch.from_numpy(g["g"])).sum().backward()
    assert rel(x.grad, g[f"dx_{tag}"]) < 1e-5
    for k in ("qkv.weight", "qkv.bias", "proj.weight", "proj.bias", "relative_position_bias_table"):
        assert rel(params[k].grad, g[f"d_{tag}/{k}"]) < 1e-5, k


# ---------------------------------------------------------------- a8 / a9
@pytest.mark.parametrize("tag,shift", [("shift0", 0), ("shift2", 2)])
def test_swin_block(tag, shift):
    g = gu.load("swin_block.npz")
    sd = gu.to_sd(g, f"sd_{tag}/")
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if v.is_floating_point() and not k.endswith("attn_mask")}
    sd2 = dict(sd)
    sd2.update(params)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = O.swin_block(x, sd2, "", (8, 8), 4, 4, shift)
    assert rel(y, g[f"y_{tag}"]) < 1e-5
    (y * torch.from_numpy(g["g"])).sum().backward()
    assert rel(x.grad, g[f"dx_{tag}"]) < 1e-5
    for k, p in params.items():
        assert rel(p.grad, g[f"d_{tag}/{k}"]) < 2e-5, k


# ---------------------------------------------------------------- a10
def test_patch_merging():
    g = gu.load("patch_merging.npz")
    sd = gu.to_sd(g)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = O.patch_merging(x, params, "", (8, 8))
    assert rel(y, g["y"]) < 1e-5
    (y * torch.from_numpy(g["g"])).sum().backward()
    assert rel(x.grad, g["dx"]) < 1e-5
    for k, p in params.items():
        assert rel(p.grad, g[f"d/{k}"]) < 1e-5, k
    # gather order is bit-exact: [(0,0),(1,0),(0,1),(1,1)]
    ids = torch.arange(4 * 4, dtype=torch.float32).reshape(1, 1, 16, 1)
    assert O.patch_merge_gather(ids, 4, 4)[0, 0].tolist() == [0.0, 4.0, 1.0, 5.0]


# ---------------------------------------------------------------- a11
def test_swin_layer_v5():
    g = gu.load("swin_layer.npz")
    sd = gu.det_fill(gu.skeleton_sd(g["keys"], g["shapes"], g["dtypes"]))
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    o1, o2 = O.swin_layer_v5(x, params, "", 4)
    assert rel(o1, g["o1"]) < 1e-5 and rel(o2, g["o2"]) < 1e-5
    ((o1 * torch.from_numpy(g["g1"])).sum() + (o2 * torch.from_numpy(g["g2"])).sum()).backward()
    assert rel(x.grad, g["dx"]) < 2e-5
    for k in ("layers.0.0.attn.relative_position_bias_table", "layers.4.1.attn.qkv.weight"):
        assert rel(params[k].grad, g["d/" + k]) < 2e-5, k
    for k, p in params.items():
        ref = g["dsum/" + k]
        assert abs(float(p.grad.abs().sum()) - float(ref[1])) <= 1e-4 * float(ref[1]) + 1e-6, k


# ---------------------------------------------------------------- a13
def test_aspp_train_and_eval():
    g = gu.load("aspp.npz")
    sd = gu.det_fill(gu.skeleton_sd(g["keys"], g["shapes"], g["dtypes"]))
    x = torch.from_numpy(g["x"])
    y = O.aspp(x, sd, "", True)
    assert rel(y, g["y_train"]) < 1e-5
    assert rel(sd["bn_conv_3x3_2.running_mean"], g["rm_after"]) < 1e-6
    y = O.aspp(x, sd, "", False)
    assert rel(y, g["y_eval"]) < 1e-5


# ---------------------------------------------------------------- a15
@pytest.mark.parametrize("tag", ["thresh_branch", "topk_branch"])
def test_ohem(tag):
    g = gu.load("ohem.npz")
    labels = torch.from_numpy(g["labels"]).long()
    logits = torch.from_numpy(g[f"{tag}_logits"])
    assert bool(g[f"{tag}_took_thresh"]) == (tag == "thresh_branch")
    lg = logits.clone().requires_grad_(True)
    loss = O.ohem_ce(lg, labels, int(g[f"{tag}_n_min"]))
    assert rel(loss, g[f"{tag}_loss"]) < 1e-6
    loss.backward()
    assert rel(lg.grad, g[f"{tag}_dlogits"]) < 1e-5


# ---------------------------------------------------------------- a12 + a14 + a15 whole model
def test_tswinplus_logits_and_loss():
    g = gu.load("tswinplus.npz")
    sd = gu.det_fill(gu.skeleton_sd(g["keys"], g["shapes"], g["dtypes"]))
    x = gu.det_tensor("tswinplus/x", (2, 4, 3, 128, 128))
    labels = torch.from_numpy(g["labels"]).long()
    with torch.no_grad():
        y = O.tswin_plus(x, sd, True)
        assert rel(y[:, :, ::2, ::2], g["y_train_sub"]) < 1e-4
        assert abs(float(y.abs().sum()) - float(g["y_train_sum"][1])) < 1e-4 * float(g["y_train_sum"][1])
        loss = O.ohem_ce(y, labels, 128 * 128 // 16)
        assert rel(loss, g["loss_train"]) < 1e-5
        # per-frame BN: running stats were updated 4x by the 4 sequential frame passes
        assert rel(sd["resnet.layer5.1.bn2.running_mean"], g["rm_after"]) < 1e-5
        assert int(sd["resnet.resnet.1.num_batches_tracked"]) == int(g["nbt_after"]) == 4
        y = O.tswin_plus(x, sd, False)
        assert rel(y[:, :, ::2, ::2], g["y_eval_sub"]) < 1e-4


# ---------------------------------------------------------------- a16 / a17
def test_regression_loss_value_and_grad():
    g = gu.load("regression_loss.npz")
    n, c, h, w = [int(v) for v in g["shape"]]
    feats = [torch.nn.functional.normalize(gu.det_tensor(f"regression/f{i}", (n, c, h, w)), dim=1) for i in range(6)]
    labs = [torch.from_numpy(g[f"l{i}"]) for i in range(6)]
    q = feats[0].clone().requires_grad_(True)
    loss = O.regression_loss(q, *feats[1:], *labs, 12)
    assert rel(loss, g["loss"]) < 1e-6
    loss.backward()
    assert rel(q.grad, g["dq"]) < 1e-5
    # SURVEY.md 8(a) a17 known answer: random normalised features + random labels -> ln 2
    assert abs(float(loss) - 0.69313) < 2e-3


# ---------------------------------------------------------------- a18-a20
def test_consistency_loss_one_step():
    g = gu.load("consistency.npz")
    sd = gu.det_fill(gu.skeleton_sd(g["keys"], g["shapes"], g["dtypes"]))
    hh, ww = [int(v) for v in g["hw"]]
    ims = [gu.det_tensor(f"consistency/im{i}", (2, 4, 3, hh, ww)) for i in range(6)]
    masks = [torch.floor(gu.det_tensor(f"consistency/mask{i}", (2, 1, hh // 8, ww // 8), "uniform", 12.0))
             .clamp(0, 11).repeat_interleave(8, 2).repeat_interleave(8, 3) for i in range(6)]
    pk = [str(k) for k in g["param_keys"]]
    w = sd["pixpro.projector.linear2.weight"].requires_grad_(True)
    loss, k1 = O.consistency_loss(ims, masks, sd, pk, int(g["k0"]), int(g["big_k"]))
    assert int(g["big_k"]) == 167625 and k1 == int(g["k1"])
    assert rel(loss, g["loss"]) < 1e-5
    loss.backward()
    assert rel(w.grad, g["d_projector_linear2"]) < 1e-4
    for key in [f for f in g.files if f.startswith("probe/")]:
        t = sd[key[len("probe/"):]].detach().double()
        ref = g[key]
        assert abs(float(t.abs().sum()) - float(ref[1])) < 1e-5 * float(ref[1]) + 1e-9, key


# ---------------------------------------------------------------- f2 LARS
def lars_fixture():
    g = gu.load("lars.npz")
    names = [str(n) for n in g["names"]]
    hyper = dict(lr=float(g["lr"]), momentum=float(g["momentum"]), trust_coef=float(g["trust_coef"]), eps=float(g["eps"]))
    return g, names, hyper, float(g["wd"])


def test_lars_three_steps():
    g, names, hyper, wd = lars_fixture()
    params = {n: torch.from_numpy(g[f"p0/{n}"]).clone() for n in names}
    groups = {True: [n for n in names if params[n].dim() != 1], False: [n for n in names if params[n].dim() == 1]}
    bufs = {n: None for n in names}
    for step in range(3):
        for adaptive, ns in groups.items():
            bl = [bufs[n] for n in ns]
            O.lars_sgd_step([params[n] for n in ns], [torch.from_numpy(g[f"g{step}/{n}"]) for n in ns], bl,
                            weight_decay=wd if adaptive else 0.0, adaptive=adaptive, **hyper)
            bufs.update(zip(ns, bl))
        for n in names:
            assert rel(params[n], g[f"p{step + 1}/{n}"]) < 1e-6 or float(np.abs(g[f"p{step + 1}/{n}"]).max()) == 0, (step, n)
    for n in names:
        assert torch.allclose(bufs[n], torch.from_numpy(g[f"buf/{n}"]), rtol=1e-5, atol=1e-7), n
    assert float(np.abs(g["p3/zero.weight"]).max()) > 0          # param_norm == 0 took the un-scaled branch and moved


# ---------------------------------------------------------------- N1 bank mode
def _tok(t):
    n, c, h, w = t.shape
    return t.permute(0, 2, 3, 1).reshape(n * h * w, c)


def test_bank_loss_with_own_key_maps_is_the_reference_regression_loss():
    """World size 1, bank = the sample's own five key maps: the bank formulation must give the reference's golden value and
    gradient (regression_loss.npz, generated by the reference itself)."""
    g = gu.load("regression_loss.npz")
    n, c, h, w = [int(v) for v in g["shape"]]
    feats = [torch.nn.functional.normalize(gu.det_tensor(f"regression/f{i}", (n, c, h, w)), dim=1) for i in range(6)]
    labs = [torch.from_numpy(g[f"l{i}"]).reshape(n * h * w).long() for i in range(6)]
    q = _tok(feats[0]).clone().requires_grad_(True)
    bank = torch.stack([_tok(f) for f in feats[1:]], 0)
    loss = O.bank_contrast_loss(q, labs[0], bank, torch.stack(labs[1:], 0), [[0, 1, 2, 3, 4]], h * w, h * w)
    assert rel(loss, g["loss"]) < 1e-6
    loss.backward()
    dq = q.grad.view(n, h, w, c).permute(0, 3, 1, 2)
    assert rel(dq, g["dq"]) < 1e-5
    # both directions at once = the sum of the two calls of ConsistencyLoss.forward (PixPro_swin_v5.py:594-595)
    six = torch.stack([_tok(f) for f in feats], 0)
    six_l = torch.stack(labs, 0)
    q2 = torch.cat([_tok(feats[0]), _tok(feats[1])], 0)
    both = O.bank_contrast_loss(q2, torch.cat([labs[0], labs[1]]), six, six_l, [[1, 2, 3, 4, 5], [0, 2, 3, 4, 5]], h * w, h * w)
    a = O.regression_loss(feats[0], feats[1], *feats[2:], *[l.view(n, 1, h, w).float() for l in labs], 12)
    b = O.regression_loss(feats[1], feats[0], *feats[2:], labs[1].view(n, 1, h, w).float(), labs[0].view(n, 1, h, w).float(),
                          *[l.view(n, 1, h, w).float() for l in labs[2:]], 12)
    assert rel(both, a + b) < 1e-6
