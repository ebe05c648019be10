"""Whole-model parity: TswinPlus logits / OHEM loss vs the reference-generated golden (fp32 path, 1e-3 relative per
BASELINE.json north_star) and the bf16 path at its documented tolerance."""
import pytest
import torch

import golden_util as gu
from stswincl_amd.net.Ours.base18 import TswinPlus
from stswincl_amd.utils.losses import OhemCELoss2D

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm())


def _model():
    g = gu.load("tswinplus.npz")
    m = TswinPlus(12, (16, 16))
    sd = gu.det_fill(gu.skeleton_sd(g["keys"], g["shapes"], g["dtypes"]))
    r = m.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys and all(k.endswith(("attn_mask", "relative_position_index")) for k in r.missing_keys)
    return g, m.cuda()


def test_tswinplus_fp32_matches_reference():
    g, m = _model()
    x = gu.det_tensor("tswinplus/x", (2, 4, 3, 128, 128)).cuda()
    labels = torch.from_numpy(g["labels"]).long().cuda()
    m.train()
    y = m(x)
    assert rel(y[:, :, ::2, ::2], g["y_train_sub"]) < 1e-3
    loss = OhemCELoss2D(128 * 128 // 16)(y, labels)
    assert abs(float(loss) - float(g["loss_train"])) < 1e-3 * float(g["loss_train"])
    assert rel(m.resnet.layer5[1].bn2.running_mean, g["rm_after"]) < 1e-3
    assert int(m.resnet.resnet[1].num_batches_tracked) == 4
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    m.eval()
    with torch.no_grad():
        assert rel(m(x)[:, :, ::2, ::2], g["y_eval_sub"]) < 1e-3


def test_tswinplus_bf16_autocast():
    g, m = _model()
    x = gu.det_tensor("tswinplus/x", (2, 4, 3, 128, 128)).cuda()
    labels = torch.from_numpy(g["labels"]).long().cuda()
    m.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(x)
        loss = OhemCELoss2D(128 * 128 // 16)(y, labels)
    # Yardstick: the REFERENCE graph itself under CPU bf16 autocast on this fixture deviates by rel_logits_128 (7.8 %) from its
    # own fp32 run (tests/golden/bf16_yardstick.npz, tools/gen_golden.py --only bf16_yardstick): untrained weights and
    # train-mode BN on 16x16 / B = 2 maps amplify bf16 rounding.  The HIP bf16 path keeps more in bf16 than autocast does
    # (BatchNorm / LayerNorm outputs, residual streams); it is deterministic since round 3 (no fp32 atomics: the same value in
    # every run, tests/test_hip_configs.py::test_bf16_training_step_is_bitwise_reproducible) and held to 1.5 x the yardstick on
    # the logits.  The tight whole-model checks are on the conditioned fixture: tests/test_hip_bf16_stages.py (eval-mode logits
    # 2e-2, every weight gradient 3e-2 against the fp32 oracle).
    yard = gu.load("bf16_yardstick.npz")
    r_log = rel(y.float()[:, :, ::2, ::2], g["y_train_sub"])
    r_loss = abs(float(loss) - float(g["loss_train"])) / float(g["loss_train"])
    print(f"bf16 128x128: logits {r_log:.4f} (reference autocast {float(yard['rel_logits_128']):.4f}) loss {r_loss:.2e}")
    assert r_log < 1.5 * float(yard["rel_logits_128"])
    assert r_loss < 1e-2
    loss.backward()


def test_tswinplus_bf16_weight_gradients_vs_reference_autocast_yardstick():
    """Weight gradients of the bf16 path against the fp32 path on the golden fixture (128x128, B = 2).  Yardstick = what the
    reference's own bf16-autocast backward loses on the same fixture (bf16_yardstick.npz: 0.14 classifier, 0.23 ASPP, 0.58-0.79 for
    the Swin / ResNet weights).  The decode-head weights are well conditioned and held to 1.3 x the yardstick.  The deep weights
    sit behind ~50 untrained layers with train-mode BatchNorm over 2 samples: a flipped bf16 rounding of one normalised activation
    changes them by O(1) (the reference's own autocast run is 0.58-0.79 away from its fp32 run) - on THIS fixture they are a
    chaos indicator, not a parity measure, and only get a sanity bound here.  Their parity is asserted where it is well posed:
    tests/test_hip_bf16_stages.py::test_tswinplus_eval_mode_bf16_weight_gradients_vs_the_fp32_oracle (every weight gradient of the
    whole model <= 3e-2 of the fp32 oracle's with the BatchNorm statistics loaded) and the per-stage table of that file."""
    yard = gu.load("bf16_yardstick.npz")
    names = [k[len("rel_grad/"):] for k in yard.files if k.startswith("rel_grad/")]
    head = ("aspp.conv_3x3_2.weight", "classifier.0.weight")
    g, m = _model()
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    x = gu.det_tensor("tswinplus/x", (2, 4, 3, 128, 128)).cuda()
    labels = torch.from_numpy(yard["labels_128"]).long().cuda()
    grads = {}
    for mode in ("fp32", "bf16"):
        m.load_state_dict(sd0)
        m.zero_grad(set_to_none=True)
        m.train()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(mode == "bf16")):
            loss = OhemCELoss2D(128 * 128 // 16)(m(x), labels)
        loss.backward()
        params = dict(m.named_parameters())
        grads[mode] = {n: params[n].grad.detach().double().cpu() for n in names}
    for n in names:
        r = float((grads["bf16"][n] - grads["fp32"][n]).norm() / grads["fp32"][n].norm())
        print(f"grad {n}: bf16 vs fp32 {r:.4f} (reference autocast {float(yard['rel_grad/' + n]):.4f})")
        assert torch.isfinite(grads["bf16"][n]).all()
        if n in head:
            assert r < 1.3 * float(yard["rel_grad/" + n]), (n, r)
        else:
            assert r < 2.5, (n, r)                       # (anti-correlated draws reach ~2; garbage or a wrong scale would not stay below)


def test_tswinplus_256_fp32_gate_and_bf16_vs_reference_autocast_yardstick():
    """256x256, B = 4 (decode-head BatchNorm on 32x32 maps): the fp32 path against the reference's fp32 logits (1e-3 gate of
    BASELINE.json north_star) and the bf16 path against what the reference's own bf16 autocast run loses (9.6 %)."""
    yard = gu.load("bf16_yardstick.npz")
    g = gu.load("tswinplus.npz")
    m = TswinPlus(12, (32, 32))
    sd = gu.det_fill(gu.skeleton_sd(g["keys"], g["shapes"], g["dtypes"]))
    m.load_state_dict(sd, strict=False)
    m = m.cuda().train()
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    x = gu.det_tensor("tswinplus/x256", (4, 4, 3, 256, 256)).cuda()
    labels = torch.from_numpy(yard["labels_256"]).long().cuda()
    crit = OhemCELoss2D(256 * 256 // 16)
    with torch.no_grad():
        y = m(x)
    assert rel(y[:, :, ::4, ::4], yard["y_sub_256"]) < 1e-3
    assert abs(float(crit(y, labels)) - float(yard["loss_256"])) < 1e-3 * float(yard["loss_256"])
    m.load_state_dict(sd0)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        yb = m(x).float()
    r_log = rel(yb[:, :, ::4, ::4], yard["y_sub_256"])
    r_loss = abs(float(crit(yb, labels)) - float(yard["loss_256"])) / float(yard["loss_256"])
    print(f"bf16 256x256: logits {r_log:.4f} (reference autocast {float(yard['rel_logits_256']):.4f}) loss {r_loss:.2e}")
    assert r_log < 1.3 * float(yard["rel_logits_256"])
    assert r_loss < 1e-2


@pytest.mark.parametrize("tag", ["thresh_branch", "topk_branch"])
def test_ohem_matches_reference(tag):
    g = gu.load("ohem.npz")
    lg = torch.from_numpy(g[f"{tag}_logits"]).cuda().requires_grad_(True)
    labels = torch.from_numpy(g["labels"]).long().cuda()
    loss = OhemCELoss2D(int(g[f"{tag}_n_min"]))(lg, labels)
    assert abs(float(loss) - float(g[f"{tag}_loss"])) < 1e-5 * abs(float(g[f"{tag}_loss"]))
    loss.backward()
    assert rel(lg.grad, g[f"{tag}_dlogits"]) < 1e-4


@pytest.mark.parametrize("mode,tol", [("fp32", 1e-3), ("bf16", 5e-2)])
def test_resnet_frames_batched_equals_per_frame_oracle(mode, tol):
    """a12: the 4 per-frame ResNet calls of base18.py:86-89 as one batch with 4 BN groups vs the oracle's sequential calls."""
    from oracle import stswin_oracle as O
    from stswincl_amd.net.Ours.resnet import ResNet18_OS8
    torch.manual_seed(3)
    net = ResNet18_OS8()
    sd = gu.det_fill(net.state_dict())
    net.load_state_dict(sd)
    x = torch.randn(2, 4, 3, 64, 96)
    sdo = {"resnet." + k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        ref = torch.stack([O.resnet18_os8(x[:, i], sdo, "resnet.", True) for i in range(4)], 1)   # (B,4,512,h,w)
    net = net.cuda().train()
    xg = x.cuda()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(mode == "bf16")):
        tok, h, w = net.forward_frames(xg)
    got = tok.view(2, 4, h, w, 512).permute(0, 1, 4, 2, 3)
    assert rel(got, ref) < tol, rel(got, ref)
    assert rel(net.layer5[1].bn2.running_mean, sdo["resnet.layer5.1.bn2.running_mean"]) < tol
    assert int(net.resnet[1].num_batches_tracked) == 4
    tok.float().square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    if mode == "fp32":   # single-frame module call == reference signature
        net2 = ResNet18_OS8()
        net2.load_state_dict(sd)
        y = net2.cuda().train()(xg[:, 0])
        assert rel(y, ref[:, 0]) < tol


def test_fused_adam_steps_reach_the_gemm_weight_cache():
    """Three training steps with FusedAdam against the same steps with torch.optim.Adam (fp32 path): the fused kernel writes
    the parameters through raw pointers, so it must bump their version counters or the cached GEMM operands (ops.wcast,
    keyed on `_version`) would keep multiplying the initial weights."""
    from stswincl_amd.optim import FusedAdam
    losses = {}
    for name in ("fused", "torch"):
        g, m = _model()
        m.train()
        opt = FusedAdam(m.parameters(), 1e-3) if name == "fused" else torch.optim.Adam(m.parameters(), 1e-3)
        x = gu.det_tensor("tswinplus/x", (2, 4, 3, 128, 128)).cuda()
        labels = torch.from_numpy(g["labels"]).long().cuda()
        v0 = next(m.swin.parameters())._version
        ls = []
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            loss = OhemCELoss2D(128 * 128 // 16)(m(x), labels)
            loss.backward()
            opt.step()
            ls.append(float(loss))
        assert next(m.swin.parameters())._version > v0
        losses[name] = ls
    assert losses["fused"][1] != losses["fused"][0], "the second step must see the updated weights"
    # Both runs are bitwise reproducible (no atomics anywhere since round 3) and see the same first-step gradients, so the first loss
    # is EQUAL; afterwards the two optimizers round differently (the fused kernel multiplies by precomputed 1 / bias-correction
    # factors, torch.optim.Adam divides: last-bit differences of the updates), which the untrained 50-layer network with train-mode
    # BatchNorm amplifies per step - measured 2e-6 / 4e-5 after one / two updates; 1e-3 leaves the amplification room and still
    # catches a stale weight cache (the loss then does not move at all) or a wrong bias correction (1e-1).
    print("losses fused", losses["fused"], "torch", losses["torch"])
    assert losses["fused"][0] == losses["torch"][0], (losses["fused"], losses["torch"])
    for a, b in zip(losses["fused"][1:], losses["torch"][1:]):
        assert abs(a - b) < 1e-3 * abs(b), (losses["fused"], losses["torch"])


@pytest.mark.parametrize("opt_name", ["torch", "fused"])
def test_reference_amp_loop_idiom_autocast_default_and_gradscaler(opt_name):
    """The training loop of seg18/train_swin.py:160-173 verbatim: `with amp.autocast():` (no dtype: fp16 on CUDA - the HIP path
    computes in bf16 under any autocast), `scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()`.  The 2^16 loss
    scale is exact in bf16 / fp32 and nothing overflows, so the steps must be taken (scale never backs off) and the losses must
    follow those of the plain bf16 loop."""
    from torch.cuda import amp
    from stswincl_amd.optim import FusedAdam
    runs = {}
    for scaled in (True, False):
        g, m = _model()
        m.train()
        opt = FusedAdam(m.parameters(), 1e-4) if opt_name == "fused" else torch.optim.Adam(m.parameters(), 1e-4)
        scaler = amp.GradScaler()
        x = gu.det_tensor("tswinplus/x", (2, 4, 3, 128, 128)).cuda()
        labels = torch.from_numpy(g["labels"]).long().cuda()
        crit = OhemCELoss2D(128 * 128 // 16)
        ls = []
        for _ in range(3):
            if scaled:
                with amp.autocast():
                    loss = crit(m(x), labels)
                opt.zero_grad()
                scaler.scale(loss).backward()
                scaler.step(opt)
                scaler.update()
            else:
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    loss = crit(m(x), labels)
                opt.zero_grad()
                loss.backward()
                opt.step()
            ls.append(float(loss.detach()))
        if scaled:
            assert scaler.get_scale() >= 65536.0, "a step was skipped: inf / nan in the scaled gradients"
        assert all(torch.isfinite(p).all() for p in m.parameters())
        runs[scaled] = ls
    assert runs[True][0] == pytest.approx(runs[False][0], rel=2e-2)
    for a, b in zip(runs[True], runs[False]):
        assert abs(a - b) < 8e-2 * abs(b), runs
    assert runs[True][2] != runs[True][0]


def test_tswinplus_reference_native_resolution_512x640():
    """The reference constructs TswinPlus with input_resolution (64, 80) = 512 x 640 frames (base18.py:57, train_swin.py:110).
    Default constructor, B = 2 clips: bf16 train step (forward, OHEM-CE, backward) and the fp32 path of the same weights -
    shapes, finite gradients for every parameter, bf16 logits within the bf16 yardstick of the fp32 ones; eval forward."""
    torch.manual_seed(0)
    m = TswinPlus(12).cuda().train()
    assert tuple(m.swin.input_resolution) == (64, 80)
    x = torch.randn(2, 4, 3, 512, 640, device="cuda")
    labels = torch.randint(0, 12, (2, 512, 640), device="cuda")
    crit = OhemCELoss2D(512 * 640 // 16)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(x)
        loss = crit(y, labels)
    assert y.shape == (2, 12, 512, 640) and torch.isfinite(y).all()
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    m2 = TswinPlus(12).cuda().train()
    m2.load_state_dict({k: v for k, v in m.state_dict().items() if "num_batches_tracked" not in k and "running_" not in k}, strict=False)
    y32 = m2(x)
    # (untrained weights, train-mode BatchNorm on 2 clips: the reference's own bf16 autocast sits 0.06-0.09 from its fp32 run at
    #  128 x 128, tests above; 0.069 measured here)
    assert rel(y.float(), y32.detach().cpu()) < 0.12
    m.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        ye = m(x[:1])
    assert ye.shape == (1, 12, 512, 640) and torch.isfinite(ye).all()


@pytest.mark.parametrize("B,hw", [(3, (64, 64)), (5, (64, 128)), (2, (128, 64))])
def test_tswinplus_odd_batches_and_shapes_vs_oracle(B, hw):
    """Edge shapes against the CPU oracle (fp32, train mode, random default init shared through the state dict): odd clip counts
    (the pair-batched Swin calls and the per-frame BatchNorm groups must not assume even B) and non-square frames whose Swin
    resolution is a single window tall or wide."""
    from oracle import stswin_oracle as O
    torch.manual_seed(B)
    h, w = hw
    m = TswinPlus(12, (h // 8, w // 8))
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.randn(B, 4, 3, h, w)
    with torch.no_grad():
        ref = O.tswin_plus(x, {k: v.clone() for k, v in sd.items()}, training=True)
    y = m.cuda().train()(x.cuda())
    assert y.shape == ref.shape == (B, 12, h, w)
    assert rel(y, ref) < 1e-3, rel(y, ref)


def test_training_drives_the_loss_down_on_a_fixed_batch():
    """100 real optimizer steps (bf16 autocast, FusedAdam) on one fixed batch whose 32 x 32 px label blocks are visible in the image:
    the OHEM loss has to fall to under a third - the gradients of the whole graph, including the linked ones (middle frame pair,
    residual joins, epilogue BatchNorm statistics), point downhill together (tools/probes/train_soak.py: 2.53 -> 0.21 in 300)."""
    from stswincl_amd.optim import FusedAdam
    torch.manual_seed(0)
    m = TswinPlus(12, (32, 32)).cuda().train()
    opt = FusedAdam(m.parameters(), 3e-4)
    y = torch.randint(0, 12, (2, 8, 8), device="cuda").repeat_interleave(32, 1).repeat_interleave(32, 2)
    x = torch.randn(2, 4, 3, 256, 256, device="cuda")
    x = x + torch.nn.functional.one_hot(y, 12).permute(0, 3, 1, 2)[:, None, :3].float() * 2.0
    crit = OhemCELoss2D(256 * 256 // 16)
    first = last = None
    for i in range(100):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = crit(m(x), y)
        loss.backward()
        opt.step()
        if i == 0:
            first = float(loss.detach())
    last = float(loss.detach())
    assert last == last and last < first / 3, (first, last)
