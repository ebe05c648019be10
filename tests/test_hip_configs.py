"""BASELINE.json configs that the bench line does not time, as `-m gpu` tests:

* configs[2]: the per-GPU batch of the 8-GPU DDP run (B = 8 clips at 512x512) - one full training step, bf16 path against
  the fp32 path of the same kernels (same assertions as the B = 4 step of test_hip_production_dispatch.py);
* configs[4]: the joint fine-tune step of seg18/train_CL_ft_mswin_sgd_minput.py (contrastive checkpoint -> TswinPlus,
  SGD with the script's 7 parameter groups, poly schedule, autocast + GradScaler) with the fp8 attention mode on and off;
* the CaDIS class counts of segcata/ (8 / 17 / 25 classes) through the model and the OHEM loss, against the CPU oracle;
* f4 post-processing at the evaluation size of seg18/test.py:155 (512x640 logits -> 1024x1280 labels)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_util as gu

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / b.norm())


def test_config2_per_gpu_batch8_step_bf16_vs_fp32_path():
    """BASELINE configs[2] per-GPU work: B = 8 clips x 4 frames x 3x512x512 (M = 131072 token rows at stage 1: three more
    rounds of 256x256 tiles per GEMM than the B = 4 bench step, other split-K factors).  Bounds as in the B = 4 test: 1.3 x the
    deviation of the reference's own bf16-autocast run from its fp32 run (tests/golden/bf16_yardstick.npz)."""
    from stswincl_amd.net.Ours.base18 import TswinPlus
    from stswincl_amd.utils.losses import OhemCELoss2D
    S, B = 512, 8
    torch.manual_seed(0)
    model = TswinPlus(12, (S // 8, S // 8)).cuda().train()
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    torch.manual_seed(4321)
    x = torch.randn(B, 4, 3, S, S, device="cuda")
    y = torch.randint(0, 12, (B, S, S), device="cuda")
    crit = OhemCELoss2D(S * S // 16)
    names = ["swin.layers.0.0.attn.qkv.weight", "swin.layers.1.1.mlp.fc1.weight", "swin.layers.5.1.mlp.fc2.weight",
             "swin.downsample.reduction.weight", "resnet.layer5.1.conv2.weight", "aspp.conv_3x3_2.weight", "classifier.0.weight"]
    res = {}
    for mode in ("fp32", "bf16"):
        model.load_state_dict(sd0)
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=BF, enabled=(mode == "bf16")):
            out = model(x)
            loss = crit(out, y)
        loss.backward()
        params = dict(model.named_parameters())
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
        res[mode] = (out.detach().float().cpu(), float(loss), {n: params[n].grad.detach().float().cpu() for n in names})
        del out, loss
        torch.cuda.empty_cache()
    (lf, lossf, gf), (lb, lossb, gb) = res["fp32"], res["bf16"]
    yard = gu.load("bf16_yardstick.npz")
    r_log, r_loss = rel(lb, lf), abs(lossb - lossf) / abs(lossf)
    r_g = {n: rel(gb[n], gf[n]) for n in names}
    print(f"B=8 bf16 vs fp32: logits {r_log:.3e} loss {r_loss:.3e} grads {r_g}")
    assert r_log < 1.3 * float(yard["rel_logits_256"]), r_log
    assert r_loss < 1e-3, (lossb, lossf)
    for n, r in r_g.items():
        assert r < 1.3 * float(yard["rel_grad/" + n]), (n, r)


@pytest.mark.parametrize("nc", [8, 17, 25])
def test_cadis_class_counts_model_and_ohem_vs_oracle(nc):
    """segcata/ (CaDIS, num_class_table: 8 / 17 / 25 classes) is the same graph with another classifier width; the classifier's
    1x1 convolution, the logits upsample and the OHEM kernels take nc as a run-time value.  fp32 path against the CPU oracle
    (1e-3 gate), train mode, forward + loss + the classifier's gradients; bf16 path: finite and close."""
    from oracle import stswin_oracle as O
    from stswincl_amd.net.Ours.base18 import TswinPlus
    from stswincl_amd.utils.losses import OhemCELoss2D
    torch.manual_seed(nc)
    hw = 128
    m = TswinPlus(nc, (hw // 8, hw // 8))
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.randn(2, 4, 3, hw, hw)
    y = torch.randint(0, nc, (2, hw, hw))
    y[0, :8, :8] = -1 if nc == 17 else y[0, :8, :8]         # (one case with ignored pixels)
    sdo = {k: v.clone() for k, v in sd.items()}
    w_last = sdo["classifier.3.weight"].requires_grad_(True)
    ref = O.tswin_plus(x, sdo, training=True)
    ref_loss = O.ohem_ce(ref, y, hw * hw // 16)
    ref_loss.backward()
    m = m.cuda().train()
    out = m(x.cuda())
    assert out.shape == (2, nc, hw, hw)
    assert rel(out, ref) < 1e-3, rel(out, ref)
    loss = OhemCELoss2D(hw * hw // 16)(out, y.cuda())
    assert abs(float(loss) - float(ref_loss)) < 1e-3 * abs(float(ref_loss))
    loss.backward()
    assert rel(m.classifier[3].weight.grad, w_last.grad) < 2e-3
    m.zero_grad(set_to_none=True)
    m.load_state_dict(sd)
    with torch.autocast("cuda", dtype=BF):
        ob = m(x.cuda())
        lb = OhemCELoss2D(hw * hw // 16)(ob, y.cuda())
    lb.backward()
    assert ob.shape == (2, nc, hw, hw) and torch.isfinite(ob).all()
    assert abs(float(lb) - float(ref_loss)) < 2e-2 * abs(float(ref_loss))
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def _contrastive_checkpoint(path, nc=12, res=(32, 32)):
    """A file in the layout main_pretrain_swinv5.py:87-103 writes, holding the encoder weights of a (seeded) TswinPlus."""
    import argparse
    from stswincl_amd.net.Ours.base18 import TswinPlus
    from stswincl_amd.utils.LoadModel import _CL_PREFIXES
    torch.manual_seed(77)
    src = TswinPlus(nc, res)
    cl = {}
    for k, v in src.state_dict().items():
        for pre, dst in _CL_PREFIXES:
            if k.startswith(dst + "."):
                cl[pre + k[len(dst):]] = v.clone()
    torch.save({"opt": argparse.Namespace(batch_size=8, amp_opt_level="O0"), "model": cl, "optimizer": {}, "scheduler": {}, "epoch": 150},
               path)
    return src


def _finetune_run(ckpt, fp8: bool, steps=3, S=256, B=2):
    """train_CL_ft_mswin_sgd_minput.py:126-129 (load_model_mswin_CL), :147-165 (7 groups, classifier lr x 10, SGD momentum +
    weight decay, OHEM, poly LR_Scheduler_Head), :192-205 (scheduler call, zero_grad, autocast, GradScaler) on the fused path."""
    from torch.cuda import amp
    from stswincl_amd.net.Ours.base18 import TswinPlus
    from stswincl_amd.optim import FusedSGD
    from stswincl_amd.utils.LoadModel import load_model_mswin_CL
    from stswincl_amd.utils.losses import OhemCELoss2D
    os.environ["STSWIN_FP8_ATTN"] = "1" if fp8 else "0"
    try:
        torch.manual_seed(5)
        model = TswinPlus(12, (S // 8, S // 8))
        model = load_model_mswin_CL(model, ckpt, log=False).cuda().train()
        lr, iters = 0.01, 100
        groups = [{"params": model.resnet.parameters(), "lr": lr}, {"params": model.aspp.parameters(), "lr": lr},
                  {"params": model.swin.parameters(), "lr": lr}, {"params": model.project1.parameters(), "lr": lr},
                  {"params": model.project2.parameters(), "lr": lr}, {"params": model.project3.parameters(), "lr": lr},
                  {"params": model.classifier.parameters(), "lr": lr * 10}]
        opt = FusedSGD(groups, lr=lr, momentum=0.9, weight_decay=1e-4)
        assert len(opt.param_groups) == 7 and opt.param_groups[-1]["lr"] == pytest.approx(10 * lr)
        crit = OhemCELoss2D(S * S // 16)
        scaler = amp.GradScaler()
        torch.manual_seed(6)
        lab = torch.randint(0, 12, (B, S // 32, S // 32), device="cuda").repeat_interleave(32, 1).repeat_interleave(32, 2)
        x = torch.randn(B, 4, 3, S, S, device="cuda") + F.one_hot(lab, 12).permute(0, 3, 1, 2)[:, None, :3].float() * 2.0
        losses = []
        for it in range(steps):
            cur = lr * pow(1 - 1.0 * it / iters, 0.9)                 # LR_Scheduler 'poly' (lr_scheduler.py:56-58) ...
            for g in opt.param_groups:                                # ... and LR_Scheduler_Head sets every group to it (:76-83)
                g["lr"] = cur
            opt.zero_grad()
            with amp.autocast():
                loss = crit(model(x), lab)
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            losses.append(float(loss.detach()))
        assert scaler.get_scale() >= 65536.0, "a step was skipped (inf / nan gradients)"
        assert all(torch.isfinite(p).all() for p in model.parameters())
        probe = model.swin.layers[0][0].attn.qkv.weight.detach().float().cpu().clone()
        return losses, probe
    finally:
        os.environ.pop("STSWIN_FP8_ATTN", None)


def test_config4_joint_finetune_steps_fp8_attention_vs_bf16(tmp_path):
    """BASELINE configs[4]: the fine-tune step of the joint script with fp8 (e4m3) attention against the same steps with the bf16
    attention core: same checkpoint, same data.  The fp8 core perturbs each attention output by ~4-5e-2 rel-L2 (kernel test);
    through the post-norm residual stream that must stay a perturbation of the trajectory: first-step losses within 2 %, every
    step within 5 %, the loss falls in both runs, and the updated weights of the first Swin block agree to 1e-2 rel-L2 after
    three SGD steps (the weights themselves, not their increments)."""
    ckpt = str(tmp_path / "ckpt_epoch_150.pth")
    src = _contrastive_checkpoint(ckpt, res=(32, 32))
    l8, w8 = _finetune_run(ckpt, fp8=True)
    l16, w16 = _finetune_run(ckpt, fp8=False)
    print(f"configs[4] losses fp8 {l8} bf16 {l16}; qkv.weight rel {rel(w8, w16):.3e}")
    w0 = src.state_dict()["swin.layers.0.0.attn.qkv.weight"]
    assert rel(w16, w0) > 1e-5, "the checkpoint was loaded and the weights moved"
    assert abs(l8[0] - l16[0]) < 2e-2 * abs(l16[0]), (l8, l16)
    for a, b in zip(l8, l16):
        assert a == a and abs(a - b) < 5e-2 * abs(b), (l8, l16)
    assert l8[-1] < l8[0] and l16[-1] < l16[0], (l8, l16)
    assert rel(w8, w16) < 1e-2


def test_config4_full_size_fp8_finetune_step_vs_the_oracle(tmp_path):
    """BASELINE configs[4] AT ITS SIZE: one joint-fine-tune step (train_CL_ft_mswin_sgd_minput.py:147-165,192-201: contrastive
    checkpoint -> TswinPlus, 7-group SGD, GradScaler) on B = 4 clips x 4 frames x 3x512x512 with the fp8 (e4m3) attention on.
    (i) every gradient finite, no skipped step; (ii) the training loss within 2e-2 of the loss of the fp32 CPU ORACLE on the same
    weights and clips (the oracle's forward: reference formulation, fp32); (iii) the attention core of EVERY block call inside
    that step (6 at stage 1: 128-token windows, head dim 128; 6 at stage 2: 32-token windows, head dim 256) against the reference
    formulation (swin_512.py:117-138) evaluated in fp32 on the very q | k | v the kernel read: rel-L2 <= 6e-2, max error <= 12 % of
    the output scale - the e4m3 bounds of the kernel test, now on the operands of a real step."""
    from torch.cuda import amp
    from oracle import stswin_oracle as O
    from stswincl_amd import ops
    from stswincl_amd.net.Ours.base18 import TswinPlus
    from stswincl_amd.optim import FusedSGD
    from stswincl_amd.utils.LoadModel import load_model_mswin_CL
    from stswincl_amd.utils.losses import OhemCELoss2D
    S, B = 512, 4
    ckpt = str(tmp_path / "ckpt_epoch_150.pth")
    _contrastive_checkpoint(ckpt, res=(S // 8, S // 8))
    torch.manual_seed(5)
    model = TswinPlus(12, (S // 8, S // 8))
    model = load_model_mswin_CL(model, ckpt, log=False)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    torch.manual_seed(6)
    lab = torch.randint(0, 12, (B, S // 32, S // 32)).repeat_interleave(32, 1).repeat_interleave(32, 2)
    x = torch.randn(B, 4, 3, S, S) + F.one_hot(lab, 12).permute(0, 3, 1, 2)[:, None, :3].float() * 2.0
    with torch.no_grad():
        ref_loss = float(O.ohem_ce(O.tswin_plus(x, sd, training=True), lab, S * S // 16))
    model = model.cuda().train()
    lr = 0.01
    groups = [{"params": model.resnet.parameters(), "lr": lr}, {"params": model.aspp.parameters(), "lr": lr},
              {"params": model.swin.parameters(), "lr": lr}, {"params": model.project1.parameters(), "lr": lr},
              {"params": model.project2.parameters(), "lr": lr}, {"params": model.project3.parameters(), "lr": lr},
              {"params": model.classifier.parameters(), "lr": lr * 10}]
    opt = FusedSGD(groups, lr=lr, momentum=0.9, weight_decay=1e-4)
    scaler = amp.GradScaler()
    # (iv, round 5) the same forward + backward first WITHOUT fp8 (bf16-stored q | k | v), nothing stepped: its gradients are what the
    # fp8 step's gradients are held against below
    sd_dev = {k: v.clone() for k, v in model.state_dict().items()}
    opt.zero_grad()
    with amp.autocast():
        loss16 = OhemCELoss2D(S * S // 16)(model(x.cuda()), lab.cuda())
    scaler.scale(loss16).backward()
    g16 = {k: p.grad.detach().float().clone() for k, p in model.named_parameters()}
    model.load_state_dict(sd_dev)                             # (running statistics / counters back: the fp8 pass sees the same state)
    os.environ["STSWIN_FP8_ATTN"] = "1"
    ops.ATTN_TAP = taps = []
    try:
        opt.zero_grad()
        with amp.autocast():
            loss = OhemCELoss2D(S * S // 16)(model(x.cuda()), lab.cuda())
        ops.ATTN_TAP = None
        scaler.scale(loss).backward()
        grads_finite = all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.parameters())
        g8 = {k: p.grad.detach().float().clone() for k, p in model.named_parameters()}
        scaler.step(opt)
        scaler.update()
    finally:
        ops.ATTN_TAP = None
        os.environ.pop("STSWIN_FP8_ATTN", None)
    assert grads_finite and scaler.get_scale() >= 65536.0, "inf / nan gradients: the step was skipped"
    assert all(bool(torch.isfinite(p).all()) for p in model.parameters())
    # fp8-step gradients against the bf16 step's (same weights, same clips, same loss scale): e4m3 q | k | v carry 2^-4 relative
    # rounding per element into every attention product; behind the decode head that is a few 1e-2 on the gradients, in front of the
    # 12 Swin blocks (ResNet) it accumulates.  Bounds per family = 1.3 x measured (measured on MI355X in brackets): classifier 0.075 (0.056),
    # projections 0.14 (0.08-0.11), ASPP 0.315 (0.24: it sits directly behind the stage-2 blocks), Swin 0.305 (0.23), ResNet 0.29 (0.22)
    # rel-L2, and every parameter's gradient keeps its direction (cosine > 0.85, measured >= 0.97) - a wrong scale factor (sq, sk, sv on
    # the wrong accumulator) or a missing straight-through term is O(1) and flips directions.  (The biases of the ASPP convolutions in
    # front of a train-mode BatchNorm have an exactly-zero true gradient - both runs hold rounding noise there - and are skipped.)
    # (round 6: the bounds are 1.3 x the values measured on the final tree - classifier 0.056, ASPP 0.241 (aspp.conv_1x1_2), projections
    #  0.078-0.107, Swin 0.234, ResNet 0.224; the kernels are deterministic, the printed line below shows the current values)
    fam_bound = {"classifier": 0.075, "aspp": 0.315, "project1": 0.14, "project2": 0.14, "project3": 0.14, "swin": 0.305, "resnet": 0.29}
    worst, worst_cos, bad = {}, {}, []
    for k in g16:
        if k.startswith("aspp.conv_") and k.endswith(".bias") and "conv_1x1_4" not in k:
            continue
        fam = k.split(".")[0]
        r = float((g8[k] - g16[k]).norm() / (g16[k].norm() + 1e-30))
        c = float((g8[k] * g16[k]).sum() / (g8[k].norm() * g16[k].norm() + 1e-30))
        worst[fam], worst_cos[fam] = max(worst.get(fam, 0.0), r), min(worst_cos.get(fam, 1.0), c)
        if not (r < fam_bound[fam] and c > 0.85):
            bad.append((k, r, c))
    print("configs[4] fp8-step vs bf16-step gradients, worst rel-L2 / lowest cosine per family: " +
          ", ".join(f"{f} {worst[f]:.3f} / {worst_cos[f]:.3f}" for f in sorted(worst)))
    assert not bad, bad
    r_loss = abs(float(loss) - ref_loss) / abs(ref_loss)
    print(f"configs[4] 512x512 B=4 fp8-attention step: loss {float(loss):.5f}, fp32 CPU oracle {ref_loss:.5f} (rel {r_loss:.2e})")
    assert r_loss < 2e-2, (float(loss), ref_loss)
    assert len(taps) == 12 and all(t["fp8"] for t in taps), [t["fp8"] for t in taps]
    rows = []
    for i, t in enumerate(taps):
        H, W, ws, shift, heads = t["geom"]
        C, N, T = t["C"], ws * ws, 2
        d = C // heads
        nW = (H // ws) * (W // ws)
        qkv = _tap_qkv_f32(t)
        nB_ = qkv.shape[0] // (T * N)
        bias = t["table"].float()[t["index"].reshape(-1)].reshape(N, N, heads).permute(2, 0, 1)        # swin_512.py:122-124
        q, k, v = qkv.reshape(nB_, T * N, 3, heads, d).permute(2, 0, 3, 1, 4)
        attn = q @ k.transpose(-2, -1) + bias.repeat(1, T, T).unsqueeze(0)
        if shift > 0:
            attn = (attn.reshape(nB_ // nW, nW, heads, T * N, T * N) + t["mask"].float().repeat(1, T, T)[None, :, None]).reshape(-1, heads, T * N, T * N)
        ref = (attn.softmax(-1) @ v).transpose(1, 2).reshape(nB_ * T * N, C)
        got = t["o"].float()
        r = float((got - ref).norm() / ref.norm())
        mx = float((got - ref).abs().max() / ref.abs().max())
        rows.append(f"block call {i:2d} (dim {C}, window {ws}, shift {shift}, {nB_} windows): rel-L2 {r:.4f}, max err {mx:.4f} of the output scale")
        assert torch.isfinite(got).all() and r < 6e-2 and mx < 0.12, rows[-1]
        del qkv, q, k, v, attn, ref, got
    print("\n".join(rows))


def _tap_qkv_f32(t):
    """q (pre-scaled) | k | v rows as fp32, as the attention kernel read them: bf16 rows, or e4m3 bytes x their (window, head) scales."""
    if t.get("qscale") is None:
        return t["qkv"].float()
    H, W, ws, shift, heads = t["geom"]
    rows, d = 2 * ws * ws, t["C"] // heads
    v = t["qkv"].view(torch.float8_e4m3fn).float()
    return v * t["qscale"].repeat_interleave(rows, 0).repeat_interleave(d, 1)


@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_postprocessing_at_the_evaluation_size_1024x1280(dtype):
    """seg18/test.py:153-157 at its real size: 512x640 logits -> bilinear (align_corners=True) to 1024x1280 -> softmax ->
    arg-max, fused in stswin_upsample_argmax.  Every pixel whose label differs from the reference formulation must be a
    numerical tie: the reference's own top-2 interpolated logits there differ by less than the fp32 rounding of the
    interpolation (1e-5 of the logit scale); and there must be only a handful of them."""
    from stswincl_amd.utils import EndoMetric as E
    torch.manual_seed(1)
    f, nc, h, w, H, W = 1, 12, 512, 640, 1024, 1280
    logits = (torch.randn(f, nc, h // 8, w // 8) * 3)
    logits = (F.interpolate(logits, (h, w), mode="bilinear") + 0.3 * torch.randn(f, nc, h, w)).to(dtype)   # smooth maps + noise
    gt = torch.randint(0, nc, (f, H // 64, W // 64)).repeat_interleave(64, 1).repeat_interleave(64, 2)
    up = F.interpolate(logits.float(), (H, W), mode="bilinear", align_corners=True)
    ref = torch.argmax(F.softmax(up, dim=1), dim=1)
    labels, dices, ious = E.predict_and_score(logits.cuda(), (H, W), gt.cuda())
    lab = labels.cpu().long()
    assert lab.shape == (f, H, W)
    mism = lab != ref
    n_mis = int(mism.sum())
    if n_mis:
        top2 = up.permute(0, 2, 3, 1)[mism].topk(2, dim=1).values
        margin = float((top2[:, 0] - top2[:, 1]).max())
        print(f"{n_mis} of {H * W} labels differ; largest top-2 margin among them {margin:.2e}")
        assert margin < 1e-5 * float(up.abs().max()), margin
    assert n_mis < 1e-5 * H * W + 5
    rd = E.general_dice(gt[0].numpy(), lab[0].numpy())
    rj = E.general_jaccard(gt[0].numpy(), lab[0].numpy())
    assert [c for c, _ in rd] == [c for c, _ in dices[0]]
    assert np.allclose([v for _, v in rd], [v for _, v in dices[0]], rtol=1e-12)
    assert np.allclose([v for _, v in rj], [v for _, v in ious[0]], rtol=1e-12)


def _seg_steps(S, B, steps, graph=False):
    from stswincl_amd.net.Ours.base18 import TswinPlus
    from stswincl_amd.optim import FusedAdam
    from stswincl_amd.utils.losses import OhemCELoss2D
    torch.manual_seed(0)
    model = TswinPlus(12, (S // 8, S // 8)).cuda().train()
    opt = FusedAdam(model.parameters(), 1e-4)
    crit = OhemCELoss2D(S * S // 16)
    torch.manual_seed(99)
    x = torch.randn(B, 4, 3, S, S, device="cuda")
    y = torch.randint(0, 12, (B, S, S), device="cuda")
    losses = []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=BF):
            loss = crit(model(x), y)
        loss.backward()
        grads = [p.grad.detach().clone() for p in model.parameters()]
        opt.step()
        losses.append(loss.detach().float().view(torch.int32).item())
    return losses, grads, [p.detach().clone() for p in model.parameters()], [b.detach().clone() for b in model.buffers()]


@pytest.mark.parametrize("S,B,steps", [(256, 2, 3), (512, 4, 1)])
def test_bf16_training_step_is_bitwise_reproducible(S, B, steps):
    """Two runs of the same bf16 training steps (same seed, same data) must agree BIT FOR BIT: loss, every gradient, every updated
    parameter, every BatchNorm running statistic.  No kernel sums with fp32 atomics (include/stswin_hip.h, 'deterministic
    cross-workgroup sums'): per-workgroup partial slabs + fixed-order folds, integer fixed-point for the OHEM sums."""
    a = _seg_steps(S, B, steps)
    b = _seg_steps(S, B, steps)
    assert a[0] == b[0], (a[0], b[0])
    names = ("gradient", "parameter", "buffer")
    for which in (1, 2, 3):
        for i, (u, v) in enumerate(zip(a[which], b[which])):
            assert torch.equal(u, v), f"{names[which - 1]} {i} differs between two runs: max |d| {float((u.float() - v.float()).abs().max())}"


def test_contrastive_step_is_bitwise_reproducible():
    """Same for the ConsistencyLoss step (batched views, bank kernel, class sums, LARS norms)."""
    import types
    from stswincl_amd.contrast.models.PixPro_swin_v5 import ConsistencyLoss
    from stswincl_amd.optim import make_contrast_optimizer
    args = types.SimpleNamespace(pixpro_p=1.0, pixpro_momentum=0.99, pixpro_clamp_value=0.0, pixpro_transform_layer=1,
                                 pixpro_ins_loss_weight=0.0, pixpro_pos_ratio=0.7, data="endo18", tag="1", pretrainpth=None,
                                 num_instances=2235, batch_size=2, epochs=150, start_epoch=1)

    def run():
        torch.manual_seed(0)
        net = ConsistencyLoss(args, input_resolution=(16, 16)).cuda().train()
        params = [p for p in net.parameters() if p.requires_grad]
        opt, _ = make_contrast_optimizer(params, batch_size=2)
        torch.manual_seed(5)
        ims = [torch.randn(2, 4, 3, 128, 128, device="cuda") for _ in range(6)]
        masks = [torch.randint(0, 12, (2, 1, 128, 128), device="cuda").float() for _ in range(6)]
        ls = []
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=BF):
                loss = net(*ims, *masks)
            loss.backward()
            opt.step()
            ls.append(loss.detach().float().view(torch.int32).item())
        return ls, [p.detach().clone() for p in net.parameters()]

    a, b = run(), run()
    assert a[0] == b[0]
    for i, (u, v) in enumerate(zip(a[1], b[1])):
        assert torch.equal(u, v), f"parameter {i} differs between two runs"


def test_bench_two_ranks_sharing_the_gpu_runs_the_data_parallel_path():
    """The N > 1 code of bench.py (launcher -> 2 ranks -> process group -> weight broadcast -> GradBucketReducer with gradients
    written straight into the all-reduce buckets -> max-over-ranks timing -> one JSON line) on the one GPU of the test box:
    STSWIN_BENCH_SHARE_GPU=1 lets both ranks use device 0 over gloo.  The 8-GPU RCCL run is the driver's; this keeps the path
    from rotting in between."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, STSWIN_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    losses = {}
    for comm, lo, hi in (("fp32", 4e8, 6e8), ("bf16", 2e8, 3e8)):
        # --graph 1 is REQUESTED: over gloo that cannot be honoured, and the line must say so (an explicit reason, never a silent
        # eager run); STSWIN_BENCH_STRICT_GRAPH=1 turns a failed capture over RCCL into an error instead of a fallback
        env["STSWIN_BENCH_STRICT_GRAPH"] = "1"
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
                              "--size", "256", "--no-cpu-baseline", "--no-secondary", "--no-profile", "--graph", "1", "--comm-dtype", comm],
                             env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')][-1]
        d = json.loads(line)
        assert d["n_gpus"] == 2 and d["dist"]["rccl_ranks"] == 2 and d["dist"]["shared_gpu_functional_test"] is True
        assert lo < d["dist"]["allreduce_bytes_per_step_per_rank"] < hi, d["dist"]       # ~125 M gradients, 4 or 2 bytes each on the wire
        assert d["dist"]["comm_dtype"] == ("float32" if comm == "fp32" else "bfloat16")
        assert d["config"]["parallelism"] == "dp2" and d["value"] > 0 and d["config"]["loss"] == d["config"]["loss"]
        assert d["config"]["graph_requested"] is True
        assert d["config"]["launch"].startswith("eager launches (hipGraph capture needs the RCCL backend"), d["config"]["launch"]
        losses[comm] = d["config"]["loss"]
    assert abs(losses["bf16"] - losses["fp32"]) < 2e-2 * abs(losses["fp32"]), losses      # bf16 on the wire: same training, rounded gradients


def test_bench_line_carries_the_round6_blocks():
    """bench.py's N = 1 line (small clips, few steps): graph-replayed headline, calibration block with the normalised value, a separate
    bracketed pass for the roofline figures, and gemm_tn's sampled work close to its launches' (the cancelled-span accounting of round 6)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "3", "--batch", "2", "--size", "256",
                          "--no-cpu-baseline", "--no-secondary", "--profile-stride", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                      # ONE JSON line on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 3 and d["value"] > 0 and d["dtype"] == "bf16"
    assert d["config"]["launch"].startswith("hipGraph replay of the whole step") and "no event brackets" in d["config"]["launch"]
    cal = d["calibration"]
    assert cal["mfma_bf16_tflops"] > 500 and cal["copy_tbps"] > 1 and cal["reference"]["mfma_bf16_tflops"] > 0
    assert abs(d["value_normalised"] / d["value"] - (cal["mfma_weight"] / cal["relative_mfma"] + 1 - cal["mfma_weight"])) < 1e-9
    assert d["profile_pass"]["steps"] == 4 and d["profile_pass"]["ms_per_step"] > 0
    r = d["roofline"]
    assert r["kernel"] == "gemm_nt_bf16" and r["bound"] == "mfma" and 0 < r["frac"] < 1 and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "gemm_tn_bf16" in r["other_kernels"]
    assert d["config"]["loss"] == d["config"]["loss"]
