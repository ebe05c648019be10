"""hipGraph replay as a TRAINING mode (round-5 verdict, item 5): N replays of a captured step == N eager steps, bit for bit.

The step-dependent scalars (Adam's step count and bias corrections, the SGD / LARS learning rate a scheduler changes on the host, the
key-encoder momentum schedule of PixPro_swin_v5.py:258-262) live in device memory (stswincl_amd.optim._Clock / EmaSchedule, csrc/optim.hip
stswin_optim_tick), so nothing of the optimizer's trajectory is frozen into kernel arguments at capture time.  Reference loops:
seg18/train_swin.py:151-173 (Adam), pixcontrast_18/main_pretrain_swinv5.py:113-153 (LARS + per-iteration cosine learning rate)."""
import math
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
STEPS = 20


def _seg_run(graphed: bool):
    from stswincl_amd.graph import GraphedStep
    from stswincl_amd.net.Ours.base18 import TswinPlus
    from stswincl_amd.optim import FusedAdam
    from stswincl_amd.utils.losses import OhemCELoss2D
    S, B = 128, 2
    torch.manual_seed(0)
    model = TswinPlus(12, (S // 8, S // 8)).cuda().train()
    opt = FusedAdam(model.parameters(), 1e-4)
    crit = OhemCELoss2D(S * S // 16)
    g = torch.Generator().manual_seed(99)
    xs = [torch.randn(B, 4, 3, S, S, generator=g) for _ in range(4)]
    ys = [torch.randint(0, 12, (B, S, S), generator=g) for _ in range(4)]
    x, y = xs[0].cuda(), ys[0].cuda()                  # static input buffers: every step copies its batch in (stream-ordered)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=BF):
            loss = crit(model(x), y)
        loss.backward()
        opt.step()
        return loss

    def lr_at(i):                                      # a host-side schedule (not in train_swin.py; exercises push_hyper)
        return 1e-4 * (1.0 - 0.02 * i)

    def before(i):                                     # host side of step i: the batch into the static buffers, the scheduler
        x.copy_(xs[i % 4], non_blocking=True)
        y.copy_(ys[i % 4], non_blocking=True)
        for grp in opt.param_groups:
            grp["lr"] = lr_at(i)

    losses = []
    if not graphed:
        for i in range(STEPS):
            before(i)
            losses.append(float(step()))
    else:
        run = GraphedStep(step, [opt], zero_grad=lambda: opt.zero_grad(set_to_none=True), warmup=2, before_step=before)
        losses += [float(v) for v in run.warmup_losses]
        for i in range(2, STEPS):
            losses.append(float(run()))
    torch.cuda.synchronize()
    sd = opt.state_dict()
    steps = sorted({int(v["step"]) for v in sd["state"].values()})
    return [p.detach().clone() for p in model.parameters()], losses, steps, [b.detach().clone() for b in model.buffers()]


def test_adam_graph_replays_equal_eager_steps_bit_for_bit():
    pe, le, se, be = _seg_run(False)
    pg, lg, sg, bg = _seg_run(True)
    assert se == sg == [STEPS], (se, sg)               # state_dict() reports the device counters after replays
    assert le == lg, (le, lg)
    assert all(torch.equal(a, b) for a, b in zip(pe, pg))
    assert all(torch.equal(a, b) for a, b in zip(be, bg))


def _contrast_run(graphed: bool):
    from stswincl_amd.contrast.lars import LARS, add_weight_decay
    from stswincl_amd.contrast.models.PixPro_swin_v5 import ConsistencyLoss
    from stswincl_amd.graph import GraphedStep
    from stswincl_amd.optim import FusedSGD
    S, B = 64, 2
    args = types.SimpleNamespace(pixpro_p=1.0, pixpro_momentum=0.99, pixpro_clamp_value=0.0, pixpro_transform_layer=1,
                                 pixpro_ins_loss_weight=0.0, pixpro_pos_ratio=0.7, data="endo18", tag="1", pretrainpth=None,
                                 num_instances=400, batch_size=B, epochs=1, start_epoch=1)          # K = 200 steps: the momentum moves visibly
    torch.manual_seed(0)
    model = ConsistencyLoss(args, input_resolution=(S // 8, S // 8)).cuda().train()
    base_lr = 0.05
    opt = LARS(FusedSGD(add_weight_decay(model.pixpro, 1e-5), lr=base_lr, momentum=0.9))
    g = torch.Generator().manual_seed(5)
    ims = [torch.randn(B, 4, 3, S, S, generator=g).cuda() for _ in range(6)]
    masks = [torch.randint(0, 12, (B, 1, S // 8, S // 8), generator=g).float().repeat_interleave(8, 2).repeat_interleave(8, 3).cuda()
             for _ in range(6)]

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=BF):
            loss = model(*ims, *masks)
        loss.backward()
        opt.step()
        return loss

    def set_lr(i):                                     # per-iteration cosine schedule (main_pretrain_swinv5.py: scheduler.step() every iteration)
        for grp in opt.param_groups:
            grp["lr"] = base_lr * 0.5 * (1.0 + math.cos(math.pi * i / 50.0))

    losses = []
    if not graphed:
        for i in range(STEPS):
            set_lr(i)
            losses.append(float(step()))
    else:
        run = GraphedStep(step, [opt], zero_grad=lambda: opt.zero_grad(set_to_none=True), warmup=2, before_step=set_lr)
        losses += [float(v) for v in run.warmup_losses]
        for i in range(2, STEPS):
            losses.append(float(run()))
    torch.cuda.synchronize()
    k = model.pixpro.sync_k()
    return [p.detach().clone() for p in model.parameters()], losses, k


def test_lars_and_momentum_schedule_graph_replays_equal_eager_steps_bit_for_bit():
    pe, le, ke = _contrast_run(False)
    pg, lg, kg = _contrast_run(True)
    assert ke == kg == STEPS, (ke, kg)                 # the key-encoder schedule advanced once per step in both modes
    assert le == lg, (le, lg)
    assert all(torch.equal(a, b) for a, b in zip(pe, pg))      # query AND momentum-key parameters


def test_graphed_step_helper_runs_warmup_capture_and_replays():
    """The helper as a user would call it (constant inputs): 2 warm-up steps + 4 replays == 6 eager steps."""
    from stswincl_amd.graph import GraphedStep
    from stswincl_amd.net.Ours.base18 import TswinPlus
    from stswincl_amd.optim import FusedAdam
    from stswincl_amd.utils.losses import OhemCELoss2D
    S, B = 64, 2
    res = []
    for graphed in (False, True):
        torch.manual_seed(0)
        model = TswinPlus(12, (S // 8, S // 8)).cuda().train()
        opt = FusedAdam(model.parameters(), 1e-3)
        crit = OhemCELoss2D(S * S // 16)
        torch.manual_seed(1)
        x, y = torch.randn(B, 4, 3, S, S, device="cuda"), torch.randint(0, 12, (B, S, S), device="cuda")

        def step():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=BF):
                loss = crit(model(x), y)
            loss.backward()
            opt.step()
            return loss

        if graphed:
            run = GraphedStep(step, [opt], zero_grad=lambda: opt.zero_grad(set_to_none=True), warmup=2)
            for _ in range(4):
                run()
            assert run.steps_run == 6
            # an eager step after replays picks the device counters up again
            step()
        else:
            for _ in range(7):
                step()
        torch.cuda.synchronize()
        res.append([p.detach().clone() for p in model.parameters()])
    assert all(torch.equal(a, b) for a, b in zip(*res))


def test_adam_bias_corrections_from_the_device_clock_match_the_host_expressions():
    from stswincl_amd import hip
    from stswincl_amd.optim import EmaSchedule, _Clock
    c = _Clock("cuda")
    for t in range(1, 40):
        hip.optim_tick(0, c.counter, c.hyper, 0.9, 0.999)
        h = c.hyper.cpu()
        want1 = torch.tensor(1.0 - 0.9 ** t, dtype=torch.float64).float()
        want2 = torch.tensor(math.sqrt(1.0 - 0.999 ** t), dtype=torch.float64).float()
        # device pow / sqrt in double, rounded to fp32: equal to the host's value or its fp32 neighbour
        assert abs(float(h[1]) - float(want1)) <= 1.2e-7 * float(want1) and abs(float(h[2]) - float(want2)) <= 1.2e-7 * float(want2), (t, h)
    assert c.sync() == 39
    e = EmaSchedule("cuda", 0.99, 167625, k=1000)
    for k in range(1000, 1010):
        h = e.tick().cpu()
        want = 1.0 - (1.0 - 0.99) * (math.cos(math.pi * k / 167625) + 1) / 2.0
        assert abs(float(h[3]) - want) <= 1.2e-7, (k, float(h[3]), want)
    assert e.sync() == 1010


def test_weight_gradient_zero_fill_and_ohem_histograms_inside_a_graph():
    """Two launcher-side fills used to be runtime memsets (hipMemset2DAsync of a weight-gradient buffer that a direct-store gemm_tn adds
    into; hipMemsetAsync of the OHEM radix-select histograms).  Captured into a hipGraph they did not reproduce the eager calls (ROCm 7.x):
    replays left parts of the gradient on stale / garbage values and the histograms uncleared, and a replayed training run went to NaN
    after ~130 steps while 20-step comparisons passed (tools/probes/tn_graph_repro.py, graph_vs_eager.py).  They are kernels now; this
    replays both against eager launches on fresh operands."""
    from stswincl_amd import hip
    from stswincl_amd.utils.losses import OhemCELoss2D
    dev = "cuda"
    frames, Hh, Ni, Cin = 4, 16, 512, 1024
    Mk = frames * Hh * Hh
    dils = (18, 12, 6)
    maps = [hip.conv_rowmap(frames, Hh, Hh, Hh, Hh, 3, 1, d, d, False, dev) for d in dils]
    g = [torch.randn(Mk, Ni, device=dev).bfloat16() for _ in dils]
    X = torch.randn(Mk, Cin, device=dev).bfloat16()
    outs = [torch.empty(Ni, 9 * Cin, dtype=torch.float32, device=dev) for _ in dils]
    crit = OhemCELoss2D(64 * 64 // 16)
    logits = torch.randn(2, 12, 64, 64, device=dev)
    labels = torch.randint(0, 12, (2, 64, 64), device=dev)
    loss_out = torch.zeros((), device=dev)

    def launches():
        hip.arena_reset(dev)               # (as a model forward does: the step's zero-initialised accumulators come from a block filled inside the step)
        for i in range(len(dils)):
            hip.gemm_tn(g[i], X, outs[i], Mk=Mk, bt_rows=maps[i], bseg=Cin, overwrite=True, tapminor=True)
        loss_out.copy_(crit(logits, labels))

    launches()
    v = hip.last_variant(1)
    assert v["slabs"] == "" and v["splits"] == 1, v          # the direct-store launch that needs the zero fill
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        launches()
    torch.cuda.synchronize()
    hip.note_capture()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        launches()
    for it in range(12):
        torch.manual_seed(100 + it)
        for t in g:
            t.copy_(torch.randn(Mk, Ni, device=dev).bfloat16())
        X.copy_(torch.randn(Mk, Cin, device=dev).bfloat16())
        # alternate between the two OHEM branches: confident logits (few hard pixels -> the n_min largest by radix select) and random ones
        lg = torch.randn(2, 12, 64, 64, device=dev)
        if it % 2:
            lg = lg + 12.0 * torch.nn.functional.one_hot(labels, 12).permute(0, 3, 1, 2)
        logits.copy_(lg)
        graph.replay()
        torch.cuda.synchronize()
        got, got_loss = [o.clone() for o in outs], float(loss_out)
        launches()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(got, outs)), it
        assert got_loss == float(loss_out) and got_loss == got_loss, (it, got_loss, float(loss_out))
