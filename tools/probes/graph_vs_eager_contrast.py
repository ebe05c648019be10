#!/usr/bin/env python3
"""graph_vs_eager.py for the contrastive pre-training step (ConsistencyLoss, LARS over SGD, per-iteration cosine learning rate, key-encoder momentum
schedule): two identical models, one eager, one replayed, compared after every step.   python tools/probes/graph_vs_eager_contrast.py [steps] [S] [B] [bank]"""
import math, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stswincl_amd.contrast.models.PixPro_swin_v5 import ConsistencyLoss
from stswincl_amd.graph import GraphedStep
from stswincl_amd.optim import make_contrast_optimizer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
bank = sys.argv[4] if len(sys.argv) > 4 else "sample"


def make():
    args = types.SimpleNamespace(pixpro_p=1.0, pixpro_momentum=0.99, pixpro_clamp_value=0.0, pixpro_transform_layer=1, pixpro_ins_loss_weight=0.0,
                                 pixpro_pos_ratio=0.7, data="endo18", tag="1", pretrainpth=None, num_instances=400, batch_size=B, epochs=4,
                                 start_epoch=1, pixpro_bank=bank)
    torch.manual_seed(0)
    m = ConsistencyLoss(args, input_resolution=(S // 8, S // 8)).cuda().train()
    params = [p for p in m.parameters() if p.requires_grad]
    opt, _ = make_contrast_optimizer(params, batch_size=B)
    base = [g["lr"] for g in opt.param_groups]
    torch.manual_seed(1)
    ims = [torch.randn(B, 4, 3, S, S, device="cuda") for _ in range(6)]
    masks = [torch.randint(0, 12, (B, 1, S // 8, S // 8), device="cuda").float().repeat_interleave(8, 2).repeat_interleave(8, 3) for _ in range(6)]

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = m(*ims, *masks)
        loss.backward()
        opt.step()
        return loss

    def before(i):
        for g, b0 in zip(opt.param_groups, base):
            g["lr"] = b0 * 0.5 * (1.0 + math.cos(math.pi * i / 400.0))
    return m, opt, step, before


me, oe, step_e, before_e = make()
mg, og, step_g, before_g = make()
run = GraphedStep(step_g, [og], zero_grad=lambda: og.zero_grad(set_to_none=True), before_step=before_g)
for i in range(2):
    before_e(i)
    step_e()
names = [n for n, _ in me.named_parameters()] + ["buffer:" + n for n, _ in me.named_buffers()]
tens = lambda m: [p for p in m.parameters()] + [b for b in m.buffers()]   # noqa: E731
for i in range(2, steps):
    before_e(i)
    a = float(step_e().detach())
    b = float(run())
    d = [(n, float((x.float() - y.float()).abs().max())) for n, x, y in zip(names, tens(me), tens(mg)) if not torch.equal(x, y)]
    if a != b or d:
        print(f"step {i}: eager loss {a!r} graph loss {b!r}; {len(d)} tensors differ", d[:8], flush=True)
        break
    if i % 10 == 0:
        print(f"step {i}: identical (loss {a:.6f})", flush=True)
else:
    print(f"{steps} steps: eager and graph identical bit for bit (k = {me.pixpro.sync_k()} / {mg.pixpro.sync_k()})")
