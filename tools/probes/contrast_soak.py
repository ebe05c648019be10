#!/usr/bin/env python3
"""A few dozen ConsistencyLoss steps (LARS over fused SGD, bf16) on one fixed synthetic batch with blocky label maps: the loss
must fall and stay finite; the momentum encoder's step counter advances."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stswincl_amd.contrast.models.PixPro_swin_v5 import ConsistencyLoss
from stswincl_amd.optim import make_contrast_optimizer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
S, B = 128, 2
args = types.SimpleNamespace(pixpro_p=1.0, pixpro_momentum=0.99, pixpro_clamp_value=0.0, pixpro_transform_layer=1,
                             pixpro_ins_loss_weight=0.0, pixpro_pos_ratio=0.7, data="endo18", tag="1", pretrainpth="none",
                             num_instances=2235, batch_size=B, epochs=150, start_epoch=1)
torch.manual_seed(0)
model = ConsistencyLoss(args, input_resolution=(S // 8, S // 8)).cuda().train()
params = [p for p in model.pixpro.parameters() if p.requires_grad]
opt, name = make_contrast_optimizer(params, batch_size=64)
lab = torch.randint(0, 12, (B, 1, 4, 4), device="cuda").float().repeat_interleave(S // 4, 2).repeat_interleave(S // 4, 3)
base = torch.randn(B, 4, 3, S, S, device="cuda") + lab[:, None] * 0.3
ims = [base + 0.1 * torch.randn_like(base) for _ in range(6)]
masks = [lab.clone() for _ in range(6)]
hist = []
for i in range(steps):
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = model(*ims, *masks)
    loss.backward()
    opt.step()
    if i % 10 == 0 or i == steps - 1:
        hist.append(float(loss.detach()))
        print(i, hist[-1], flush=True)
print(name, "first", hist[0], "last", hist[-1], "k", model.pixpro.k, "OK" if hist[-1] == hist[-1] and hist[-1] < hist[0] else "NOT FALLING")
