#!/usr/bin/env python3
"""Does a few hundred real optimizer steps on a FIXED synthetic batch: the loss has to fall (the model memorises the batch) and
stay finite - an end-to-end check that the gradients, including the linked ones (frame pair, residual joins, epilogue
statistics), point downhill.  bf16 autocast, FusedAdam 3e-4; labels are 32 x 32 blocks that the image carries in its first three channels, 2 clips x 4 frames x 256 x 256."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stswincl_amd.net.Ours.base18 import TswinPlus
from stswincl_amd.utils.losses import OhemCELoss2D
from stswincl_amd.optim import FusedAdam

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.manual_seed(0)
m = TswinPlus(12, (32, 32)).cuda().train()
opt = FusedAdam(m.parameters(), 3e-4)
x = torch.randn(2, 4, 3, 256, 256, device="cuda")
y = torch.randint(0, 12, (2, 8, 8), device="cuda").repeat_interleave(32, 1).repeat_interleave(32, 2)      # 32 x 32 px label blocks
x = x + torch.nn.functional.one_hot(y, 12).permute(0, 3, 1, 2)[:, None, :3].float() * 2.0                    # (and an image that shows them)
crit = OhemCELoss2D(256 * 256 // 16)
hist = []
for i in range(steps):
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = crit(m(x), y)
    loss.backward()
    opt.step()
    if i % 20 == 0 or i == steps - 1:
        hist.append((i, float(loss.detach())))
        print(i, hist[-1][1], flush=True)
assert all(v == v for _, v in hist), "NaN"
print("first", hist[0][1], "last", hist[-1][1], "OK" if hist[-1][1] < 0.5 * hist[0][1] else "NOT FALLING")
