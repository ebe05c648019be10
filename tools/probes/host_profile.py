#!/usr/bin/env python3
"""cProfile of the host side of eager training steps (where do the ~16 ms of enqueue time per step go?)."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stswincl_amd.net.Ours.base18 import TswinPlus
from stswincl_amd.utils.losses import OhemCELoss2D
from stswincl_amd.optim import FusedAdam
m = TswinPlus(12, (64, 64)).cuda().train()
opt = FusedAdam(m.parameters(), 1e-4)
crit = OhemCELoss2D(512 * 512 // 16)
x = torch.randn(4, 4, 3, 512, 512, device="cuda"); y = torch.randint(0, 12, (4, 512, 512), device="cuda")
def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = crit(m(x), y)
    loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
