#!/usr/bin/env python3
"""How long does the host take to ENQUEUE one eager training step (no synchronisation inside the loop)?  If this approaches the GPU
step time, eager launches (multi-GPU ranks, profiled runs) become host-bound."""
import os, sys, time
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
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / 10:.2f} ms/step, GPU-complete {1e3 * (t2 - t0) / 10:.2f} ms/step, threads {torch.get_num_threads()}")
