#!/usr/bin/env python3
"""Host time to ENQUEUE one training step (eager launches) against the GPU time of the step: how far ahead of the device does Python run?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd.net.Ours.base18 import TswinPlus
from stswincl_amd.utils.losses import OhemCELoss2D
from stswincl_amd.optim import FusedAdam

dev, S, B = "cuda", 512, 4
torch.manual_seed(0)
model = TswinPlus(12, (S // 8, S // 8)).to(dev).train()
opt = FusedAdam(model.parameters(), 1e-4)
crit = OhemCELoss2D(S * S // 16)
x = torch.randn(B, 4, 3, S, S, device=dev)
y = torch.randint(0, 12, (B, S, S), device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = crit(model(x), y)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
per = []
for _ in range(n):
    a = time.perf_counter()
    step()
    per.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
# host-only time: enqueue with the device far behind is what `per` shows once the queue is deep; measure again with a sync before each step
hs = []
for _ in range(8):
    torch.cuda.synchronize()
    a = time.perf_counter()
    step()
    hs.append(time.perf_counter() - a)
torch.cuda.synchronize()
print(f"{n} steps back to back: enqueue loop {1e3 * (t1 - t0) / n:.2f} ms/step, until the device is done {1e3 * (t2 - t0) / n:.2f} ms/step")
print(f"host time of one step enqueued onto an idle device (no back-pressure): {1e3 * sorted(hs)[len(hs) // 2]:.2f} ms (median of 8)")
