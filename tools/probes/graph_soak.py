#!/usr/bin/env python3
"""Soak of the hipGraph TRAINING mode (stswincl_amd.graph.GraphedStep, round 6): the bench configuration (B = 4 x 4 frames x 512 x 512, bf16, OHEM-CE,
FusedAdam) trained for N replays on a fixed synthetic batch whose labels the image carries, with a host-side cosine learning-rate schedule pushed
before every replay.  The loss has to fall and stay finite, Adam's device step counter has to equal the number of steps, the memory in use must not
grow, and the replay time is reported.   python tools/probes/graph_soak.py [steps]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stswincl_amd.graph import GraphedStep
from stswincl_amd.net.Ours.base18 import TswinPlus
from stswincl_amd.optim import FusedAdam
from stswincl_amd.utils.losses import OhemCELoss2D

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
EAGER = len(sys.argv) > 2 and sys.argv[2] == "eager"        # the same trajectory with eager launches (bit-identical by construction)
LR = float(os.environ.get("SOAK_LR", "3e-4"))
S, B = 512, 4
torch.manual_seed(0)
m = TswinPlus(12, (S // 8, S // 8)).cuda().train()
opt = FusedAdam(m.parameters(), LR)
y = torch.randint(0, 12, (B, S // 32, S // 32), device="cuda").repeat_interleave(32, 1).repeat_interleave(32, 2)
x = torch.randn(B, 4, 3, S, S, device="cuda") + torch.nn.functional.one_hot(y, 12).permute(0, 3, 1, 2)[:, None, :3].float() * 2.0
crit = OhemCELoss2D(S * S // 16)


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = crit(m(x), y)
    loss.backward()
    opt.step()
    return loss


def before(i):
    for g in opt.param_groups:
        g["lr"] = LR * 0.5 * (1.0 + math.cos(math.pi * i / steps))


if EAGER:
    class _Eager:
        warmup_losses = []
        n = 0
        def __call__(self):
            before(self.n); self.n += 1
            return step().detach()
    run = _Eager()
    run.warmup_losses = [run(), run()]
else:
    run = GraphedStep(step, [opt], zero_grad=lambda: opt.zero_grad(set_to_none=True), before_step=before)
hist = [(i, float(v)) for i, v in enumerate(run.warmup_losses)]
torch.cuda.synchronize()
mem0 = torch.cuda.memory_allocated()
t0 = time.perf_counter()
for i in range(2, steps):
    loss = run()
    if i % 20 == 0 or i == steps - 1:
        hist.append((i, float(loss)))                 # (a host read: one synchronisation per 100 steps)
        print(i, hist[-1][1], flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
counts = sorted({int(v["step"]) for v in opt.state_dict()["state"].values()})
print(f"{steps - 2} replays in {dt:.2f} s = {1e3 * dt / (steps - 2):.3f} ms per step ({B * 4 * (steps - 2) / dt:.1f} frames/s incl. the loss reads); "
      f"Adam step counts {counts}; memory allocated {mem0 / 2**30:.2f} -> {torch.cuda.memory_allocated() / 2**30:.2f} GiB")
assert all(v == v for _, v in hist), "NaN"
assert counts == [steps], counts
print("first", hist[0][1], "last", hist[-1][1], "OK" if hist[-1][1] < 0.5 * hist[0][1] else "NOT FALLING")
