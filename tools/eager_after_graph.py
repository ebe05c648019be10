"""Why are eager steps slower after a hipGraph capture of the same step?  (bench.py's bracketed pass: 27.3-28.6 ms against 25.7 without a capture)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd.graph import GraphedStep
from stswincl_amd.net.Ours.base18 import TswinPlus
from stswincl_amd.optim import FusedAdam
from stswincl_amd.utils.losses import OhemCELoss2D

S, B = 512, 4
torch.manual_seed(0)
model = TswinPlus(12, (S // 8, S // 8)).cuda().train()
opt = FusedAdam(model.parameters(), 1e-4)
crit = OhemCELoss2D(S * S // 16)
x = torch.randn(B, 4, 3, S, S, device="cuda"); y = torch.randint(0, 12, (B, S, S), device="cuda")


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = crit(model(x), y)
    loss.backward()
    opt.step()
    return loss


def timeit(fn, n=12, tag=""):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
    t0 = time.perf_counter()
    for _ in range(n): fn()
    host = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    print(f"{tag:60s} {dt:7.2f} ms/step (host enqueue {host:6.2f} ms)", flush=True)


timeit(step, tag="eager, before any capture")
run = GraphedStep(step, [opt], zero_grad=lambda: opt.zero_grad(set_to_none=True))
timeit(run, tag="graph replay")
timeit(step, tag="eager after the capture (default stream)")
with torch.cuda.stream(run.stream):
    timeit(step, tag="eager after the capture (the capture's warm-up stream)")
if len(sys.argv) > 1:
    del run
    import gc; gc.collect(); torch.cuda.empty_cache()
    timeit(step, tag="eager after dropping the graph + empty_cache")
