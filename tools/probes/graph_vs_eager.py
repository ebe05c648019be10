#!/usr/bin/env python3
"""Where does a hipGraph-replayed training run leave the eager trajectory?  Bench configuration (B x 4 frames x S x S, bf16, OHEM-CE, FusedAdam,
cosine lr pushed per step), two identical models: one stepped eagerly, one through stswincl_amd.graph.GraphedStep, compared after EVERY step
(loss bits, then every parameter and buffer).  Prints the first step at which they differ and the tensors that differ.
  python tools/probes/graph_vs_eager.py [steps] [S] [B]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stswincl_amd.graph import GraphedStep
from stswincl_amd.net.Ours.base18 import TswinPlus
from stswincl_amd.optim import FusedAdam
from stswincl_amd.utils.losses import OhemCELoss2D

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
LR = 3e-4


def make():
    torch.manual_seed(0)
    m = TswinPlus(12, (S // 8, S // 8)).cuda().train()
    opt = FusedAdam(m.parameters(), LR)
    torch.manual_seed(1)
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
            g["lr"] = LR * 0.5 * (1.0 + math.cos(math.pi * i / 1500.0))
    return m, opt, step, before


me, oe, step_e, before_e = make()
mg, og, step_g, before_g = make()
run = GraphedStep(step_g, [og], zero_grad=lambda: og.zero_grad(set_to_none=True), before_step=before_g)
le = []
for i in range(2):
    before_e(i)
    le.append(float(step_e().detach()))
lg = [float(v) for v in run.warmup_losses]
names = [n for n, _ in me.named_parameters()] + ["buffer:" + n for n, _ in me.named_buffers()]


def tensors(m):
    return [p for p in m.parameters()] + [b for b in m.buffers()]


def diff():
    out = []
    for n, a, b in zip(names, tensors(me), tensors(mg)):
        if not torch.equal(a, b):
            d = (a.float() - b.float()).abs().max()
            out.append((n, float(d), float(a.float().abs().max()), bool(torch.isfinite(b).all())))
    return out


print("after the 2 warm-up steps:", le, lg, "differing tensors:", len(diff()), flush=True)
for i in range(2, steps):
    before_e(i)
    a = float(step_e().detach())
    b = float(run())
    d = diff() if (a != b or i % 25 == 0 or i == steps - 1 or os.environ.get('EVERY') == '1') else None
    if a != b or d:
        print(f"step {i}: eager loss {a!r} graph loss {b!r}; {len(d)} tensors differ", flush=True)
        for row in d[:int(os.environ.get('ROWS', '25'))]:
            print("   ", row)
        pe, pg = dict(me.named_parameters()), dict(mg.named_parameters())
        for n, *_ in d[:6]:
            if n in pe and pe[n].grad is not None and pg[n].grad is not None:
                ge, gg = pe[n].grad.reshape(-1), pg[n].grad.reshape(-1)
                bad = (ge != gg).nonzero().reshape(-1)
                wbad = (pe[n].detach().reshape(-1) != pg[n].detach().reshape(-1)).nonzero().reshape(-1)
                print(f"    {n}: grad elements differing {bad.numel()} of {ge.numel()}"
                      + (f" (first {int(bad[0])}, last {int(bad[-1])})" if bad.numel() else "")
                      + f"; weight elements differing {wbad.numel()}" + (f" (first {int(wbad[0])}, last {int(wbad[-1])}, "
                      f"contiguous {bool(int(wbad[-1]) - int(wbad[0]) + 1 == wbad.numel())})" if wbad.numel() else ""))
        break
    if i % 25 == 0:
        print(f"step {i}: identical (loss {a:.6f})", flush=True)
else:
    print(f"{steps} steps: eager and graph identical bit for bit")
