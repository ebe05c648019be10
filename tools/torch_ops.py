#!/usr/bin/env python3
"""Where do the remaining torch (aten) kernels of the training step come from?  Runs 2 steps under torch.profiler with
stacks and prints, per aten op, the innermost stswincl_amd frame that issued it with call count and device time."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from stswincl_amd import hip
from stswincl_amd.net.Ours.base18 import TswinPlus
from stswincl_amd.utils.losses import OhemCELoss2D
from stswincl_amd.optim import FusedAdam

dev = "cuda"
if len(sys.argv) > 1 and sys.argv[1] == "contrast":          # the contrastive pre-training step of bench.py --workload contrast
    import types
    from stswincl_amd.contrast.models.PixPro_swin_v5 import ConsistencyLoss
    S, B = 256, 8
    args = types.SimpleNamespace(pixpro_p=1.0, pixpro_momentum=0.99, pixpro_clamp_value=0.0, pixpro_transform_layer=1,
                                 pixpro_ins_loss_weight=0.0, pixpro_pos_ratio=0.7, data="endo18", tag="1", pretrainpth=None,
                                 num_instances=2235, batch_size=B, epochs=150, start_epoch=1, pixpro_bank="sample")
    torch.manual_seed(0)
    model = ConsistencyLoss(args, input_resolution=(S // 8, S // 8)).to(dev).train()
    params = [p for p in model.parameters() if p.requires_grad]
    from stswincl_amd.optim import make_contrast_optimizer
    opt, _ = make_contrast_optimizer(params, batch_size=B)      # LARS over SGD-momentum, as bench.py's secondary workload
    ims = [torch.randn(B, 4, 3, S, S, device=dev) for _ in range(6)]
    masks = [torch.randint(0, 12, (B, 1, S, S), device=dev).float() for _ in range(6)]

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = model(*ims, *masks)
        loss.backward()
        opt.step()
else:
    S, B = 512, 4
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


for _ in range(2):
    step()
torch.cuda.synchronize()
NSTEP = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    for _ in range(NSTEP):
        step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith("aten::"):
        continue
    dt = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
    if dt <= 0:
        continue
    # innermost-first stack: the first two repo frames (a custom autograd Function's forward / backward + its caller)
    fr2 = [fr.split("stswincl_amd/")[-1][:60] for fr in (ev.stack or []) if "stswincl_amd/" in fr and "hip.py" not in fr][:2]
    frame = " <- ".join(fr2) if fr2 else (ev.stack[0][-70:] if ev.stack else "?")
    k = (ev.name, frame)
    agg[k][0] += 1
    agg[k][1] += dt
tot = sum(v[1] for v in agg.values())
print(f"aten device time: {tot / NSTEP / 1e3:.3f} ms/step")
for (name, frame), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{t / NSTEP / 1e3:8.3f} ms/step {c // NSTEP:4d}x  {name:24s} {frame}")
