import sys; sys.path.insert(0,'/root/repo')
import torch, torch.nn as nn
from torch.profiler import profile, ProfilerActivity
from stswincl_amd import headops as H
conv = nn.Conv2d(64,64,3,1,1,bias=False).cuda()
bn = nn.BatchNorm2d(64).cuda()
x = torch.randn(16*64*64, 64, device="cuda").bfloat16().requires_grad_(True)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y, ho, wo, tab = H.conv_tokens(x, conv, 16, 64, 64, stats=True)
        z = H.batchnorm_tokens(y, bn, relu=True, groups=4, stats=tab)
    z.float().sum().backward()
step(); step()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step()
    torch.cuda.synchronize()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::copy_") and ev.stack:
        print(ev.name, getattr(ev, "self_device_time_total", 0), [s.split("/")[-1][:60] for s in ev.stack if "stswincl" in s or "fillprobe" in s][:5])
