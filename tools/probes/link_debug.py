import sys; sys.path.insert(0,'/root/repo')
import torch
from stswincl_amd import headops as H
from stswincl_amd.net.Ours.resnet import ResNet18_OS8
torch.manual_seed(0)
net = ResNet18_OS8().cuda()
x0 = torch.randn(4, 3, 64, 64, device="cuda")
gout = None
def rel(a,b): return float((a.double()-b.double()).norm()/(b.double().norm()+1e-30))
def run(link):
    global gout
    H._RESID_GRAD_LINK = link
    net.zero_grad(set_to_none=True)
    x = x0.clone().requires_grad_(True)
    img = x * 1.0
    tok, h, w = net.forward_tokens(img, groups=2)
    if gout is None: gout = torch.randn(tok.shape, device="cuda")
    (tok.float() * gout).sum().backward()
    return {k: p.grad.clone() for k, p in net.named_parameters()}
a, b, c = run(True), run(False), run(False)
for k in a:
    print(f"{k:40s} link-vs-plain {rel(a[k], b[k]):.2e}   plain-vs-plain {rel(c[k], b[k]):.2e}")
