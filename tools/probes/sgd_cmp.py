import sys, torch
sys.path.insert(0, "/root/repo")
from stswincl_amd.optim import FusedSGD
torch.manual_seed(0)
shapes = [(512, 512), (2048,), (64, 3, 7, 7), (1000, 17), (5,)]
pa = [torch.randn(*s, device="cuda").requires_grad_(True) for s in shapes]
pb = [p.detach().clone().requires_grad_(True) for p in pa]
oa = torch.optim.SGD(pa, 0.05, momentum=0.9, weight_decay=1e-5)
ob = FusedSGD(pb, 0.05, momentum=0.9, weight_decay=1e-5)
for step in range(4):
    gs = [torch.randn_like(p) for p in pa]
    for p, q, g in zip(pa, pb, gs):
        if step == 2 and p.dim() == 1:      # a parameter without gradient in one step
            p.grad = None; q.grad = None
        else:
            p.grad = g.clone(); q.grad = g.clone()
    oa.step(); ob.step()
    print(step, [float((p - q).abs().max()) for p, q in zip(pa, pb)])
