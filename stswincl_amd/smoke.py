"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0, checked against the CPU oracle."""
from __future__ import annotations

import torch


def run() -> None:
    from oracle import stswin_oracle as O   # checker only
    from . import hip
    from .net.Ours.swin_512 import SwinTransformerLayerv5
    hip.load()
    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    torch.manual_seed(0)
    net = SwinTransformerLayerv5(dim=128, input_resolution=(16, 16), num_heads=4)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.randn(2, 4, 128, 16, 16)
    with torch.no_grad():
        r1, r2 = O.swin_layer_v5(x, sd, "", 4)
    net = net.cuda()
    xg = x.cuda().requires_grad_(True)
    o1, o2 = net(xg)                                   # fp32 path: exact-f32 MFMA kernels
    (o1.sum() + o2.sum()).backward()
    e1 = float((o1.cpu() - r1).norm() / r1.norm())
    e2 = float((o2.cpu() - r2).norm() / r2.norm())
    assert e1 < 1e-3 and e2 < 1e-3, (e1, e2)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        b1, b2 = net(xg)                               # bf16 MFMA path
    eb = float((b2.float().cpu() - r2).norm() / r2.norm())
    assert eb < 5e-2, eb
    assert torch.isfinite(xg.grad).all()
    print(f"[smoke] swin layer v5 on {torch.cuda.get_device_name(0)}: fp32 rel err {e1:.2e}/{e2:.2e}, bf16 {eb:.2e} - OK")
