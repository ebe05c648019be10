"""Where does bf16 error accumulate in TswinPlus?  (diagnostic; GPU)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import golden_util as gu
from stswincl_amd.net.Ours.base18 import TswinPlus, decode_tokens
from stswincl_amd import headops as H

def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / b.norm())

g = gu.load("tswinplus.npz")
m = TswinPlus(12, (16, 16))
m.load_state_dict(gu.det_fill(gu.skeleton_sd(g["keys"], g["shapes"], g["dtypes"])), strict=False)
m = m.cuda().train()
x = gu.det_tensor("tswinplus/x", (2, 4, 3, 128, 128)).cuda()
import copy
def run(ac_resnet, ac_rest):
    mm = copy.deepcopy(m)
    with torch.no_grad():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=ac_resnet):
            seq = [mm.resnet(x[:, i].contiguous(memory_format=torch.channels_last)) for i in range(4)]
        seq = [s.float() for s in seq]
        b, c, h, w = seq[0].shape
        tem = torch.stack([H.to_tokens(s).view(b, h * w, c) for s in seq], 1)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=ac_rest):
            t1, t2 = mm.swin.forward_tokens(tem)
        return tem, t1.float(), t2.float()
ref = run(False, False)
for name, cfg in (("resnet bf16 only", (True, False)), ("swin bf16 only", (False, True)), ("both", (True, True))):
    out = run(*cfg)
    print(name, "feat %.3e swin1 %.3e swin2 %.3e" % tuple(rel(a, b) for a, b in zip(out, ref)))
with torch.no_grad():
    yf = m(x)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yb = copy.deepcopy(m)(x)
print("logits bf16 vs fp32 (HIP):", rel(yb.float(), yf), " |logits| rms", float(yf.pow(2).mean().sqrt()))
